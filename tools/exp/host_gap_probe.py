#!/usr/bin/env python3
"""Where a synchronous cfg2 decode step spends host time: feature head, decode call (enqueue), wait for the ids."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, ROOT)
import torch
import bench
from capdec import engine

dev = torch.device("cuda:0")
model = bench.build_model(dev)
B, E, D, LEN = 1024, 512, 2048, 20
g = torch.Generator().manual_seed(1236)
feats = torch.rand(B, D, generator=g).to(dev)
ref = engine.TfModelRef(model)
acc = [0.0, 0.0, 0.0, 0.0]
N = 30
for it in range(N + 5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        enc = model.encode_features(feats).view(B, 1, E)
    t1 = time.perf_counter()
    ids, lens, nd = engine.tf_greedy_ids(ref, enc, LEN, 1, 2, to_cpu=False, resolve_dirty=False)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if it >= 5:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
tt = 0.0
for it in range(N):
    t0 = time.perf_counter()
    with torch.no_grad():
        enc = model.encode_features(feats).view(B, 1, E)
    engine.tf_greedy_ids(ref, enc, LEN, 1, 2, to_cpu=True)
    tt += time.perf_counter() - t0
print(f"feature head (host) {acc[0] / N * 1e6:.0f} us, decode call returns after {acc[1] / N * 1e6:.0f} us, "
      f"GPU done {acc[2] / N * 1e6:.0f} us later; synchronous step {tt / N * 1e3:.3f} ms")
# the same synchronous step with the benchmark's HIP events around each vocabulary kernel
probe = engine.VocabProbe(LEN)
tp = 0.0
for it in range(N):
    t0 = time.perf_counter()
    with torch.no_grad():
        enc = model.encode_features(feats).view(B, 1, E)
    engine.tf_greedy_ids(ref, enc, LEN, 1, 2, to_cpu=True, probe=probe)
    tp += time.perf_counter() - t0
print(f"synchronous step with {2 * LEN} event records in the stream {tp / N * 1e3:.3f} ms")
