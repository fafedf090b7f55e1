#!/usr/bin/env python3
"""cfg2 greedy decode of one B=1024 batch: launched kernel by kernel, and replayed as one HIP graph."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from capdec import engine

dev = torch.device("cuda:0")
model = bench.build_model(dev)
B, D, E, LEN = bench.B, bench.D, bench.E, bench.LEN
g = torch.Generator().manual_seed(1234 + 2)
feats = torch.rand(B, D, generator=g).to(dev)


def step():
    with torch.no_grad():
        enc = model.encode_features(feats).view(B, 1, E)
        return engine.tf_greedy_ids(model, enc, LEN, 1, 2, to_cpu=False, resolve_dirty=False)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t1 = timeit(step)
print(f"launched : {t1 * 1e3:.3f} ms  {B / t1:.0f} captions/s", flush=True)
ids_ref = step()[0].clone()
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    step()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    out = step()
torch.cuda.synchronize()
t2 = timeit(graph.replay)
print(f"graph    : {t2 * 1e3:.3f} ms  {B / t2:.0f} captions/s  same ids: {bool(torch.equal(out[0], ids_ref))}", flush=True)
