#!/usr/bin/env python3
"""cfg2 greedy decode of one batch of 1024 as n equal parts on n HIP streams (one host thread enqueues them in turn):
does the latency-bound layers' launch of one part fill in beside the vocabulary kernel of another?"""
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
with torch.no_grad():
    enc = model.encode_features(feats).view(B, 1, E).contiguous()
torch.cuda.synchronize()
base = None
for n in (1, 2, 4, 2, 1):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    rows = B // n

    def run():
        outs = []
        cur = torch.cuda.current_stream(dev)
        for i, st in enumerate(streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(engine.tf_greedy_ids(ref, enc[i * rows:(i + 1) * rows], LEN, 1, 2, to_cpu=False, resolve_dirty=False))
        for st in streams:
            cur.wait_stream(st)
        return outs

    for _ in range(3):
        o = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        o = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    ids = torch.cat([x[0] for x in o]).cpu()
    if base is None:
        base = ids
    print(f"{n} parts of {rows}: {dt * 1e3:.3f} ms per 1024 captions = {B / dt / 1e3:.1f} k captions/s; same ids as one part: {bool((ids == base).all())}", flush=True)
