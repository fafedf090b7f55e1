#!/usr/bin/env python3
"""cfg2 greedy decode at small batches: eager launches against a replayed HIP graph of the same call."""
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
E, LEN = 512, 20
ref = engine.TfModelRef(model)
for B in (32, 64, 1024):
    g = torch.Generator().manual_seed(B)
    enc = torch.randn(B, 1, E, generator=g).to(dev)

    def run():
        return engine.tf_greedy_ids(ref, enc, LEN, 1, 2, to_cpu=False, resolve_dirty=False)

    for _ in range(3):
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = run()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    ids0 = out[0].clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gout = run()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 20
    print(f"B={B}: eager {eager * 1e3:.3f} ms, graph replay {rep * 1e3:.3f} ms; same ids: {bool((gout[0] == ids0).all())}", flush=True)
