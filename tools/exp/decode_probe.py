#!/usr/bin/env python3
"""cfg2 greedy decode alone (for a kernel trace of the decode loop)."""
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
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        enc = model.encode_features(feats).view(B, 1, E)
        engine.tf_greedy_ids(model, enc, LEN, 1, 2, to_cpu=True)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) * 1e3:.3f} ms")
