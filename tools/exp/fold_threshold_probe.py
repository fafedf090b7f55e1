#!/usr/bin/env python3
"""Greedy decode (B=1024, len 20, L=2) over S memory rows: projected keys/values against the folded form."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from capdec import engine
from models.vitattn_model import VITAttentionModel

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VITAttentionModel(bench.E, bench.V, bench.H, 2).to(dev).eval()
B, E, LEN = bench.B, bench.E, bench.LEN
g = torch.Generator().manual_seed(3)


def timeit(enc, **kw):
    for _ in range(2):
        engine.tf_greedy_ids(m, enc, LEN, 1, 2, to_cpu=True, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        engine.tf_greedy_ids(m, enc, LEN, 1, 2, to_cpu=True, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 4 * 1e3


for S in [int(x) for x in os.environ.get("PROBE_S", "16,32,48,64,96,128,197,261").split(",")]:
    enc = torch.randn(B, S, E, generator=g).to(dev)
    a = timeit(enc, no_s1_shortcut=True)
    b = timeit(enc, fold_memory=True)
    print(f"S={S:4d}: projected {a:7.3f} ms   folded {b:7.3f} ms", flush=True)
