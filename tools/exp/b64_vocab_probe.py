#!/usr/bin/env python3
"""Vocabulary projection backward at the B=64 training shard (704 live rows of a 1280-row slab, V=10000, E=512):
input gradient (split contraction, float atomics) and weight gradient, per tile form (tuning build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from capdec import _lib, ops
from gemm_bench import timeit

dev = "cuda"
g = torch.Generator().manual_seed(0)
V, E = 10000, 512
for live, cap in ((704, 1280), (960, 1280), (7488, 10240)):
    dy = torch.randn(cap, V, generator=g).to(dev)
    dy[live:] = 0
    W = torch.randn(V, E, generator=g).to(dev)
    x = torch.randn(cap, E, generator=g).to(dev)
    rows = torch.tensor([live], dtype=torch.int32, device=dev)
    dW = torch.zeros(V, E, device=dev)
    fl = 2.0 * live * V * E
    with _lib.tuning_build() as lib:
        for cfg in (0, 11, 116, 216, 32, 316):
            lib.capdec_gemm_force_config(cfg)
            a = timeit(lambda: ops.linear_dgrad(dy, W, rows_dev=rows), reps=20)
            b = timeit(lambda: ops.linear_wgrad(dy, x, out=dW, rows_dev=rows), reps=20)
            print(f"live {live:5d} cfg {cfg:3d}: dgrad {a:7.1f} us {fl / a / 1e6:6.1f} TF   wgrad {b:7.1f} us {fl / b / 1e6:6.1f} TF", flush=True)
        lib.capdec_gemm_force_config(0)
