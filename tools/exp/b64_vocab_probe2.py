#!/usr/bin/env python3
"""Weight-gradient tile forms against the contraction length (live rows): vocabulary (10000x512) and layer (512x512,
1536x512) outputs; vocabulary input gradient at small row counts.  Tuning build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from capdec import _lib, ops
from gemm_bench import timeit

dev = "cuda"
g = torch.Generator().manual_seed(0)
E = 512
with _lib.tuning_build() as lib:
    for N in (10000, 512, 1536):
        for live in (704, 1408, 2816, 4224, 5632, 7488):
            cap = (live + 255) // 256 * 256
            dy = torch.randn(cap, N, generator=g).to(dev)
            dy[live:] = 0
            x = torch.randn(cap, E, generator=g).to(dev)
            rows = torch.tensor([live], dtype=torch.int32, device=dev)
            dW = torch.zeros(N, E, device=dev)
            fl = 2.0 * live * N * E
            cols = []
            for cfg in (0, 11, 116, 216, 32, 316):
                lib.capdec_gemm_force_config(cfg)
                b = timeit(lambda: ops.linear_wgrad(dy, x, out=dW, rows_dev=rows), reps=40, warm=10)
                cols.append(f"{cfg}:{b:7.1f}us {fl / b / 1e6:5.1f}TF")
            print(f"wgrad N={N:5d} live {live:5d}  " + "  ".join(cols), flush=True)
    V = 10000
    W = torch.randn(V, E, generator=g).to(dev)
    for live in (352, 704, 1408, 2816):
        cap = (live + 255) // 256 * 256
        dy = torch.randn(cap, V, generator=g).to(dev)
        dy[live:] = 0
        rows = torch.tensor([live], dtype=torch.int32, device=dev)
        fl = 2.0 * live * V * E
        cols = []
        for cfg in (0, 11, 116, 216, 32, 316):
            lib.capdec_gemm_force_config(cfg)
            a = timeit(lambda: ops.linear_dgrad(dy, W, rows_dev=rows), reps=40, warm=10)
            cols.append(f"{cfg}:{a:7.1f}us {fl / a / 1e6:5.1f}TF")
        print(f"vocab dgrad live {live:5d} cap {cap}  " + "  ".join(cols), flush=True)
    lib.capdec_gemm_force_config(0)
