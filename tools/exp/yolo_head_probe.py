#!/usr/bin/env python3
"""The 86436-deep YOLO projection (1024 x 512 outputs, fixed-order split): the contraction as it is (ragged last k-tile:
operands staged through registers) against its first 86432 columns (whole k-tiles: loaded straight into LDS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from capdec import ops
from gemm_bench import timeit

g = torch.Generator().manual_seed(0)
x = torch.rand(1024, 88484, generator=g).cuda()
W = (torch.randn(512, 86436, generator=g) * 0.01).cuda()
W2 = W[:, :86432].contiguous()
out = torch.empty(1024, 512, device="cuda")
for name, fn, K in (("K=86436", lambda: ops.linear(x[:, :86436], W, out=out, split_k=True), 86436),
                    ("K=86432", lambda: ops.linear(x[:, :86432], W2, out=out, split_k=True), 86432),
                    ("K=86436", lambda: ops.linear(x[:, :86436], W, out=out, split_k=True), 86436),
                    ("K=86432", lambda: ops.linear(x[:, :86432], W2, out=out, split_k=True), 86432)):
    us = timeit(fn, reps=20, warm=5)
    print(f"{name}: {us:8.1f} us  {2.0 * 1024 * 512 * K / us / 1e6:6.1f} TF", flush=True)
