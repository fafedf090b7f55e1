#!/usr/bin/env python3
"""Vocabulary input gradient and weight gradient of one step: back to back on one stream against side by side on two."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib, ops

dev = "cuda"
g = torch.Generator().manual_seed(0)
V, E = 10000, 512
side = torch.cuda.Stream()
for live, cap in ((704, 1280), (1408, 2560), (2816, 5120), (7488, 10240)):
    dy = torch.randn(cap, V, generator=g).to(dev)
    dy[live:] = 0
    W = torch.randn(V, E, generator=g).to(dev)
    x = torch.randn(cap, E, generator=g).to(dev)
    rows = torch.tensor([live], dtype=torch.int32, device=dev)
    dW = torch.zeros(V, E, device=dev)

    def serial():
        ops.linear_wgrad(dy, x, out=dW, rows_dev=rows)
        ops.linear_dgrad(dy, W, rows_dev=rows)

    def pair():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.linear_wgrad(dy, x, out=dW, rows_dev=rows)
        ops.linear_dgrad(dy, W, rows_dev=rows)
        main.wait_stream(side)

    for name, fn in (("serial", serial), ("pair", pair), ("serial", serial), ("pair", pair)):
        for _ in range(10):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(40):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 40
        print(f"live {live:5d} {name:6s}: {us:8.1f} us  {4.0 * live * V * E / us / 1e6:6.1f} TF", flush=True)
