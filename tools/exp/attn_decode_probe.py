#!/usr/bin/env python3
"""Decode-step attention (one query row per sample and head) over Tk cached or memory rows: achieved GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import ops

M, E, H = 1024, 512, 4
g = torch.Generator().manual_seed(0)
x = torch.randn(M, E, generator=g).cuda()
for Tk in (5, 10, 20, 32, 50, 64, 128, 197):
    kc, vc = torch.randn(M * Tk, E, generator=g).cuda(), torch.randn(M * Tk, E, generator=g).cuda()
    for _ in range(5):
        ops.attention(x, kc, vc, M, H, 1, Tk)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        ops.attention(x, kc, vc, M, H, 1, Tk)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    nbytes = (2.0 * Tk + 2) * M * E * 4
    print(f"Tk={Tk:4d}: {us:7.1f} us  {nbytes / us / 1e3:6.0f} GB/s  ({100 * nbytes / us / 1e3 / 8000:.0f} % of 8 TB/s)", flush=True)
