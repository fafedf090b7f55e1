#!/usr/bin/env python3
"""vocab_reduce_kernel at 1024 rows x 10000 ids over contraction depths 256..2048: the slope is the main loop, the
intercept what a launch pays outside it (dispatch, first tiles, arg-max epilogue, tail)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import ctypes
import torch
from capdec import _lib

lib = _lib.load()
M, V = 1024, 10000
s = torch.cuda.current_stream().cuda_stream
for K in (256, 512, 1024, 2048):
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(V, K, device="cuda")
    b = torch.randn(V, device="cuda")
    n = lib.capdec_vocab_partial_elems(M, V)
    pv = torch.empty(n, device="cuda")
    pi = torch.empty(n, dtype=torch.int32, device="cuda")

    def go():
        rc = lib.capdec_vocab_argmax_partial(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), M, V, K, pv.data_ptr(), pi.data_ptr(), s)
        assert rc == 0, rc

    for _ in range(5):
        go()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        go()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1e3 / 50
    print(f"K={K}: {us:.1f} us, {2.0 * M * V * K / us / 1e6:.1f} TFLOP/s", flush=True)
