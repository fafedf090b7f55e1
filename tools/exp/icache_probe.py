#!/usr/bin/env python3
"""Does a kernel pay for a cold instruction cache?  The decode step's 1024x512x512 contraction, launched (a) back to
back and (b) alternating with a different, large contraction kernel on a tiny problem.  Read the per-kernel durations
from a rocprofv3 kernel trace of this script."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib

lib = _lib.tuning_build().__enter__()      # libcapdec_tune.so: the build that has the tile overrides
s = torch.cuda.current_stream().cuda_stream
E = 512


def group(M, x, W, y):
    g = (_lib.GemmGroup * 1)()
    g[0].P, g[0].Q, g[0].C = x.data_ptr(), W.data_ptr(), y.data_ptr()
    g[0].ldp, g[0].ldq, g[0].ldci, g[0].ldcj, g[0].I, g[0].J = E, E, E, 1, M, E
    return g


x, y = torch.randn(1024, E, device="cuda"), torch.empty(1024, E, device="cuda")
Ws = [torch.randn(E, E, device="cuda") for _ in range(12)]
gs = [group(1024, x, W, y) for W in Ws]
xs, ys = torch.randn(128, E, device="cuda"), torch.empty(128, E, device="cuda")
small = group(128, xs, Ws[0], ys)
ex = _lib.GemmExtra()
mode = sys.argv[1]
for rep in range(30):
    for g in gs:
        lib.capdec_gemm_force_config(0)
        lib.capdec_gemm_ex(g, 1, E, 0, ctypes.byref(ex), s)
        if mode == "alt":
            for cfg in (22, 216, 11):
                lib.capdec_gemm_force_config(cfg)
                lib.capdec_gemm_ex(small, 1, E, 0, ctypes.byref(ex), s)
torch.cuda.synchronize()
lib.capdec_gemm_force_config(0)
