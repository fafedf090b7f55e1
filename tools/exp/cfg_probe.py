#!/usr/bin/env python3
"""Dependent chains of x W^T + b contractions (K = N = 512, 12 different weights) at several row counts, per tile
configuration."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib

_lib.LIB_TUNE_PATH = os.environ.get("CAPDEC_EXP_LIB", _lib.LIB_TUNE_PATH)   # an experimental build of the same library
lib = _lib.tuning_build().__enter__()      # libcapdec_tune.so: the build that has the tile overrides
s = torch.cuda.current_stream().cuda_stream
E = 512
Ws = [torch.randn(E, E, device="cuda") * E ** -0.5 for _ in range(12)]
bias = torch.zeros(E, device="cuda")


def chain(M, ngroups, cfg, reps=30):
    bufs = [torch.randn(M, E, device="cuda") for _ in range(2 + ngroups)]
    calls = []
    for k in range(12):
        g = (_lib.GemmGroup * ngroups)()
        for q in range(ngroups):
            x, y = bufs[k % 2], bufs[2 + q]
            g[q].P, g[q].Q, g[q].C = x.data_ptr(), Ws[(k + q) % 12].data_ptr(), y.data_ptr()
            g[q].ldp, g[q].ldq, g[q].ldci, g[q].ldcj, g[q].I, g[q].J = E, E, E, 1, M, E
            g[q].bias_j = bias.data_ptr()
        calls.append(g)
    ex = _lib.GemmExtra()
    lib.capdec_gemm_force_config(cfg)

    def run():
        for g in calls:
            lib.capdec_gemm_ex(g, ngroups, E, 0, ctypes.byref(ex), s)

    for _ in range(5):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        run()
    b.record()
    torch.cuda.synchronize()
    lib.capdec_gemm_force_config(0)
    return a.elapsed_time(b) * 1e3 / (reps * 12)


cfgs = (32, 316, 116, 11)
print("rows x groups        " + "".join(f"cfg {c:<8d}" for c in cfgs))
shapes = ((256, 1), (512, 1), (1024, 1), (1536, 1), (2048, 1), (3072, 1), (4096, 1), (1024, 3), (2048, 3), (3072, 3))
if os.environ.get("CAPDEC_PROBE_SHORT"):
    shapes = ((1024, 1), (1024, 3))
for M, ng in shapes:
    print(f"M={M:5d} x{ng}:        " + "".join(f"{chain(M, ng, c):7.2f} us  " for c in cfgs), flush=True)
