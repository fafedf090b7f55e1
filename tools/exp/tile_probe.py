#!/usr/bin/env python3
"""One contraction shape per tile configuration of gemm_kernel: forward (n,n) x W^T and input gradient (n,t) dy W at
M rows, 512 x 512 weights (the 10240-row family of the cfg3 training step), back-to-back launches on different buffers."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib

lib = _lib.tuning_build().__enter__()
s = torch.cuda.current_stream().cuda_stream
E = 512
TQ = 2      # CAPDEC_GEMM_TQ


def run(M, cfg, tq, reps=20, nbuf=6):
    xs = [torch.randn(M, E, device="cuda") for _ in range(nbuf)]
    ys = [torch.empty(M, E, device="cuda") for _ in range(nbuf)]
    Ws = [torch.randn(E, E, device="cuda") * E ** -0.5 for _ in range(nbuf)]
    bias = torch.zeros(E, device="cuda")
    calls = []
    for k in range(nbuf):
        g = (_lib.GemmGroup * 1)()
        g[0].P, g[0].Q, g[0].C = xs[k].data_ptr(), Ws[k].data_ptr(), ys[k].data_ptr()
        g[0].ldp, g[0].ldq, g[0].ldci, g[0].ldcj, g[0].I, g[0].J = E, E, E, 1, M, E
        g[0].bias_j = 0 if tq else bias.data_ptr()
        calls.append(g)
    ex = _lib.GemmExtra()
    lib.capdec_gemm_force_config(cfg)
    flags = _lib.GEMM_TQ if tq else 0

    def go():
        for g in calls:
            rc = lib.capdec_gemm_ex(g, 1, E, flags, ctypes.byref(ex), s)
            assert rc == 0, rc

    for _ in range(3):
        go()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        go()
    b.record()
    torch.cuda.synchronize()
    lib.capdec_gemm_force_config(0)
    us = a.elapsed_time(b) * 1e3 / (reps * nbuf)
    return us, 2.0 * M * E * E / us / 1e6


cfgs = (116, 11, 2116, 2132, 216)
for M in (7488, 10240):
    for tq in (False, True):
        row = f"M={M:6d} {'dgrad (n,t)' if tq else 'forward (n,n)'}: "
        for c in cfgs:
            us, tf = run(M, c, tq)
            row += f"cfg {c}: {us:6.1f} us {tf:5.1f} TF ({tf / 157.3 * 100:4.1f} %)   "
        print(row, flush=True)
