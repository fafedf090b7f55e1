#!/usr/bin/env python3
"""The two big non-atomic contractions of a training step per tile configuration of gemm_kernel: the logits
h W^T ([M,512] x [V,512]^T) and their input gradient dlogits W ([M,V] x [V,512]), back-to-back launches."""
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


def run(M, V, cfg, dgrad, reps=6, nbuf=2):
    hs = [torch.randn(M, E, device="cuda") for _ in range(nbuf)]
    ls = [torch.randn(M, V, device="cuda") for _ in range(nbuf)]
    W = torch.randn(V, E, device="cuda") * E ** -0.5
    bias = torch.zeros(V, device="cuda")
    calls = []
    for k in range(nbuf):
        g = (_lib.GemmGroup * 1)()
        if dgrad:
            g[0].P, g[0].Q, g[0].C = ls[k].data_ptr(), W.data_ptr(), hs[k].data_ptr()
            g[0].ldp, g[0].ldq, g[0].ldci, g[0].ldcj, g[0].I, g[0].J = V, E, E, 1, M, E
        else:
            g[0].P, g[0].Q, g[0].C = hs[k].data_ptr(), W.data_ptr(), ls[k].data_ptr()
            g[0].ldp, g[0].ldq, g[0].ldci, g[0].ldcj, g[0].I, g[0].J = E, E, V, 1, M, V
            g[0].bias_j = bias.data_ptr()
        calls.append(g)
    ex = _lib.GemmExtra()
    assert lib.capdec_gemm_force_config(cfg) == 0
    flags = _lib.GEMM_TQ if dgrad else 0
    K = V if dgrad else E

    def go():
        for g in calls:
            rc = lib.capdec_gemm_ex(g, 1, K, flags, ctypes.byref(ex), s)
            assert rc == 0, rc

    for _ in range(2):
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
    return us, 2.0 * M * E * V / us / 1e6


cfgs = (0, 116, 11, 2116, 2132, 216, 22)
for V in (10000, 10004):
    for M in (7488, 10240):
        for dgrad in (False, True):
            row = f"V={V} M={M:6d} {'dgrad (n,t)' if dgrad else 'logits (n,n)'}: "
            for c in cfgs:
                us, tf = run(M, V, c, dgrad)
                row += f"cfg {c}: {us:6.1f} us ({tf / 157.3 * 100:4.1f} %)  "
            print(row, flush=True)
