#!/usr/bin/env python3
"""The two vocabulary contractions of the cfg3 training step that run on gemm_kernel's automatic tile rule - logits
(M x V x E, forward layout) and their input gradient (M x E x V, (n,t) layout) - per forced tile configuration."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib

lib = _lib.tuning_build().__enter__()
s = torch.cuda.current_stream().cuda_stream
E, V = 512, 10000


def run(M, I, J, K, cfg, tq, reps=10):
    P = torch.randn(I, K, device="cuda")
    Q = torch.randn(K, J, device="cuda") if tq else torch.randn(J, K, device="cuda")
    C = torch.empty(I, J, device="cuda")
    g = (_lib.GemmGroup * 1)()
    g[0].P, g[0].Q, g[0].C = P.data_ptr(), Q.data_ptr(), C.data_ptr()
    g[0].ldp, g[0].ldq, g[0].ldci, g[0].ldcj, g[0].I, g[0].J = K, (J if tq else K), J, 1, I, J
    ex = _lib.GemmExtra()
    lib.capdec_gemm_force_config(cfg)
    flags = _lib.GEMM_TQ if tq else 0

    def go():
        rc = lib.capdec_gemm_ex(g, 1, K, flags, ctypes.byref(ex), s)
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
    us = a.elapsed_time(b) * 1e3 / reps
    return us, 2.0 * I * J * K / us / 1e6


for M in (7488, 10240):
    for name, (I, J, K, tq) in (("logits", (M, V, E, False)), ("input gradient", (M, E, V, True))):
        cfgs = (116, 11, 2116, 2132, 216, 0)
        best = {c: 1e9 for c in cfgs}
        for rnd in range(3):                       # every configuration in every round: the first one pays no warm-up alone
            for c in cfgs:
                us, _ = run(M, I, J, K, c, tq)
                best[c] = min(best[c], us)
        print(f"M={M:6d} {name:15s}: " + "   ".join(f"cfg {c}: {best[c]:6.1f} us {2.0 * I * J * K / best[c] / 1e6 / 157.3 * 100:4.1f} %" for c in cfgs), flush=True)
