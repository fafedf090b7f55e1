#!/usr/bin/env python3
"""Dependent chain of [M,K] x [512,K]^T contractions at several K: fixed cost per launch against cost per k-tile."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib

_lib.LIB_TUNE_PATH = os.environ.get("CAPDEC_EXP_LIB", _lib.LIB_TUNE_PATH)
lib = _lib.tuning_build().__enter__()
s = torch.cuda.current_stream().cuda_stream
N = 512


def chain(M, K, cfg, reps=30, bias=True, resid=False):
    Ws = [torch.randn(N, K, device="cuda") * K ** -0.5 for _ in range(4)]
    b = torch.zeros(N, device="cuda")
    x = torch.randn(M, max(K, N), device="cuda")
    y = torch.randn(M, max(K, N), device="cuda")
    calls = []
    for k in range(12):
        g = (_lib.GemmGroup * 1)()
        src, dst = (x, y) if k % 2 == 0 else (y, x)
        g[0].P, g[0].Q, g[0].C = src.data_ptr(), Ws[k % 4].data_ptr(), dst.data_ptr()
        g[0].ldp, g[0].ldq, g[0].ldci, g[0].ldcj, g[0].I, g[0].J = max(K, N), K, max(K, N), 1, M, N
        if bias:
            g[0].bias_j = b.data_ptr()
        if resid:
            g[0].R, g[0].ldri, g[0].ldrj = dst.data_ptr(), max(K, N), 1
        calls.append(g)
    ex = _lib.GemmExtra()
    lib.capdec_gemm_force_config(cfg)

    def run():
        for g in calls:
            lib.capdec_gemm_ex(g, 1, K, 0, ctypes.byref(ex), s)

    for _ in range(5):
        run()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    lib.capdec_gemm_force_config(0)
    return a.elapsed_time(e) * 1e3 / (reps * 12)


# an empty-ish kernel chain for the launch floor: layernorm rows of 512 on M rows is the lightest kernel here
for M in (256, 1024):
    print(f"M={M}: " + "  ".join(f"K={K}: {chain(M, K, 32):6.2f} us" for K in (32, 64, 128, 256, 512, 1024, 2048)), flush=True)
print("cfg 316, M=1024: " + "  ".join(f"K={K}: {chain(1024, K, 316):6.2f} us" for K in (32, 128, 512, 2048)), flush=True)
print("no bias, M=1024: " + "  ".join(f"K={K}: {chain(1024, K, 32, bias=False):6.2f} us" for K in (32, 512)), flush=True)
print("residual, M=1024: " + "  ".join(f"K={K}: {chain(1024, K, 32, resid=True):6.2f} us" for K in (32, 512)), flush=True)
