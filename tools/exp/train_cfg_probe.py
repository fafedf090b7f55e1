#!/usr/bin/env python3
"""Tile configurations at the training step's 512-wide shapes (rows x 512 x 512): forward, input gradient, weight
gradient, each timed as 12 back-to-back launches on different operands."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib, ops

lib = _lib.tuning_build().__enter__()
E = 512


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for M in (7488, 10240, 2560):
    xs = [torch.randn(M, E, device="cuda") for _ in range(4)]
    dys = [torch.randn(M, E, device="cuda") for _ in range(4)]
    Ws = [torch.randn(E, E, device="cuda") * E ** -0.5 for _ in range(4)]
    outs = [torch.empty(M, E, device="cuda") for _ in range(4)]
    gws = [torch.zeros(E, E, device="cuda") for _ in range(4)]
    b = torch.zeros(E, device="cuda")
    gf = 2.0 * M * E * E / 1e9
    print(f"rows {M} ({gf:.2f} GFLOP per launch)")
    for cfg in (11, 116, 22, 216):
        lib.capdec_gemm_force_config(cfg)
        f = timed(lambda: [ops.linear(xs[k], Ws[k], b, out=outs[k]) for k in range(4)]) / 4
        d = timed(lambda: [ops.gemm_raw(dys[k], Ws[k], outs[k], M, E, E, E, E, _lib.GEMM_TQ) for k in range(4)]) / 4
        w = timed(lambda: [ops.linear_wgrad(dys[k], xs[k], out=gws[k]) for k in range(4)]) / 4
        print(f"  cfg {cfg:3d}: forward {f:6.1f} us ({gf / f * 1e3:5.1f} TF/s)   dgrad {d:6.1f} us ({gf / d * 1e3:5.1f})   "
              f"wgrad {w:6.1f} us ({gf / w * 1e3:5.1f})", flush=True)
    lib.capdec_gemm_force_config(0)
    f = timed(lambda: [ops.linear(xs[k], Ws[k], b, out=outs[k]) for k in range(4)]) / 4
    d = timed(lambda: [ops.gemm_raw(dys[k], Ws[k], outs[k], M, E, E, E, E, _lib.GEMM_TQ) for k in range(4)]) / 4
    w = timed(lambda: [ops.linear_wgrad(dys[k], xs[k], out=gws[k]) for k in range(4)]) / 4
    print(f"  auto   : forward {f:6.1f} us   dgrad {d:6.1f} us   wgrad {w:6.1f} us", flush=True)
