#!/usr/bin/env python3
"""In-situ cost of the decode step's 1024x512x512 contractions: a dependent chain over 12 different weight matrices
(more than one L2 holds), with and without the device-side row count, bias and residual."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
E = 512
Ws = [torch.randn(E, E, device="cuda") * E ** -0.5 for _ in range(12)]
bs = [torch.zeros(E, device="cuda") for _ in range(12)]
rows = torch.tensor([1024], dtype=torch.int32, device="cuda")


def chain(M, use_rows, use_bias, use_res, reps=40):
    bufs = [torch.randn(M, E, device="cuda") for _ in range(3)]
    calls = []
    for k in range(12):
        g = (_lib.GemmGroup * 1)()
        x, y, r = bufs[k % 3], bufs[(k + 1) % 3], bufs[(k + 2) % 3]
        g[0].P, g[0].Q, g[0].C = x.data_ptr(), Ws[k].data_ptr(), y.data_ptr()
        g[0].ldp, g[0].ldq, g[0].ldci, g[0].ldcj, g[0].I, g[0].J = E, E, E, 1, M, E
        if use_bias:
            g[0].bias_j = bs[k].data_ptr()
        if use_res:
            g[0].R, g[0].ldri, g[0].ldrj = r.data_ptr(), E, 1
        ex = _lib.GemmExtra()
        if use_rows:
            ex.rows_dev = rows.data_ptr()
        calls.append((g, ex))

    def run():
        for g, ex in calls:
            lib.capdec_gemm_ex(g, 1, E, 0, ctypes.byref(ex), s)

    for _ in range(5):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * 12)


for name, args in (("M=1024 plain", (1024, False, False, False)),
                   ("M=1024 bias", (1024, False, True, False)),
                   ("M=1024 bias+res", (1024, False, True, True)),
                   ("M=1280 rows_dev=1024", (1280, True, False, False)),
                   ("M=1280 rows_dev bias+res", (1280, True, True, True))):
    print(f"{name:28s} {chain(*args):6.2f} us / launch", flush=True)
