#!/usr/bin/env python3
"""Latency anatomy of a small contraction: time per launch vs K and tile config with minimal host overhead
(the ctypes argument block is built once)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib


def main():
    lib = _lib.tuning_build().__enter__()      # libcapdec_tune.so: the build that has the tile overrides
    s = torch.cuda.current_stream().cuda_stream
    for M, N in ((1024, 512), (64, 64), (4096, 512)):
        for K in (32, 64, 128, 256, 512, 1024, 2048):
            x = torch.randn(M, K, device="cuda")
            W = torch.randn(N, K, device="cuda")
            out = torch.empty(M, N, device="cuda")
            g = (_lib.GemmGroup * 1)()
            g[0].P, g[0].Q, g[0].C = x.data_ptr(), W.data_ptr(), out.data_ptr()
            g[0].ldp, g[0].ldq, g[0].ldci, g[0].ldcj, g[0].I, g[0].J = K, K, N, 1, M, N
            line = f"M={M:5d} N={N:4d} K={K:5d}:"
            for cfg in (11, 32):
                lib.capdec_gemm_force_config(cfg)
                for _ in range(20):
                    lib.capdec_gemm(g, 1, K, 0, s)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                reps = 200
                for _ in range(reps):
                    lib.capdec_gemm(g, 1, K, 0, s)
                b.record()
                torch.cuda.synchronize()
                line += f"  cfg{cfg}: {a.elapsed_time(b) * 1e3 / reps:7.2f} us"
            print(line, flush=True)
    lib.capdec_gemm_force_config(0)


if __name__ == "__main__":
    main()
