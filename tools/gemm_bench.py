#!/usr/bin/env python3
"""Micro-benchmark of the contraction kernels at the shapes the decoder uses (run on the GPU box).
Prints one line per case: average microseconds and algorithmic TFLOP/s (HIP events around `reps` launches)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib, ops


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*s):
        return torch.randn(*s, generator=g).to(dev)

    cases = []
    for (M, N, K, tag) in [(1024, 512, 512, "decode proj"), (1024, 2048, 512, "decode 4x"), (10240, 512, 512, "train proj"),
                           (10240, 10000, 512, "train logits"), (1024, 512, 2048, "head cnn"), (512, 512, 1000, "head vit"),
                           (4096, 4096, 4096, "square 4k")]:
        x, W, b = rnd(M, K), rnd(N, K), rnd(N)
        out = torch.empty(M, N, device=dev)
        cases.append((f"fwd  {tag:14s} M={M} N={N} K={K}", 2.0 * M * N * K, lambda x=x, W=W, b=b, out=out: ops.linear(x, W, b, out=out)))
    M, E = 1024, 512
    x = rnd(M, E)
    Ws = [rnd(E, E) for _ in range(3)]
    outs = [torch.empty(M, E, device=dev) for _ in range(3)]
    cases.append((f"fwd  grouped qkv    M={M} N=3x{E} K={E}", 6.0 * M * E * E,
                  lambda: ops.gemm_groups([dict(P=x, Q=w, C=o) for w, o in zip(Ws, outs)], E)))
    for (M, N, K, tag) in [(10240, 512, 512, "train proj"), (10240, 10000, 512, "train vocab")]:
        dy, W, xx = rnd(M, N), rnd(N, K), rnd(M, K)
        cases.append((f"dgrad {tag:13s} M={M} N={N} K={K}", 2.0 * M * N * K, lambda dy=dy, W=W: ops.linear_dgrad(dy, W)))
        dW = torch.zeros(N, K, device=dev)
        cases.append((f"wgrad {tag:13s} M={M} N={N} K={K}", 2.0 * M * N * K, lambda dy=dy, xx=xx, dW=dW: ops.linear_wgrad(dy, xx, out=dW)))
    x, W, b = rnd(1024, 512), rnd(10000, 512), rnd(10000)
    cases.append(("vocab argmax        M=1024 V=10000 K=512", 2.0 * 1024 * 10000 * 512, lambda: ops.vocab_argmax(x, W, b)))
    x2 = rnd(10240, 512)
    tg = torch.randint(0, 10000, (10240,), generator=g).to(dev)
    cases.append(("vocab nll           M=10240 V=10000 K=512", 2.0 * 10240 * 10000 * 512, lambda: ops.vocab_nll(x2, W, b, tg)))
    lib = _lib.tuning_build().__enter__()      # libcapdec_tune.so: the build that has the tile overrides
    print(f"{'case':48s} " + " ".join(f"{'cfg ' + str(c):>22s}" for c in (0, 11, 116, 32, 216)))
    for name, flops, fn in cases:
        cols = []
        for cfg in (0, 11, 116, 32, 216):
            lib.capdec_gemm_force_config(cfg)
            us = timeit(fn, reps=30 if flops < 5e10 else 10)
            cols.append(f"{us:9.1f} us {flops / us / 1e6:6.1f} TF")
        lib.capdec_gemm_force_config(0)
        print(f"{name:48s} " + " ".join(f"{c:>22s}" for c in cols), flush=True)


if __name__ == "__main__":
    _lib.load()
    main()
