#!/usr/bin/env python3
"""Ragged self-attention backward at the cfg3 shape (512 samples, 4 heads of 128, captions of 8-19 live rows of 20) alone:
the matrix-core form against the row-by-row one-pass kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib, ops

lib = _lib.tuning_build().__enter__()
B, T, H, dk = 512, 20, 4, 128
E = H * dk
g = torch.Generator().manual_seed(0)
ids = torch.randint(4, 100, (B, T), generator=g)
tgt = torch.randint(4, 100, (B, T), generator=g)
for b in range(B):
    n = int(torch.randint(8, 20, (1,), generator=g))
    ids[b, n:] = 0
    tgt[b, n:] = 0
rr = ops.RaggedRows(ids.cuda(), tgt.cuda(), 0)
M = B * T
q, k, v, do = (torch.randn(M, E, generator=g).cuda() for _ in range(4))
o, lse = ops.attention_ragged(q, k, v, rr, H, want_lse=True)
live = int(rr.counts[1])
nbytes = 7 * live * E * 4     # q, k, v, dO read; dq, dk, dv written


def timeit(reps=50):
    for _ in range(5):
        ops.attention_bwd_ragged(q, k, v, do, lse, rr, H)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        ops.attention_bwd_ragged(q, k, v, do, lse, rr, H)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for rnd in range(2):
    lib.capdec_attention_bwd_force_row_form(0)
    t_m = timeit()
    lib.capdec_attention_bwd_force_row_form(1)
    t_r = timeit()
    lib.capdec_attention_bwd_force_row_form(0)
    print(f"live rows {live}: matrix-core form {t_m:.1f} us ({nbytes / t_m / 1e3:.0f} GB/s), row form {t_r:.1f} us "
          f"({nbytes / t_r / 1e3:.0f} GB/s)", flush=True)
