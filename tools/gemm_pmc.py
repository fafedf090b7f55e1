#!/usr/bin/env python3
"""A handful of contraction launches for counter collection (run under rocprofv3 --pmc on the GPU box).
Usage: gemm_pmc.py [cfg]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib, ops

lib = _lib.tuning_build().__enter__()      # libcapdec_tune.so: the build that has the tile overrides
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 11
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()
lib.capdec_gemm_force_config(cfg)
x, W, b = rnd(10240, 512), rnd(512, 512), rnd(512)
out = torch.empty(10240, 512, device="cuda")
for _ in range(3):
    ops.linear(x, W, b, out=out)
x4, W4 = rnd(4096, 4096), rnd(4096, 4096)
o4 = torch.empty(4096, 4096, device="cuda")
for _ in range(3):
    ops.linear(x4, W4, None, out=o4)
xs, Ws = rnd(1024, 512), rnd(512, 512)
os_ = torch.empty(1024, 512, device="cuda")
for _ in range(3):
    ops.linear(xs, Ws, b, out=os_)
torch.cuda.synchronize()
