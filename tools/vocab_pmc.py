#!/usr/bin/env python3
"""A few launches of the fused vocabulary projection + arg-max at the benchmark shape, for counter collection
(run under rocprofv3 --pmc on the GPU box).  Usage: vocab_pmc.py [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib, ops

_lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 512, generator=g).cuda()
W = torch.randn(10000, 512, generator=g).cuda()
b = torch.randn(10000, generator=g).cuda()
for _ in range(5):
    ops.vocab_argmax(x, W, b)
torch.cuda.synchronize()
