#!/usr/bin/env python3
"""Times the vocabulary projection + arg-max launch (1024 x 10000 x 512) of whatever libcapdec.so is in place and checks
the ids against a float64 product (ties aside)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import ops

torch.manual_seed(0)
M, V, E = 1024, 10000, 512
x = torch.randn(M, E, device="cuda")
W = torch.randn(V, E, device="cuda") * 0.05
b = torch.randn(V, device="cuda") * 0.1
ids = ops.vocab_argmax(x, W, b)
ids = ids[0] if isinstance(ids, tuple) else ids
ref = (x.double() @ W.double().t() + b.double()).argmax(1)
bad = int((ids.long() != ref).sum())
for _ in range(20):
    ops.vocab_argmax(x, W, b)
torch.cuda.synchronize()
n = 300
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.vocab_argmax(x, W, b)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(json.dumps({"us_per_call_incl_finalize": us, "mismatch_vs_f64": bad, "tflops": 2.0 * M * V * E / us / 1e6}))
