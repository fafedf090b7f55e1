#!/usr/bin/env python3
"""cfg2-shaped Transformer greedy decode at batch 32 (the reference's evaluation batch), for a rocprofv3 kernel trace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-captioning_amd")):
    sys.path.insert(0, p)
import torch
from models.cnnattn_model import CNNAttentionModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = CNNAttentionModel(512, 10000, 4, 2).eval().to("cuda")
fd = torch.rand(B, 2048, generator=torch.Generator().manual_seed(1236)).to("cuda")
for _ in range(5):
    m.caption_ids(fd, max_length=20)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    m.caption_ids(fd, max_length=20)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per batch of {B}")
