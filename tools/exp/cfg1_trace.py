#!/usr/bin/env python3
"""BASELINE configs[0] on the GPU (CNN-RNN greedy decode, B=32, len 20) a few times, for a rocprofv3 kernel trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-captioning_amd")):
    sys.path.insert(0, p)
import time
import torch
from models.cnnrnn_model import CNNtoRNN

torch.manual_seed(0)
m = CNNtoRNN(256, 512, 5000).eval().to("cuda")
fd = torch.rand(32, 2048, generator=torch.Generator().manual_seed(1235)).to("cuda")
for _ in range(5):
    m.caption_ids(fd, max_length=20)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    m.caption_ids(fd, max_length=20)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per batch of 32")
