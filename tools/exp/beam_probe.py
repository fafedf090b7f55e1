#!/usr/bin/env python3
"""cfg2 beam search (width 3, B=1024, len 20, KV-cached) alone, for a kernel trace:
rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o p -- python3 tools/exp/beam_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
model = bench.build_model(dev)
B, D, LEN = 1024, 2048, 20
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator().manual_seed(1236)
feats = torch.rand(B, D, generator=g).to(dev)
for _ in range(2):
    model.caption_ids_beam(feats, beam_width=W, max_length=LEN)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    model.caption_ids_beam(feats, beam_width=W, max_length=LEN)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"beam W={W}: {dt * 1e3:.2f} ms per {B} captions = {B / dt / 1e3:.1f} k captions/s")
