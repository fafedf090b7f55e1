#!/usr/bin/env python3
"""Time beam search and greedy decode at a few sizes (run on the GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from models.cnnattn_model import CNNAttentionModel

torch.manual_seed(0)
m = CNNAttentionModel(512, 10000, 4, 2).cuda().eval()
g = torch.Generator().manual_seed(1)
for B, W, L in ((32, 3, 50), (256, 3, 50), (1024, 3, 20), (1024, 1, 20)):
    feats = torch.rand(B, 2048, generator=g).cuda()
    for name, fn in (("beam", lambda: m.caption_ids_beam(feats, beam_width=W, max_length=L)),
                     ("greedy", lambda: m.caption_ids(feats, max_length=L))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:6s} B={B:5d} W={W} len={L}: {dt * 1e3:8.2f} ms  {B / dt:10.0f} captions/s", flush=True)
