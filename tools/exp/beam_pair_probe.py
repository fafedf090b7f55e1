#!/usr/bin/env python3
"""Beam search (width 3, len 20, B=1024) of the two-memory-row model (ViTCNN-Attn) and of the one-row model."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from models.cnnattn_model import CNNAttentionModel
from models.vitcnnattn_model import VITCNNAttentionModel

torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
for name, m, D in (("cnn-attn S=1", CNNAttentionModel(512, 10000, 4, 2), 2048),
                   ("vitcnn-attn S=2", VITCNNAttentionModel(512, 10000, 4, 2), 3048)):
    m = m.cuda().eval()
    feats = torch.rand(1024, D, generator=g).cuda()
    fn = lambda: m.caption_ids_beam(feats, beam_width=3, max_length=20)
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt * 1e3:8.2f} ms  {1024 / dt:10.0f} captions/s", flush=True)
