#!/usr/bin/env python3
"""Host time to enqueue one training step against its GPU time (cfg3 shapes, padded captions)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec.trainer import CaptionTrainer
from models.vitattn_model import VITAttentionModel

E, V, H, L, T, Bt = 512, 10000, 4, 3, 20, 512
torch.manual_seed(0)
model = VITAttentionModel(E, V, H, L).cuda()
tr = CaptionTrainer(model, lr=1e-3)
g = torch.Generator().manual_seed(1)
feats = torch.rand(Bt, 1000, generator=g).cuda()
caps = torch.zeros(Bt, T + 1, dtype=torch.long)
for b in range(Bt):
    n = int(torch.randint(8, 20, (1,), generator=g))
    caps[b, 0] = 1
    caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
    caps[b, 1 + n] = 2
caps = caps.cuda()
for _ in range(5):
    tr.step(feats, caps)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.step(feats, caps)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, wall {1e3 * (t2 - t0) / n:.2f} ms/step")

if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        tr.step(feats, caps)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
