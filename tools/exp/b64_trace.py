#!/usr/bin/env python3
"""A few cfg3 training steps at B=64 (the strong-scaling shard), launched kernel by kernel, for a rocprofv3 kernel trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-captioning_amd")):
    sys.path.insert(0, p)
import torch
from capdec.trainer import CaptionTrainer
from models.vitattn_model import VITAttentionModel

E, H, L, V, T = 512, 4, 3, 10000, 20
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = VITAttentionModel(E, V, H, L).to(dev)
tr = CaptionTrainer(m, lr=1e-3, graph=graph)
g = torch.Generator().manual_seed(3)
feats = torch.rand(B, 1000, generator=g).to(dev)
caps = torch.zeros(B, T + 1, dtype=torch.long)
for b in range(B):
    n = int(torch.randint(8, 20, (1,), generator=g))
    caps[b, 0] = 1
    caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
    caps[b, 1 + n] = 2
caps = caps.to(dev)
for _ in range(8):
    loss = tr.step(feats, caps)
torch.cuda.synchronize()
print("loss", float(loss.item()))
