#!/usr/bin/env python3
"""A few greedy decodes over a 64-cell feature grid at B=1024, len 20 (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import engine
from models.cnnattn_model import CNNAttentionModel

torch.manual_seed(0)
B, S, E = 1024, 64, 512
m = CNNAttentionModel(E, 10000, 4, 2).cuda().eval()
feats = torch.rand(B * S, 2048, generator=torch.Generator().manual_seed(1)).cuda()
with torch.no_grad():
    for _ in range(4):
        enc = m.encode_features(feats).view(B, S, E)
        engine.tf_greedy_ids(m, enc, 20, 1, 2, to_cpu=True)
torch.cuda.synchronize()
