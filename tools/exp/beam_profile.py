#!/usr/bin/env python3
"""A few width-3 beam searches at B samples (argument, default 1024), len 20 (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from models.cnnattn_model import CNNAttentionModel

torch.manual_seed(0)
m = CNNAttentionModel(512, 10000, 4, 2).cuda().eval()
feats = torch.rand(int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 2048, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(4):
    m.caption_ids_beam(feats, beam_width=3, max_length=20)
torch.cuda.synchronize()
