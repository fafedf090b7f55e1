#!/usr/bin/env python3
"""BASELINE.json configs[3] as worded (S=261) alone: training step and greedy decode variants of bench.py."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
print(json.dumps(bench.train_memory_variant(dev, 4, 2, parts=((197, 768), (64, 2048)), layers=2, name="ViTCNN-Attn")), flush=True)
torch.cuda.empty_cache()
print(json.dumps(bench.decode_memory_variant(dev, 4, 2, ((197, 768), (64, 2048)), 2, "ViTCNN-Attn")), flush=True)
print(json.dumps(bench.train_tokens197_variant(dev, 4, 2)), flush=True)
