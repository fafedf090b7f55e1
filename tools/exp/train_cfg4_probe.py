#!/usr/bin/env python3
"""The cfg4 (ViTCNN-Attn, two memory rows, L=2) training leg of bench.py alone (for a kernel trace)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

r = bench.train_bench(torch.device("cuda:0"), 0, 1, 20, 4, torch.cuda.synchronize, model_kind="vitcnn", layers=2)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step")}))
