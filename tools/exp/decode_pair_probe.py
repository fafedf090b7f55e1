#!/usr/bin/env python3
"""The two-memory-row decode variants of bench.py alone (cfg4 ViTCNN-Attn, cfg5 YOLOCNN-Attn shard)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
for kind in ("vitcnn", "yolocnn"):
    r = bench.decode_variant(kind, dev, 6, 2)
    print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "head_ms")}), flush=True)
