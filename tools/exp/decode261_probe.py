#!/usr/bin/env python3
"""Greedy decode over a 261-row projected memory alone (for a kernel trace)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

print(json.dumps(bench.decode_memory_variant(torch.device("cuda:0"), 2, 1, ((197, 768), (64, 2048)), 2, "ViTCNN-Attn")))
