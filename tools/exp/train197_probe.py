#!/usr/bin/env python3
"""The S=197 training variant of bench.py alone (for a kernel trace of just that step)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

print(json.dumps(bench.train_tokens197_variant(torch.device("cuda:0"), steps=4, warmup=2)))
