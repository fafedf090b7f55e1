#!/usr/bin/env python3
"""The cfg3 training leg of bench.py alone (for a kernel trace of just that step)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

r = bench.train_bench(torch.device("cuda:0"), 0, 1, 4, 2, torch.cuda.synchronize)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "executed_mfma_frac")}))
