#!/usr/bin/env python3
"""cfg3 training step at B=512: launched kernel by kernel against replayed as one HIP graph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
for graph in (False, True, False, True):
    r = bench.train_bench(dev, 0, 1, 10, 4, torch.cuda.synchronize, graph=graph)
    print(f"graph={graph}: {r['value']:.0f} samples/s  {r['ms_per_step']:.3f} ms  loss {r['loss_first']:.4f} -> {r['loss_last']:.4f}",
          flush=True)
