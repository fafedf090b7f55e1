#!/usr/bin/env python3
"""cfg3 training step with the main chain on a high-priority HIP stream (the weight-gradient stream keeps the default
priority) against both on default-priority streams."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for prio in (0, -1, 0, -1):
    st = torch.cuda.Stream(device=dev, priority=prio)
    with torch.cuda.stream(st):
        r = bench.train_bench(dev, 0, 1, 10, 4, torch.cuda.synchronize)
    print(f"main stream priority {prio}: {r['value']:.0f} samples/s  {r['ms_per_step']:.3f} ms", flush=True)
