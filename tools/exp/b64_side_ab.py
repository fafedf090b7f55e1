#!/usr/bin/env python3
"""The launched (and, at 64 samples, the graph-replayed) training step with and without the second stream for the
weight gradients (CAPDEC_WGRAD_STREAM=0/1), over the batch size."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def fence():
    torch.cuda.synchronize()


for B, graph in ((64, True), (64, False), (128, False), (256, False), (512, False)):
    r = bench.train_bench(dev, 0, 1, 40, 5, fence, batch=B, graph=graph)
    print("wgrad stream", os.environ.get("CAPDEC_WGRAD_STREAM", "1"), "B", B, "graph", graph, "%.3f ms" % r["ms_per_step"], flush=True)
