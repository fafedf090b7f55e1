#!/usr/bin/env python3
"""Step time of CaptionTrainer at the cfg3 shapes, launched kernel by kernel vs replayed as one HIP graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-captioning_amd")):
    sys.path.insert(0, p)
import torch
from capdec.trainer import CaptionTrainer
from models.vitattn_model import VITAttentionModel

E, H, L, V, T = 512, 4, 3, 10000, 20
dev = torch.device("cuda", 0)


def run(B, graph, steps=30, side=None):
    torch.manual_seed(0)
    m = VITAttentionModel(E, V, H, L).to(dev)
    tr = CaptionTrainer(m, lr=1e-3, graph=graph, wgrad_stream=side)
    g = torch.Generator().manual_seed(3)
    feats = torch.rand(B, 1000, generator=g).to(dev)
    caps = torch.zeros(B, T + 1, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(8, 20, (1,), generator=g))
        caps[b, 0] = 1
        caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
        caps[b, 1 + n] = 2
    caps = caps.to(dev)
    for _ in range(4):
        loss = tr.step(feats, caps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(feats, caps)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B:4d} graph={graph!s:5} side={side}: {dt / steps * 1e3:7.3f} ms/step ({B * steps / dt:9.0f} samples/s), "
          f"host enqueue {th / steps * 1e3:6.3f} ms/step, loss {float(loss.item()):.4f}", flush=True)


if len(sys.argv) > 1:      # one configuration: B graph side   (for a profiler run)
    run(int(sys.argv[1]), sys.argv[2] == "1", side=sys.argv[3] == "1")
else:
    for B in (64, 128, 512):
        for graph in (False, True):
            for side in (True, False):
                run(B, graph, side=side)
