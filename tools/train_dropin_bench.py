#!/usr/bin/env python3
"""The reference's inner training loop as written (train.py:252-262: model(imgs, captions) -> CrossEntropyLoss ->
backward -> torch.optim.Adam) on the drop-in class, next to CaptionTrainer on the same workload (run on the GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec.trainer import CaptionTrainer
from models.vitattn_model import VITAttentionModel

B, T, E, V, L = 512, 20, 512, 10000, 3
g = torch.Generator().manual_seed(0)
feats = torch.rand(B, 1000, generator=g).cuda()
caps = torch.randint(4, V, (B, T + 1), generator=g)
caps[:, 0] = 1
caps = caps.cuda()


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


torch.manual_seed(0)
m = VITAttentionModel(E, V, 4, L).cuda().train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
crit = torch.nn.CrossEntropyLoss(ignore_index=0)


def dropin():
    out = m(feats, caps[:, :-1], "precomputed")
    loss = crit(out.reshape(-1, out.shape[2]), caps[:, 1:].reshape(-1))
    opt.zero_grad()
    loss.backward()
    opt.step()


dt = timed(dropin)
print(f"drop-in loop (autograd bridge + torch CE + torch Adam): {dt * 1e3:.2f} ms/step, {B / dt:.0f} samples/s")
torch.manual_seed(0)
m2 = VITAttentionModel(E, V, 4, L).cuda()
tr = CaptionTrainer(m2, lr=1e-3)
dt = timed(lambda: tr.step(feats, caps))
print(f"CaptionTrainer (fused loss, flat buffers, fused Adam):   {dt * 1e3:.2f} ms/step, {B / dt:.0f} samples/s")
