#!/usr/bin/env python3
"""Decoder-only training step with a long memory (ViT token sequence, S=197): forward, loss, backward of the
Transformer decoder with enc [B,S,E] given (run on the GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from capdec import autograd as AG
from models.vitattn_model import VITAttentionModel

B, T, S, E, V, L = 512, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 197, 512, 10000, 3
torch.manual_seed(0)
m = VITAttentionModel(E, V, 4, L).cuda().train()
g = torch.Generator().manual_seed(1)
enc = torch.randn(B, S, E, generator=g).cuda()
caps = torch.randint(4, V, (B, T + 1), generator=g).cuda()
ids, tgt = caps[:, :-1].contiguous(), caps[:, 1:].contiguous()


def step():
    grads = {}
    x, tape = AG.tf_forward(m, enc, ids, 0.1, 1234)
    loss, saved = AG.vocab_loss(m, x, tgt)
    dx = AG.vocab_loss_bwd(m, x, saved, grads)
    AG.tf_backward(m, tape, dx, grads, need_denc=True)
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
f_fwd = L * (16 * T * E * E + 4 * S * E * E + 4 * T * T * E + 4 * T * S * E) + 2 * T * E * V
print(f"S={S}: {dt * 1e3:.2f} ms/step, {B / dt:.0f} samples/s, {3 * f_fwd * B / dt / 1e12:.1f} algorithmic TFLOP/s")
