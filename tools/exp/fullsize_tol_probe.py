#!/usr/bin/env python3
"""How far are the full-size (cfg3) training gradients of the HIP path from the oracle's fp32 and fp64 gradients,
tensor by tensor?  Diagnostic for the tolerance of tests/test_gpu_fullsize.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "image-captioning_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import decoder_ref as R
from capdec.trainer import CaptionTrainer
from models.vitattn_model import VITAttentionModel
import test_gpu_fullsize as F

torch.manual_seed(0)
m = VITAttentionModel(F.E, F.V, F.H, 3, dropout=0.0)
g = torch.Generator().manual_seed(1234 + 3)
feats = torch.rand(512, 1000, generator=g)
caps = F._captions(g, 512, 20, F.V)
sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
check = [k for k, v in m.named_parameters()]
head = lambda sd, f: R.head_single(sd, f, "fc_vit", training=True)
torch.set_num_threads(16)
l32, g32, _ = F._oracle_grads(sd0, head, feats, caps, torch.float32)
l64, g64, _ = F._oracle_grads(sd0, head, feats, caps, torch.float64)
m = m.cuda()
tr = CaptionTrainer(m, lr=1e-3)
got = tr.step(feats.cuda(), caps.cuda())
print("loss hip %.7f oracle32 %.7f oracle64 %.7f" % (float(got.item()), l32, l64))
big = max(float(g64[k].abs().max()) for k in check)
for k in check:
    mine = tr.grads[k].detach().cpu().double()
    sc = float(g64[k].abs().max())
    if sc < 1e-6 * big:
        continue
    e_h32 = (mine - g32[k]).abs(); e_h64 = (mine - g64[k]).abs(); e_3264 = (g32[k] - g64[k]).abs()
    tol = 1e-4 * sc
    print(f"{k:48s} scale {sc:.2e}  hip-vs-32 max {float(e_h32.max())/sc:.1e} bad {float((e_h32>tol).double().mean()):.2%} | "
          f"hip-vs-64 max {float(e_h64.max())/sc:.1e} bad {float((e_h64>tol).double().mean()):.2%} | "
          f"32-vs-64 max {float(e_3264.max())/sc:.1e} bad {float((e_3264>tol).double().mean()):.2%}")
