#!/usr/bin/env python3
"""Would pipelining two half batches through the training step pay?  Two independent trainers (B=256 each) on two
host threads / streams against one trainer at B=512 (cfg3 shapes, padded captions)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec.trainer import CaptionTrainer
from models.vitattn_model import VITAttentionModel

E, V, H, L, T = 512, 10000, 4, 3, 20


def make(Bt, seed):
    torch.manual_seed(seed)
    model = VITAttentionModel(E, V, H, L).cuda()
    tr = CaptionTrainer(model, lr=1e-3)
    g = torch.Generator().manual_seed(seed)
    feats = torch.rand(Bt, 1000, generator=g).cuda()
    caps = torch.zeros(Bt, T + 1, dtype=torch.long)
    for b in range(Bt):
        n = int(torch.randint(8, 20, (1,), generator=g))
        caps[b, 0] = 1
        caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
        caps[b, 1 + n] = 2
    return tr, feats, caps.cuda()


def loop(tr, feats, caps, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            tr.step(feats, caps)


def timed(parts, n=20):
    streams = [torch.cuda.Stream() for _ in parts]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=loop, args=(*p, s, n if rep else 3)) for p, s in zip(parts, streams)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return sum(p[1].shape[0] for p in parts) * n / dt


print("one trainer,  B=512       : %.0f samples/s" % timed([make(512, 1)]))
print("two trainers, B=256 each  : %.0f samples/s" % timed([make(256, 2), make(256, 3)]))
print("two trainers, B=512 each  : %.0f samples/s" % timed([make(512, 4), make(512, 5)]))
