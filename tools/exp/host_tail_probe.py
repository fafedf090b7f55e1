#!/usr/bin/env python3
"""Host time of the tail of a synchronous decode call: pinned allocation, copy enqueue, wait, flag scan."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, ROOT)
import torch

dev = torch.device("cuda:0")
B, ML = 1024, 20
n_ids = 2 * B * ML
pack = torch.zeros((n_ids + 2 * B,), dtype=torch.int32, device=dev)
acc = [0.0] * 5
N = 200
for it in range(N + 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = torch.empty(pack.shape, dtype=torch.int32, pin_memory=True)
    t1 = time.perf_counter()
    host.copy_(pack, non_blocking=True)
    t2 = time.perf_counter()
    torch.cuda.current_stream(dev).synchronize()
    t3 = time.perf_counter()
    rows = torch.nonzero(host[n_ids + B:], as_tuple=False).flatten()
    n = int(rows.numel())
    t4 = time.perf_counter()
    a, b, c = host[:n_ids].view(torch.int64).view(B, ML), host[n_ids:n_ids + B], host[n_ids + B:]
    t5 = time.perf_counter()
    if it >= 20:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[i] += d
print("pinned alloc %.1f us, copy enqueue %.1f us, wait %.1f us, flag scan %.1f us, views %.1f us" % tuple(x / N * 1e6 for x in acc))
keep = torch.empty(pack.shape, dtype=torch.int32, pin_memory=True)
t0 = time.perf_counter()
for it in range(N):
    keep.copy_(pack, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    bad = bool(keep[n_ids + B:].any())
print("with a kept pinned buffer and .any(): %.1f us per call" % ((time.perf_counter() - t0) / N * 1e6))
