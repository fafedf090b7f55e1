#!/usr/bin/env python3
"""cfg2 greedy decode of one B=1024 batch: as one launch sequence, and as NP parts decoded at the same time (one host
thread and one HIP stream per part)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from capdec import engine

dev = torch.device("cuda:0")
model = bench.build_model(dev)
B, D, E, LEN = bench.B, bench.D, bench.E, bench.LEN
g = torch.Generator().manual_seed(1234 + 2)
feats = torch.rand(B, D, generator=g).to(dev)



def whole():
    with torch.no_grad():
        enc = model.encode_features(feats).view(B, 1, E)
        return engine.tf_greedy_ids(model, enc, LEN, 1, 2, to_cpu=True)


def parts(n, streams, results):
    with torch.no_grad():
        enc = model.encode_features(feats).view(B, 1, E)
    torch.cuda.current_stream().synchronize()
    per = B // n

    def work(i):
        with torch.cuda.stream(streams[i]), torch.no_grad():
            results[i] = engine.tf_greedy_ids(model, enc[i * per:(i + 1) * per], LEN, 1, 2, to_cpu=True)

    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for x in th:
        x.start()
    for x in th:
        x.join()


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t1 = timeit(whole)
print(f"one sequence : {t1 * 1e3:.3f} ms  {B / t1:.0f} captions/s", flush=True)
ids_ref = whole()[0]
for n in (2, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    res = [None] * n
    tn = timeit(lambda: parts(n, streams, res))
    ids = torch.cat([r[0] for r in res])
    print(f"{n} parts      : {tn * 1e3:.3f} ms  {B / tn:.0f} captions/s  same ids: {bool(torch.equal(ids, ids_ref))}", flush=True)
