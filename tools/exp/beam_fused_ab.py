#!/usr/bin/env python3
"""KV-cached beam search (width 3, len 20) with the decoder layers of a step as one launch against separate launches
(tuning build), at several batch sizes; the PAD-redo is excluded (the cached entry point alone is timed)."""
import os
import sys
import time
import ctypes

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, ROOT)
import torch
import bench
from capdec import _lib, engine

dev = torch.device("cuda:0")
model = bench.build_model(dev)
E, W, LEN = 512, 3, 20
ref = engine.TfModelRef(model)
with _lib.tuning_build() as lib:
    for B in (32, 64, 256, 1024):
        g = torch.Generator().manual_seed(B)
        enc = torch.randn(B, 1, E, generator=g).to(dev)
        nbytes = lib.capdec_tf_beam_cached_workspace_bytes(ctypes.byref(ref.c), B, W, 1, LEN)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ids = torch.empty((B, LEN), dtype=torch.int64, device=dev)
        ln = torch.empty((B,), dtype=torch.int32, device=dev)
        sc = torch.empty((B, W), dtype=torch.float32, device=dev)
        left = torch.empty((1,), dtype=torch.int32, device=dev)
        pad = torch.zeros((B,), dtype=torch.int32, device=dev)
        res = {}
        for unf in (0, 1, 0, 1):
            lib.capdec_beam_force_unfused(unf)

            def run():
                _lib.check(lib.capdec_tf_beam_search_cached(ctypes.byref(ref.c), enc.data_ptr(), B, W, 1, LEN, 1, 2, 0, LEN,
                                                            ws.data_ptr(), ws.numel(), ids.data_ptr(), ln.data_ptr(),
                                                            sc.data_ptr(), left.data_ptr(), pad.data_ptr(),
                                                            torch.cuda.current_stream().cuda_stream), "beam")
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            res[unf] = (time.perf_counter() - t0) / 10
        print(f"B={B}: one launch {res[0] * 1e3:.2f} ms, separate launches {res[1] * 1e3:.2f} ms; samples with <PAD>: {int(pad.sum())}", flush=True)
