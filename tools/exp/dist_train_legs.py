#!/usr/bin/env python3
"""The training leg of bench.py through the multi-rank code path with one RCCL rank, several times in one process:
does the first leg differ from the ones after it, and does the decode leg before it matter?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29551")
import torch
import torch.distributed as dist
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "dist"
if mode == "dist":
    if os.environ.get("CAPDEC_EXP_NCCL_HIPRI") == "1":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev,
                                pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
    else:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    bench.DIST_ON = True


def fence():
    torch.cuda.synchronize()
    if bench.DIST_ON:
        dist.barrier()
    torch.cuda.synchronize()


import time
from capdec.trainer import CaptionTrainer as _CT
_orig = _CT.step
_host = []


def _timed(self, *a, **k):
    t0 = time.perf_counter()
    r = _orig(self, *a, **k)
    _host.append(time.perf_counter() - t0)
    return r


_CT.step = _timed
burn = int(sys.argv[2]) if len(sys.argv) > 2 else 0
held = [torch.cuda.Stream(device=dev) for _ in range(burn)]          # streams taken from torch's pool before the trainer's
for leg in range(2):
    # CAPDEC_EXP_B64=1: the 64-sample shard, replayed as a HIP graph; =seq: bench.py's order at 8 ranks (512 launched, then 64 as a graph)
    small = os.environ.get("CAPDEC_EXP_B64") == "1" or (os.environ.get("CAPDEC_EXP_B64") == "seq" and leg == 1)
    r = bench.train_bench(dev, 0, 1, 30, 3, fence, batch=64, graph=True) if small else bench.train_bench(dev, 0, 1, 30, 3, fence)
    if small:
        print("collectives inside the graph:", r["config"]["allreduces_captured_in_graph"], "segments", r["config"]["allreduce_segments"])
    print(mode, "burn", burn, "leg", leg, "%.3f ms" % r["ms_per_step"], "host, timed calls: median %.3f max %.3f ms" % (sorted(_host[-30:])[15] * 1e3, max(_host[-30:]) * 1e3), flush=True)
    del _host[:]
if bench.DIST_ON:
    dist.destroy_process_group()
