#!/usr/bin/env python3
"""Spin-kernel overlap of torch pool streams with the default stream (what trainer._second_stream measures), with and
without a one-rank RCCL group created first: time of one spin alone, then default + candidate for 8 pool streams."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29553")
import torch
import torch.distributed as dist

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if len(sys.argv) > 1 and sys.argv[1] == "dist":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.ones(4, device=dev)
    dist.all_reduce(t, async_op=True).wait()
main = torch.cuda.current_stream(dev)
cycles = 200_000


def spin_ms(streams):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    a.record(main)
    torch.cuda._sleep(cycles)
    for st in streams:
        with torch.cuda.stream(st):
            torch.cuda._sleep(cycles)
    for st in streams:
        main.wait_stream(st)
    b.record(main)
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b)


spin_ms(())
print("alone %.3f ms" % min(spin_ms(()), spin_ms(())))
for i in range(8):
    c = torch.cuda.Stream(device=dev)
    print("pool stream %d: %.3f ms" % (i, min(spin_ms((c,)), spin_ms((c,)))))
if dist.is_initialized():
    dist.destroy_process_group()
