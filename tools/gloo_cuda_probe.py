#!/usr/bin/env python3
"""Two gloo ranks sharing GPU 0: do several outstanding async all-reduces on slices of one CUDA buffer complete?"""
import os
import sys
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
flat = torch.full((20 << 20,) if len(sys.argv) < 2 else (int(sys.argv[1]),), float(rank + 1), device="cuda")
n = flat.numel() // 5
print(f"[rank {rank}] start", file=sys.stderr, flush=True)
works = [dist.all_reduce(flat[i * n:(i + 1) * n], async_op=True) for i in range(5)]
print(f"[rank {rank}] issued", file=sys.stderr, flush=True)
for w in works:
    w.wait()
torch.cuda.synchronize()
print(f"[rank {rank}] done {flat[0].item()} {flat[-1].item()}", file=sys.stderr, flush=True)
dist.destroy_process_group()
