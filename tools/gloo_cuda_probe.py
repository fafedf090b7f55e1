#!/usr/bin/env python3
"""Two gloo ranks sharing GPU 0: do several outstanding async all-reduces on slices of one CUDA buffer complete
when GPU work and allocations are interleaved between them (the pattern of the trainer's overlapped exchange)?
Usage: gloo_cuda_probe.py [plain|busy|sizes]"""
import os
import sys
import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
rank = int(os.environ["RANK"])
world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
sizes = [5130000, 2630656, 2630656, 2630656, 5120000, 513536] if mode != "plain" else [4 << 20] * 5
flat = torch.full((sum(sizes),), float(rank + 1), device="cuda")
a = torch.randn(4096, 4096, device="cuda")
print(f"[rank {rank}] start {mode}", file=sys.stderr, flush=True)
works, off = [], 0
for n in sizes:
    if mode == "busy":
        for _ in range(20):
            b = a @ a                      # GPU work + fresh allocations between the collectives
            c = torch.empty(1 << 22, device="cuda")
    works.append(dist.all_reduce(flat[off:off + n], async_op=True))
    off += n
print(f"[rank {rank}] issued", file=sys.stderr, flush=True)
for w in works:
    w.wait()
torch.cuda.synchronize()
print(f"[rank {rank}] done {flat[0].item()} {flat[-1].item()}", file=sys.stderr, flush=True)
dist.destroy_process_group()
