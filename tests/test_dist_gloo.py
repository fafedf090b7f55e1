"""world_size-2 checks of the multi-GPU plumbing on CPU (gloo): gradient averaging with the reference's
per-rank-mean loss semantics, batch sharding and the final caption gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT, load_golden, sub_sd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from capdec import dist as cdist
        from oracle import decoder_ref as R
        torch.set_num_threads(1)
        fx = load_golden("f4_decoder_forward")
        H = int(fx["H"])
        caps = torch.from_numpy(fx["captions"])
        enc = torch.from_numpy(fx["enc_S2"])

        def grads_of(rows):
            sd = {k: v.clone().requires_grad_(v.is_floating_point() and k != "positional_encoding.pe")
                  for k, v in sub_sd(fx, "sd__").items()}
            logits = R.decoder_forward(sd, enc[rows], caps[rows][:, :-1], H)
            R.caption_loss(logits, caps[rows][:, 1:]).backward()      # mean over THIS shard's non-PAD targets
            names = [k for k, v in sd.items() if v.requires_grad]
            return names, torch.cat([(sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])).reshape(-1)
                                     for k in names])

        # 1. DDP semantics: all-reduce mean of per-rank gradients
        lo, hi = cdist.shard_bounds(4, rank, world)
        _, mine = grads_of(slice(lo, hi))
        flat = mine.clone()
        cdist.allreduce_mean_(flat)
        parts = [grads_of(slice(*cdist.shard_bounds(4, r, world)))[1] for r in range(world)]
        want = sum(parts) / world
        ok_mean = torch.allclose(flat, want, rtol=1e-6, atol=1e-9)
        # ... which is NOT the single-process gradient of the concatenated batch when shards hold different
        # numbers of non-PAD targets (the reference has the same property)
        _, single = grads_of(slice(0, 4))
        differs = not torch.allclose(want, single, rtol=1e-3, atol=1e-7)

        # 2. batch-sharded decode: ragged shards, gather to rank 0 restores the row order
        n = 7
        ids = torch.arange(n * 5).view(n, 5)
        lens = torch.arange(n, dtype=torch.int32)
        lo, hi = cdist.shard_bounds(n, rank, world)
        g_ids, g_len = cdist.gather_captions(ids[lo:hi].clone(), lens[lo:hi].clone(), n)
        ok_gather = True
        if rank == 0:
            ok_gather = torch.equal(g_ids, ids) and torch.equal(g_len, lens)
        else:
            ok_gather = g_ids is None
        # 3. broadcast of the initial weights
        w = torch.full((6,), float(rank))
        cdist.broadcast_(w)
        ok_bcast = bool((w == 0).all())
        q.put((rank, ok_mean, differs, ok_gather, ok_bcast))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_mean_gather_and_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_mean, differs, ok_gather, ok_bcast in results:
        assert ok_mean, f"rank {rank}: all-reduce mean mismatch"
        assert ok_gather, f"rank {rank}: gather mismatch"
        assert ok_bcast, f"rank {rank}: broadcast mismatch"
    assert any(r[2] for r in results)


def test_shard_bounds_cover_every_row_once():
    from capdec import dist as cdist
    for n in (0, 1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = cdist.shard_bounds(n, r, world)
                seen.extend(range(lo, hi))
            assert seen == list(range(n))
