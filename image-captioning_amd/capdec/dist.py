"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).  Only two exchange patterns exist on the hot path:

* training: SUM all-reduce of flat fp32 gradient segments, then 1/world in the optimizer
  (DDP semantics of reference train.py:224-226,261);
* decode: batch-sharded, no collective until one final gather of the [B,max_len] token ids and lengths
  (the reference gathers nothing: train.py:289-292 scores rank 0's shard only).
"""
import torch
import torch.distributed as dist


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n_rows, rank_, world):
    """Contiguous, near-equal row ranges: the first n_rows % world ranks get one extra row."""
    q, r = divmod(n_rows, world)
    lo = rank_ * q + min(rank_, r)
    return lo, lo + q + (1 if rank_ < r else 0)


def shard_rows(t, group=None):
    """This rank's contiguous slice of dim 0."""
    lo, hi = shard_bounds(t.shape[0], rank(group), world_size(group))
    return t[lo:hi]


def broadcast_(t, src=0, group=None):
    if world_size(group) > 1:
        dist.broadcast(t, src=src, group=group)
    return t


def allreduce_mean_(flat, group=None):
    """In-place mean over ranks of a flat gradient buffer (SUM all-reduce, then scale)."""
    w = world_size(group)
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / w)
    return flat


def gather_captions(ids, lengths, n_total, dst=0, group=None):
    """Gather ragged row shards (made by ``shard_bounds``) of token ids [b_r, L] / lengths [b_r] to ``dst``.
    Returns (ids [n_total, L], lengths [n_total]) on dst, (None, None) elsewhere."""
    w, r = world_size(group), rank(group)
    if w == 1:
        return ids, lengths
    L = ids.shape[1]
    sizes = [shard_bounds(n_total, k, w) for k in range(w)]
    cap = max(hi - lo for lo, hi in sizes)
    pad_ids = torch.full((cap, L), -1, dtype=ids.dtype, device=ids.device)
    pad_len = torch.zeros((cap,), dtype=lengths.dtype, device=ids.device)
    pad_ids[: ids.shape[0]] = ids
    pad_len[: lengths.shape[0]] = lengths
    if r == dst:
        all_ids = [torch.empty_like(pad_ids) for _ in range(w)]
        all_len = [torch.empty_like(pad_len) for _ in range(w)]
    else:
        all_ids = all_len = None
    dist.gather(pad_ids, all_ids, dst=dst, group=group)
    dist.gather(pad_len, all_len, dst=dst, group=group)
    if r != dst:
        return None, None
    out_ids = torch.cat([a[: hi - lo] for a, (lo, hi) in zip(all_ids, sizes)], dim=0)
    out_len = torch.cat([a[: hi - lo] for a, (lo, hi) in zip(all_len, sizes)], dim=0)
    return out_ids, out_len
