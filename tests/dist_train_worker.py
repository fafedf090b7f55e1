"""Worker of tests/test_gpu_dist.py: one of two data-parallel ranks sharing GPU 0 (gloo, host-staged collectives).
Each rank builds the model under its OWN seed (the trainer must replace it with rank 0's weights), trains two steps
on its own batch and rank 0 saves the result.  argv: output .npz path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import numpy as np
import torch
import torch.distributed as dist


def batch(rank, B=10, T=8, V=90):
    g = torch.Generator().manual_seed(100 + rank)
    feats = torch.rand(B, 3048, generator=g)
    caps = torch.zeros(B, T + 1, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(2, T - 1, (1,), generator=g))
        caps[b, 0] = 1
        caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
        caps[b, 1 + n] = 2
    return feats, caps


def main():
    from capdec.trainer import CaptionTrainer
    from models.vitcnnattn_model import VITCNNAttentionModel
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(7 + rank)                     # ranks start different: rank 0's weights must win
    m = VITCNNAttentionModel(64, 90, 4, 2, dropout=0.0).cuda()
    tr = CaptionTrainer(m, lr=1e-3)
    feats, caps = batch(rank)
    losses = [float(tr.step(feats.cuda(), caps.cuda()).item()) for _ in range(2)]
    gathered = [None] * world
    dist.all_gather_object(gathered, losses)
    if rank == 0:
        out = {"loss_r%d" % r: np.asarray(l) for r, l in enumerate(gathered)}
        out.update({"sd__" + k: v.detach().cpu().numpy() for k, v in m.state_dict().items()})
        np.savez(sys.argv[1], **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
