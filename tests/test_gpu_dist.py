"""Data-parallel training semantics on the GPU: two ranks (sharing GPU 0, gloo with host-staged collectives) against
a CPU simulation of what the reference gets from accelerate/DDP (train.py:176,224-226,261): every rank starts from
rank 0's weights, computes the loss mean over its OWN non-PAD targets with its OWN BatchNorm batch statistics, the
gradients are averaged over ranks, Adam steps on the average; the BatchNorm running buffers follow rank 0."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import decoder_ref as R

pytestmark = pytest.mark.gpu


def test_two_rank_training_matches_ddp_semantics(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_train_worker import batch
    from models.vitcnnattn_model import VITCNNAttentionModel
    out = str(tmp_path / "dp.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tests", "dist_train_worker.py"), out]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-3000:]
    got = np.load(out)
    # CPU simulation
    H = 4
    torch.manual_seed(7)                                         # rank 0's seed
    m = VITCNNAttentionModel(64, 90, H, 2, dropout=0.0)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k and k != "positional_encoding.pe")
          for k, v in m.state_dict().items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    batches = [batch(r) for r in range(2)]
    ref_losses = [[], []]
    for _ in range(2):
        grads = [torch.zeros_like(p) for p in params]
        new0 = None
        for r, (feats, caps) in enumerate(batches):
            enc, new = R.head_pair(sd, feats, "fc_vit", 1000, training=True)
            loss = R.caption_loss(R.decoder_forward(sd, enc, caps[:, :-1], H), caps[:, 1:])
            gs = torch.autograd.grad(loss, params, allow_unused=True)
            for acc, gr in zip(grads, gs):
                if gr is not None:
                    acc += gr / 2
            ref_losses[r].append(float(loss.detach()))
            if r == 0:
                new0 = new
        for p, gr in zip(params, grads):
            p.grad = gr
        opt.step()
        for k, v in new0.items():                                # buffers follow rank 0
            sd[k] = v
    for r in range(2):
        for a, b in zip(got["loss_r%d" % r], ref_losses[r]):
            assert abs(a - b) <= 1e-4 * abs(b), (r, got["loss_r%d" % r], ref_losses[r])
    lr, steps = 1e-3, 2
    for k, v in sd.items():
        if v.is_floating_point():
            d = np.abs(got["sd__" + k].astype(np.float64) - v.detach().double().numpy())
            assert d.max() <= 2.2 * lr * steps, (k, d.max())
            if not k.endswith(("W_k.bias", "fc_vit.bias", "fc_cnn.bias", "batchnorm.running_mean")):
                assert d.mean() <= 3e-6, (k, d.mean())


def test_rccl_one_rank_group_runs_the_device_collective_branch(tmp_path):
    """The branch that real multi-GPU runs take (RCCL reducing the device buffers in place: asynchronous per-segment
    all-reduces issued inside the backward pass, the whole-buffer form, the graph-replayed step that holds its
    per-segment all-reduces and Adam, and the graph of forward + backward with the exchange launched after it) on one GPU,
    in a one-rank "nccl" group where every collective is the identity: losses and parameters must equal those of the
    trainer that issues no collective; a launched or captured step issues one all-reduce per gradient segment, a
    replayed one none from the host."""
    out = str(tmp_path / "one_rank.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_one_rank_worker.py"), out], env=env,
                         capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-3000:]
    got = np.load(out)
    ref_l, ref_p = got["loss_plain"], got["p_plain"]
    assert ref_l[-1] < ref_l[0] * 1.5 and np.isfinite(ref_p).all()
    nseg = int(got["nseg_graph"])
    assert got["calls_plain"].tolist() == [0, 0, 0, 0]
    assert got["calls_segments"].tolist() == [nseg] * 4
    assert got["calls_whole"].tolist() == [1, 1, 1, 1]
    assert int(got["inside_graph"]) == 1, "RCCL all-reduces were not captured with the step"
    assert got["calls_graph"].tolist() == [nseg, nseg, 0, 0]           # launched, captured, replayed, replayed
    assert int(got["inside_graph_outside"]) == 0 and got["calls_graph_outside"].tolist() == [1, 1, 1, 1]
    for name in ("segments", "whole", "graph", "graph_outside"):
        assert np.allclose(got["loss_" + name], ref_l, rtol=2e-5, atol=0), (name, got["loss_" + name], ref_l)
        d = np.abs(got["p_" + name] - ref_p)
        # same gradients up to the rounding of the weight gradients' float atomics; Adam turns a sign flip of a
        # rounding-level gradient into a step of up to lr
        assert d.max() <= 2.2 * 1e-3 * 4 and d.mean() <= 3e-6, (name, d.max(), d.mean())


def test_bench_multi_rank_flow_runs_under_rccl_with_one_rank(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, RCCL process group, id gather, timer
    all-reduce, trainer collectives, strong-scaling leg), dry-run with one rank: the JSON line must carry every leg."""
    import json
    env = dict(os.environ, CAPDEC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup",
           "1", "--train-steps", "2", "--no-variants", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0
    for leg in ("train", "train_strong"):
        assert line[leg]["value"] > 0 and "error" not in line[leg], line[leg]
        assert line[leg]["loss_last"] < line[leg]["loss_first"]
