"""GPU parity of the training path: teacher-forced forward, loss and every gradient against the golden
vectors produced by the reference (autograd on its own classes).  Tolerance: 1e-4 relative (per tensor,
relative to the tensor's max magnitude), dropout disabled (dropout parity is statistical only)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub_sd
from gpu_util import REL, close, decoder_from_sd

pytestmark = pytest.mark.gpu


def t(x):
    return torch.from_numpy(np.asarray(x))


def _noise_floor(fx, key_fmt, names):
    """Some gradients are mathematically zero (e.g. the key bias: softmax is invariant to a constant added to
    every score of a row) and hold only rounding noise (~1e-10) in the reference too; compare those against
    1e-6 of the largest gradient magnitude in the model instead of against their own magnitude."""
    big = max(float(np.abs(fx[key_fmt.format(n)]).max()) for n in names if key_fmt.format(n) in fx)
    return 1e-6 * big


def _check_grads(named_params, fx, key_fmt, rtol=REL):
    named_params = list(named_params)
    atol = _noise_floor(fx, key_fmt, [n for n, _ in named_params])
    n = 0
    for name, prm in named_params:
        key = key_fmt.format(name)
        if key not in fx:
            continue
        g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        close(g, fx[key], rtol=rtol, atol=atol, what=name)
        n += 1
    return n


@pytest.mark.parametrize("S", [1, 2, 5])
def test_f4_logits_loss_and_all_gradients_via_autograd(S):
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"])).train()
    caps = t(fx["captions"]).cuda()
    enc = t(fx[f"enc_S{S}"]).cuda().requires_grad_(True)
    logits = dec.decoder_forward(enc, caps[:, :-1])
    close(logits, fx[f"logits_S{S}"], what="logits")
    V = logits.shape[-1]
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, V), caps[:, 1:].reshape(-1), ignore_index=0)
    close(loss, fx[f"loss_S{S}"], rtol=1e-5, what="loss")
    loss.backward()
    close(enc.grad, fx[f"denc_S{S}"], what="denc")
    n = _check_grads(dec.named_parameters(), fx, "grad__{}@S" + str(S))
    assert n == len(list(dec.parameters()))
    if S == 1:   # softmax over one memory row is constant: exactly zero gradient for the cross W_q / W_k
        for l in range(2):
            assert float(dec.decoder_layers[l].cross_attn.W_q.weight.grad.abs().max()) == 0.0
            assert float(dec.decoder_layers[l].cross_attn.W_k.weight.grad.abs().max()) == 0.0


@pytest.mark.parametrize("S,shortcut,materialize,skip", [(1, True, True, True), (1, False, False, False), (5, True, False, False),
                                                         (5, True, True, True), (5, True, True, False)])
def test_f4_fused_loss_path(S, shortcut, materialize, skip):
    """Projection + log-sum-exp + NLL without logits, and its backward, against the same golden vectors; with one
    memory row both the closed form of the cross-attention sub-layer and the general kernels are checked."""
    from capdec import autograd as AG
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"])).train()
    caps = t(fx["captions"]).cuda()
    enc = t(fx[f"enc_S{S}"]).cuda()
    x, tape = AG.tf_forward(dec, enc, caps[:, :-1].contiguous(), 0.0, 0, single_row_shortcut=shortcut)
    loss, saved = AG.vocab_loss(dec, x, caps[:, 1:].contiguous(), materialize=materialize, skip_ignored=skip)
    close(loss[0], fx[f"loss_S{S}"], rtol=1e-5, what="loss")
    grads = {}
    dx = AG.vocab_loss_bwd(dec, x, saved, grads)
    denc = AG.tf_backward(dec, tape, dx, grads)
    close(denc, fx[f"denc_S{S}"], what="denc")
    names = [n for n, _ in dec.named_parameters()]
    atol = _noise_floor(fx, "grad__{}@S" + str(S), names)
    for name in names:
        close(grads[name], fx[f"grad__{name}@S{S}"], atol=atol, what=name)


@pytest.mark.parametrize("M,frac_pad", [(70, 0.4), (1000, 0.3), (333, 1.0), (257, 0.0), (4096, 0.5)])
def test_vocab_loss_skips_ignored_rows(M, frac_pad):
    """Rows with an ignored target are compacted away on the device (no host sync): loss, d(loss)/dx and the fc2
    gradients against nn.CrossEntropyLoss(ignore_index=0) in fp64 (train.py:216,254-256) and against the dense path."""
    from capdec import autograd as AG
    g = torch.Generator().manual_seed(M)
    E, V = 64, 300
    fc2 = torch.nn.Linear(E, V)
    holder = type("Holder", (), {})()
    holder.fc2 = fc2.cuda()
    x = torch.randn(M, E, generator=g)
    tgt = torch.randint(1, V, (M,), generator=g)
    tgt[torch.rand(M, generator=g) < frac_pad] = 0
    if frac_pad >= 1.0:
        tgt[:] = 0
    xd = x.double().requires_grad_(True)
    Wd, bd = fc2.weight.detach().double().cpu().requires_grad_(True), fc2.bias.detach().double().cpu().requires_grad_(True)
    outs = {}
    for skip in (True, False):
        grads = {}
        loss, saved = AG.vocab_loss(holder, x.cuda(), tgt.cuda(), skip_ignored=skip)
        dx = AG.vocab_loss_bwd(holder, x.cuda(), saved, grads)
        outs[skip] = (loss, dx, grads["fc2.weight"], grads["fc2.bias"])
    if frac_pad >= 1.0:      # no counted target: the reference's mean over zero terms is NaN; every gradient is NaN or 0
        assert not torch.isfinite(outs[True][0]).any() and not torch.isfinite(outs[False][0]).any()
        return
    ref = torch.nn.functional.cross_entropy(xd @ Wd.t() + bd, tgt, ignore_index=0)
    ref.backward()
    for skip in (True, False):
        loss, dx, dw, db = outs[skip]
        close(loss[0], ref.detach(), rtol=1e-5, what=f"loss skip={skip}")
        close(dx, xd.grad, what=f"dx skip={skip}")
        close(dw, Wd.grad, what=f"dW skip={skip}")
        close(db, bd.grad, what=f"db skip={skip}")
    assert float(outs[True][1][tgt == 0].abs().max() if (tgt == 0).any() else 0.0) == 0.0


@pytest.mark.parametrize("V", [300, 5124, 10000, 10240, 10244])
def test_softmax_nll_gradient_in_place_every_row_form(V):
    """logits -> d(mean NLL)/d(logits) in place (train.py:216,254-256): the two-pass kernel (short and very long rows)
    and the one that keeps a row in registers (5120 < V <= 10240), with ignored targets, against fp64 autograd; then with
    a device-side row limit: rows up to the next multiple of 32 zeroed, the rest untouched."""
    from capdec import ops
    g = torch.Generator().manual_seed(V)
    M = 70
    logits = torch.randn(M, V, generator=g) * 3
    tgt = torch.randint(1, V, (M,), generator=g)
    tgt[::4] = 0
    tgt[1] = V - 1
    ld = logits.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ld, tgt, ignore_index=0)
    ref.backward()
    work = logits.clone().cuda()
    sums, gscale = ops.softmax_nll_grad_(work, tgt.cuda())
    close(work, ld.grad, atol=1e-9, what="d logits")
    close(sums[0] / sums[1], ref.detach(), rtol=1e-5, what="loss")
    assert int(sums[1]) == int((tgt != 0).sum())
    live = 37
    rows = torch.tensor([live], dtype=torch.int32, device="cuda")
    work = logits.clone().cuda()
    work[64:] = 7.0
    tl = tgt.clone()
    ops.softmax_nll_grad_(work, tl.cuda(), rows_dev=rows)
    ld2 = logits[:live].double().requires_grad_(True)
    torch.nn.functional.cross_entropy(ld2, tgt[:live], ignore_index=0, reduction="sum").backward()
    n_all = float((tgt != 0).sum())             # the count runs over every target handed in
    close(work[:live], ld2.grad / n_all, atol=1e-9, what="d logits, limited rows")
    assert float(work[live:64].abs().max()) == 0.0 and bool((work[64:] == 7.0).all())


def _run_stack(dec, enc, caps, ragged, shortcut=True, absorb=None, pair=None):
    from capdec import autograd as AG
    from capdec import ops
    ids, tgt = caps[:, :-1].contiguous(), caps[:, 1:].contiguous()
    rg = ops.RaggedRows(ids, tgt, 0) if ragged else None
    x, tape = AG.tf_forward(dec, enc, ids, 0.0, 0, single_row_shortcut=shortcut, ragged=rg, absorb_cross=absorb,
                            pair_cross=pair)
    assert absorb is None or tape.absorbed == absorb
    assert pair is None or tape.pair == pair
    if ragged:
        loss, saved = AG.vocab_loss(dec, x, rg.targets_c, rows=rg.rows)
    else:
        loss, saved = AG.vocab_loss(dec, x, tgt)
    grads = {}
    dx = AG.vocab_loss_bwd(dec, x, saved, grads)
    denc = AG.tf_backward(dec, tape, dx, grads)
    return loss, denc, grads, rg


@pytest.mark.parametrize("S,shortcut", [(1, True), (1, False), (2, True), (5, True)])
def test_f4_ragged_stack_against_golden(S, shortcut):
    """The decoder stack on the live rows only: loss and every gradient against the golden vectors (one memory row
    through the closed form and through the general cross-attention kernels; two and five memory rows)."""
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"])).train()
    caps = t(fx["captions"]).cuda()
    loss, denc, grads, rg = _run_stack(dec, t(fx[f"enc_S{S}"]).cuda(), caps, True, shortcut)
    close(loss[0], fx[f"loss_S{S}"], rtol=1e-5, what="loss")
    close(denc, fx[f"denc_S{S}"], what="denc")
    names = [n for n, _ in dec.named_parameters()]
    atol = _noise_floor(fx, "grad__{}@S" + str(S), names)
    for name in names:
        close(grads[name], fx[f"grad__{name}@S{S}"], atol=atol, what=name)


@pytest.mark.parametrize("B,T,S", [(37, 20, 1), (256, 12, 1), (5, 50, 1), (37, 20, 2), (64, 20, 64), (16, 20, 197)])
def test_ragged_stack_equals_full_rows(B, T, S):
    """Captions of every length, with <PAD> tails, one caption with a <PAD> inside (it keeps its full length: the
    reference lets that query row attend over all T keys), one empty caption and one that fills all T positions."""
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"])).train()
    V, E = dec.fc2.weight.shape[0], dec.fc2.weight.shape[1]
    g = torch.Generator().manual_seed(B * 100 + T)
    caps = torch.zeros(B, T + 1, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(0, T, (1,), generator=g))
        caps[b, 0] = 1
        caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
        caps[b, 1 + n] = 2
    caps[1, 1:] = 0                                   # nothing but <SOS>: one live row (target <PAD>... none counted)
    caps[2, 1:] = torch.randint(4, V, (T,), generator=g)   # no <EOS>, no <PAD>: all T rows live
    caps[3, 1:6] = torch.tensor([7, 0, 9, 11, 2])[:min(5, T)]     # <PAD> inside the caption
    caps[3, 6:] = 0
    caps = caps.cuda()
    enc = torch.randn(B, S, E, generator=g).cuda()
    loss_d, denc_d, grads_d, _ = _run_stack(dec, enc, caps, False)
    loss_r, denc_r, grads_r, rg = _run_stack(dec, enc, caps, True)
    counts = rg.counts.cpu()
    live = int(counts[1])
    assert int(counts[0]) == min(B * T, (live + 31) // 32 * 32) and live < B * T
    off = rg.seq_off.cpu()
    assert int(off[B]) == live and int(off[4] - off[3]) == T and int(off[3] - off[2]) == T
    close(loss_r[0], loss_d[0].cpu().numpy(), rtol=1e-6, what="loss")
    close(denc_r, denc_d.cpu().numpy(), what="denc")
    # (the key-bias gradients are zero in exact arithmetic -- a softmax ignores a shift of all its scores -- and
    #  rounding noise in fp32: an absolute floor tied to the largest gradient covers them)
    floor = 1e-7 * max(float(gd.abs().max()) for gd in grads_d.values())
    for name, gd in grads_d.items():
        close(grads_r[name], gd.detach().cpu().numpy(), atol=floor, what=name)


@pytest.mark.parametrize("ragged,pair,side", [(False, True, False), (True, True, False), (False, False, False),
                                              (True, False, False), (True, True, True), (False, True, True)])
def test_f4_two_memory_rows_against_golden(ragged, pair, side):
    """Two memory rows (the ViT+CNN / YOLO+CNN heads): the logistic closed form of the cross-attention and the general
    kernels, dense and on the live rows only, with and without the second stream: loss and every gradient against the
    reference's."""
    from capdec import autograd as AG
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"])).train()
    caps = t(fx["captions"]).cuda()
    if side:
        AG.use_side_stream(torch.cuda.Stream())
    try:
        loss, denc, grads, _ = _run_stack(dec, t(fx["enc_S2"]).cuda(), caps, ragged, pair=pair)
        AG.join_side_stream()
    finally:
        AG.use_side_stream(None)
    torch.cuda.synchronize()
    close(loss[0], fx["loss_S2"], rtol=1e-5, what="loss")
    close(denc, fx["denc_S2"], what="denc")
    names = [n for n, _ in dec.named_parameters()]
    atol = _noise_floor(fx, "grad__{}@S2", names)
    for name in names:
        close(grads[name], fx[f"grad__{name}@S2"], atol=atol, what=name)


@pytest.mark.parametrize("B,T,ragged", [(37, 20, True), (37, 20, False), (130, 5, True), (3, 50, False)])
def test_two_memory_rows_closed_form_equals_general_kernels(B, T, ragged):
    """The closed form against the general cross-attention kernels on captions of every length and memories whose two
    rows are far apart (probabilities near 0 and 1) or identical (exactly one half)."""
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"])).train()
    V, E = dec.fc2.weight.shape[0], dec.fc2.weight.shape[1]
    g = torch.Generator().manual_seed(B * 10 + T)
    caps = torch.zeros(B, T + 1, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(0, T, (1,), generator=g))
        caps[b, 0] = 1
        caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
        caps[b, 1 + n] = 2
    caps = caps.cuda()
    enc = torch.randn(B, 2, E, generator=g)
    enc[0] *= 6.0
    enc[1, 1] = enc[1, 0]
    enc = enc.cuda()
    loss_g, denc_g, grads_g, _ = _run_stack(dec, enc, caps, ragged, pair=False)
    loss_p, denc_p, grads_p, _ = _run_stack(dec, enc, caps, ragged, pair=True)
    close(loss_p[0], loss_g[0].cpu().numpy(), rtol=1e-6, what="loss")
    close(denc_p, denc_g.cpu().numpy(), what="denc")
    floor = 1e-6 * max(float(gd.abs().max()) for gd in grads_g.values())
    for name, gd in grads_g.items():
        close(grads_p[name], gd.detach().cpu().numpy(), atol=floor, what=name)


@pytest.mark.parametrize("S,ragged", [(2, False), (5, False), (2, True), (5, True)])
def test_f4_folded_cross_attention_against_golden(S, ragged):
    """The cross-attention with the key / value projections folded onto the query side (the form taken for long
    memories), forced here on the golden fixture's short memories: loss and every gradient against the reference's."""
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"])).train()
    caps = t(fx["captions"]).cuda()
    loss, denc, grads, _ = _run_stack(dec, t(fx[f"enc_S{S}"]).cuda(), caps, ragged, absorb=True)
    close(loss[0], fx[f"loss_S{S}"], rtol=1e-5, what="loss")
    close(denc, fx[f"denc_S{S}"], what="denc")
    names = [n for n, _ in dec.named_parameters()]
    atol = _noise_floor(fx, "grad__{}@S" + str(S), names)
    for name in names:
        close(grads[name], fx[f"grad__{name}@S{S}"], atol=atol, what=name)


@pytest.mark.parametrize("B,T,S,ragged", [(16, 20, 197, True), (16, 20, 197, False), (5, 7, 65, True), (3, 33, 300, True),
                                          (9, 4, 1000, False)])
def test_folded_cross_attention_equals_projected_form(B, T, S, ragged):
    """Long memories: the folded form against the form that projects the memory to keys and values (different order
    of the same sums), on captions of every length."""
    from capdec import autograd as AG
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"])).train()
    V, E = dec.fc2.weight.shape[0], dec.fc2.weight.shape[1]
    g = torch.Generator().manual_seed(B * 1000 + S)
    caps = torch.zeros(B, T + 1, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(0, T, (1,), generator=g))
        caps[b, 0] = 1
        caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
        caps[b, 1 + n] = 2
    caps[1, 1:] = 0
    caps[2, 1:] = torch.randint(4, V, (T,), generator=g)
    caps = caps.cuda()
    enc = torch.randn(B, S, E, generator=g).cuda()
    loss_p, denc_p, grads_p, _ = _run_stack(dec, enc, caps, ragged, absorb=False)
    loss_f, denc_f, grads_f, _ = _run_stack(dec, enc, caps, ragged, absorb=True)
    close(loss_f[0], loss_p[0].cpu().numpy(), rtol=1e-6, what="loss")
    close(denc_f, denc_p.cpu().numpy(), what="denc")
    floor = 1e-6 * max(float(gd.abs().max()) for gd in grads_p.values())
    for name, gd in grads_p.items():
        close(grads_f[name], gd.detach().cpu().numpy(), atol=floor, what=name)
    # the choice made without forcing: folded exactly when it saves at least half of the multiply-adds
    H = dec.num_heads
    assert AG.cross_absorbed(197, 20, 512, 4) and not AG.cross_absorbed(64, 20, 512, 4) and not AG.cross_absorbed(1, 20, 512, 4)
    assert AG.cross_absorbed(S, T, E, H) == (S > 64 and 2 * T * (H * S + E) <= S * E + T * S)


@pytest.mark.parametrize("B,H,Tq,Tk,E", [(3, 4, 20, 197, 512), (2, 4, 5, 65, 64), (2, 2, 64, 300, 128), (1, 4, 33, 1000, 256)])
def test_long_memory_attention_forward_chunked_vs_row_kernel(B, H, Tq, Tk, E, tuning_lib):
    """Unmasked attention over a long memory (transformer.py:13-22 with mask None): the chunked kernel (keys streamed
    through LDS, running softmax) against fp64 and against the one-wavefront-per-row kernel, output and log-sum-exp."""
    from capdec import _lib, ops
    lib = tuning_lib
    g = torch.Generator().manual_seed(Tk)
    q, k, v = (torch.randn(B * n, E, generator=g) for n in (Tq, Tk, Tk))
    dk = E // H
    qd, kd, vd = (x.double().view(B, -1, H, dk).transpose(1, 2) for x in (q, k, v))
    sc = qd @ kd.transpose(-1, -2) / dk ** 0.5
    ref = (torch.softmax(sc, -1) @ vd).transpose(1, 2).reshape(B * Tq, E)
    ref_lse = torch.logsumexp(sc, -1)
    outs = []
    try:
        for row_kernel in (0, 1):
            lib.capdec_attention_force_row_kernel(row_kernel)
            out, lse = ops.attention(q.cuda(), k.cuda(), v.cuda(), B, H, Tq, Tk, want_lse=True)
            close(out, ref, what=f"out row_kernel={row_kernel}")
            close(lse, ref_lse, rtol=1e-5, what=f"lse row_kernel={row_kernel}")
            outs.append(out)
    finally:
        lib.capdec_attention_force_row_kernel(0)
    close(outs[0], outs[1].cpu().numpy(), rtol=1e-5, what="chunked vs row kernel")


@pytest.mark.parametrize("B,T,seed", [(1, 1, 0), (7, 5, 1), (300, 20, 2), (2050, 3, 3), (64, 50, 4)])
def test_compaction_maps_against_a_python_restatement(B, T, seed):
    """capdec_compact_prefix / capdec_compact_targets: every output array against a plain Python loop, on random
    batches with empty captions, <PAD> inside a caption and captions that fill all T positions."""
    from capdec import ops
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 6, (B, T), generator=g)          # 0 = <PAD> is frequent: many inside-prefix cases
    tgt = torch.randint(0, 3, (B, T), generator=g)
    for b in range(0, B, 3):                                # a third of the samples get a clean <PAD> tail
        n = int(torch.randint(0, T + 1, (1,), generator=g))
        ids[b, :n] = torch.randint(1, 6, (n,), generator=g)
        tgt[b, :n] = torch.randint(1, 3, (n,), generator=g)
        tgt[b, n:] = 0
        ids[b, n:] = 0
    rr = ops.RaggedRows(ids.cuda(), tgt.cuda(), 0)
    M = B * T
    lens = []
    for b in range(B):
        nz = [p for p in range(T) if int(tgt[b, p]) != 0]
        n = nz[-1] + 1 if nz else 0
        if any(int(ids[b, p]) == 0 for p in range(n)):
            n = T
        lens.append(n)
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    live = off[-1]
    assert rr.counts.cpu().tolist() == [min(M, (live + 31) // 32 * 32), live]
    assert rr.seq_off.cpu().tolist() == off
    order, slot, sample = rr.order.cpu().tolist(), rr.slot.cpu().tolist(), rr.sample.cpu().tolist()
    ids_c, tgt_c = rr.ids_c.cpu().tolist(), rr.targets_c.cpu().tolist()
    for b in range(B):
        for p in range(T):
            r = b * T + p
            if p < lens[b]:
                s_ = off[b] + p
                assert slot[r] == s_ and order[s_] == r and sample[s_] == b
                assert ids_c[s_] == int(ids[b, p]) and tgt_c[s_] == int(tgt[b, p])
            else:
                assert slot[r] == M
    assert all(t_ == 0 for t_ in tgt_c[live:]) and all(0 <= o < M for o in order[live:])
    # loss-level compaction of the flat targets
    flat = tgt.reshape(-1)
    o2, s2, t2, c2 = ops.compact_targets(flat.cuda(), 0)
    keep = [r for r in range(M) if int(flat[r]) != 0]
    assert int(c2) == len(keep) and o2.cpu().tolist()[:len(keep)] == keep
    assert t2.cpu().tolist() == [int(flat[r]) for r in keep] + [0] * (M - len(keep))
    want_slot = [M] * M
    for k_, r in enumerate(keep):
        want_slot[r] = k_
    assert s2.cpu().tolist() == want_slot


@pytest.mark.parametrize("B,T", [(1, 1), (9, 7), (64, 20)])
def test_compaction_reads_the_two_views_of_a_caption_matrix_in_place(B, T):
    """capdec_compact_prefix_strided on captions[:, :-1] / captions[:, 1:] (row stride T + 1, train.py:253-256) gives
    every array of the packed call."""
    from capdec import ops
    g = torch.Generator().manual_seed(B * 31 + T)
    caps = torch.randint(0, 5, (B, T + 1), generator=g)
    for b in range(0, B, 2):
        n = int(torch.randint(0, T + 2, (1,), generator=g))
        caps[b, n:] = 0
    caps = caps.cuda()
    a = ops.RaggedRows(caps[:, :-1], caps[:, 1:], 0)
    b_ = ops.RaggedRows(caps[:, :-1].contiguous(), caps[:, 1:].contiguous(), 0)
    live = int(b_.counts[1])
    for name in ("order", "slot", "seq_off", "sample", "ids_c", "targets_c", "counts"):
        assert torch.equal(getattr(a, name), getattr(b_, name)), name
    assert live <= B * T


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("ragged", [False, True])
def test_embedding_position_dropout_in_one_launch(p, ragged):
    """capdec_embed_train_fwd / _bwd against the launches they replace (embedding + position, move to the slots,
    dropout; dropout, scatter-add): bit for bit forward, and backward up to the order of the float atomics."""
    from capdec import ops
    B, T, E, V = 37, 11, 64, 50
    g = torch.Generator().manual_seed(5)
    caps = torch.randint(0, V, (B, T + 1), generator=g).cuda()
    emb = torch.randn(V, E, generator=g).cuda()
    pe = torch.randn(T + 3, E, generator=g).cuda()
    seed = 0x1234567
    seed_dev = torch.tensor([99], dtype=torch.int64, device="cuda")
    ids = caps[:, :-1]
    order = torch.randperm(B * T, generator=g).cuda() if ragged else None
    x, ids_out = ops.embed_train(ids, emb, pe, p, seed, seed_dev=seed_dev, order=order, want_ids=True)
    want = ops.embed_pe(ids.contiguous(), emb, pe)
    flat_ids = ids.contiguous().view(-1)
    if ragged:
        want = want.index_select(0, order)
        flat_ids = flat_ids.index_select(0, order)
    if p > 0:
        want = ops.dropout(want, p, seed, seed_dev=seed_dev)
        assert 0.03 < float((x == 0).float().mean()) < 0.2
    assert torch.equal(x, want) and torch.equal(ids_out, flat_ids)
    # backward
    dx = torch.randn(B * T, E, generator=g).cuda()
    rows = torch.tensor([B * T - 5], dtype=torch.int32, device="cuda")
    got = ops.embed_bwd(ids_out, dx, torch.zeros(V, E, device="cuda"), rows_dev=rows, p=p, seed=seed, seed_dev=seed_dev)
    dxm = ops.dropout(dx, p, seed, seed_dev=seed_dev) if p > 0 else dx
    ref = torch.zeros(V, E, dtype=torch.float64, device="cuda")
    ref.index_add_(0, ids_out[:B * T - 5], dxm[:B * T - 5].double())
    close(got, ref.cpu().numpy(), rtol=1e-6, what="embedding gradient")


def test_f6_cnn_attn_train_mode_batchnorm_and_head_gradients():
    from models.cnnattn_model import CNNAttentionModel
    fx = load_golden("f6_heads")
    m = CNNAttentionModel(64, 50, int(fx["H"]), 2, dropout=0.0)
    m.load_state_dict(sub_sd(fx, "cnn__sd__"))
    m = m.cuda().train()
    feats, caps = t(fx["cnn_feats"]).cuda(), t(fx["captions"]).cuda()
    logits = m(feats, caps[:, :-1])
    close(logits, fx["cnn_logits_train"], what="logits")
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 50), caps[:, 1:].reshape(-1), ignore_index=0)
    close(loss, fx["cnn_loss_train"], rtol=1e-5)
    loss.backward()
    n = _check_grads(m.named_parameters(), fx, "cnn_grad__{}")
    assert n == len(list(m.parameters()))
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        close(getattr(m.batchnorm, k).float(), fx["cnn_after__batchnorm." + k].astype(np.float32), rtol=1e-5, what=k)


def test_f6_pair_head_train_mode():
    from models.vitcnnattn_model import VITCNNAttentionModel
    fx = load_golden("f6_heads")
    sd = sub_sd(fx, "pair__sd__")
    split = int(fx["pair_split"])
    m = VITCNNAttentionModel(64, 50, int(fx["H"]), 1, dropout=0.0)
    m.vit_out_size = split
    m.fc_vit = torch.nn.Linear(split, 64)
    m.fc_cnn = torch.nn.Linear(sd["fc_cnn.weight"].shape[1], 64)
    m.load_state_dict(sd)
    m = m.cuda().train()
    feats, caps = t(fx["pair_feats"]).cuda(), t(fx["captions"]).cuda()
    logits = m(feats, caps[:, :-1], "precomputed")
    close(logits, fx["pair_logits_train"], what="logits")
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 50), caps[:, 1:].reshape(-1), ignore_index=0)
    close(loss, fx["pair_loss_train"], rtol=1e-5)
    loss.backward()
    n = _check_grads(m.named_parameters(), fx, "pair_grad__{}")
    assert n == len(list(m.parameters()))


def test_dropout_statistics_and_mask_replay():
    from capdec import ops
    x = torch.ones(1 << 20, device="cuda")
    y = ops.dropout(x, 0.1, 12345)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.9) < 2e-3
    assert torch.all((y == 0) | ((y - 1 / 0.9).abs() < 1e-6))
    assert torch.equal(y, ops.dropout(x, 0.1, 12345))            # same seed -> same mask (backward replays it)
    assert not torch.equal(y, ops.dropout(x, 0.1, 12346))
    # LayerNorm with fused dropout: s = add + drop(x) uses the same generator
    g, b = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
    xx = torch.randn(64, 512, device="cuda")
    add = torch.randn(64, 512, device="cuda")
    _, s, _, _ = ops.add_layernorm_train(xx, add, g, b, 0.1, 777)
    close(s, add + ops.dropout(xx, 0.1, 777), rtol=1e-6)
    dy = torch.randn(64, 512, device="cuda")
    dg, db = torch.zeros(512, device="cuda"), torch.zeros(512, device="cuda")
    _, _, mean, rstd = ops.add_layernorm_train(xx, add, g, b, 0.1, 777)
    d_add, d_x = ops.layernorm_bwd(dy, s, mean, rstd, g, 0.1, 777, dg, db)
    close(d_x, ops.dropout(d_add, 0.1, 777), rtol=1e-6)


@pytest.mark.parametrize("M,E", [(1, 64), (37, 256), (1000, 512), (10240, 512), (700, 1024)])
def test_layernorm_backward_two_stage_column_sums(M, E):
    """The LayerNorm backward kernel hands dgamma / dbeta / the bias-gradient column sums to a second launch through a
    per-workgroup slab: against fp64 arithmetic (autograd of nn.LayerNorm, transformer.py:85,87,89), against the form
    with float atomics, accumulating into what the outputs already hold, bit-identical from run to run, and with a
    device-side row count."""
    from capdec import ops
    g = torch.Generator().manual_seed(M + E)
    s = torch.randn(M, E, generator=g).cuda()
    dy = torch.randn(M, E, generator=g).cuda()
    gam = (1 + 0.1 * torch.randn(E, generator=g)).cuda()
    zero = torch.zeros(E, device="cuda")
    _, _, mean, rstd = ops.add_layernorm_train(s, torch.zeros_like(s), gam, zero, 0.0, 1)
    base = torch.randn(3, E, generator=g).cuda()

    def run(staged, rows_dev=None):
        out = base.clone()
        d_add, d_x = ops.layernorm_bwd(dy, s, mean, rstd, gam, 0.0, 1, out[0], out[1], dxsum=out[2], staged=staged,
                                       rows_dev=rows_dev)
        return d_add, d_x, out

    d_add, d_x, out = run(True)
    d_add2, _, out2 = run(False)
    assert torch.equal(d_add, d_add2) and torch.equal(d_add, d_x)
    s64, dy64, g64 = s.double().cpu(), dy.double().cpu(), gam.double().cpu()
    mu = s64.mean(1, keepdim=True)
    rs = 1.0 / torch.sqrt(s64.var(1, unbiased=False, keepdim=True) + 1e-5)
    xh = (s64 - mu) * rs
    gg = dy64 * g64
    want = rs * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True))
    close(d_add, want, rtol=2e-5, what="d_add")
    cols = torch.stack([(dy64 * xh).sum(0), dy64.sum(0), want.sum(0)]) + base.double().cpu()
    for k, name in enumerate(("dgamma", "dbeta", "dxsum")):
        close(out[k], cols[k], rtol=2e-5, what=name + " (slab)")
        close(out2[k], cols[k], rtol=2e-5, what=name + " (atomics)")
    assert torch.equal(run(True)[2], out)
    if M >= 37:
        live = M - M // 3
        rows_dev = torch.tensor([live], dtype=torch.int32, device="cuda")
        _, _, out3 = run(True, rows_dev)
        cols3 = torch.stack([(dy64 * xh)[:live].sum(0), dy64[:live].sum(0), want[:live].sum(0)]) + base.double().cpu()
        for k in range(3):
            close(out3[k], cols3[k], rtol=2e-5, what=f"column sums {k} over the live rows")


def test_train_mode_with_dropout_runs_and_is_seeded():
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"]), dropout=0.1).train()
    caps = t(fx["captions"]).cuda()
    enc = t(fx["enc_S2"]).cuda()
    torch.manual_seed(3)
    a = dec.decoder_forward(enc, caps[:, :-1])
    torch.manual_seed(3)
    b = dec.decoder_forward(enc, caps[:, :-1])
    c = dec.decoder_forward(enc, caps[:, :-1])
    assert torch.equal(a, b) and not torch.equal(a, c)
    a.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in dec.parameters())


def test_train_state_advance_and_adam_from_device_state():
    """The device-resident per-step scalars: step count, bias corrections, learning rate and a fresh seed per step; Adam
    reading them gives torch.optim.Adam's parameters."""
    from capdec import ops
    st = ops.TrainState(torch.device("cuda"), 1e-3, seed=12345, step=0)
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(5000, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    seeds = [st.read()[0]]
    for step in range(1, 6):
        if step == 4:                      # StepLR changed the rate: the host writes it, the kernel reads it
            st.set(lr=5e-4)
            opt.param_groups[0]["lr"] = 5e-4
        st.advance(0.9, 0.999)
        seed, t_dev, lr, b1, b2s = st.read()
        assert t_dev == step and 0 <= seed < 2 ** 62 and seed not in seeds
        seeds.append(seed)
        f1, f2 = float(np.float32(0.9)), float(np.float32(0.999))      # the betas cross the C ABI as floats
        assert abs(b1 - (1 - f1 ** step)) <= 1e-7 and abs(b2s - (1 - f2 ** step) ** 0.5) <= 1e-7
        assert abs(lr - (5e-4 if step >= 4 else 1e-3)) <= 1e-10
        grad = torch.randn(5000, generator=g)
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step_dev(p, grad.cuda(), m, v, st)
    close(p, ref.detach(), rtol=1e-6)
    # the device seed is added to the by-value one: same sum, same mask
    x = torch.ones(1 << 16, device="cuda")
    assert torch.equal(ops.dropout(x, 0.3, 1000 + seeds[-1]), ops.dropout(x, 0.3, 1000, seed_dev=st.seed_dev))
    assert not torch.equal(ops.dropout(x, 0.3, 1000), ops.dropout(x, 0.3, 1000, seed_dev=st.seed_dev))


def test_graph_trainer_draws_new_dropout_masks_and_tracks_the_launched_trainer():
    """Dropout on (p = 0.1), six steps on changing batches: a trainer whose step is captured in a HIP graph and one that
    launches kernel by kernel start from the same weights and the same seed; they must see the same masks at every
    step (the seed lives on the device and advances inside the graph), so losses agree to rounding, and the losses of
    one batch fed twice must differ (new masks at every replay)."""
    from capdec.trainer import CaptionTrainer
    from models.vitattn_model import VITAttentionModel
    E, V, H, L, B, T = 64, 150, 4, 2, 24, 11
    torch.manual_seed(21)
    a = VITAttentionModel(E, V, H, L, dropout=0.1).cuda()
    b = VITAttentionModel(E, V, H, L, dropout=0.1).cuda()
    b.load_state_dict(a.state_dict())
    torch.manual_seed(22)
    ta = CaptionTrainer(a, lr=1e-3, graph=False)
    torch.manual_seed(22)
    tb = CaptionTrainer(b, lr=1e-3, graph=True)
    assert ta.state.read()[0] == tb.state.read()[0]
    g = torch.Generator().manual_seed(23)
    la, lb = [], []
    for step in range(6):
        if step in (0, 1, 2):       # step 1 and 2 reuse batch 0: a replay with the same inputs
            if step == 0:
                feats = torch.rand(B, 1000, generator=g).cuda()
                caps = torch.randint(4, V, (B, T + 1), generator=g)
                caps[:, 0] = 1
                for r in range(B):
                    n = int(torch.randint(3, T, (1,), generator=g))
                    caps[r, n] = 2
                    caps[r, n + 1:] = 0
                caps = caps.cuda()
        else:
            feats = torch.rand(B, 1000, generator=g).cuda()
        if step == 4:
            ta.scheduler_step(4)    # epoch 4 done: StepLR(5, 0.8) lowers the rate, on the host and in device memory
            tb.scheduler_step(4)
        la.append(float(ta.step(feats, caps).item()))
        lb.append(float(tb.step(feats, caps).item()))
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-5 * abs(x), (la, lb)
    assert abs(lb[1] - lb[2]) > 1e-6 * abs(lb[1])           # two replays on the same batch: different masks, and
    assert tb.t == 6 and tb.state.read()[1] == 6             # different weights
    assert abs(tb.lr - 8e-4) < 1e-12 and abs(tb.state.read()[2] - 8e-4) < 1e-9
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        if va.is_floating_point():
            assert float((va - vb).abs().max()) <= 2.2 * 1e-3 * 6 * 1e-2 + 1e-6, k
    assert int(b.state_dict()["batchnorm.num_batches_tracked"]) == 6


def test_scheduler_step_is_steplr_5_08():
    """StepLR(step_size=5, gamma=0.8) stepped once per epoch (reference train.py:218,366), against torch's own."""
    from capdec.trainer import CaptionTrainer
    from models.cnnattn_model import CNNAttentionModel
    m = CNNAttentionModel(32, 40, 4, 1).cuda()
    tr = CaptionTrainer(m, lr=1e-3)
    w = torch.zeros(1, requires_grad=True)
    opt = torch.optim.Adam([w], lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.8)
    for epoch in range(23):
        opt.step()
        sched.step()
        got = tr.scheduler_step(epoch)
        assert abs(got - sched.get_last_lr()[0]) <= 1e-12, epoch
    assert abs(tr.state.read()[2] - sched.get_last_lr()[0]) <= 1e-9


def test_adam_step_matches_torch_formula():
    from capdec import ops
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(10000, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        grad = torch.randn(10000, generator=g)
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step(p, grad.cuda(), m, v, 1e-3, step)
    close(p, ref.detach(), rtol=1e-6)


@pytest.mark.parametrize("graph", [False, True])
def test_trainer_steps_match_oracle_training_loop(graph):
    """CaptionTrainer (flat buffers, fused loss, HIP Adam) against the oracle forward + torch autograd + torch.optim.Adam
    on the CPU, 4 steps of the reference's inner loop (train.py:252-262), dropout off; launch by launch and with the
    step captured in a HIP graph (first step launched, second captured and replayed, third and fourth replayed)."""
    from capdec.trainer import CaptionTrainer
    from models.vitcnnattn_model import VITCNNAttentionModel
    from oracle import decoder_ref as R
    torch.manual_seed(5)
    E, V, H, L, B, T = 64, 120, 4, 2, 12, 9
    m = VITCNNAttentionModel(E, V, H, L, dropout=0.0)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    feats = torch.rand(B, 3048, generator=g)
    caps = torch.zeros(B, T + 1, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(2, T - 1, (1,), generator=g))
        caps[b, 0] = 1
        caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
        caps[b, 1 + n] = 2
    # reference loop on the CPU
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and k != "positional_encoding.pe")
          for k, v in sd0.items()}
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=1e-3)
    ref_losses = []
    for _ in range(4):
        enc, new = R.head_pair(sd, feats, "fc_vit", 1000, training=True)
        loss = R.caption_loss(R.decoder_forward(sd, enc, caps[:, :-1], H), caps[:, 1:])
        opt.zero_grad()
        loss.backward()
        opt.step()
        for k, v in new.items():
            sd[k] = v
        ref_losses.append(float(loss.detach()))
    # HIP trainer
    m = m.cuda()
    tr = CaptionTrainer(m, lr=1e-3, graph=graph)
    losses = [tr.step(feats.cuda(), caps.cuda()) for _ in range(4)]
    losses = [float(l.item()) for l in losses]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-4 * abs(b), (losses, ref_losses)
    assert ref_losses[-1] < ref_losses[0]
    got = m.state_dict()
    # Parameters after Adam: m / (sqrt(v) + eps) maps every gradient element to a step of about +-lr whatever its
    # magnitude, so elements whose gradient is at the rounding-noise level (e.g. the key biases, whose gradient is
    # mathematically zero) may move by up to lr per step in either direction on both sides.  Hence: the bulk must
    # agree tightly (mean |diff| <= 2e-6), no element may differ by more than the total possible movement.
    lr, steps = 1e-3, 4
    for k, v in sd.items():
        if v.is_floating_point():
            d = (got[k].detach().cpu().double() - v.detach().double()).abs()
            assert float(d.max()) <= 2.2 * lr * steps, (k, float(d.max()))
            # zero-gradient parameters: key biases (softmax shift invariance) and the biases of the projections that
            # feed the training-mode BatchNorm (a constant shift is removed by the batch mean)
            # (the running mean follows those projection biases)
            if not k.endswith(("W_k.bias", "fc_vit.bias", "fc_cnn.bias", "batchnorm.running_mean")):
                assert float(d.mean()) <= 2e-6, (k, float(d.mean()))
    assert int(got["batchnorm.num_batches_tracked"]) == 4


@pytest.mark.parametrize("ragged", [False, True])
def test_stacked_self_attention_projections_against_golden(ragged):
    """With the three self-attention projection weights (and biases, and their gradients) laid out back to back, as the
    trainer does, the backward pass runs them as one stacked layer: every gradient against the reference's."""
    from capdec import autograd as AG
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"])).train()
    E = dec.fc2.weight.shape[1]
    grads = {}
    with torch.no_grad():
        for l, ly in enumerate(dec.decoder_layers):
            lins = (ly.self_attn.W_q, ly.self_attn.W_k, ly.self_attn.W_v)
            pw, pb = torch.empty(3 * E * E, device="cuda"), torch.empty(3 * E, device="cuda")
            gw, gb = torch.zeros(3 * E * E, device="cuda"), torch.zeros(3 * E, device="cuda")
            for i, (ln, nm) in enumerate(zip(lins, ("W_q", "W_k", "W_v"))):
                w, b = pw[i * E * E:(i + 1) * E * E].view(E, E), pb[i * E:(i + 1) * E]
                w.copy_(ln.weight)
                b.copy_(ln.bias)
                ln.weight.data, ln.bias.data = w, b
                grads[f"decoder_layers.{l}.self_attn.{nm}.weight"] = gw[i * E * E:(i + 1) * E * E].view(E, E)
                grads[f"decoder_layers.{l}.self_attn.{nm}.bias"] = gb[i * E:(i + 1) * E]
            assert AG._stacked_projections(grads, f"decoder_layers.{l}.self_attn.", lins) is not None
    caps = t(fx["captions"]).cuda()
    ids, tgt = caps[:, :-1].contiguous(), caps[:, 1:].contiguous()
    from capdec import ops
    rg = ops.RaggedRows(ids, tgt, 0) if ragged else None
    x, tape = AG.tf_forward(dec, t(fx["enc_S5"]).cuda(), ids, 0.0, 0, ragged=rg)
    if ragged:
        loss, saved = AG.vocab_loss(dec, x, rg.targets_c, rows=rg.rows)
    else:
        loss, saved = AG.vocab_loss(dec, x, tgt)
    dx = AG.vocab_loss_bwd(dec, x, saved, grads)
    denc = AG.tf_backward(dec, tape, dx, grads)
    close(loss[0], fx["loss_S5"], rtol=1e-5, what="loss")
    close(denc, fx["denc_S5"], what="denc")
    names = [n for n, _ in dec.named_parameters()]
    atol = _noise_floor(fx, "grad__{}@S5", names)
    for name in names:
        close(grads[name], fx[f"grad__{name}@S5"], atol=atol, what=name)


def _rnn_from_fixture(fx):
    from models.cnnrnn_model import CNNtoRNN
    sd = sub_sd(fx, "sd__")
    Hd, E = sd["lstm.weight_hh_l0"].shape[1], sd["embed.weight"].shape[1]
    m = CNNtoRNN(E, Hd, sd["embed.weight"].shape[0])
    m.decoderRNN.dropout.p = 0.0
    m.decoderRNN.load_state_dict(sd)
    return m.cuda()


def test_f7_cnn_rnn_train_logits_loss_and_all_gradients():
    """DecoderRNN.forward in training mode (reference cnnrnn_model.py:20-34) + CrossEntropyLoss + backward through
    the LSTM, the embedding and the BatchNorm head, against the reference's autograd."""
    fx = load_golden("f7_decoder_rnn")
    m = _rnn_from_fixture(fx).train()
    feats, caps = t(fx["feats"]).cuda(), t(fx["captions"]).cuda()
    logits = m(feats, caps[:, :-1])
    close(logits, fx["logits_train"], what="logits")
    V = logits.shape[-1]
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, V), caps[:, 1:].reshape(-1), ignore_index=0)
    close(loss, fx["loss_train"], rtol=1e-5, what="loss")
    loss.backward()
    n = _check_grads(m.decoderRNN.named_parameters(), fx, "grad__{}")
    assert n == len(list(m.parameters()))


@pytest.mark.parametrize("graph", [False, True])
def test_cnn_rnn_trainer_matches_oracle_training_loop(graph):
    """CaptionTrainer on CNNtoRNN (fused loss on time-major rows, flat buffers, HIP Adam) against the oracle forward +
    torch autograd + torch.optim.Adam on the CPU; launch by launch and captured in a HIP graph."""
    from capdec.trainer import CaptionTrainer
    from oracle import decoder_ref as R
    fx = load_golden("f7_decoder_rnn")
    m = _rnn_from_fixture(fx)
    feats, caps = t(fx["feats"]), t(fx["captions"])
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k)
          for k, v in sub_sd(fx, "sd__").items()}
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=1e-3)
    ref_losses = []
    for _ in range(3):
        logits, new = R.decoder_rnn_forward(sd, feats, caps[:, :-1], training=True, with_buffers=True)
        loss = R.caption_loss(logits, caps[:, 1:])
        opt.zero_grad()
        loss.backward()
        opt.step()
        for k, v in new.items():
            sd[k] = v
        ref_losses.append(float(loss.detach()))
    tr = CaptionTrainer(m, lr=1e-3, graph=graph)
    losses = [float(tr.step(feats.cuda(), caps.cuda()).item()) for _ in range(3)]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-4 * abs(b), (losses, ref_losses)
    got = m.decoderRNN.state_dict()
    for k, v in sd.items():
        if v.is_floating_point():
            d = (got[k].detach().cpu().double() - v.detach().double()).abs()
            assert float(d.max()) <= 2.2 * 1e-3 * 3, (k, float(d.max()))
            if k != "batchnorm.running_mean":
                assert float(d.mean()) <= 3e-6, (k, float(d.mean()))


@pytest.mark.parametrize("B,H,Tq,Tk,dk,causal", [(5, 4, 20, 20, 128, True), (3, 4, 32, 32, 128, False), (3, 2, 7, 5, 16, False), (2, 4, 50, 50, 128, True),
                                                (4, 4, 20, 2, 64, False), (3, 4, 20, 197, 128, False), (2, 2, 9, 70, 32, False)])
def test_attention_backward_one_pass_matches_two_pass(B, H, Tq, Tk, dk, causal, tuning_lib):
    """The LDS-resident backward for short sequences against the general two-pass kernels (both are also pinned by
    the golden gradients through the decoder tests, which use short sequences)."""
    from capdec import _lib, ops
    g = torch.Generator().manual_seed(B * 100 + Tq)
    E = H * dk
    q, do = torch.randn(B * Tq, E, generator=g).cuda(), torch.randn(B * Tq, E, generator=g).cuda()
    k, v = torch.randn(B * Tk, E, generator=g).cuda(), torch.randn(B * Tk, E, generator=g).cuda()
    ids32 = None
    if causal:
        ids = torch.randint(1, 9, (B, Tq), generator=g)
        ids[0, Tq // 2] = 0
        ids[-1, -1] = 0
        ids32 = ids.to(torch.int32).cuda()
    o, lse = ops.attention(q, k, v, B, H, Tq, Tk, causal=causal, ids32=ids32, want_lse=True)
    lib = tuning_lib
    one = ops.attention_bwd(q, k, v, do, lse, B, H, Tq, Tk, causal=causal, ids32=ids32, out=o)
    lib.capdec_attention_bwd_force_two_pass(1)
    try:
        two = ops.attention_bwd(q, k, v, do, lse, B, H, Tq, Tk, causal=causal, ids32=ids32)
    finally:
        lib.capdec_attention_bwd_force_two_pass(0)
    for a, b, name in zip(one, two, ("dq", "dk", "dv")):
        close(a, b.cpu().numpy(), rtol=1e-5, what=name)


@pytest.mark.parametrize("B,T,H", [(1, 1, 1), (9, 20, 4), (33, 32, 2), (64, 7, 4)])
def test_ragged_self_attention_on_the_matrix_cores(B, T, H, tuning_lib):
    """Head width 128, at most 32 positions (the caption decoder's own shape): the MFMA forms of the ragged self-attention
    forward and backward against the row-by-row one-pass kernel and against fp64 arithmetic (transformer.py:18-27 differentiated by
    hand: causal rows, <PAD> query rows uniform over the sample's keys), on ragged batches with empty samples."""
    from capdec import ops
    dk = 128
    E = H * dk
    g = torch.Generator().manual_seed(B * 7 + T)
    ids = torch.randint(1, 6, (B, T), generator=g)
    tgt = torch.randint(1, 3, (B, T), generator=g)
    for b in range(B):
        n = int(torch.randint(0, T + 1, (1,), generator=g))
        tgt[b, n:] = 0
        ids[b, n:] = 0
        if n > 2 and b % 3 == 0:
            ids[b, n // 2] = 0                           # a <PAD> input inside the prefix: the sample keeps all T rows
    rr = ops.RaggedRows(ids.cuda(), tgt.cuda(), 0)
    M = B * T
    q, k, v, do = (torch.randn(M, E, generator=g).cuda() for _ in range(4))
    o, lse = ops.attention_ragged(q, k, v, rr, H, want_lse=True)
    got = ops.attention_bwd_ragged(q, k, v, do, lse, rr, H)
    tuning_lib.capdec_attention_bwd_force_row_form(1)
    tuning_lib.capdec_attention_ragged_force_row_form(1)
    try:
        o_rows, lse_rows = ops.attention_ragged(q, k, v, rr, H, want_lse=True)
        rows = ops.attention_bwd_ragged(q, k, v, do, lse, rr, H)
    finally:
        tuning_lib.capdec_attention_bwd_force_row_form(0)
        tuning_lib.capdec_attention_ragged_force_row_form(0)
    live = int(rr.counts[1])
    up = int(rr.counts[0])
    # forward: output rows (zero between the last sample and the row limit) and the log-sum-exp of the live rows
    close(o[:up], o_rows[:up].cpu().numpy(), rtol=2e-6, what="forward output vs the row form")
    assert up == live or float(o[live:up].abs().max()) == 0.0
    lens = (rr.seq_off[1:] - rr.seq_off[:-1]).cpu()
    live_mask = (torch.arange(T)[None, None, :] < lens[:, None, None]).expand(B, H, T)
    close(lse.cpu()[live_mask], lse_rows.cpu()[live_mask].numpy(), rtol=2e-6, what="log-sum-exp vs the row form")
    for a_, b_, name in zip(got, rows, ("dq", "dk", "dv")):
        close(a_[:up], b_[:up].cpu().numpy(), rtol=2e-6, what=name + " vs the row form")
        assert float(a_[live:up].abs().max()) == 0.0 if up > live else True
    # fp64, sample by sample
    off = rr.seq_off.cpu().tolist()
    idc = rr.ids_c.cpu()
    want = [torch.zeros(M, E, dtype=torch.float64) for _ in range(3)]
    want_o = torch.zeros(M, E, dtype=torch.float64)
    q64, k64, v64, do64 = (x.double().cpu() for x in (q, k, v, do))
    for b in range(B):
        s0, s1 = off[b], off[b + 1]
        n = s1 - s0
        if n == 0:
            continue
        for h in range(H):
            cs = slice(h * dk, (h + 1) * dk)
            qq = q64[s0:s1, cs].clone().requires_grad_(True)
            kk = k64[s0:s1, cs].clone().requires_grad_(True)
            vv = v64[s0:s1, cs].clone().requires_grad_(True)
            sc = qq @ kk.t() / dk ** 0.5
            mask = torch.tril(torch.ones(n, n, dtype=torch.bool))
            mask[idc[s0:s1] == 0, :] = False
            sc = sc.masked_fill(~mask, -1e9)
            out = torch.softmax(sc, dim=-1) @ vv
            out.backward(do64[s0:s1, cs])
            want[0][s0:s1, cs], want[1][s0:s1, cs], want[2][s0:s1, cs] = qq.grad, kk.grad, vv.grad
            want_o[s0:s1, cs] = out.detach()
    for a_, w_, name in zip(got, want, ("dq", "dk", "dv")):
        close(a_[:live], w_[:live].numpy(), rtol=1e-5, what=name + " vs fp64")
    close(o[:live], want_o[:live].numpy(), rtol=1e-5, what="forward output vs fp64")


def test_the_trainers_second_stream_runs_beside_the_current_one():
    """HIP multiplexes streams onto a few hardware queues and two streams on one queue run in order; the trainer picks
    the stream of its weight gradients by trying: a spin kernel on each of the two streams must take the time of one."""
    from capdec import trainer
    dev = torch.device("cuda:0")
    held = [torch.cuda.Stream(device=dev) for _ in range(3)]           # other users of torch's stream pool
    side = trainer._second_stream(dev)
    main = torch.cuda.current_stream(dev)
    assert side.cuda_stream != main.cuda_stream

    def spin_ms(cand):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        a.record(main)
        torch.cuda._sleep(1_000_000)
        if cand is not None:
            with torch.cuda.stream(cand):
                torch.cuda._sleep(1_000_000)
            main.wait_stream(cand)
        b.record(main)
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b)

    spin_ms(None)
    alone = min(spin_ms(None) for _ in range(3))
    both = min(spin_ms(side) for _ in range(3))
    assert both < 1.5 * alone, (alone, both)
    del held


def test_checkpoint_roundtrip_resumes_training_exactly(tmp_path):
    """Save after two steps in the reference's checkpoint schema, load into a fresh model + trainer, take one more
    step on both: weights, Adam moments and step counter must carry over (dropout off, so steps are deterministic
    up to the float atomics in the weight gradients)."""
    from capdec.checkpoint import load_reference_checkpoint, save_reference_checkpoint
    from capdec.trainer import CaptionTrainer
    from models.vitattn_model import VITAttentionModel
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(12)
    feats = torch.rand(16, 1000, generator=g).cuda()
    caps = torch.randint(4, 90, (16, 9), generator=g)
    caps[:, 0] = 1
    caps[:, -1] = 2
    caps = caps.cuda()
    a = VITAttentionModel(64, 90, 4, 2, dropout=0.0).cuda()
    ta = CaptionTrainer(a, lr=2e-3)
    for _ in range(2):
        ta.step(feats, caps)
    path = str(tmp_path / "ck.pth.tar")
    save_reference_checkpoint(path, a, ta, step=2)
    b = VITAttentionModel(64, 90, 4, 2, dropout=0.0).cuda()
    tb = CaptionTrainer(b, lr=1e-3)
    assert load_reference_checkpoint(path, b, tb) == 2
    assert tb.t == 2 and abs(tb.lr - 2e-3) < 1e-12
    la, lb = ta.step(feats, caps), tb.step(feats, caps)
    close(lb, la.detach().cpu().numpy(), rtol=1e-5, what="loss after resume")
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        if va.is_floating_point():
            assert float((va - vb).abs().max()) <= 2.2 * 2e-3 * 1e-2 + 1e-6, k      # same step up to atomics' rounding


def test_feature_store_batches_on_gpu(tmp_path):
    from capdec.feature_store import FeatureStore
    g = torch.Generator().manual_seed(5)
    feats = torch.rand(1000, 128, generator=g)
    st = FeatureStore.from_arrays(str(tmp_path / "fs"), list(range(1000)), feats.numpy())
    order = torch.randperm(1000, generator=g).numpy()
    n = 0
    for r, x in st.batches(192, "cuda", order=order):
        assert x.is_cuda and torch.equal(x.cpu(), feats[torch.from_numpy(r)])
        n += len(r)
    assert n == 1000


def test_out_of_range_target_gives_nan_loss_and_no_stray_read():
    """torch's CrossEntropyLoss raises on a target outside [0, V); these kernels cannot tell the host, so the loss of
    that row -- and with it the batch loss -- is NaN, and nothing outside the logits row / the weight matrix is read."""
    from capdec import ops
    g = torch.Generator().manual_seed(4)
    for V in (300, 7000, 10000, 12000):                 # two-pass, row-in-registers and streaming kernels
        M = 40
        tg = torch.randint(1, V, (M,), generator=g)
        tg[3] = 0
        good = torch.randn(M, V, generator=g)
        sums, _ = ops.softmax_nll_grad_(good.clone().cuda(), tg.cuda(), 0)
        assert bool(torch.isfinite(sums).all())
        for bad in (V, V + 12345, -7):
            t2 = tg.clone()
            t2[5] = bad
            sums, _ = ops.softmax_nll_grad_(good.clone().cuda(), t2.cuda(), 0)
            assert bool(torch.isnan(sums[0])), (V, bad)
    x, W, b = torch.randn(16, 64, generator=g).cuda(), torch.randn(500, 64, generator=g).cuda(), torch.zeros(500).cuda()
    t3 = torch.randint(1, 500, (16,), generator=g)
    assert bool(torch.isfinite(ops.vocab_nll(x, W, b, t3.cuda(), 0)[2]).all())
    t3[2] = 500
    assert bool(torch.isnan(ops.vocab_nll(x, W, b, t3.cuda(), 0)[2][0]))
