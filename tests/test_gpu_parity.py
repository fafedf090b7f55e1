"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Tolerance (north star): fp32 logits within 1e-4 relative (of the tensor's max magnitude); greedy token
ids identical wherever the oracle's top-2 logit margin exceeds that tolerance.
"""
import numpy as np
import pytest
import torch

from conftest import Vocab, load_golden, sub_sd, unpad
from gpu_util import REL, check_greedy, close, cpu_sd, decoder_from_sd, plain_decoder
from oracle import decoder_ref as R

pytestmark = pytest.mark.gpu


def t(x):
    return torch.from_numpy(np.asarray(x))


@pytest.fixture(scope="module")
def ops():
    from capdec import ops as _ops
    return _ops


# ---------------------------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(1, 1, 4), (7, 5, 3), (64, 64, 32), (100, 130, 36), (257, 513, 1000),
                                   (1024, 512, 512), (33, 64, 2050), (1024, 1536, 512), (2048, 2048, 256)])
def test_gemm_plain_and_epilogues(ops, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    sc = torch.rand(N, generator=g) + 0.5
    sh = torch.randn(N, generator=g)
    xd, Wd = x.cuda(), W.cuda()
    ref = (x.double() @ W.double().t())
    close(ops.linear(xd, Wd), ref, what="plain")
    close(ops.linear(xd, Wd, b.cuda()), ref + b.double(), what="bias")
    close(ops.linear(xd, Wd, b.cuda(), relu=True), torch.relu(ref + b.double()), what="relu")
    close(ops.linear(xd, Wd, b.cuda(), residual=res.cuda()), ref + b.double() + res.double(), what="residual")
    close(ops.linear(xd, Wd, b.cuda(), scale=sc.cuda(), shift=sh.cuda()),
          (ref + b.double()) * sc.double() + sh.double(), what="scale/shift")


@pytest.mark.parametrize("M,N,K", [(33, 64, 4100), (100, 40, 5003 * 4), (64, 64, 4096), (7, 5, 4097 * 4 + 28)])
def test_gemm_fixed_order_split_with_a_ragged_last_k_tile(ops, M, N, K):
    """Long contractions split over workgroups with their partial sums added in a fixed order: when K is not a whole
    number of 32-deep tiles the first stage takes the whole tiles and the second stage adds the remaining k.  Against
    fp64, with the epilogue (bias, scale / shift, residual), and bit for bit from run to run."""
    g = torch.Generator().manual_seed(K + M)
    x = torch.randn(M, K + 8, generator=g)[:, :K]            # a strided view, as the feature heads pass
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b, sc, sh = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = (x.double() @ W.double().t() + b.double()) * sc.double() + sh.double() + res.double()
    args = (x.cuda(), W.cuda(), b.cuda())
    kw = dict(scale=sc.cuda(), shift=sh.cuda(), residual=res.cuda(), split_k=True)
    y = ops.linear(*args, **kw)
    close(y, ref)
    assert torch.equal(y, ops.linear(*args, **kw))
    close(ops.linear(args[0], args[1], split_k=True), x.double() @ W.double().t(), what="plain")


def test_gemm_strided_views_and_groups(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(50, 300, generator=g).cuda()
    W1 = torch.randn(40, 100, generator=g).cuda()
    W2 = torch.randn(24, 100, generator=g).cuda()
    out = torch.zeros(50, 64).cuda()
    # two problems sharing K in one launch, strided inputs (column slices) and outputs
    ops.gemm_groups([dict(P=x[:, :100], Q=W1, C=out[:, :40]), dict(P=x[:, 200:300], Q=W2, C=out[:, 40:])], 100)
    ref = torch.cat((x[:, :100].cpu().double() @ W1.cpu().double().t(),
                     x[:, 200:300].cpu().double() @ W2.cpu().double().t()), dim=1)
    close(out, ref)
    # unaligned column offset -> scalar-load kernel
    y = ops.linear(x[:, 1:98], W1[:, 3:100])
    close(y, x[:, 1:98].cpu().double() @ W1[:, 3:100].cpu().double().t())


@pytest.mark.parametrize("M,E", [(1, 64), (5, 256), (1000, 512), (33, 1024)])
def test_add_layernorm(ops, M, E):
    g = torch.Generator().manual_seed(M + E)
    x, a = torch.randn(M, E, generator=g) * 3 + 1, torch.randn(M, E, generator=g)
    gam, bet = torch.rand(E, generator=g) + 0.5, torch.randn(E, generator=g)
    y, mean, rstd = ops.add_layernorm(x.cuda(), gam.cuda(), bet.cuda(), add=a.cuda(), want_stats=True)
    ref = R.layer_norm((x + a).double(), gam.double(), bet.double())
    close(y, ref, rtol=1e-5)
    close(mean, (x + a).double().mean(-1), rtol=1e-5)
    y2 = ops.add_layernorm(x.cuda(), gam.cuda(), bet.cuda())
    close(y2, R.layer_norm(x.double(), gam.double(), bet.double()), rtol=1e-5)


def test_embed_pe(ops):
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(50, 64, generator=g)
    pe = R.positional_encoding_table(20, 64)
    ids = torch.randint(0, 50, (6, 9), generator=g)
    for dt in (torch.int64, torch.int32):
        x = ops.embed_pe(ids.to(dt).cuda(), emb.cuda(), pe.cuda(), pos0=2)
        assert torch.equal(x.cpu().view(6, 9, 64), emb[ids] + pe[2:11])
    x = ops.embed_pe(ids.cuda(), emb.cuda(), None)
    assert torch.equal(x.cpu().view(6, 9, 64), emb[ids])


def test_f1_self_attention_pad_rows(ops):
    fx = load_golden("f1_mha_self")
    sd = {k: v.cuda() for k, v in sub_sd(fx, "sd__").items()}
    x = t(fx["x"]).cuda()
    B, T, E = x.shape
    H = int(fx["H"])
    ids32 = t(fx["ids"]).to(torch.int32).cuda()
    x2 = x.view(B * T, E)
    q = ops.linear(x2, sd["W_q.weight"], sd["W_q.bias"])
    k = ops.linear(x2, sd["W_k.weight"], sd["W_k.bias"])
    v = ops.linear(x2, sd["W_v.weight"], sd["W_v.bias"])
    o = ops.attention(q, k, v, B, H, T, T, causal=True, ids32=ids32)
    y = ops.linear(o, sd["W_o.weight"], sd["W_o.bias"]).view(B, T, E)
    close(y, fx["y"])


def test_f2_cross_attention(ops):
    fx = load_golden("f2_mha_cross")
    sd = {k: v.cuda() for k, v in sub_sd(fx, "sd__").items()}
    x = t(fx["x"]).cuda()
    B, T, E = x.shape
    H = int(fx["H"])
    for S in (1, 2, 5):
        enc = t(fx[f"enc_S{S}"]).cuda().view(B * S, E)
        q = ops.linear(x.view(B * T, E), sd["W_q.weight"], sd["W_q.bias"])
        k = ops.linear(enc, sd["W_k.weight"], sd["W_k.bias"])
        v = ops.linear(enc, sd["W_v.weight"], sd["W_v.bias"])
        o = ops.attention(q, k, v, B, H, T, S)
        y = ops.linear(o, sd["W_o.weight"], sd["W_o.bias"]).view(B, T, E)
        close(y, fx[f"y_S{S}"], what=f"S={S}")


@pytest.mark.parametrize("B,H,Tq,Tk,dk", [(2, 4, 1, 37, 128), (3, 2, 20, 20, 64), (2, 1, 5, 300, 32), (1, 4, 50, 50, 256)])
def test_attention_shapes(ops, B, H, Tq, Tk, dk):
    g = torch.Generator().manual_seed(B + Tq + Tk + dk)
    E = H * dk
    q, k, v = (torch.randn(B * n, E, generator=g) for n in (Tq, Tk, Tk))
    causal = Tq == Tk
    ids = torch.randint(0, 3, (B, Tq), generator=g, dtype=torch.int32)
    o, lse = ops.attention(q.cuda(), k.cuda(), v.cuda(), B, H, Tq, Tk, causal=causal, ids32=ids.cuda(), want_lse=True)
    qh = q.view(B, Tq, H, dk).transpose(1, 2).double()
    kh = k.view(B, Tk, H, dk).transpose(1, 2).double()
    vh = v.view(B, Tk, H, dk).transpose(1, 2).double()
    s = qh @ kh.transpose(-1, -2) / dk ** 0.5
    mask = (ids != 0).view(B, 1, Tq, 1).expand(B, 1, Tq, Tk).clone()
    if causal:
        mask = mask & torch.tril(torch.ones(Tq, Tk, dtype=torch.bool))
    s = s.masked_fill(~mask, -1e9)
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * Tq, E)
    close(o, ref, rtol=2e-5)


def test_vocab_argmax_and_nll(ops):
    g = torch.Generator().manual_seed(11)
    # (up to 32 rows with K a multiple of 32: the eight-wavefront small-batch kernel, with one to four of its k-groups
    #  at work; other K: the general one)
    for M, V, K in ((5, 50, 64), (300, 10000, 512), (64, 5000, 512), (1024, 777, 128), (32, 5000, 512), (17, 777, 136),
                    (1, 10000, 512), (32, 333, 8), (9, 130, 132), (20, 300, 32), (32, 130, 96), (31, 4097, 160)):
        x = torch.randn(M, K, generator=g)
        W = torch.randn(V, K, generator=g) / K ** 0.5
        b = torch.randn(V, generator=g) * 0.1
        logits = x.double() @ W.double().t() + b.double()
        idx, val = ops.vocab_argmax(x.cuda(), W.cuda(), b.cuda())
        close(val, logits.max(-1).values, rtol=1e-5)
        top2 = logits.topk(2, -1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-5 * logits.abs().max()
        assert torch.equal(idx.cpu().long()[safe], logits.argmax(-1)[safe])
        tg = torch.randint(0, V, (M,), generator=g)
        tg[::3] = 0
        lse, nll, sums = ops.vocab_nll(x.cuda(), W.cuda(), b.cuda(), tg.cuda(), 0)
        close(lse, torch.logsumexp(logits, -1), rtol=1e-5)
        ref_nll = (torch.logsumexp(logits, -1) - logits.gather(1, tg[:, None])[:, 0]) * (tg != 0)
        close(nll, ref_nll, rtol=1e-5)
        close(sums, torch.stack((ref_nll.sum(), (tg != 0).sum().double())), rtol=1e-5)


def test_argmax_ties_resolve_to_lowest_index(ops):
    x = torch.ones(70, 8).cuda()
    W = torch.zeros(300, 8)
    W[[17, 150, 299]] = 1.0      # three exactly equal winners in different tiles / lane halves
    idx, _ = ops.vocab_argmax(x, W.cuda(), torch.zeros(300).cuda())
    assert torch.all(idx.cpu() == 17)
    W32 = torch.zeros(300, 32)                                   # K a multiple of 32: the small-batch kernel, one and two row blocks
    W32[[17, 150, 299]] = 1.0
    for m in (29, 47, 128, 256):
        idx, _ = ops.vocab_argmax(torch.ones(m, 32).cuda(), W32.cuda(), torch.zeros(300).cuda())
        assert torch.all(idx.cpu() == 17)


def test_small_batch_vocab_argmax_against_the_general_kernel(ops):
    """Batches of at most 256 rows (blocks of 32) take a kernel of their own (a workgroup of eight wavefronts per vocabulary tile, the
    contraction split four ways); the tuning build's staging override sends the same call through the general kernel.
    Same winners (near-ties aside), values within rounding of each other."""
    from capdec import _lib
    g = torch.Generator().manual_seed(23)
    for M, V, K in ((32, 5000, 512), (32, 10000, 512), (7, 1000, 256), (64, 5000, 512), (33, 1000, 256), (128, 5000, 512), (100, 10000, 512), (256, 5000, 512)):
        x = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(V, K, generator=g) / K ** 0.5).cuda()
        b = (torch.randn(V, generator=g) * 0.1).cuda()
        with _lib.tuning_build() as lib:
            idx_s, val_s = ops.vocab_argmax(x, W, b)
            lib.capdec_gemm_force_staging(1)
            try:
                idx_g, val_g = ops.vocab_argmax(x, W, b)
            finally:
                lib.capdec_gemm_force_staging(0)
        close(val_s, val_g.cpu().double(), rtol=1e-5)
        logits = x.cpu().double() @ W.cpu().double().t() + b.cpu().double()
        top2 = logits.topk(2, -1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-5 * logits.abs().max()
        assert torch.equal(idx_s.cpu()[safe], idx_g.cpu()[safe])


def test_batchnorm_fold_and_train(ops):
    g = torch.Generator().manual_seed(2)
    B, C = 37, 200
    x = torch.randn(B, C, generator=g) * 2 + 1
    sd = {"bn.weight": torch.rand(C, generator=g) + 0.5, "bn.bias": torch.randn(C, generator=g),
          "bn.running_mean": torch.randn(C, generator=g), "bn.running_var": torch.rand(C, generator=g) + 0.5,
          "bn.num_batches_tracked": torch.tensor(0)}
    sc, sh = ops.bn_fold(*(sd["bn." + k].cuda() for k in ("weight", "bias", "running_mean", "running_var")))
    ref, _ = R.batchnorm1d(x, sd, "bn", False)
    close(x.cuda() * sc + sh, ref, rtol=1e-5)
    rm, rv = sd["bn.running_mean"].clone().cuda(), sd["bn.running_var"].clone().cuda()
    y, mean, rstd = ops.bn_train_forward(x.cuda(), sd["bn.weight"].cuda(), sd["bn.bias"].cuda(), rm, rv)
    ref, new = R.batchnorm1d(x, sd, "bn", True)
    close(y, ref, rtol=1e-5)
    close(rm, new["bn.running_mean"], rtol=1e-5)
    close(rv, new["bn.running_var"], rtol=1e-5)


def test_f9_lstm_step(ops):
    fx = load_golden("f7_decoder_rnn")
    sd = {k: v.cuda() for k, v in sub_sd(fx, "sd__").items()}
    w = (sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"], sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"])
    h1, c1 = ops.lstm_step(t(fx["step_x"]).cuda(), t(fx["step_h0"]).cuda(), t(fx["step_c0"]).cuda(), *w)
    close(h1, fx["step_h1"])
    close(c1, fx["step_c1"])
    # zero initial state == NULL state pointers
    z = torch.zeros_like(h1)
    ha, ca = ops.lstm_step(t(fx["step_x"]).cuda(), None, None, *w)
    hb, cb = ops.lstm_step(t(fx["step_x"]).cuda(), z, z, *w)
    close(ha, hb, rtol=1e-6)
    close(ca, cb, rtol=1e-6)


@pytest.mark.parametrize("B,E,Hd", [(300, 256, 512), (257, 64, 96), (1024, 1024, 512)])
def test_lstm_step_tiles_loaded_straight_into_lds(ops, B, E, Hd, tuning_lib):
    """E and Hd multiples of 32 take the kernel whose loads write LDS directly: against the fp64 cell equations
    (cnnrnn_model.py:17,29-30: nn.LSTM, gates i,f,g,o) and bit for bit against the register-staged kernel."""
    from capdec import _lib
    lib = tuning_lib
    g = torch.Generator().manual_seed(B + E)
    x, h0, c0 = (torch.randn(B, n, generator=g) for n in (E, Hd, Hd))
    w_ih, w_hh = torch.randn(4 * Hd, E, generator=g) * E ** -0.5, torch.randn(4 * Hd, Hd, generator=g) * Hd ** -0.5
    b_ih, b_hh = torch.randn(4 * Hd, generator=g), torch.randn(4 * Hd, generator=g)
    gates = x.double() @ w_ih.double().t() + b_ih.double() + h0.double() @ w_hh.double().t() + b_hh.double()
    gi, gf, gg, go = gates.split(Hd, dim=1)
    c_ref = torch.sigmoid(gf) * c0.double() + torch.sigmoid(gi) * torch.tanh(gg)
    h_ref = torch.sigmoid(go) * torch.tanh(c_ref)
    args = [v.cuda() for v in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
    outs = []
    try:
        for staging in (1, 0):
            lib.capdec_lstm_force_staging(staging)
            h1, c1, gt = ops.lstm_step(*args, want_gates=True)
            close(h1, h_ref, what=f"h staging {staging}")
            close(c1, c_ref, what=f"c staging {staging}")
            close(gt, gates, what=f"gates staging {staging}")
            h1z, c1z = ops.lstm_step(args[0], None, None, *args[3:])
            outs.append((h1, c1, gt, h1z, c1z))
    finally:
        lib.capdec_lstm_force_staging(0)
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------
# decoder_forward (teacher forced, inference) against golden logits
# ---------------------------------------------------------------------------------------------
def test_f4_decoder_forward_logits():
    fx = load_golden("f4_decoder_forward")
    dec = decoder_from_sd(sub_sd(fx, "sd__"), int(fx["H"]))
    caps = t(fx["captions"]).cuda()
    with torch.no_grad():
        for S in (1, 2, 5):
            logits = dec.decoder_forward(t(fx[f"enc_S{S}"]).cuda(), caps[:, :-1])
            close(logits, fx[f"logits_S{S}"], what=f"S={S}")
        last = dec.decoder_forward(t(fx["enc_S2"]).cuda(), caps[:, :5], take_last=True)
        close(last, fx["last_logits_S2_T5"])


def test_f10_medium_decoder_default_init():
    """E=512, H=4, L=2, V=10000 with default init under seed 0: also pins the construction order."""
    fx = load_golden("f10_medium")
    torch.manual_seed(int(fx["seed"]))
    dec = plain_decoder(512, 10000, 4, 2).cuda().eval()
    with torch.no_grad():
        logits = dec.decoder_forward(t(fx["enc"]).cuda(), t(fx["captions"]).cuda())
    flat = logits.reshape(-1).cpu()
    close(flat[t(fx["sample_idx"])], fx["sample_logits"])
    assert abs(float(flat.double().sum()) - float(fx["logits_sum"])) <= 1e-4 * float(fx["logits_abs_sum"])
    ids, lengths = dec.greedy_ids(t(fx["enc"]).cuda(), 1, 2, 8)
    from capdec.engine import ids_to_lists
    check_greedy(ids_to_lists(ids, lengths), cpu_sd(dec), t(fx["enc"]), 4, 8, want_rows=unpad(fx["greedy_ids"]))


@pytest.mark.parametrize("B,E,Hd", [(1, 256, 512), (7, 64, 96), (32, 256, 512), (32, 1024, 512), (20, 8, 8), (31, 40, 24), (9, 32, 32),
                                    (32, 96, 160), (33, 256, 512), (64, 256, 512), (50, 64, 96), (128, 256, 512), (97, 32, 64), (256, 256, 512)])
def test_lstm_step_small_batch_kernel(ops, B, E, Hd, tuning_lib):
    """Up to 256 samples (the reference's loaders default to batches of 32 and 64), in blocks of 32, take the kernels that
    give a workgroup 8 hidden units and split the contraction over its wavefronts: against the fp64 cell equations (cnnrnn_model.py:17,
    29-30) and, within rounding, against the large-batch kernel."""
    lib = tuning_lib
    g = torch.Generator().manual_seed(B * 7 + E)
    x, h0, c0 = (torch.randn(B, n, generator=g) for n in (E, Hd, Hd))
    w_ih, w_hh = torch.randn(4 * Hd, E, generator=g) * E ** -0.5, torch.randn(4 * Hd, Hd, generator=g) * Hd ** -0.5
    b_ih, b_hh = torch.randn(4 * Hd, generator=g), torch.randn(4 * Hd, generator=g)
    gates = x.double() @ w_ih.double().t() + b_ih.double() + h0.double() @ w_hh.double().t() + b_hh.double()
    gi, gf, gg, go = gates.split(Hd, dim=1)
    c_ref = torch.sigmoid(gf) * c0.double() + torch.sigmoid(gi) * torch.tanh(gg)
    h_ref = torch.sigmoid(go) * torch.tanh(c_ref)
    args = [v.cuda() for v in (x, h0, c0, w_ih, w_hh, b_ih, b_hh)]
    h1, c1, gt = ops.lstm_step(*args, want_gates=True)
    close(h1, h_ref, what="h")
    close(c1, c_ref, what="c")
    close(gt, gates, what="gates")
    # no previous state: the recurrent half of the contraction is skipped
    h1z, c1z = ops.lstm_step(args[0], None, None, *args[3:])
    gz = x.double() @ w_ih.double().t() + b_ih.double() + b_hh.double()
    zi, zf, zg, zo = gz.split(Hd, dim=1)
    cz = torch.sigmoid(zi) * torch.tanh(zg)
    close(c1z, cz, what="c from a zero state")
    close(h1z, torch.sigmoid(zo) * torch.tanh(cz), what="h from a zero state")
    try:
        lib.capdec_lstm_force_staging(1)         # the large-batch, register-staged kernel
        h1b, c1b, gtb = ops.lstm_step(*args, want_gates=True)
    finally:
        lib.capdec_lstm_force_staging(0)
    close(h1, h1b.cpu().numpy(), rtol=1e-5, what="h, small against large-batch kernel")
    close(gt, gtb.cpu().numpy(), rtol=1e-5, what="gates, small against large-batch kernel")
    if E % 32 == 0 and Hd % 32 == 0 and B <= 32:
        # whole-line loads through wavefront-private LDS against fragments read straight from memory: the same products
        # added in the same order within a 32-float block, the blocks in the same order
        try:
            lib.capdec_lstm_force_staging(2)
            h1c, c1c, gtc = ops.lstm_step(*args, want_gates=True)
        finally:
            lib.capdec_lstm_force_staging(0)
        close(h1, h1c.cpu().numpy(), rtol=1e-6, what="h, the two small-batch kernels")
        close(gt, gtc.cpu().numpy(), rtol=1e-6, what="gates, the two small-batch kernels")


# ---------------------------------------------------------------------------------------------
# greedy decode
# ---------------------------------------------------------------------------------------------
def _greedy(dec, enc, max_len, **kw):
    from capdec.engine import ids_to_lists, tf_greedy_ids
    ids, lengths, n_dirty = tf_greedy_ids(dec, enc.cuda(), max_len, R.SOS, R.EOS, **kw)
    return ids_to_lists(ids, lengths), n_dirty


def test_f5_greedy_random_weights_cached_and_exact():
    fx = load_golden("f5_greedy")
    sd = sub_sd(fx, "a__sd__")
    dec = decoder_from_sd(sd, int(fx["H"]))
    for S in (1, 2, 5):
        enc = t(fx[f"a_enc_S{S}"])
        want = unpad(fx[f"a_ids_S{S}"])
        # (fold_memory: the long-memory form of the cross-attention, forced here on the fixture's 5-row memory)
        for kw in ({}, {"exact": True}, {"no_s1_shortcut": True}) + (({"fold_memory": True},) if S >= 3 else ()):
            got, _ = _greedy(dec, enc, 12, **kw)
            check_greedy(got, sd, enc, int(fx["H"]), 12, want_rows=want)


def test_two_memory_rows_closed_form_against_the_oracle_and_the_general_kernels():
    """S = 2 (the ViT+CNN and YOLO+CNN heads): the cached decoder folds the cross-attention block into the LayerNorms
    around it (a per-sample constant plus one logistic gate per head).  Same tokens as the oracle (near-ties aside) and
    as the general cross-attention kernels, at the benchmark's widths."""
    torch.manual_seed(5)
    dec = plain_decoder(512, 10000, 4, 2).cuda().eval()
    g = torch.Generator().manual_seed(77)
    enc = torch.randn(24, 2, 512, generator=g)
    got, _ = _greedy(dec, enc, 12)
    check_greedy(got, cpu_sd(dec), enc, 4, 12)
    general, _ = _greedy(dec, enc, 12, no_s1_shortcut=True)
    same = sum(a == b for a, b in zip(got, general))
    assert same >= 23, f"{same} of 24 rows equal"
    # memory rows that differ a lot (saturated gates) and rows that are equal (gate exactly one half)
    enc2 = torch.cat([enc[:8] * 30.0, enc[8:16, :1].expand(8, 2, 512)], 0).contiguous()
    got2, _ = _greedy(dec, enc2, 12)
    check_greedy(got2, cpu_sd(dec), enc2, 4, 12)


@pytest.mark.parametrize("S,B", [(261, 12), (96, 9), (100, 5), (33, 7)])
def test_long_memory_decode_attends_to_the_raw_memory(S, B):
    """From 48 memory rows on the cached decoder folds the key / value projections into the query and output
    projections and reads the raw memory once per step: same tokens as the oracle (near-ties aside) and as the
    projected form, at the benchmark's widths; forced on a shorter memory too.  Rows that generate <PAD> are re-run
    through the same kernels."""
    torch.manual_seed(11)
    dec = plain_decoder(512, 10000, 4, 2).cuda().eval()
    g = torch.Generator().manual_seed(S)
    enc = torch.randn(B, S, 512, generator=g)
    kw = {} if S >= 48 else {"fold_memory": True}
    got, _ = _greedy(dec, enc, 10, **kw)
    check_greedy(got, cpu_sd(dec), enc, 4, 10)
    projected, _ = _greedy(dec, enc, 10, no_s1_shortcut=True)
    assert sum(a == b for a, b in zip(got, projected)) >= B - 1
    with pytest.raises(RuntimeError):                 # nothing to fold with one or two memory rows
        _greedy(dec, enc[:, :2].contiguous(), 4, fold_memory=True)


def test_long_memory_decode_with_generated_pad_rows():
    fx = load_golden("f5_greedy")
    sd = sub_sd(fx, "b__sd__")
    dec = decoder_from_sd(sd, int(fx["H"]))
    enc = t(fx["b_enc"])
    if enc.shape[1] < 3:
        enc = enc[:, :1].expand(-1, 3, -1).contiguous()       # three equal memory rows attend like one
    want = unpad(fx["b_ids"])
    got, n_fallback = _greedy(dec, enc, 14, extra_cap=256, fold_memory=True)
    assert n_fallback == 0
    check_greedy(got, sd, enc, int(fx["H"]), 14, want_rows=want if t(fx["b_enc"]).shape[1] >= 3 else None)
    assert sum(0 in row for row in got) >= len(got) // 2          # the fixture's point: most rows emit <PAD>


@pytest.mark.parametrize("S", [1, 2, 64, 261])
def test_greedy_decode_is_bit_identical_from_run_to_run(S):
    """No kernel of the decode path adds floating-point numbers in an order that depends on timing (no float atomics):
    the same batch decoded three times gives the same tokens and lengths, for every form of the cross-attention (one
    memory row, two, a feature grid, a long token memory)."""
    from capdec.engine import tf_greedy_ids
    torch.manual_seed(3)
    dec = plain_decoder(512, 10000, 4, 2).cuda().eval()
    g = torch.Generator().manual_seed(100 + S)
    enc = torch.randn(96, S, 512, generator=g).cuda()
    first = tf_greedy_ids(dec, enc, 20, R.SOS, R.EOS)
    for _ in range(2):
        again = tf_greedy_ids(dec, enc, 20, R.SOS, R.EOS)
        assert torch.equal(again[0], first[0]) and torch.equal(again[1], first[1])


def test_greedy_outputs_packed_on_the_device_and_prebuilt_parameter_struct():
    """ids, lengths and flags share one device allocation: the host copies (one pinned transfer), the device views
    (to_cpu=False, what a cross-rank gather takes) and a call through a parameter struct built once must agree."""
    from capdec.engine import TfModelRef, tf_greedy_ids
    fx = load_golden("f5_greedy")
    dec = decoder_from_sd(sub_sd(fx, "a__sd__"), int(fx["H"]))
    enc = t(fx["a_enc_S2"]).cuda()
    ids, lens, _ = tf_greedy_ids(dec, enc, 12, R.SOS, R.EOS)
    assert not ids.is_cuda and ids.dtype == torch.int64 and lens.dtype == torch.int32 and ids.shape == (enc.shape[0], 12)
    ids_d, lens_d, _ = tf_greedy_ids(dec, enc, 12, R.SOS, R.EOS, to_cpu=False)
    assert ids_d.is_cuda and torch.equal(ids_d.cpu(), ids) and torch.equal(lens_d.cpu(), lens)
    ref = TfModelRef(dec)
    for kw in ({}, {"exact": True}, {"resolve_dirty": False}):
        ids2, lens2, _ = tf_greedy_ids(ref, enc, 12, R.SOS, R.EOS, **kw)
        assert torch.equal(ids2, ids) and torch.equal(lens2, lens), kw
    e0, l0, _ = tf_greedy_ids(ref, enc[:0], 12, R.SOS, R.EOS)
    assert e0.shape == (0, 12) and l0.shape == (0,)


def test_f5_greedy_generated_pad_rows_are_rerun_inside_the_cached_loop():
    """Weights crafted so that most rows emit <PAD> mid-sequence: the cached decoder must reproduce the reference
    (which masks the PAD query row to a uniform softmax over the CURRENT length) without the host fallback when
    the per-step row list is large enough, and through the exact fallback when it is not."""
    fx = load_golden("f5_greedy")
    sd = sub_sd(fx, "b__sd__")
    dec = decoder_from_sd(sd, int(fx["H"]))
    enc = t(fx["b_enc"])
    want = unpad(fx["b_ids"])
    got, n_fallback = _greedy(dec, enc, 14, extra_cap=256)
    assert n_fallback == 0
    assert check_greedy(got, sd, enc, int(fx["H"]), 14, want_rows=want) == 0
    assert got == want
    got, n_fallback = _greedy(dec, enc, 14, extra_cap=8)          # list overflows -> exact re-decode of those samples
    assert n_fallback > 0
    assert got == want
    got, _ = _greedy(dec, enc, 14, extra_cap=8, resolve_dirty=False)
    assert got != want                                            # without the fallback the overflowed samples are wrong


def test_f5_greedy_rows_finish_early():
    fx = load_golden("f5_greedy")
    sd = sub_sd(fx, "c__sd__")
    dec = decoder_from_sd(sd, int(fx["H"]))
    got, _ = _greedy(dec, t(fx["c_enc"]), 30)
    check_greedy(got, sd, t(fx["c_enc"]), int(fx["H"]), 30, want_rows=unpad(fx["c_ids"]))


def test_greedy_loop_ends_on_the_device_once_every_caption_has_finished():
    """The reference leaves the loop when every caption holds an <EOS> (transformer.py:166-168).  The cached decoder
    enqueues all max_length steps ahead of time; once the device-side count of running captions is zero the remaining
    steps run on no rows.  Same tokens as with the loop run to the end (an <EOS> id that never occurs keeps every row
    alive, its tokens up to each caption's <EOS> position must agree), same tokens as the oracle, and a fraction of the
    time."""
    import time
    from capdec.engine import tf_greedy_ids
    torch.manual_seed(21)
    V, E, H, L, B, ML = 10000, 512, 4, 2, 512, 50
    dec = plain_decoder(E, V, H, L).cuda().eval()
    with torch.no_grad():
        # logits of a default-init decoder: standard deviation 0.58, the largest of 10000 about 2.3
        dec.fc2.bias[R.EOS] += 1.9
    g = torch.Generator().manual_seed(9)
    enc = torch.randn(B, 1, E, generator=g)
    # some rows of a random decoder repeat one token for ever; a row's caption does not depend on the rest of the batch,
    # so the batch is cut down to the rows that finish within 30 steps: finishing times spread over 0 .. 29
    _, lens0, _ = tf_greedy_ids(dec, enc.cuda(), ML, R.SOS, R.EOS)
    keep = lens0 < 30
    enc = enc[keep].contiguous()
    B = enc.shape[0]
    assert B >= 64 and int(lens0[keep].max()) >= 3, (B, lens0[keep].max())
    enc_d = enc.cuda()

    def timed(eos):
        tf_greedy_ids(dec, enc_d, ML, R.SOS, eos)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = tf_greedy_ids(dec, enc_d, ML, R.SOS, eos)
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    (ids, lens, _), dt = timed(R.EOS)
    (ids_full, lens_full, _), dt_full = timed(V + 7)       # no such token: every row runs all 50 steps
    lens_l = lens.tolist()
    assert lens_l == lens0[keep].tolist() and max(lens_l) < 30
    assert min(lens_full.tolist()) == ML
    for b in range(B):
        n = lens_l[b]
        assert ids[b, :n].tolist() == ids_full[b, :n].tolist()
        assert int(ids_full[b, n]) == R.EOS
    got = [row[:n] for row, n in zip(ids.tolist(), lens_l)]
    check_greedy(got[:48], cpu_sd(dec), enc[:48], H, ML)
    print(f"early exit: {B} captions, the longest {max(lens_l)} of {ML} steps, {dt * 1e3:.2f} ms against {dt_full * 1e3:.2f} ms")
    assert dt < 0.85 * dt_full
    (ids2, lens2, _), _ = timed(R.EOS)                     # and the same again from run to run
    assert torch.equal(ids2, ids) and torch.equal(lens2, lens)


def test_greedy_api_strings_and_limits():
    fx = load_golden("f5_greedy")
    sd = sub_sd(fx, "c__sd__")
    dec = decoder_from_sd(sd, int(fx["H"]))
    vocab = Vocab(50)
    text = dec.greedy_inference(t(fx["c_enc"]).cuda(), vocab, 30)
    assert text == R.detokenize(unpad(fx["c_ids"]), vocab.itos)
    with pytest.raises(RuntimeError):          # longer than the positional-encoding buffer
        dec.greedy_inference(t(fx["c_enc"]).cuda(), vocab, 51)
    with pytest.raises(RuntimeError):          # CPU tensors are refused: no CPU path
        dec.greedy_inference(t(fx["c_enc"]), vocab, 5)
    assert dec.greedy_inference(t(fx["c_enc"])[:0].cuda(), vocab, 5) == []


# ---------------------------------------------------------------------------------------------
# wrappers (feature heads) and CNN-RNN
# ---------------------------------------------------------------------------------------------
def test_f6_cnn_attn_wrapper_eval_logits_and_captions():
    from models.cnnattn_model import CNNAttentionModel
    fx = load_golden("f6_heads")
    sd = sub_sd(fx, "cnn__sd__")
    m = CNNAttentionModel(64, 50, int(fx["H"]), 2, dropout=0.0)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    feats, caps = t(fx["cnn_feats"]).cuda(), t(fx["captions"]).cuda()
    with torch.no_grad():
        close(m(feats, caps[:, :-1]), fx["cnn_logits_eval"])
    vocab = Vocab(50)
    assert m.caption_images(feats, vocab, max_length=10) == R.detokenize(unpad(fx["cnn_greedy_ids"]), vocab.itos)


def test_f6_pair_head_eval():
    from models.vitcnnattn_model import VITCNNAttentionModel
    fx = load_golden("f6_heads")
    sd = sub_sd(fx, "pair__sd__")
    split = int(fx["pair_split"])
    m = VITCNNAttentionModel(64, 50, int(fx["H"]), 1, dropout=0.0)
    # scaled-down split/width of the fixture: replace the two projections' shapes, keep the layout
    m.vit_out_size = split
    m.fc_vit = torch.nn.Linear(split, 64)
    m.fc_cnn = torch.nn.Linear(sd["fc_cnn.weight"].shape[1], 64)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    feats, caps = t(fx["pair_feats"]).cuda(), t(fx["captions"]).cuda()
    with torch.no_grad():
        close(m.encode_features(feats), fx["pair_enc_eval"], rtol=1e-5)
        close(m(feats, caps[:, :-1], "precomputed"), fx["pair_logits_eval"])


def test_f7_cnn_rnn_eval_logits_and_greedy():
    from models.cnnrnn_model import CNNtoRNN
    fx = load_golden("f7_decoder_rnn")
    sd = sub_sd(fx, "sd__")
    E, Hd, V = sd["fc1.weight"].shape[0], sd["lstm.weight_hh_l0"].shape[1], sd["embed.weight"].shape[0]
    m = CNNtoRNN(E, Hd, V)
    m.decoderRNN.load_state_dict(sd)
    m = m.cuda().eval()
    feats, caps = t(fx["feats"]).cuda(), t(fx["captions"]).cuda()
    with torch.no_grad():
        close(m(feats, caps[:, :-1]), fx["logits_eval"])
    sd2 = dict(sd)
    for k in ("fc2.bias", "batchnorm.running_mean", "batchnorm.running_var"):
        sd2[k] = t(fx["greedy__" + k])
    m.decoderRNN.load_state_dict(sd2)
    m = m.cuda()
    ids, lengths = m.caption_ids(feats, max_length=15)
    from capdec.engine import ids_to_lists
    assert ids_to_lists(ids, lengths) == unpad(fx["greedy_ids"])


def test_cnn_rnn_greedy_loop_ends_on_the_device_once_every_caption_has_finished():
    """cnnrnn_model.py:101-103 leaves the loop when every caption holds an <EOS>.  The decoder enqueues all max_length
    steps; once all captions have finished, the later steps' kernels find a row count of zero.  Same tokens as with the
    loop run to the end (an <EOS> id that never occurs), same tokens as the oracle's restatement of the reference loop,
    in less time.  Batches of 32 and 96 (one and three blocks of the small-batch kernels)."""
    import time
    from capdec.engine import rnn_greedy_ids
    from models.cnnrnn_model import CNNtoRNN
    E, Hd, V, ML = 256, 512, 5000, 40
    torch.manual_seed(4)
    m = CNNtoRNN(E, Hd, V).eval()
    with torch.no_grad():
        m.decoderRNN.fc2.bias[R.EOS] += 0.12       # default-init logits: the largest of 5000 is about 0.15-0.2
    sd = {k[len("decoderRNN."):]: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    rnn = m.decoderRNN
    g = torch.Generator().manual_seed(6)
    pool = torch.rand(512, 2048, generator=g)
    with torch.no_grad():
        x_pool = rnn.image_input(pool.cuda())
    # rows that finish within 25 steps (a caption does not depend on the rest of the batch)
    _, lens0 = rnn_greedy_ids(rnn, x_pool, ML, R.SOS, R.EOS)
    keep = torch.nonzero(lens0 < 25).flatten()
    assert keep.numel() >= 96 and int(lens0[keep].max()) >= 2, (keep.numel(), lens0[keep].max())
    for B in (32, 96):
        rows = keep[:B]
        feats, x0 = pool[rows], x_pool[rows.cuda()].contiguous()
        want = R.cnnrnn_greedy_ids(sd, feats, ML)

        def timed(eos):
            best = None
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = rnn_greedy_ids(rnn, x0, ML, R.SOS, eos)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            return out, best

        (ids, lens), dt = timed(R.EOS)
        (ids_full, lens_full), dt_full = timed(V + 7)
        lens_l = lens.tolist()
        assert max(lens_l) < 25 and min(lens_full.tolist()) == ML
        got = [row[:n] for row, n in zip(ids.tolist(), lens_l)]
        assert got == want
        for b in range(B):
            assert ids_full[b, :lens_l[b]].tolist() == got[b] and int(ids_full[b, lens_l[b]]) == R.EOS
        print(f"CNN-RNN early exit, batch {B}: longest caption {max(lens_l)} of {ML} steps, "
              f"{dt * 1e3:.3f} ms against {dt_full * 1e3:.3f} ms")
        assert dt < dt_full


# ---------------------------------------------------------------------------------------------
# BASELINE-sized workloads: size-independent properties + oracle on a row subset
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S", [1, 64])
def test_full_size_decode_cfg2(S):
    """B=1024, E=512, L=2, V=10000, len 20: rows are independent, so (i) a 6-row subset must match the
    oracle, (ii) decoding a sub-batch alone must give the same ids, (iii) cached == exact on a sub-batch."""
    torch.manual_seed(0)
    dec = plain_decoder(512, 10000, 4, 2).cuda().eval()
    g = torch.Generator().manual_seed(1234 + 2)
    enc = torch.randn(1024, S, 512, generator=g)
    got, n_dirty = _greedy(dec, enc, 20)
    assert len(got) == 1024 and all(len(r) <= 20 for r in got)
    sd = cpu_sd(dec)
    rows = [0, 1, 511, 777, 1022, 1023]
    check_greedy([got[r] for r in rows], sd, enc[rows], 4, 20)
    sub, _ = _greedy(dec, enc[256:320], 20)
    exact, _ = _greedy(dec, enc[256:320], 20, exact=True)
    check_greedy(exact, sd, enc[256:320], 4, 20, want_rows=sub, tol=REL)
    assert sum(a != b for a, b in zip(sub, got[256:320])) <= 1   # batch-size independent up to a near-tie


def test_gemm_split_k_atomic_and_forced_tiles(ops, tuning_lib):
    """Long contraction split over workgroups (partial sums to a scratch tensor, added in a fixed order; float atomics
    for the weight gradients) and every tile configuration of the kernel."""
    from capdec import _lib
    lib = tuning_lib
    g = torch.Generator().manual_seed(77)
    x = torch.randn(96, 9000, generator=g)
    W = torch.randn(80, 9000, generator=g)
    b, sc, sh = torch.randn(80, generator=g), torch.rand(80, generator=g) + 0.5, torch.randn(80, generator=g)
    ref = (x.double() @ W.double().t() + b.double()) * sc.double() + sh.double()
    try:
        for cfg in (0, 11, 116, 32, 316, 22, 216):
            lib.capdec_gemm_force_config(cfg)
            y = ops.linear(x.cuda(), W.cuda(), b.cuda(), scale=sc.cuda(), shift=sh.cuda(), split_k=True)
            close(y, ref, what=f"split-k cfg {cfg}")
            # partial sums are added in a fixed order: bit-identical from run to run
            assert torch.equal(y, ops.linear(x.cuda(), W.cuda(), b.cuda(), scale=sc.cuda(), shift=sh.cuda(), split_k=True))
            yr = ops.linear(x.cuda(), W.cuda(), b.cuda(), relu=True, residual=y, split_k=True)
            close(yr, torch.relu(x.double() @ W.double().t() + b.double()) + y.double().cpu(), what=f"split-k relu+residual cfg {cfg}")
            y2 = ops.linear(x.cuda(), W.cuda(), b.cuda(), relu=True)
            close(y2, torch.relu(x.double() @ W.double().t() + b.double()), what=f"plain cfg {cfg}")
            dy = torch.randn(300, 70, generator=g)
            xx = torch.randn(300, 130, generator=g)
            close(ops.linear_wgrad(dy.cuda(), xx.cuda()), dy.double().t() @ xx.double(), what=f"wgrad cfg {cfg}")
            Wd = torch.randn(70, 130, generator=g)
            close(ops.linear_dgrad(dy.cuda(), Wd.cuda()), dy.double() @ Wd.double(), what=f"dgrad cfg {cfg}")
    finally:
        lib.capdec_gemm_force_config(0)


def test_gemm_tiles_loaded_straight_into_lds(ops, tuning_lib):
    """Shapes made of whole, aligned k-tiles take the path whose loads write LDS directly: every tile configuration and
    operand layout against fp64, and bit for bit against the register-staged path (same order of sums)."""
    from capdec import _lib
    lib = tuning_lib
    g = torch.Generator().manual_seed(78)
    x, W, b = torch.randn(200, 1056, generator=g), torch.randn(136, 1056, generator=g), torch.randn(136, generator=g)
    dy, xx = torch.randn(352, 72, generator=g), torch.randn(352, 132, generator=g)
    dy2, Wd = torch.randn(300, 96, generator=g), torch.randn(96, 132, generator=g)
    try:
        for cfg in (0, 11, 116, 32, 316, 22, 216):
            lib.capdec_gemm_force_config(cfg)
            outs = []
            for staging in (1, 0):
                lib.capdec_gemm_force_staging(staging)
                y = ops.linear(x.cuda(), W.cuda(), b.cuda(), relu=True)
                close(y, torch.relu(x.double() @ W.double().t() + b.double()), what=f"plain cfg {cfg} staging {staging}")
                ys = ops.linear(x.cuda(), W.cuda(), b.cuda(), split_k=True)
                close(ys, x.double() @ W.double().t() + b.double(), what=f"split-k cfg {cfg} staging {staging}")
                dw = ops.linear_wgrad(dy.cuda(), xx.cuda())
                close(dw, dy.double().t() @ xx.double(), what=f"wgrad cfg {cfg} staging {staging}")
                dx = ops.linear_dgrad(dy2.cuda(), Wd.cuda())
                close(dx, dy2.double() @ Wd.double(), what=f"dgrad cfg {cfg} staging {staging}")
                outs.append((y, dx))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), f"cfg {cfg}"
    finally:
        lib.capdec_gemm_force_config(0)
        lib.capdec_gemm_force_staging(0)


def test_gemm_direct_to_lds_edge_shapes_and_device_limits(ops, tuning_lib):
    """Smallest whole-k-tile shapes, one-row / four-column outputs, and the device-side row and contraction limits
    (0, inside a tile, across tiles) of the contraction kernels."""
    from capdec import _lib
    lib = tuning_lib
    g = torch.Generator().manual_seed(79)
    try:
        for cfg in (0, 11, 116, 32, 316, 216):
            lib.capdec_gemm_force_config(cfg)
            for (M, N, K) in [(1, 4, 16), (3, 8, 32), (65, 68, 48), (130, 260, 64)]:
                x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
                close(ops.linear(x.cuda(), W.cuda(), b.cuda()), x.double() @ W.double().t() + b.double(),
                      what=f"cfg {cfg} {M}x{N}x{K}")
            x, W = torch.randn(300, 64, generator=g), torch.randn(72, 64, generator=g)
            for live in (0, 5, 64, 299):
                rows = torch.tensor([live], dtype=torch.int32, device="cuda")
                out = torch.full((300, 72), 7.0, device="cuda")
                ops.linear(x.cuda(), W.cuda(), None, out=out, rows_dev=rows)
                close(out[:live], x[:live].double() @ W.double().t(), what=f"cfg {cfg} rows {live}")
                assert bool((out[live:] == 7.0).all()), f"cfg {cfg}: rows past the limit were written"
            dy, xx = torch.randn(320, 72, generator=g), torch.randn(320, 64, generator=g)
            for live in (0, 7, 33, 320):
                rows = torch.tensor([live], dtype=torch.int32, device="cuda")
                dyz = dy.clone()
                dyz[live:] = 0                 # the contract: zeros from the limit to the next multiple of 32
                dw = ops.linear_wgrad(dyz.cuda(), xx.cuda(), rows_dev=rows)
                close(dw, dy[:live].double().t() @ xx[:live].double(), atol=1e-6, what=f"cfg {cfg} k limit {live}")
    finally:
        lib.capdec_gemm_force_config(0)


def test_gemm_epilogue_residual_accumulate_and_mask_in_every_tile(ops, tuning_lib):
    """Epilogue operands that some tile configurations fetch before the contraction and others in a pass of their own:
    residual, in-place accumulation (R == C), ReLU mask, with and without a device-side row limit."""
    from capdec import _lib
    lib = tuning_lib
    g = torch.Generator().manual_seed(80)
    M, N, K = 200, 136, 96
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    dy, Wd = torch.randn(M, N, generator=g), torch.randn(N, K, generator=g)
    mask = (torch.rand(M, K, generator=g) > 0.4).float()
    acc0 = torch.randn(M, K, generator=g)
    try:
        for cfg in (0, 11, 116, 32, 316, 22, 216):
            lib.capdec_gemm_force_config(cfg)
            for staging in (1, 0):
                lib.capdec_gemm_force_staging(staging)
                tag = f"cfg {cfg} staging {staging}"
                y = ops.linear(x.cuda(), W.cuda(), b.cuda(), residual=R.cuda())
                close(y, x.double() @ W.double().t() + b.double() + R.double(), what=f"residual {tag}")
                for live in (0, 33, 200):
                    rows = torch.tensor([live], dtype=torch.int32, device="cuda")
                    out = torch.full((M, N), 7.0, device="cuda")
                    ops.linear(x.cuda(), W.cuda(), b.cuda(), residual=R.cuda(), out=out, rows_dev=rows)
                    close(out[:live], (x.double() @ W.double().t() + b.double() + R.double())[:live],
                          what=f"residual rows {live} {tag}")
                    assert bool((out[live:] == 7.0).all()), f"{tag}: rows past the limit were written"
                dx = ops.linear_dgrad(dy.cuda(), Wd.cuda(), mask=mask.cuda())
                close(dx, (dy.double() @ Wd.double()) * mask.double(), what=f"masked dgrad {tag}")
                acc = acc0.clone().cuda()
                ops.linear_dgrad(dy.cuda(), Wd.cuda(), accum=acc)
                close(acc, dy.double() @ Wd.double() + acc0.double(), what=f"accumulating dgrad {tag}")
    finally:
        lib.capdec_gemm_force_config(0)
        lib.capdec_gemm_force_staging(0)


def test_maximum_length_50_forward_greedy_and_beam():
    """The positional-encoding buffer holds 50 positions (transformer.py:58): teacher-forced logits at T=50, and greedy
    and beam decodes of the full 50 steps (cached, exact, with PAD rows mid-sequence) against the oracle."""
    fx = load_golden("f5_greedy")
    sd = sub_sd(fx, "b__sd__")                      # the weight set that emits <PAD> often
    H = int(fx["H"])
    dec = decoder_from_sd(sd, H)
    g = torch.Generator().manual_seed(50)
    enc = torch.randn(4, 2, 64, generator=g)
    ids = torch.randint(0, 50, (4, 50), generator=g)
    ids[:, 0] = 1
    with torch.no_grad():
        close(dec.decoder_forward(enc.cuda(), ids.cuda()), R.decoder_forward(sd, enc, ids, H), what="logits T=50")
    got, _ = _greedy(dec, enc, 50)
    check_greedy(got, sd, enc, H, 50)
    exact, _ = _greedy(dec, enc, 50, exact=True)
    check_greedy(exact, sd, enc, H, 50)
    from capdec.engine import ids_to_lists
    want = R.beam_search_ids(sd, enc, H, 2, 50)
    for ex in (False, True):
        b_ids, b_len = dec.beam_ids(enc.cuda(), R.SOS, R.EOS, 2, 50, exact=ex)
        assert ids_to_lists(b_ids, b_len) == want, ex
