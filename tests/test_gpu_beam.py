"""GPU parity of beam search (Transformer and CNN-RNN) against the golden vectors produced by the reference's
beam_search_inference / caption_images_beam_search, and against the oracle's hypothesis scores."""
import numpy as np
import pytest
import torch

from conftest import Vocab, load_golden, sub_sd, unpad
from gpu_util import close, decoder_from_sd
from oracle import decoder_ref as R

pytestmark = pytest.mark.gpu


def t(x):
    return torch.from_numpy(np.asarray(x))


def _rows(ids, lengths):
    from capdec.engine import ids_to_lists
    return ids_to_lists(ids, lengths)


@pytest.mark.parametrize("case", ["a_S1_W3", "a_S2_W3", "a_S2_W2", "b", "b2", "c"])
def test_f11_transformer_beam_search(case):
    fx = load_golden("f11_beam")
    H = int(fx["H"])
    if case.startswith("a_"):
        _, S, W = case.split("_")
        W = int(W[1:])
        sd, enc, want = sub_sd(fx, "a__sd__"), t(fx[f"a_enc_{S}_W{W}"]), unpad(fx[f"a_ids_{S}_W{W}"])
        ml = fx[f"a_ids_{S}_W{W}"].shape[1]
    else:
        W = 3
        sd = sub_sd(fx, ("b" if case == "b2" else case) + "__sd__")
        if case == "b2":
            sd["fc2.bias"] = t(fx["b2__fc2.bias"])
        enc, want, ml = t(fx[case + "_enc"]), unpad(fx[case + "_ids"]), fx[case + "_ids"].shape[1]
    dec = decoder_from_sd(sd, H)
    _, _, want_scores = R.beam_search_ids(sd, enc, H, W, ml, return_state=True)
    for exact in (False, True):          # KV-cached (with the <PAD> samples redone) and full-prefix everywhere
        ids, lengths, scores = dec.beam_ids(enc.cuda(), R.SOS, R.EOS, W, ml, return_scores=True, exact=exact)
        assert _rows(ids, lengths) == want, exact
        # the hypothesis scores (sums of log-probabilities) against the oracle's, which
        # tests/test_oracle_golden.py::test_f12_beam_search_after_eos pins to the reference's own
        close(scores, want_scores, rtol=1e-4, atol=1e-5, what="scores")


def test_f11_beam_api_strings_and_repeated_memory():
    """beam_search_inference takes the memory repeated per hypothesis, as the reference's callers pass it."""
    fx = load_golden("f11_beam")
    H = int(fx["H"])
    sd = sub_sd(fx, "a__sd__")
    dec = decoder_from_sd(sd, H)
    enc = t(fx["a_enc_S2_W3"]).cuda()
    B = enc.shape[0]
    rep = enc.unsqueeze(1).expand(-1, 3, -1, -1).reshape(B * 3, 2, -1)
    vocab = Vocab(50)
    text = dec.beam_search_inference(rep, B, vocab, 3, 12)
    want = [" ".join(vocab.itos[i] for i in row) for row in unpad(fx["a_ids_S2_W3"])]
    assert text == want
    assert dec.beam_search_inference(enc, B, vocab, 3, 12) == want
    with pytest.raises(RuntimeError):
        dec.beam_ids(enc, R.SOS, R.EOS, 9, 12)
    with pytest.raises(RuntimeError):
        dec.beam_ids(enc, R.SOS, R.EOS, 3, 51)


def test_beam_width_one_is_greedy():
    fx = load_golden("f11_beam")
    H = int(fx["H"])
    sd = sub_sd(fx, "c__sd__")
    dec = decoder_from_sd(sd, H)
    enc = t(fx["c_enc"])
    got = _rows(*dec.beam_ids(enc.cuda(), R.SOS, R.EOS, 1, 12))
    assert got == R.greedy_inference_ids(sd, enc, H, 12)


@pytest.mark.parametrize("case", ["r0", "r"])
def test_f11_cnn_rnn_beam_search(case):
    from models.cnnrnn_model import CNNtoRNN
    fx = load_golden("f11_beam")
    sd = sub_sd(fx, "r__sd__")
    if case == "r":
        sd["fc2.bias"] = t(fx["r__fc2.bias_biased"])
    E, Hd, V = sd["fc1.weight"].shape[0], sd["lstm.weight_hh_l0"].shape[1], sd["embed.weight"].shape[0]
    m = CNNtoRNN(E, Hd, V)
    m.decoderRNN.load_state_dict(sd)
    m = m.cuda().eval()
    feats = t(fx["r_feats"])
    got = _rows(*m.caption_ids_beam(feats.cuda(), beam_width=3, max_length=15))
    assert got == unpad(fx[case + "_ids"])
    vocab = Vocab(V)
    text = m.caption_images_beam_search(feats.cuda(), vocab, beam_width=3, max_length=15)
    assert text == [" ".join(vocab.itos[i] for i in row) for row in got]


def test_beam_medium_size_against_oracle():
    """E=512, V=10000, L=2, B=6, W=3, 10 steps: best hypotheses and scores against the oracle."""
    from gpu_util import cpu_sd, plain_decoder
    torch.manual_seed(0)
    dec = plain_decoder(512, 10000, 4, 2).cuda().eval()
    sd = cpu_sd(dec)
    g = torch.Generator().manual_seed(77)
    enc = torch.randn(6, 2, 512, generator=g)
    want, _, want_scores = R.beam_search_ids(sd, enc, 4, 3, 10, return_state=True)
    for exact in (False, True):
        ids, lengths, scores = dec.beam_ids(enc.cuda(), R.SOS, R.EOS, 3, 10, return_scores=True, exact=exact)
        close(scores, want_scores, rtol=1e-4, atol=1e-4, what="scores")
        assert _rows(ids, lengths) == want


@pytest.mark.parametrize("W", [2, 5, 8])
def test_beam_widths_of_every_top_w_list_size_against_oracle(W):
    """The log-softmax / top-W sweep keeps per-lane lists of 2, 4 or 8 candidates depending on the width, and takes a
    workgroup per hypothesis row at these sizes: widths 2, 5 and 8 at V = 2000 against the oracle (hypotheses, scores)."""
    from gpu_util import cpu_sd, plain_decoder
    torch.manual_seed(10 + W)
    dec = plain_decoder(512, 2000, 4, 2).cuda().eval()
    sd = cpu_sd(dec)
    g = torch.Generator().manual_seed(90 + W)
    enc = torch.randn(5, 1, 512, generator=g)
    want, _, want_scores = R.beam_search_ids(sd, enc, 4, W, 7, return_state=True)
    ids, lengths, scores = dec.beam_ids(enc.cuda(), R.SOS, R.EOS, W, 7, return_scores=True)
    close(scores, want_scores, rtol=1e-4, atol=1e-4, what="scores")
    assert _rows(ids, lengths) == want


def test_cached_beam_equals_full_prefix_beam_at_batch_256():
    """Rows are independent and the two forms compute the same function: identical best hypotheses (up to a
    near-tie) and scores on a larger batch with one and with 64 memory rows."""
    from gpu_util import plain_decoder
    torch.manual_seed(2)
    dec = plain_decoder(512, 10000, 4, 2).cuda().eval()
    g = torch.Generator().manual_seed(78)
    for S in (1, 2, 64):
        enc = torch.randn(256, S, 512, generator=g).cuda()
        a = dec.beam_ids(enc, R.SOS, R.EOS, 3, 20, return_scores=True)
        b = dec.beam_ids(enc, R.SOS, R.EOS, 3, 20, return_scores=True, exact=True)
        ra, rb = _rows(a[0], a[1]), _rows(b[0], b[1])
        assert sum(x != y for x, y in zip(ra, rb)) <= 2
        same = [i for i, (x, y) in enumerate(zip(ra, rb)) if x == y]
        close(a[2][same], b[2][same], rtol=1e-4, atol=1e-4, what="scores")


@pytest.mark.parametrize("S,W,B", [(1, 3, 100), (2, 2, 37), (1, 4, 16)])
def test_cached_beam_one_launch_layers_against_separate_launches(S, W, B):
    """Shapes the one-launch decoder layers support (E=512, H=4, L<=2, one or two memory rows) take them inside the
    KV-cached beam search; the tuning build can keep the launch-per-operation layers: same hypotheses and scores."""
    from capdec import _lib
    from gpu_util import plain_decoder
    torch.manual_seed(5)
    dec = plain_decoder(512, 3000, 4, 2).cuda().eval()
    g = torch.Generator().manual_seed(79 + S)
    enc = torch.randn(B, S, 512, generator=g).cuda()
    with _lib.tuning_build() as lib:
        a = dec.beam_ids(enc, R.SOS, R.EOS, W, 15, return_scores=True)
        lib.capdec_beam_force_unfused(1)
        b = dec.beam_ids(enc, R.SOS, R.EOS, W, 15, return_scores=True)
    ra, rb = _rows(a[0], a[1]), _rows(b[0], b[1])
    assert sum(x != y for x, y in zip(ra, rb)) <= 1
    same = [i for i, (x, y) in enumerate(zip(ra, rb)) if x == y]
    close(a[2][same], b[2][same], rtol=1e-4, atol=1e-4, what="scores")


@pytest.mark.parametrize("L,S", [(2, 1), (2, 2), (1, 1)])
def test_cached_beam_follows_pad_hypotheses(L, S):
    """<PAD> tokens inside hypotheses (vocabulary bias raised): the cached search with the one-launch layers carries
    those positions along instead of handing the sample to the full-prefix form; hypotheses and scores against the
    full-prefix form and the oracle."""
    import ctypes
    from capdec import _lib, engine
    from gpu_util import cpu_sd, plain_decoder
    torch.manual_seed(11 + L)
    dec = plain_decoder(512, 1500, 4, L).cuda().eval()
    with torch.no_grad():
        dec.fc2.bias[R.PAD] += 1.5
    B, W, ML = 40, 3, 10
    g = torch.Generator().manual_seed(80 + S)
    enc = torch.randn(B, S, 512, generator=g)
    # the cached entry point alone: which samples does it report?
    lib = _lib.load()
    ref = engine.TfModelRef(dec)
    encd = enc.cuda().contiguous()
    ws = torch.empty(lib.capdec_tf_beam_cached_workspace_bytes(ctypes.byref(ref.c), B, W, S, ML), dtype=torch.uint8, device="cuda")
    ids = torch.empty((B, ML), dtype=torch.int64, device="cuda")
    ln = torch.empty((B,), dtype=torch.int32, device="cuda")
    sc = torch.empty((B, W), dtype=torch.float32, device="cuda")
    left = torch.empty((1,), dtype=torch.int32, device="cuda")
    pad = torch.zeros((B,), dtype=torch.int32, device="cuda")
    _lib.check(lib.capdec_tf_beam_search_cached(ctypes.byref(ref.c), encd.data_ptr(), B, W, S, ML, R.SOS, R.EOS, 0, ML,
                                                ws.data_ptr(), ws.numel(), ids.data_ptr(), ln.data_ptr(), sc.data_ptr(),
                                                left.data_ptr(), pad.data_ptr(), torch.cuda.current_stream().cuda_stream), "beam")
    reported = int(pad.sum())
    a = dec.beam_ids(encd, R.SOS, R.EOS, W, ML, return_scores=True)
    b = dec.beam_ids(encd, R.SOS, R.EOS, W, ML, return_scores=True, exact=True)
    ra, rb = _rows(a[0], a[1]), _rows(b[0], b[1])
    with_pad = sum(R.PAD in row for row in rb)
    print(f"L={L} S={S}: {with_pad} best hypotheses hold a <PAD>, {reported} samples reported by the cached search")
    assert with_pad >= 5 and reported <= B // 4
    assert sum(x != y for x, y in zip(ra, rb)) <= 2
    same = [i for i, (x, y) in enumerate(zip(ra, rb)) if x == y]
    close(a[2][same], b[2][same], rtol=1e-4, atol=1e-4, what="scores")
    want, _, want_scores = R.beam_search_ids(cpu_sd(dec), enc[:8], 4, W, ML, return_state=True)
    assert sum(x != y for x, y in zip(ra[:8], want)) <= 1


def _peaked_sd(fx, prefix, scale, eos_bias):
    sd = sub_sd(fx, prefix)
    sd["fc2.weight"] = sd["fc2.weight"] * float(scale)
    b = sd["fc2.bias"].clone()
    b[2] += float(eos_bias)
    sd["fc2.bias"] = b
    return sd


def test_topk_rows_orders_equal_values_as_torch_topk_on_the_cpu():
    """The selection routine of the beam search, alone: rows full of ties for every width up to 8."""
    import ctypes
    from capdec import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for W in range(1, 9):
        n = W * W
        rows = 512
        levels = torch.tensor([-0.5, -1.0, -2.25, -3.0, float("-inf")])
        x = levels[torch.randint(0, 5, (rows, n), generator=g)]
        x[: rows // 4] = torch.randn(rows // 4, n, generator=g)
        # beam-like rows: groups of W equal values next to groups of falling values
        grp = -5 * torch.rand(rows // 4, W, 1, generator=g)
        fall = grp - torch.cumsum(torch.rand(rows // 4, W, W, generator=g), dim=2)
        frozen = (torch.rand(rows // 4, W, 1, generator=g) < 0.5)
        x[rows // 4: rows // 2] = torch.where(frozen, grp.expand(-1, -1, W), fall).reshape(rows // 4, n)
        want_v, want_i = x.topk(W, dim=-1)
        xd = x.cuda().contiguous()
        ov = torch.empty((rows, W), dtype=torch.float32, device="cuda")
        oi = torch.empty((rows, W), dtype=torch.int32, device="cuda")
        _lib.check(lib.capdec_topk_rows(xd.data_ptr(), rows, n, W, ov.data_ptr(), oi.data_ptr(),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "capdec_topk_rows")
        assert torch.equal(oi.cpu().long(), want_i), W
        assert torch.equal(ov.cpu(), want_v), W


@pytest.mark.parametrize("exact", [False, True])
def test_f12_transformer_beam_search_after_eos(exact):
    """Peaked output distributions (hypotheses end at every length): captions and final scores of the reference's own
    run, widths 2, 3, 5, 8; and the one-sample search the reference leaves early (its state must stay frozen through
    the steps that were enqueued past that point)."""
    fx = load_golden("f12_beam_peaked")
    H = int(fx["H"])
    enc = t(fx["t_enc"])
    for n, (scale, eos_bias, W) in enumerate(fx["tf_cases"]):
        dec = decoder_from_sd(_peaked_sd(fx, "t__sd__", scale, eos_bias), H)
        ml = fx[f"t{n}_ids"].shape[1]
        ids, lengths, scores = dec.beam_ids(enc.cuda(), R.SOS, R.EOS, int(W), ml, return_scores=True, exact=exact)
        assert _rows(ids, lengths) == unpad(fx[f"t{n}_ids"]), (n, exact)
        close(scores, t(fx[f"t{n}_scores"]), rtol=1e-4, atol=1e-5, what=f"scores of case {n}")
    dec = decoder_from_sd(_peaked_sd(fx, "t__sd__", 6.0, 3.0), H)
    b, W = int(fx["te_row"]), int(fx["te_W"])
    ml = fx["te_ids"].shape[1]
    ids, lengths, scores = dec.beam_ids(enc[b:b + 1].cuda(), R.SOS, R.EOS, W, ml, return_scores=True, exact=exact)
    assert _rows(ids, lengths) == unpad(fx["te_ids"])
    close(scores, t(fx["te_scores"]), rtol=1e-4, atol=1e-5, what="scores of the search that ends early")


def test_f12_cnn_rnn_beam_search_after_eos():
    from models.cnnrnn_model import CNNtoRNN
    fx = load_golden("f12_beam_peaked")
    feats = t(fx["r_feats"]).cuda()
    for n, (scale, eos_bias, W) in enumerate(fx["rnn_cases"]):
        sd = _peaked_sd(fx, "r__sd__", scale, eos_bias)
        E, Hd, V = sd["fc1.weight"].shape[0], sd["lstm.weight_hh_l0"].shape[1], sd["embed.weight"].shape[0]
        m = CNNtoRNN(E, Hd, V)
        m.decoderRNN.load_state_dict(sd)
        m = m.cuda().eval()
        ml = fx[f"r{n}_ids"].shape[1]
        ids, lengths, scores = m.caption_ids_beam(feats, beam_width=int(W), max_length=ml, return_scores=True)
        assert _rows(ids, lengths) == unpad(fx[f"r{n}_ids"]), n
        close(scores, t(fx[f"r{n}_scores"]), rtol=1e-4, atol=1e-5, what=f"scores of case {n}")
