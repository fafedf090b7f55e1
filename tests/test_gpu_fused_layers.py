"""All decoder layers of a cached decode step as one launch (csrc/decode_fused.hip; taken by itself at E = 512, H = 4,
at most two layers, one memory row) against the launch-per-operation form and the oracle (transformer.py:141-175).
Generated <PAD> tokens are the hard case: the reference lets a <PAD> query row attend uniformly over the CURRENT length,
which the one-launch form follows by carrying the <PAD> positions along as side rows of their sample's team."""
import warnings

import pytest
import torch

from gpu_util import check_greedy, cpu_sd, plain_decoder
from oracle import decoder_ref as R

pytestmark = pytest.mark.gpu

E, H = 512, 4


def _decoder(L, V, seed, pad_bias=0.0, eos_bias=0.0):
    torch.manual_seed(seed)
    dec = plain_decoder(E, V, H, L).cuda().eval()
    with torch.no_grad():
        dec.fc2.bias[R.PAD] += pad_bias
        dec.fc2.bias[R.EOS] += eos_bias
    return dec


def _ids(dec, enc, max_len, **kw):
    from capdec.engine import ids_to_lists, tf_greedy_ids
    with warnings.catch_warnings():
        warnings.simplefilter("error")            # a wait of the one-launch form that runs out warns: never expected
        ids, lengths, n_dirty = tf_greedy_ids(dec, enc.cuda(), max_len, R.SOS, R.EOS, **kw)
    return ids_to_lists(ids, lengths), n_dirty


@pytest.mark.parametrize("L,B", [(2, 1), (2, 33), (2, 100), (1, 40)])
def test_one_launch_layers_follow_generated_pad_tokens(L, B):
    V, ML = 2000, 12
    dec = _decoder(L, V, 40 + L, pad_bias=1.2)
    g = torch.Generator().manual_seed(B)
    enc = torch.randn(B, 1, E, generator=g)
    fused, nd = _ids(dec, enc, ML)
    # (a team of 16 samples carries up to 16 <PAD> positions; beyond that the surplus samples go to the exact path)
    pads_per_team = [sum(row.count(R.PAD) for row in fused[i:i + 16]) for i in range(0, B, 16)]
    print(f"L={L} B={B}: <PAD> positions per team {pads_per_team}, {nd} samples through the exact path")
    if L == 1 or max(pads_per_team) <= 16:
        assert nd == 0
    plain, _ = _ids(dec, enc, ML, fused_layers=False)
    sd = cpu_sd(dec)
    ties = check_greedy(fused, sd, enc, H, ML)
    check_greedy(plain, sd, enc, H, ML)
    assert sum(a == b for a, b in zip(fused, plain)) >= B - max(1, ties)
    if B >= 33:
        # the point of the test: <PAD> in the middle of captions that go on
        assert sum(R.PAD in row[:-2] for row in fused) >= (2 if B < 64 else 4), [row for row in fused][:4]


@pytest.mark.parametrize("L,B", [(2, 70), (1, 20)])
def test_one_launch_layers_with_two_memory_rows(L, B):
    """Two memory rows (the ViT+CNN and YOLO+CNN heads): the logistic closed form of the cross-attention block inside the
    one-launch layers, main rows and <PAD> side rows alike, against the launch-per-operation form, the general
    cross-attention kernels and the oracle."""
    V, ML = 2000, 12
    dec = _decoder(L, V, 50 + L, pad_bias=1.2)
    g = torch.Generator().manual_seed(B + 1)
    enc = torch.randn(B, 2, E, generator=g)
    fused, nd = _ids(dec, enc, ML)
    plain, _ = _ids(dec, enc, ML, fused_layers=False)
    general, _ = _ids(dec, enc, ML, no_s1_shortcut=True)
    sd = cpu_sd(dec)
    ties = check_greedy(fused, sd, enc, H, ML)
    check_greedy(plain, sd, enc, H, ML)
    assert sum(a == b for a, b in zip(fused, plain)) >= B - max(1, ties)
    assert sum(a == b for a, b in zip(fused, general)) >= B - max(1, ties)
    if L == 2:
        assert sum(R.PAD in row[:-2] for row in fused) >= 2


def test_one_launch_layers_rows_of_the_last_step_and_bit_identical_runs():
    """The rows handed to the vocabulary projection agree with the launch-per-operation form to rounding, and two runs
    agree bit for bit (fixed summation orders, no float atomics)."""
    from capdec.engine import tf_greedy_ids
    V, B = 3000, 70
    dec = _decoder(2, V, 7, pad_bias=1.2)
    g = torch.Generator().manual_seed(5)
    enc = torch.randn(B, 1, E, generator=g).cuda()
    for n in (1, 2, 5):
        wa, wb, wc = [], [], []
        tf_greedy_ids(dec, enc, n, R.SOS, R.EOS, fused_layers=False, keep_workspace=wa)
        tf_greedy_ids(dec, enc, n, R.SOS, R.EOS, keep_workspace=wb)
        tf_greedy_ids(dec, enc, n, R.SOS, R.EOS, keep_workspace=wc)
        torch.cuda.synchronize()
        xa, xb, xc = (w[0][:B * E * 4].view(torch.float32).view(B, E).cpu() for w in (wa, wb, wc))
        assert torch.equal(xb, xc)
        err = float((xa - xb).abs().max()) / float(xa.abs().max())
        assert err <= 1e-5, (n, err)


def test_one_launch_layers_more_pad_positions_than_side_rows():
    """More than 16 <PAD> positions among the 16 samples of a team: the surplus samples are flagged and re-decoded on the
    exact path; the captions are the oracle's either way."""
    V, B, ML = 500, 64, 10
    dec = _decoder(2, V, 3, pad_bias=1.5)
    g = torch.Generator().manual_seed(8)
    enc = torch.randn(B, 1, E, generator=g)
    got, n_dirty = _ids(dec, enc, ML)
    assert n_dirty > 0
    assert max(sum(row.count(R.PAD) for row in got[i:i + 16]) for i in range(0, B, 16)) > 16
    assert len({tuple(row) for row in got}) > B // 2          # (not a degenerate all-<PAD> batch)
    check_greedy(got, cpu_sd(dec), enc, H, ML)


def test_one_launch_layers_all_captions_finish_and_ragged_batches():
    """Captions that end: the device-side end of the loop (a row count of zero) reaches the one-launch form too; batch
    sizes that are no multiple of a team's 16 rows."""
    V, ML = 1000, 25
    dec = _decoder(2, V, 11, eos_bias=1.4, pad_bias=0.8)
    for B in (5, 31, 65):
        g = torch.Generator().manual_seed(B)
        enc = torch.randn(B, 1, E, generator=g)
        got, _ = _ids(dec, enc, ML)
        plain, _ = _ids(dec, enc, ML, fused_layers=False)
        ties = check_greedy(got, cpu_sd(dec), enc, H, ML)
        assert sum(a == b for a, b in zip(got, plain)) >= B - max(1, ties)
        assert min(len(r) for r in got) < ML


def test_one_launch_layers_at_the_benchmark_batch():
    V, B, ML = 10000, 1024, 20
    dec = _decoder(2, V, 0)
    g = torch.Generator().manual_seed(1236)
    enc = torch.randn(B, 1, E, generator=g)
    fused, nd = _ids(dec, enc, ML)
    plain, _ = _ids(dec, enc, ML, fused_layers=False)
    assert nd == 0
    same = sum(a == b for a, b in zip(fused, plain))
    assert same >= B - 4, same
    check_greedy(fused[:40], cpu_sd(dec), enc[:40], H, ML)


def test_one_launch_layers_bit_identical_under_uneven_load():
    """The hand-offs between the workgroups of a team must not depend on what else the chip is doing: the same decode,
    alone and beside a stream of large contractions that come and go (other CUs busy at times, two teams per CU at
    1024 rows, caches warm from the previous run), gives the same rows bit for bit and the same captions."""
    from capdec.engine import tf_greedy_ids
    V, B, ML = 3000, 1024, 12
    dec = _decoder(2, V, 9, pad_bias=0.9)
    g = torch.Generator().manual_seed(15)
    enc = torch.randn(B, 1, E, generator=g).cuda()
    quiet = []
    ids0, len0, _ = tf_greedy_ids(dec, enc, ML, R.SOS, R.EOS, keep_workspace=quiet)
    torch.cuda.synchronize()
    x0 = quiet[0][:B * E * 4].clone()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    b = torch.randn(4096, 4096, device="cuda")
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(3 + 2 * rep):
                a @ b                                   # bursts of different lengths beside the decode
        ws = []
        ids, lens, _ = tf_greedy_ids(dec, enc, ML, R.SOS, R.EOS, keep_workspace=ws)
        torch.cuda.synchronize()
        assert torch.equal(ids, ids0) and torch.equal(lens, len0), rep
        assert torch.equal(ws[0][:B * E * 4], x0), rep


def test_one_launch_layers_at_the_longest_supported_captions():
    """max_length 64 (the limit of the one-launch form): every pass of the attention stage over 16 keys and every block
    of values is exercised; captions against the launch-per-operation form and, for a few rows, the oracle."""
    V, B, ML = 800, 24, 64
    torch.manual_seed(21)
    dec = plain_decoder(E, V, H, 2, max_len=ML).cuda().eval()
    with torch.no_grad():
        dec.fc2.bias[R.PAD] += 0.6
        dec.fc2.bias[R.EOS] -= 5.0
    g = torch.Generator().manual_seed(64)
    enc = torch.randn(B, 1, E, generator=g)
    fused, _ = _ids(dec, enc, ML)
    plain, _ = _ids(dec, enc, ML, fused_layers=False)
    assert max(len(r) for r in fused) == ML
    ties = check_greedy(fused[:6], cpu_sd(dec), enc[:6], H, ML)
    assert sum(a == b for a, b in zip(fused, plain)) >= B - max(2, ties)


def test_greedy_decode_enqueued_ahead_equals_the_synchronous_call():
    """tf_greedy_ids_async: three batches enqueued before the first result is asked for, one of them with samples the
    cached decode hands to the exact path; ids and lengths equal the synchronous wrapper's."""
    from capdec.engine import TfModelRef, tf_greedy_ids, tf_greedy_ids_async
    V, ML = 500, 10
    dec = _decoder(2, V, 3, pad_bias=1.5)
    ref = TfModelRef(dec)
    g = torch.Generator().manual_seed(8)
    encs = [torch.randn(b, 1, E, generator=g).cuda() for b in (64, 17, 40)]
    pending = [tf_greedy_ids_async(ref, e, ML, R.SOS, R.EOS) for e in encs]
    redone = 0
    for e, p in zip(encs, pending):
        ids, lens, nd = p.result()
        want = tf_greedy_ids(ref, e, ML, R.SOS, R.EOS)
        assert torch.equal(ids, want[0]) and torch.equal(lens, want[1]) and nd == want[2]
        redone += nd
    assert redone > 0


def test_one_launch_layers_random_shapes_repeat_bit_for_bit_beside_other_work():
    """A short form of tools/exp/fused_soak.py: random batch sizes (1 ... 2048 rows: one and two workgroups per CU,
    several dispatch rounds), lengths, layer counts and memory rows; every decode is repeated beside bursts of large
    contractions on a second stream and must give the same ids, lengths and final rows bit for bit; against the
    launch-per-operation form only arg-max near-ties may differ."""
    import random
    from capdec.engine import tf_greedy_ids
    rng = random.Random(7)
    a = torch.randn(2048, 2048, device="cuda")
    side = torch.cuda.Stream()
    decs = {}
    differing = rows = 0
    for case in range(36):
        L = rng.choice((1, 2, 2))
        S = rng.choice((1, 1, 2))
        V = rng.choice((500, 3000))
        B = rng.choice((1, 7, 16, 17, 64, 100, 256, 513, 777, 1024, 1500, 2048))
        ML = rng.choice((2, 5, 12, 20, 33, 50))
        pad_bias = rng.choice((0.0, 0.8, 1.3))
        key = (L, V, pad_bias)
        if key not in decs:
            torch.manual_seed(200 + len(decs))
            d = plain_decoder(E, V, H, L, max_len=50).cuda().eval()
            with torch.no_grad():
                d.fc2.bias[R.PAD] += pad_bias
            decs[key] = d
        dec = decs[key]
        g = torch.Generator().manual_seed(case)
        enc = torch.randn(B, S, E, generator=g).cuda()
        w0, w1 = [], []
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            ids0, len0, nd0 = tf_greedy_ids(dec, enc, ML, R.SOS, R.EOS, keep_workspace=w0)
            torch.cuda.synchronize()
            x0 = w0[0][:B * E * 4].clone()
            with torch.cuda.stream(side):
                for _ in range(rng.randint(1, 6)):
                    a @ a
            ids1, len1, nd1 = tf_greedy_ids(dec, enc, ML, R.SOS, R.EOS, keep_workspace=w1)
            torch.cuda.synchronize()
            idp, lenp, _ = tf_greedy_ids(dec, enc, ML, R.SOS, R.EOS, fused_layers=False)
        what = (case, L, S, V, B, ML, pad_bias)
        assert torch.equal(ids0, ids1) and torch.equal(len0, len1) and nd0 == nd1, what
        assert torch.equal(w1[0][:B * E * 4], x0), what
        d = int(((ids0 != idp).any(dim=1) | (len0 != lenp)).sum())
        assert d <= max(2, B // 100), what + (d,)
        differing += d
        rows += B
    assert differing <= rows // 2000 + 2, (differing, rows)
