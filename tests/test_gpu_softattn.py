"""Soft-attention LSTM decoder (no counterpart in the reference: PARITY UNPINNED).  The HIP path is checked against
a plain-PyTorch evaluation of the same formulas (oracle/soft_attention_ref.py): logits within 1e-4 relative, greedy
ids identical up to near-ties."""
import numpy as np
import pytest
import torch

from gpu_util import REL, close
from oracle import soft_attention_ref as SR

pytestmark = pytest.mark.gpu


def _model(D, E, Hd, A, V, seed):
    from models.softattn_model import SoftAttentionLSTMDecoder
    torch.manual_seed(seed)
    return SoftAttentionLSTMDecoder(D, E, Hd, A, V)


@pytest.mark.parametrize("B,S,D,E,Hd,A,V", [(5, 7, 40, 32, 48, 24, 60), (3, 64, 128, 64, 64, 64, 200)])
def test_additive_attention_and_forward_logits(B, S, D, E, Hd, A, V):
    m = _model(D, E, Hd, A, V, 3).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    feats = torch.rand(B, S, D, generator=g)
    caps = torch.randint(4, V, (B, 6), generator=g)
    want = SR.forward_logits(sd, feats, caps)
    m = m.cuda()
    with torch.no_grad():
        got = m(feats.cuda(), caps.cuda())
    close(got, want, what="logits")
    # the attention kernel alone, with the weights
    from capdec import _lib
    lib = _lib.load()
    enc, pf, h0, _ = SR.prepare(sd, feats)
    ctx_want, alpha_want = SR.attend(sd, h0, enc, pf)
    u = SR.linear(h0, sd["att_h.weight"], sd["att_h.bias"]).cuda()
    ctx = torch.empty(B, E, device="cuda")
    alpha = torch.empty(B, S, device="cuda")
    encd, pfd, v = enc.contiguous().cuda(), pf.contiguous().cuda(), sd["att_v.weight"].reshape(-1).cuda()
    _lib.check(lib.capdec_additive_attention(u.data_ptr(), A, pfd.data_ptr(), v.data_ptr(), encd.data_ptr(), B, S, A, E,
                                             ctx.data_ptr(), E, alpha.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "capdec_additive_attention")
    close(ctx, ctx_want, what="context")
    close(alpha, alpha_want, what="alpha")


def test_soft_attention_greedy_decode():
    B, S, D, E, Hd, A, V = 8, 64, 256, 128, 128, 64, 300
    m = _model(D, E, Hd, A, V, 5).eval()
    with torch.no_grad():
        m.fc2.bias[2] += 1.5                         # some rows end early
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    feats = torch.rand(B, S, D, generator=g)
    want = SR.greedy_ids(sd, feats, 12)
    m = m.cuda()
    ids, lengths = m.caption_ids(feats.cuda(), max_length=12)
    from capdec.engine import ids_to_lists
    got = ids_to_lists(ids, lengths)
    for b, (gr, wr) in enumerate(zip(got, want)):
        if gr == wr:
            continue
        n = min(len(gr), len(wr))
        s = next((i for i in range(n) if gr[i] != wr[i]), n)
        pre = torch.tensor([[SR.SOS] + wr[:s]], dtype=torch.long)
        lg = SR.forward_logits(sd, feats[b:b + 1], pre)[0, -1]
        tok = gr[s] if s < len(gr) else SR.EOS
        assert float(lg.max() - lg[tok]) <= REL * float(lg.abs().max()), f"row {b} step {s}"
    assert any(len(r) < 12 for r in got)
