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


def _caps_with_pad(g, B, T, V):
    caps = torch.zeros(B, T + 1, dtype=torch.long)
    for b in range(B):
        n = T - 1 if b == 0 else int(torch.randint(1, T - 1, (1,), generator=g))
        caps[b, 0] = 1
        caps[b, 1:1 + n] = torch.randint(4, V, (n,), generator=g)
        caps[b, 1 + n] = 2
    return caps


def _oracle_loss_and_grads(sd0, feats, caps):
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    logits = SR.forward_logits(sd, feats, caps[:, :-1])
    V = logits.shape[-1]
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, V), caps[:, 1:].reshape(-1), ignore_index=0)
    loss.backward()
    return logits.detach(), float(loss.detach()), {k: v.grad for k, v in sd.items()}


@pytest.mark.parametrize("B,S,D,E,Hd,A,V,T", [(6, 7, 40, 32, 48, 24, 60, 8), (4, 64, 96, 64, 64, 128, 150, 10),
                                              (3, 5, 32, 32, 32, 1024, 40, 5)])
def test_soft_attention_training_gradients_match_autograd_of_the_formulas(B, S, D, E, Hd, A, V, T):
    """Teacher-forced logits, loss and EVERY parameter gradient (attention scores, keys, query projection, initial-state
    projections, memory projection, LSTM, embedding, output layer) against torch autograd on the plain-PyTorch
    evaluation of the same formulas; through loss.backward() and through the fused-loss training step."""
    from capdec import autograd as AG
    m = _model(D, E, Hd, A, V, 7)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(8)
    feats = torch.rand(B, S, D, generator=g)
    caps = _caps_with_pad(g, B, T, V)
    want_logits, want_loss, want = _oracle_loss_and_grads(sd0, feats, caps)
    m = m.cuda().train()
    fd, cd = feats.cuda(), caps.cuda()
    logits = m(fd, cd[:, :-1])
    close(logits, want_logits, what="training-mode logits")
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, V), cd[:, 1:].reshape(-1), ignore_index=0)
    assert abs(float(loss.item()) - want_loss) <= 1e-5 * abs(want_loss)
    loss.backward()
    big = max(float(v.abs().max()) for v in want.values())
    for k, prm in m.named_parameters():
        close(prm.grad, want[k], atol=1e-6 * big, what=f"grad {k} (autograd bridge)")
    grads = {}
    loss2 = AG.satt_train_step(m, fd, cd, grads)
    assert abs(float(loss2.item()) - want_loss) <= 1e-5 * abs(want_loss)
    assert set(grads) == set(want)
    for k in want:
        close(grads[k], want[k], atol=1e-6 * big, what=f"grad {k} (fused-loss step)")


def test_soft_attention_without_context_is_the_plain_lstm_decoder():
    """With the context half of the LSTM's input weights at zero the attention operand has no effect on the outputs:
    the same step kernels over the same tape give the logits of an LSTM over the embeddings alone (the oracle's
    lstm_cell, which the DecoderRNN golden vectors pin), and the attention parameters receive exactly-zero gradients."""
    B, S, D, E, Hd, A, V, T = 5, 9, 24, 32, 40, 16, 50, 7
    m = _model(D, E, Hd, A, V, 9)
    with torch.no_grad():
        m.lstm.weight_ih_l0[:, E:] = 0.0
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(10)
    feats = torch.rand(B, S, D, generator=g)
    caps = _caps_with_pad(g, B, T, V)
    enc, pf, h, c = SR.prepare(sd, feats)
    w = (sd["lstm.weight_ih_l0"][:, :E].contiguous(), sd["lstm.weight_hh_l0"], sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"])
    outs = []
    for t in range(T):
        h, c = SR.lstm_cell(sd["embed.weight"][caps[:, t]], h, c, *w)
        outs.append(h)
    want = SR.linear(torch.stack(outs, 1), sd["fc2.weight"], sd["fc2.bias"])
    m = m.cuda().train()
    logits = m(feats.cuda(), caps[:, :-1].cuda())
    close(logits, want, what="logits without context")
    logits.square().mean().backward()
    for k in ("att_h.weight", "att_h.bias", "att_v.weight", "att_f.weight", "att_f.bias"):
        assert float(dict(m.named_parameters())[k].grad.abs().max()) == 0.0, k
