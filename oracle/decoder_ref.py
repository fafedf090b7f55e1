"""CPU restatement (PyTorch fp32, functional) of the reference caption decoders.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Parity status: pinned
by ``tests/golden/*.npz`` (generated from the reference's own classes).

Every function works on a plain ``dict`` of tensors keyed by the reference's
``state_dict`` names, so no ``nn.Module`` of the reference is restated as a
class.  Citations are ``file:line`` under the reference checkout.

Dropout is the identity everywhere here (``model.eval()`` or ``dropout=0``):
dropout parity is statistical only and is not part of this oracle.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

SD = Dict[str, torch.Tensor]

PAD, SOS, EOS, UNK = 0, 1, 2, 3  # get_loader.py:28-29


# --------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------
def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``nn.Linear``: y = x W^T + b, W is [out, in]."""
    y = x.matmul(w.t())
    return y if b is None else y + b


def layer_norm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """``nn.LayerNorm(E)`` over the last dim, biased variance (transformer.py:78-80)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * g + b


def batchnorm1d(x: torch.Tensor, sd: SD, prefix: str, training: bool,
                momentum: float = 0.1, eps: float = 1e-5) -> Tuple[torch.Tensor, Optional[SD]]:
    """``nn.BatchNorm1d`` on [B, C] (cnnattn_model.py:13,29; cnnrnn_model.py:14,24).

    eval: running statistics.  train: biased batch variance for the
    normalisation, unbiased for the running update, momentum 0.1.  Returns the
    output and, in training, the updated running buffers (not applied in place).
    """
    g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if not training:
        rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
        return (x - rm) / torch.sqrt(rv + eps) * g + b, None
    n = x.shape[0]
    mu = x.mean(dim=0)
    var_b = ((x - mu) ** 2).mean(dim=0)
    y = (x - mu) / torch.sqrt(var_b + eps) * g + b
    with torch.no_grad():
        var_u = var_b * (n / max(n - 1, 1))
        new = {
            prefix + ".running_mean": (1 - momentum) * sd[prefix + ".running_mean"] + momentum * mu,
            prefix + ".running_var": (1 - momentum) * sd[prefix + ".running_var"] + momentum * var_u,
            prefix + ".num_batches_tracked": sd[prefix + ".num_batches_tracked"] + 1,
        }
    return y, new


def positional_encoding_table(max_len: int, E: int) -> torch.Tensor:
    """Sinusoid buffer ``pe[max_len, E]`` (transformer.py:60-65)."""
    pe = torch.zeros(max_len, E)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, E, 2).float() * -(math.log(10000.0) / E))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


# --------------------------------------------------------------------------
# Transformer decoder (models/modules/transformer.py)
# --------------------------------------------------------------------------
def tgt_mask_from_ids(ids: torch.Tensor) -> torch.Tensor:
    """``generate_mask`` target half (transformer.py:117-123).

    mask[b,0,i,j] = (ids[b,i] != 0) and (j <= i): PAD masks the *query row* i.
    """
    T = ids.shape[1]
    row_ok = (ids != 0).unsqueeze(1).unsqueeze(3)             # [B,1,T,1]
    causal = torch.tril(torch.ones(1, T, T, dtype=torch.bool))  # j <= i
    return row_ok & causal                                     # [B,1,T,T]


def mha(sd: SD, prefix: str, q_in: torch.Tensor, kv_in: torch.Tensor, H: int,
        mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``MultiHeadAttention.forward`` (transformer.py:21-44).

    Scores are divided by sqrt(d_k) *after* the matmul; masked entries are set
    to -1e9 (not -inf), so a fully masked row softmaxes to uniform.
    """
    B, Tq, E = q_in.shape
    Tk = kv_in.shape[1]
    dk = E // H
    q = linear(q_in, sd[prefix + ".W_q.weight"], sd[prefix + ".W_q.bias"]).view(B, Tq, H, dk).transpose(1, 2)
    k = linear(kv_in, sd[prefix + ".W_k.weight"], sd[prefix + ".W_k.bias"]).view(B, Tk, H, dk).transpose(1, 2)
    v = linear(kv_in, sd[prefix + ".W_v.weight"], sd[prefix + ".W_v.bias"]).view(B, Tk, H, dk).transpose(1, 2)
    s = q.matmul(k.transpose(-2, -1)) / math.sqrt(dk)
    if mask is not None:
        s = s.masked_fill(mask == 0, -1e9)
    p = torch.softmax(s, dim=-1)
    o = p.matmul(v).transpose(1, 2).contiguous().view(B, Tq, E)
    return linear(o, sd[prefix + ".W_o.weight"], sd[prefix + ".W_o.bias"])


def decoder_layer(sd: SD, l: int, x: torch.Tensor, enc: torch.Tensor, tgt_mask: torch.Tensor, H: int) -> torch.Tensor:
    """``DecoderLayer.forward``: post-LN, d_ff = E (transformer.py:83-90, :96)."""
    p = f"decoder_layers.{l}"
    a = mha(sd, p + ".self_attn", x, x, H, tgt_mask)
    x = layer_norm(x + a, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    a = mha(sd, p + ".cross_attn", x, enc, H, None)            # cross-attn never masked (:86)
    x = layer_norm(x + a, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    f = linear(torch.relu(linear(x, sd[p + ".feed_forward.fc1.weight"], sd[p + ".feed_forward.fc1.bias"])),
               sd[p + ".feed_forward.fc2.weight"], sd[p + ".feed_forward.fc2.bias"])
    return layer_norm(x + f, sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])


def num_layers_of(sd: SD) -> int:
    n = 0
    while f"decoder_layers.{n}.norm1.weight" in sd:
        n += 1
    return n


def decoder_forward(sd: SD, enc: torch.Tensor, ids: torch.Tensor, H: int, take_last: bool = False) -> torch.Tensor:
    """``DecoderTransformer.decoder_forward`` (transformer.py:125-138).

    enc [B,S,E] fp32, ids [B,T] int64 -> logits [B,T,V] (or [B,V] with take_last).
    """
    T = ids.shape[1]
    pe = sd["positional_encoding.pe"][0]                       # [max_len, E]
    x = sd["decoder_embedding.weight"][ids] + pe[:T]
    mask = tgt_mask_from_ids(ids)
    for l in range(num_layers_of(sd)):
        x = decoder_layer(sd, l, x, enc, mask, H)
    if take_last:
        x = x[:, -1, :]
    return linear(x, sd["fc2.weight"], sd["fc2.bias"])


def greedy_inference_ids(sd: SD, enc: torch.Tensor, H: int, max_length: int,
                         sos: int = SOS, eos: int = EOS) -> List[List[int]]:
    """``DecoderTransformer.greedy_inference`` up to detokenisation (transformer.py:141-171).

    Full-prefix recompute every step; finished rows keep feeding their own
    predictions back; a row's result is every predicted id before its first
    <EOS>; the loop stops early only when all rows are done.
    """
    B = enc.shape[0]
    caption = torch.full((B, 1), sos, dtype=torch.long)
    result: List[List[int]] = [[] for _ in range(B)]
    done = [False] * B
    for _ in range(max_length):
        out = decoder_forward(sd, enc, caption, H, take_last=True)
        pred = out.argmax(dim=-1)
        for i in range(B):
            if not done[i]:
                t = int(pred[i].item())
                if t == eos:
                    done[i] = True
                else:
                    result[i].append(t)
        if all(done):
            break
        caption = torch.cat([caption, pred.unsqueeze(1)], dim=1)
    return result


def detokenize(ids: Sequence[Sequence[int]], itos) -> List[str]:
    """transformer.py:174 / cnnrnn_model.py:109."""
    return [" ".join(itos[i] for i in row) for row in ids]


# --------------------------------------------------------------------------
# feature heads of the Transformer wrappers
# --------------------------------------------------------------------------
def head_single(sd: SD, feats: torch.Tensor, fc: str, training: bool = False):
    """fc -> dropout(identity) -> BatchNorm1d(E) -> [B,1,E].

    fc = "fc1" (no bias, cnnattn_model.py:27-30), "fc_vit" (vitattn_model.py:43-47)
    or "fc_yolo" (yoloattn_model.py:70-74).
    """
    y = linear(feats, sd[fc + ".weight"], sd.get(fc + ".bias"))
    y, new = batchnorm1d(y, sd, "batchnorm", training)
    return y.unsqueeze(1), new


def head_pair(sd: SD, feats: torch.Tensor, fc_a: str, split: int, training: bool = False):
    """Two heads on a column split, concat, BatchNorm1d(2E), reshape [B,2,E].

    ("fc_vit", 1000): vitcnnattn_model.py:47-53; ("fc_yolo", 86436): yolocnnattn_model.py:75-81.
    Row 0 of the memory is the first head, row 1 is fc_cnn.
    """
    a = linear(feats[:, :split], sd[fc_a + ".weight"], sd[fc_a + ".bias"])
    c = linear(feats[:, split:], sd["fc_cnn.weight"], sd["fc_cnn.bias"])
    y = torch.cat((a, c), dim=1)
    y, new = batchnorm1d(y, sd, "batchnorm", training)
    return y.reshape(y.shape[0], 2, -1), new


# --------------------------------------------------------------------------
# CNN-RNN (models/cnnrnn_model.py); keys are those under "decoderRNN."
# --------------------------------------------------------------------------
def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """One ``nn.LSTM`` step, gate order i,f,g,o (cnnrnn_model.py:17)."""
    Hd = h.shape[1]
    g = linear(x, w_ih, b_ih) + linear(h, w_hh, b_hh)
    i, f, gg, o = g[:, :Hd], g[:, Hd:2 * Hd], g[:, 2 * Hd:3 * Hd], g[:, 3 * Hd:]
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def _lstm_w(sd: SD):
    return (sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"], sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"])


def rnn_image_state(sd: SD, feats: torch.Tensor, training: bool = False):
    """dropout(identity) -> fc1 (no bias) -> BatchNorm1d -> one LSTM step from zero state
    (cnnrnn_model.py:22-25,29 and :69-75)."""
    x = linear(feats, sd["fc1.weight"])
    x, new = batchnorm1d(x, sd, "batchnorm", training)
    Hd = sd["lstm.weight_hh_l0"].shape[1]
    z = torch.zeros(x.shape[0], Hd)
    h, c = lstm_cell(x, z, z, *_lstm_w(sd))
    return h, c, new


def decoder_rnn_forward(sd: SD, feats: torch.Tensor, ids: torch.Tensor, training: bool = False,
                        with_buffers: bool = False):
    """``DecoderRNN.forward`` (cnnrnn_model.py:20-34): logits [B,T,V]; with ``with_buffers`` also the updated
    BatchNorm buffers of a training-mode call."""
    h, c, new = rnn_image_state(sd, feats, training)
    emb = sd["embed.weight"][ids]
    outs = []
    for t in range(ids.shape[1]):
        h, c = lstm_cell(emb[:, t], h, c, *_lstm_w(sd))
        outs.append(h)
    logits = linear(torch.stack(outs, dim=1), sd["fc2.weight"], sd["fc2.bias"])
    return (logits, new) if with_buffers else logits


def cnnrnn_logits_after(sd: SD, feats: torch.Tensor, prefix: List[int], sos: int = SOS) -> torch.Tensor:
    """Logits [B,V] of the greedy loop (cnnrnn_model.py:77-87) after feeding <SOS> and then ``prefix`` (the same
    tokens for every row): what the reference ranks when it picks token number len(prefix)."""
    B = feats.shape[0]
    h, c, _ = rnn_image_state(sd, feats, False)
    out = None
    for tok in [sos] + list(prefix):
        emb = sd["embed.weight"][torch.full((B,), int(tok), dtype=torch.long)]
        h, c = lstm_cell(emb, h, c, *_lstm_w(sd))
        out = linear(h, sd["fc2.weight"], sd["fc2.bias"])
    return out


def cnnrnn_greedy_ids(sd: SD, feats: torch.Tensor, max_length: int, sos: int = SOS, eos: int = EOS) -> List[List[int]]:
    """``CNNtoRNN.caption_images`` up to detokenisation (cnnrnn_model.py:57-106)."""
    B = feats.shape[0]
    h, c, _ = rnn_image_state(sd, feats, False)
    emb = sd["embed.weight"][torch.full((B,), sos, dtype=torch.long)]
    result: List[List[int]] = [[] for _ in range(B)]
    done = [False] * B
    for _ in range(max_length):
        h, c = lstm_cell(emb, h, c, *_lstm_w(sd))
        out = linear(h, sd["fc2.weight"], sd["fc2.bias"])
        pred = torch.log_softmax(out, dim=-1).argmax(dim=-1)
        for i in range(B):
            if not done[i]:
                t = int(pred[i].item())
                if t == eos:
                    done[i] = True
                else:
                    result[i].append(t)
        if all(done):
            break
        emb = sd["embed.weight"][pred]
    return result


# --------------------------------------------------------------------------
# loss (train.py:216,252-256)
# --------------------------------------------------------------------------
# --------------------------------------------------------------------------
# beam search (transformer.py:177-244, cnnrnn_model.py:112-209): identical selection logic in both
# --------------------------------------------------------------------------
def beam_select(step: int, scores: torch.Tensor, fin: torch.Tensor, logits: torch.Tensor, W: int):
    """One selection step.  scores [B,W]; fin [B,W]: the hypothesis' LAST token is <EOS>; logits [B,W,V].

    Each hypothesis proposes its W most probable next tokens (log-softmax, ``topk``).  What the reference's code does
    with a hypothesis that has produced <EOS> is narrower than its comments say.  ``mask = done.unsqueeze(-1)
    .expand(-1, -1, beam_width)`` is a stride-0 VIEW of ``done``, so ``mask[:, :, 0] = False`` (transformer.py:199-201,
    cnnrnn_model.py:161-163) clears ``done`` itself, for every hypothesis, at every step.  Consequences, all reproduced
    here:
      * ``done`` only ever holds "the token chosen at the previous step was <EOS>" (transformer.py:226-227 ORs the new
        <EOS> flags into an all-False tensor);
      * in the one step after its <EOS> a hypothesis adds ``(1 - 1) * log_prob`` (transformer.py:194-196, evaluated before
        the write): its W proposals all carry its unchanged score and ALL stay eligible -- the mask is all False by then,
        so nothing is set to -inf except at the first step (transformer.py:202-205: only hypothesis 0 expands);
      * from the next step on it accumulates log-probabilities again;
      * ``if done.all(): break`` (transformer.py:231) fires only when every hypothesis of every sample produced <EOS>
        in the same step.
    The best W of the W*W candidates survive, ordered as ``torch.topk`` orders them (among equal scores: the CPU
    kernel's order, restated in oracle/topk_order.py).
    Returns (new scores [B,W], parent hypothesis [B,W], new token [B,W]).
    (transformer.py:189-219 == cnnrnn_model.py:151-180)"""
    B = scores.shape[0]
    lp, tok = torch.log_softmax(logits, dim=-1).topk(W, dim=2)                       # [B,W,W]
    cand = scores.unsqueeze(-1) + (1.0 - fin.unsqueeze(-1).float()) * lp
    if step == 0:
        cand[:, 1:, :] = float("-inf")
    cand = cand.reshape(B, W * W)
    new_scores, flat = cand.topk(W, dim=-1)
    parent = flat // W
    new_tok = tok.reshape(B, W * W).gather(1, flat)
    return new_scores, parent, new_tok


def _beam_result(seqs: torch.Tensor, eos: int) -> List[List[int]]:
    """Best hypothesis of every sample, without <SOS>, cut at its first <EOS> (transformer.py:233-241)."""
    out = []
    for row in seqs[:, 0].tolist():
        out.append(row[1:row.index(eos)] if eos in row else row[1:])
    return out


def beam_search_ids(sd: SD, enc: torch.Tensor, H: int, W: int, max_length: int, sos: int = SOS, eos: int = EOS,
                    return_state: bool = False):
    """``DecoderTransformer.beam_search_inference`` up to detokenisation (transformer.py:177-244).
    enc [B,S,E] (one memory per sample; the reference's callers repeat it W times, e.g. cnnattn_model.py:65-66).
    Full-prefix recompute of all B*W hypotheses every step; every hypothesis keeps being extended whether or not it
    contains <EOS> (see ``beam_select`` for what <EOS> does to the scores); the loop stops early only when all B*W
    hypotheses produce <EOS> in the same step."""
    B = enc.shape[0]
    enc_w = enc.unsqueeze(1).expand(-1, W, -1, -1).reshape(B * W, enc.shape[1], enc.shape[2])
    seqs = torch.full((B, W, 1), sos, dtype=torch.long)
    scores = torch.zeros(B, W)
    done = torch.zeros(B, W, dtype=torch.bool)
    for step in range(max_length):
        logits = decoder_forward(sd, enc_w, seqs.reshape(B * W, -1), H, take_last=True).reshape(B, W, -1)
        scores, parent, new_tok = beam_select(step, scores, done, logits, W)
        seqs = torch.cat([seqs.gather(1, parent.unsqueeze(-1).expand(-1, -1, seqs.shape[2])), new_tok.unsqueeze(-1)], dim=2)
        done = new_tok == eos          # transformer.py:226-227 on a `done` that :201 has cleared
        if bool(done.all()):
            break
    res = _beam_result(seqs, eos)
    return (res, seqs, scores) if return_state else res


def cnnrnn_beam_search_ids(sd: SD, feats: torch.Tensor, W: int, max_length: int, sos: int = SOS, eos: int = EOS,
                           return_state: bool = False):
    """``CNNtoRNN.caption_images_beam_search`` up to detokenisation (cnnrnn_model.py:112-209): the LSTM state of
    every hypothesis follows its parent."""
    B = feats.shape[0]
    h, c, _ = rnn_image_state(sd, feats, False)
    Hd = h.shape[1]
    h = h.unsqueeze(1).expand(-1, W, -1).reshape(B * W, Hd)
    c = c.unsqueeze(1).expand(-1, W, -1).reshape(B * W, Hd)
    seqs = torch.full((B, W, 1), sos, dtype=torch.long)
    scores = torch.zeros(B, W)
    done = torch.zeros(B, W, dtype=torch.bool)
    for step in range(max_length):
        emb = sd["embed.weight"][seqs[:, :, -1].reshape(B * W)]
        h, c = lstm_cell(emb, h, c, *_lstm_w(sd))
        logits = linear(h, sd["fc2.weight"], sd["fc2.bias"]).reshape(B, W, -1)
        scores, parent, new_tok = beam_select(step, scores, done, logits, W)
        seqs = torch.cat([seqs.gather(1, parent.unsqueeze(-1).expand(-1, -1, seqs.shape[2])), new_tok.unsqueeze(-1)], dim=2)
        idx = (torch.arange(B).unsqueeze(1) * W + parent).reshape(B * W)
        h, c = h[idx], c[idx]
        done = new_tok == eos          # cnnrnn_model.py:191-192 on a `done` that :163 has cleared
        if bool(done.all()):
            break
    res = _beam_result(seqs, eos)
    return (res, seqs, scores) if return_state else res


def caption_loss(logits: torch.Tensor, targets: torch.Tensor, pad: int = PAD) -> torch.Tensor:
    """``CrossEntropyLoss(ignore_index=pad)``: mean NLL over non-PAD targets."""
    V = logits.shape[-1]
    lg = logits.reshape(-1, V)
    tg = targets.reshape(-1)
    lse = torch.logsumexp(lg, dim=-1)
    picked = lg.gather(1, tg.unsqueeze(1)).squeeze(1)
    valid = tg != pad
    return ((lse - picked) * valid).sum() / valid.sum()
