"""TEST INFRASTRUCTURE ONLY.  PyTorch-CPU composition of the soft-attention LSTM decoder's formulas.

PARITY UNPINNED: the reference repository has no additive-attention LSTM decoder (its "Attn" models are Transformer
decoders); the model exists here because the project's goals name it.  This file therefore pins nothing against the
reference: it only restates, with plain torch operators, the formulas the HIP kernels of
``models/softattn_model.py`` implement, so that the kernels can be checked against an independent evaluation.
The product never imports this module.
"""
from typing import Dict, List

import torch

from .decoder_ref import EOS, SOS, linear, lstm_cell

SD = Dict[str, torch.Tensor]


def prepare(sd: SD, feats: torch.Tensor):
    """feats [B,S,D] -> (enc [B,S,E], pf [B,S,A], h0 [B,Hd], c0 [B,Hd])."""
    enc = linear(feats, sd["fc_feat.weight"], sd["fc_feat.bias"])
    pf = linear(enc, sd["att_f.weight"], sd["att_f.bias"])
    mean = enc.mean(dim=1)
    return enc, pf, linear(mean, sd["init_h.weight"], sd["init_h.bias"]), linear(mean, sd["init_c.weight"], sd["init_c.bias"])


def attend(sd: SD, h: torch.Tensor, enc: torch.Tensor, pf: torch.Tensor):
    u = linear(h, sd["att_h.weight"], sd["att_h.bias"])                                   # [B,A]
    e = (torch.tanh(u.unsqueeze(1) + pf) * sd["att_v.weight"].reshape(1, 1, -1)).sum(-1)   # [B,S]
    alpha = torch.softmax(e, dim=1)
    return (alpha.unsqueeze(-1) * enc).sum(1), alpha


def _w(sd: SD):
    return (sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"], sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"])


def forward_logits(sd: SD, feats: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """Teacher-forced logits [B,T,V]: step t sees token ids[:,t] and the context attended with the previous state."""
    enc, pf, h, c = prepare(sd, feats)
    outs = []
    for t in range(ids.shape[1]):
        ctx, _ = attend(sd, h, enc, pf)
        x = torch.cat([sd["embed.weight"][ids[:, t]], ctx], dim=1)
        h, c = lstm_cell(x, h, c, *_w(sd))
        outs.append(h)
    return linear(torch.stack(outs, 1), sd["fc2.weight"], sd["fc2.bias"])


def greedy_ids(sd: SD, feats: torch.Tensor, max_length: int, sos: int = SOS, eos: int = EOS) -> List[List[int]]:
    """Greedy loop with the reference's bookkeeping (finished rows keep feeding their prediction back)."""
    enc, pf, h, c = prepare(sd, feats)
    B = feats.shape[0]
    tok = torch.full((B,), sos, dtype=torch.long)
    res: List[List[int]] = [[] for _ in range(B)]
    done = [False] * B
    for _ in range(max_length):
        ctx, _ = attend(sd, h, enc, pf)
        h, c = lstm_cell(torch.cat([sd["embed.weight"][tok], ctx], dim=1), h, c, *_w(sd))
        tok = linear(h, sd["fc2.weight"], sd["fc2.bias"]).argmax(-1)
        for i in range(B):
            if not done[i]:
                if int(tok[i]) == eos:
                    done[i] = True
                else:
                    res[i].append(int(tok[i]))
        if all(done):
            break
    return res
