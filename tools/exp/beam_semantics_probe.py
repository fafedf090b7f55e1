#!/usr/bin/env python3
"""Build-container experiment: how often does a restatement of the beam-search state machine agree with the
reference's own beam_search_inference on peaked (trained-like) output distributions, under different tie rules
for the W*W candidate selection?  Prints mismatch counts; writes nothing."""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

import torch

from gen_golden import Vocab, ids_from_text, import_reference, make_decoder
from oracle import decoder_ref as R


def select(step, scores, fin, logits, W, rule):
    B = scores.shape[0]
    lp, tok = torch.log_softmax(logits, dim=-1).topk(W, dim=2)
    cand = scores.unsqueeze(-1) + (1.0 - fin.unsqueeze(-1).float()) * lp
    if step == 0:
        cand[:, 1:, :] = float("-inf")
    cand = cand.reshape(B, W * W)
    if rule == "topk":
        new_scores, flat = cand.topk(W, dim=-1)
    else:
        s, i = torch.sort(cand, dim=-1, descending=True, stable=True)
        new_scores, flat = s[:, :W], i[:, :W]
    return new_scores, flat // W, tok.reshape(B, W * W).gather(1, flat)


def beam(sd, enc, H, W, max_length, rule, sos=1, eos=2):
    B = enc.shape[0]
    enc_w = enc.unsqueeze(1).expand(-1, W, -1, -1).reshape(B * W, enc.shape[1], enc.shape[2])
    seqs = torch.full((B, W, 1), sos, dtype=torch.long)
    scores = torch.zeros(B, W)
    fin = torch.zeros(B, W, dtype=torch.bool)
    for step in range(max_length):
        logits = R.decoder_forward(sd, enc_w, seqs.reshape(B * W, -1), H, take_last=True).reshape(B, W, -1)
        scores, parent, new_tok = select(step, scores, fin, logits, W, rule)
        seqs = torch.cat([seqs.gather(1, parent.unsqueeze(-1).expand(-1, -1, seqs.shape[2])), new_tok.unsqueeze(-1)], 2)
        fin = new_tok == eos
        if bool(fin.all()):
            break
    return R._beam_result(seqs, eos), scores


def main():
    T, Rf, A = import_reference("/root/reference")
    torch.set_num_threads(4)
    E, H, V = 64, 4, 50
    vocab = Vocab(V)
    g = torch.Generator().manual_seed(5)
    tot = {"topk": 0, "stable": 0}
    n = 0
    for scale in (1.0, 3.0, 6.0):
        for eos_bias in (0.0, 1.0, 2.0):
            for W in (2, 3, 5):
                torch.manual_seed(int(scale * 100 + eos_bias * 10 + W))
                dec = make_decoder(T, E, V, H, 2, dropout=0.0).eval()
                with torch.no_grad():
                    dec.fc2.weight *= scale
                    dec.fc2.bias[2] += eos_bias
                sd = {k: v.detach().clone() for k, v in dec.state_dict().items()}
                enc = torch.randn(32, 2, E, generator=g)
                rep = enc.unsqueeze(1).expand(-1, W, -1, -1).reshape(32 * W, 2, E)
                with torch.no_grad():
                    want = ids_from_text(dec.beam_search_inference(rep, 32, vocab, W, 14), vocab)
                n += 32
                line = f"scale {scale} eos_bias {eos_bias} W {W}:"
                for rule in ("topk", "stable"):
                    got, _ = beam(sd, enc, H, W, 14, rule)
                    bad = sum(1 for a, b in zip(got, want) if a != b)
                    tot[rule] += bad
                    line += f" {rule} {bad}"
                old = R.beam_search_ids(sd, enc, H, W, 14)
                line += f" old {sum(1 for a, b in zip(old, want) if a != b)}"
                print(line, " mean len", sum(len(w) for w in want) / 32)
    print("total mismatches of", n, tot)


if __name__ == "__main__":
    main()
