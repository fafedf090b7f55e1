#!/usr/bin/env python3
"""Golden vectors for beam search on PEAKED (trained-like) output distributions, produced by the reference's own
code (run in the build container only; the reference never travels).  Writes tests/golden/f12_beam_peaked.npz.

With default-init weights the output distribution is nearly uniform and almost no hypothesis produces <EOS>, so the
f11 vectors do not exercise what the reference does after an <EOS> (its `done` tensor is cleared through a stride-0
view at every step: transformer.py:199-201, cnnrnn_model.py:161-163).  Here the output layer's weight is scaled by
3 or 6 and <EOS> gets a bias, so hypotheses end at every length, are frozen for one step, compete with W equal-score
proposals and resume accumulating.

Besides the captions, the final hypothesis scores and the number of steps the reference's loop ran are recorded by
wrapping ``torch.Tensor.topk`` for the duration of the call: the reference's second ``topk`` of every step
(transformer.py:211, cnnrnn_model.py:172) is the only one on a 2-D tensor, and its values are the scores.

Usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_beam_peaked.py [--ref /root/reference]
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch
import torch.nn as nn

from gen_golden import Vocab, ids_from_text, import_reference, make_decoder, pad_ragged, save, sd_np


class ScoreTap:
    """Records the values of every 2-D topk while active."""

    def __enter__(self):
        self.calls = []
        self._orig = torch.Tensor.topk
        tap = self

        def topk(t, *a, **k):
            out = tap._orig(t, *a, **k)
            if t.dim() == 2:
                tap.calls.append(out[0].detach().clone())
            return out

        torch.Tensor.topk = topk
        return self

    def __exit__(self, *exc):
        torch.Tensor.topk = self._orig


TF_CASES = ((6.0, 1.0, 2), (6.0, 1.0, 3), (6.0, 1.0, 5), (3.0, 1.0, 3), (6.0, 0.0, 5), (6.0, 2.0, 2), (6.0, 2.0, 8))
RNN_CASES = ((6.0, 0.55, 2), (6.0, 0.65, 3), (6.0, 0.65, 5), (12.0, 0.85, 3), (12.0, 0.95, 5), (6.0, 0.45, 8))
TF_LEN, RNN_LEN = 14, 15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    T, R, A = import_reference(args.ref)
    torch.set_num_threads(1)
    E, H, V, B = 64, 4, 50, 24
    vocab = Vocab(V)
    g = torch.Generator().manual_seed(12)
    arrs = dict(H=np.int64(H), tf_cases=np.asarray(TF_CASES, dtype=np.float64),
                rnn_cases=np.asarray(RNN_CASES, dtype=np.float64))

    torch.manual_seed(1201)
    base = make_decoder(T, E, V, H, 2, dropout=0.0).eval()
    arrs.update({"t__sd__" + k[4:]: v for k, v in sd_np(base).items()})
    enc = torch.randn(B, 2, E, generator=g)
    arrs["t_enc"] = enc.numpy()
    w0, b0 = base.fc2.weight.detach().clone(), base.fc2.bias.detach().clone()
    for n, (scale, eos_bias, W) in enumerate(TF_CASES):
        with torch.no_grad():
            base.fc2.weight.copy_(w0 * scale)
            base.fc2.bias.copy_(b0)
            base.fc2.bias[2] += eos_bias
        rep = enc.unsqueeze(1).expand(-1, W, -1, -1).reshape(B * W, 2, E)
        with torch.no_grad(), ScoreTap() as tap:
            ids = ids_from_text(base.beam_search_inference(rep, B, vocab, W, TF_LEN), vocab)
        arrs[f"t{n}_ids"] = pad_ragged(ids, TF_LEN)
        arrs[f"t{n}_scores"] = tap.calls[-1].numpy()
        arrs[f"t{n}_steps"] = np.int64(len(tap.calls))
        print(f"tf case {n} scale {scale} eos_bias {eos_bias} W {W}: steps {len(tap.calls)}, lengths",
              [len(r) for r in ids])

    # one more where the whole batch ends in the same step (the reference's early exit): a single sample, strong bias
    with torch.no_grad():
        base.fc2.weight.copy_(w0 * 6.0)
        base.fc2.bias.copy_(b0)
        base.fc2.bias[2] += 3.0
    found = False
    for b in range(B):
        one = enc[b:b + 1]
        for W in (2, 3):
            rep = one.expand(W, -1, -1).contiguous()
            with torch.no_grad(), ScoreTap() as tap:
                ids = ids_from_text(base.beam_search_inference(rep, 1, vocab, W, TF_LEN), vocab)
            if 2 < len(tap.calls) < TF_LEN:
                arrs["te_row"], arrs["te_W"] = np.int64(b), np.int64(W)
                arrs["te_ids"] = pad_ragged(ids, TF_LEN)
                arrs["te_scores"] = tap.calls[-1].numpy()
                arrs["te_steps"] = np.int64(len(tap.calls))
                print(f"early exit: sample {b} W {W} stops after {len(tap.calls)} steps, ids {ids}")
                found = True
                break
        if found:
            break
    assert found, "no early-exit case found"

    Er, Hd, Vr = 48, 80, 60
    vocr = Vocab(Vr)
    # per-sample gain and sparsity, so that the samples' LSTM states (and with them the caption lengths) differ
    featr = torch.rand(B, 2048, generator=g) * (0.3 + 3.0 * torch.rand(B, 1, generator=g))
    featr = featr * (torch.rand(B, 2048, generator=g) < 0.5).float()
    arrs["r_feats"] = featr.numpy()

    class Holder(nn.Module):
        def __init__(self, d):
            super().__init__()
            self.decoderRNN = d

    torch.manual_seed(1202)
    rnn = R.DecoderRNN(Er, Hd, Vr, dropout=0.0)
    with torch.no_grad():
        rnn.batchnorm.running_mean.normal_(0, 0.3)
        rnn.batchnorm.running_var.uniform_(0.5, 1.5)
    arrs.update({"r__sd__" + k[4:]: v for k, v in sd_np(rnn).items()})
    w0, b0 = rnn.fc2.weight.detach().clone(), rnn.fc2.bias.detach().clone()
    for n, (scale, eos_bias, W) in enumerate(RNN_CASES):
        with torch.no_grad():
            rnn.fc2.weight.copy_(w0 * scale)
            rnn.fc2.bias.copy_(b0)
            rnn.fc2.bias[2] += eos_bias
        with ScoreTap() as tap:
            text = R.CNNtoRNN.caption_images_beam_search(Holder(rnn), featr, vocr, beam_width=W, max_length=RNN_LEN)
        ids = ids_from_text(text, vocr)
        arrs[f"r{n}_ids"] = pad_ragged(ids, RNN_LEN)
        arrs[f"r{n}_scores"] = tap.calls[-1].numpy()
        arrs[f"r{n}_steps"] = np.int64(len(tap.calls))
        print(f"rnn case {n} scale {scale} eos_bias {eos_bias} W {W}: steps {len(tap.calls)}, lengths",
              [len(r) for r in ids])
    save("f12_beam_peaked", **arrs)


if __name__ == "__main__":
    main()
