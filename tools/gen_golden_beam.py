#!/usr/bin/env python3
"""Golden vectors for beam search, produced by the reference's own code (run in the build container only; the
reference never travels).  Writes tests/golden/f11_beam.npz.

  Transformer: DecoderTransformer.beam_search_inference (transformer.py:177-244) on a trivial concrete subclass,
               memory repeated per hypothesis exactly as the wrappers do (cnnattn_model.py:65-66);
  CNN-RNN:     CNNtoRNN.caption_images_beam_search (cnnrnn_model.py:112-209) as an unbound method on a holder that
               carries the reference's DecoderRNN (the real constructor would fetch InceptionV3 weights).

Usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_beam.py [--ref /root/reference]
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


def tf_beam(dec, enc, vocab, W, max_len):
    B = enc.shape[0]
    rep = enc.unsqueeze(1).expand(-1, W, -1, -1).reshape(B * W, enc.shape[1], enc.shape[2])
    with torch.no_grad():
        return ids_from_text(dec.beam_search_inference(rep, B, vocab, W, max_len), vocab)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    T, R, A = import_reference(args.ref)
    torch.set_num_threads(1)
    E, H, V = 64, 4, 50
    vocab = Vocab(V)
    g = torch.Generator().manual_seed(11)
    arrs = dict(H=np.int64(H))

    # (a) default-init weights, S in {1,2}, widths 3 and 2
    torch.manual_seed(1011)
    dec = make_decoder(T, E, V, H, 2, dropout=0.0).eval()
    arrs.update({"a__sd__" + k[4:]: v for k, v in sd_np(dec).items()})
    for S, W, ml in ((1, 3, 12), (2, 3, 12), (2, 2, 9)):
        enc = torch.randn(5, S, E, generator=g)
        arrs[f"a_enc_S{S}_W{W}"] = enc.numpy()
        arrs[f"a_ids_S{S}_W{W}"] = pad_ragged(tf_beam(dec, enc, vocab, W, ml), ml)

    # (b) <EOS> bias: some hypotheses finish early (frozen scores, single surviving proposal) while others run to
    #     the length limit; (b2) a stronger bias where every hypothesis of every sample finishes and the loop stops
    enc = torch.randn(6, 2, E, generator=g)
    for tag, eos_bias, ml in (("b", 0.6, 20), ("b2", 1.0, 20)):
        torch.manual_seed(1012)
        decb = make_decoder(T, E, V, H, 2, dropout=0.0).eval()
        with torch.no_grad():
            decb.fc2.bias[2] += eos_bias
        ids = tf_beam(decb, enc, vocab, 3, ml)
        print("F11" + tag, "eos_bias", eos_bias, ids)
        if tag == "b":
            arrs.update({"b__sd__" + k[4:]: v for k, v in sd_np(decb).items()})
        else:                                # same weights as (b) except the biased output layer
            arrs["b2__fc2.bias"] = decb.fc2.bias.detach().numpy().copy()
        arrs[tag + "_enc"] = enc.numpy()
        arrs[tag + "_ids"] = pad_ragged(ids, ml)

    # (c) <PAD> bias: hypotheses contain PAD tokens mid-sequence (PAD query rows attend uniformly)
    enc = torch.randn(6, 1, E, generator=g)
    found = None
    for pad_bias in (0.6, 0.8, 1.0, 1.2, 1.4):
        torch.manual_seed(1013)
        decc = make_decoder(T, E, V, H, 2, dropout=0.0).eval()
        with torch.no_grad():
            decc.fc2.bias[0] += pad_bias
            decc.fc2.bias[2] += 0.4
        ids = tf_beam(decc, enc, vocab, 3, 12)
        mid = sum(1 for r in ids if 0 in r and any(t != 0 for t in r[r.index(0) + 1:]))
        if mid >= 2:
            found = ids
            break
    assert found is not None, "no PAD-in-beam fixture found"
    print("F11c pad_bias", pad_bias, found)
    arrs.update({"c__sd__" + k[4:]: v for k, v in sd_np(decc).items()})
    arrs["c_enc"] = enc.numpy()
    arrs["c_ids"] = pad_ragged(found, 12)

    # (r) CNN-RNN beam search
    Er, Hd, Vr = 48, 80, 60
    vocr = Vocab(Vr)
    featr = torch.rand(6, 2048, generator=g)

    class Holder(nn.Module):
        def __init__(self, d):
            super().__init__()
            self.decoderRNN = d

    for tag, eos_bias in (("r0", 0.0), ("r", 0.1)):
        torch.manual_seed(1014)
        rnn = R.DecoderRNN(Er, Hd, Vr, dropout=0.0)
        with torch.no_grad():
            rnn.batchnorm.running_mean.normal_(0, 0.3)
            rnn.batchnorm.running_var.uniform_(0.5, 1.5)
            rnn.fc2.bias[2] += eos_bias
        text = R.CNNtoRNN.caption_images_beam_search(Holder(rnn), featr, vocr, beam_width=3, max_length=15)
        ids = ids_from_text(text, vocr)
        print("F11" + tag, "eos_bias", eos_bias, ids)
        if tag == "r0":
            arrs.update({"r__sd__" + k[4:]: v for k, v in sd_np(rnn).items()})
        else:
            arrs["r__fc2.bias_biased"] = rnn.fc2.bias.detach().numpy().copy()
        arrs[tag + "_ids"] = pad_ragged(ids, 15)
    arrs["r_feats"] = featr.numpy()
    save("f11_beam", **arrs)


if __name__ == "__main__":
    main()
