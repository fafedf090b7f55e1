#!/usr/bin/env python3
"""Soak of the one-launch decoder layers: many decodes at random batch sizes, lengths, layer counts and memory rows,
each run three times - alone, beside bursts of large contractions on another stream, and beside a second decode on a
third stream - and compared bit for bit (token ids, lengths, the rows handed to the vocabulary projection); once more
through the launch-per-operation form, where only near-ties of the arg-max may differ.  A wait between workgroups that
ran out would show as a warning, turned into an error here.   usage: fused_soak.py [cases] [seed]"""
import os
import random
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch
from capdec.engine import tf_greedy_ids
from gpu_util import plain_decoder

E, H, PAD, SOS, EOS = 512, 4, 0, 1, 2
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
warnings.simplefilter("error")

a = torch.randn(4096, 4096, device="cuda")
b = torch.randn(4096, 4096, device="cuda")
side, third = torch.cuda.Stream(), torch.cuda.Stream()
decs = {}
t0 = time.time()
diff_plain = rows_plain = 0
for case in range(cases):
    L = rng.choice((1, 2, 2, 2))
    S = rng.choice((1, 1, 1, 2))
    V = rng.choice((500, 3000, 10000))
    B = rng.choice((1, 7, 16, 17, 33, 64, 100, 255, 256, 512, 513, 777, 1024, 1024, 1500, 2048))
    ML = rng.choice((2, 5, 12, 20, 20, 33, 50))
    pad_bias = rng.choice((0.0, 0.0, 0.8, 1.3))
    key = (L, V, pad_bias)
    if key not in decs:
        torch.manual_seed(100 + len(decs))
        d = plain_decoder(E, V, H, L, max_len=50).cuda().eval()
        with torch.no_grad():
            d.fc2.bias[PAD] += pad_bias
        decs[key] = d
    dec = decs[key]
    g = torch.Generator().manual_seed(case)
    enc = torch.randn(B, S, E, generator=g).cuda()
    w0, w1, w2 = [], [], []
    ids0, len0, nd0 = tf_greedy_ids(dec, enc, ML, SOS, EOS, keep_workspace=w0)
    torch.cuda.synchronize()
    x0 = w0[0][:B * E * 4].clone()
    with torch.cuda.stream(side):
        for _ in range(rng.randint(1, 8)):
            a @ b
    ids1, len1, nd1 = tf_greedy_ids(dec, enc, ML, SOS, EOS, keep_workspace=w1)
    torch.cuda.synchronize()
    assert torch.equal(ids0, ids1) and torch.equal(len0, len1) and nd0 == nd1, ("beside contractions", case, L, S, V, B, ML)
    assert torch.equal(w1[0][:B * E * 4], x0), ("rows beside contractions", case, L, S, V, B, ML)
    enc_b = torch.randn(rng.choice((64, 512, 1024)), 1, E, generator=g).cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(third):
        tf_greedy_ids(dec, enc_b, ML, SOS, EOS)
        ids2, len2, nd2 = tf_greedy_ids(dec, enc, ML, SOS, EOS, keep_workspace=w2)
    with torch.cuda.stream(side):
        tf_greedy_ids(dec, enc_b, ML, SOS, EOS)
    torch.cuda.synchronize()
    assert torch.equal(ids0, ids2) and torch.equal(len0, len2) and nd0 == nd2, ("beside a decode", case, L, S, V, B, ML)
    assert torch.equal(w2[0][:B * E * 4], x0), ("rows beside a decode", case, L, S, V, B, ML)
    idp, lenp, _ = tf_greedy_ids(dec, enc, ML, SOS, EOS, fused_layers=False)
    torch.cuda.synchronize()
    d = int(((ids0 != idp).any(dim=1) | (len0 != lenp)).sum())
    diff_plain += d
    rows_plain += B
    assert d <= max(2, B // 100), ("against the launch-per-operation form", case, L, S, V, B, ML, d)
    if case % 10 == 9 or case == cases - 1:
        print(f"case {case + 1}/{cases}  {time.time() - t0:5.0f} s  last: L={L} S={S} V={V} B={B} len={ML} pad_bias={pad_bias} "
              f"redone on the exact path {nd0};  captions differing from the launch-per-operation form so far "
              f"{diff_plain} of {rows_plain}", flush=True)
print(f"soak ok: {cases} cases, every repeat bit-identical; {diff_plain} of {rows_plain} captions differ from the "
      f"launch-per-operation form (arg-max near-ties)")
