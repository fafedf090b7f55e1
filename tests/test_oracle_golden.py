"""The CPU oracle against vectors produced by the reference's own classes (tools/gen_golden.py)."""
import numpy as np
import torch

from conftest import load_golden, sub_sd, unpad
from oracle import decoder_ref as R

RT, AT = 2e-5, 2e-6


def close(a, b, rtol=RT, atol=AT):
    a = torch.as_tensor(np.asarray(a))
    b = torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = float(b.abs().max()) if b.numel() else 1.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= atol + rtol * max(scale, 1e-30), f"max err {err:.3e} vs scale {scale:.3e}"


def t(x):
    return torch.from_numpy(np.asarray(x))


def test_f1_self_attention_pad_rows():
    fx = load_golden("f1_mha_self")
    sd = {"a." + k: v for k, v in sub_sd(fx, "sd__").items()}
    ids = t(fx["ids"])
    mask = R.tgt_mask_from_ids(ids)
    assert torch.equal(mask, t(fx["mask"]))
    y = R.mha(sd, "a", t(fx["x"]), t(fx["x"]), int(fx["H"]), mask)
    close(y, fx["y"])


def test_f2_cross_attention():
    fx = load_golden("f2_mha_cross")
    sd = {"a." + k: v for k, v in sub_sd(fx, "sd__").items()}
    for S in (1, 2, 5):
        y = R.mha(sd, "a", t(fx["x"]), t(fx[f"enc_S{S}"]), int(fx["H"]), None)
        close(y, fx[f"y_S{S}"])


def test_f3_decoder_layer():
    fx = load_golden("f3_decoder_layer")
    sd = sub_sd(fx, "sd__")
    mask = R.tgt_mask_from_ids(t(fx["ids"]))
    y = R.decoder_layer(sd, 0, t(fx["x"]), t(fx["enc"]), mask, int(fx["H"]))
    close(y, fx["y"])


def test_pe_table_matches_reference_buffer():
    fx = load_golden("f4_decoder_forward")
    pe = fx["sd__positional_encoding.pe"][0]
    mine = R.positional_encoding_table(pe.shape[0], pe.shape[1])
    assert np.array_equal(mine.numpy(), pe)


def test_f4_decoder_forward_loss_grads():
    fx = load_golden("f4_decoder_forward")
    H = int(fx["H"])
    caps = t(fx["captions"])
    for S in (1, 2, 5):
        sd = {k: v.clone().requires_grad_(v.is_floating_point() and k != "positional_encoding.pe")
              for k, v in sub_sd(fx, "sd__").items()}
        enc = t(fx[f"enc_S{S}"]).clone().requires_grad_(True)
        logits = R.decoder_forward(sd, enc, caps[:, :-1], H)
        close(logits.detach(), fx[f"logits_S{S}"])
        loss = R.caption_loss(logits, caps[:, 1:])
        close(loss.detach(), fx[f"loss_S{S}"])
        loss.backward()
        close(enc.grad, fx[f"denc_S{S}"], rtol=1e-4)
        for k, v in sd.items():
            key = f"grad__{k}@S{S}"
            if key not in fx:
                continue
            g = v.grad if v.grad is not None else torch.zeros_like(v)
            close(g, fx[key], rtol=1e-4, atol=1e-7)
        if S == 1:  # cross-attention W_q/W_k get exactly zero gradient when S == 1
            for l in range(2):
                for w in ("W_q", "W_k"):
                    assert float(np.abs(fx[f"grad__decoder_layers.{l}.cross_attn.{w}.weight@S1"]).max()) == 0.0


def test_f4_take_last_and_prefix_invariance():
    fx = load_golden("f4_decoder_forward")
    sd = sub_sd(fx, "sd__")
    caps = t(fx["captions"])
    out = R.decoder_forward(sd, t(fx["enc_S2"]), caps[:, :5], int(fx["H"]), take_last=True)
    close(out, fx["last_logits_S2_T5"])


def test_f5_greedy_random_weights():
    fx = load_golden("f5_greedy")
    sd = sub_sd(fx, "a__sd__")
    for S in (1, 2, 5):
        got = R.greedy_inference_ids(sd, t(fx[f"a_enc_S{S}"]), int(fx["H"]), 12)
        assert got == unpad(fx[f"a_ids_S{S}"])


def test_f5_greedy_emits_pad_mid_sequence():
    fx = load_golden("f5_greedy")
    want = unpad(fx["b_ids"])
    assert any(0 in r and any(x != 0 for x in r[r.index(0) + 1:]) for r in want)
    got = R.greedy_inference_ids(sub_sd(fx, "b__sd__"), t(fx["b_enc"]), int(fx["H"]), 14)
    assert got == want


def test_f5_greedy_all_done_early_exit():
    fx = load_golden("f5_greedy")
    want = unpad(fx["c_ids"])
    got = R.greedy_inference_ids(sub_sd(fx, "c__sd__"), t(fx["c_enc"]), int(fx["H"]), 30)
    assert got == want and max(len(r) for r in got) < 30


def test_f6_cnn_attn_head_eval_train_and_greedy():
    fx = load_golden("f6_heads")
    H = int(fx["H"])
    caps = t(fx["captions"])
    feats = t(fx["cnn_feats"])
    sd = sub_sd(fx, "cnn__sd__")
    enc, _ = R.head_single(sd, feats, "fc1", training=False)
    close(R.decoder_forward(sd, enc, caps[:, :-1], H), fx["cnn_logits_eval"])
    got = R.greedy_inference_ids(sd, enc, H, 10)
    assert got == unpad(fx["cnn_greedy_ids"])
    # training mode: batch statistics, running-stat update, gradients
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and k != "positional_encoding.pe")
           for k, v in sd.items()}
    enc, new = R.head_single(sdg, feats, "fc1", training=True)
    logits = R.decoder_forward(sdg, enc, caps[:, :-1], H)
    close(logits.detach(), fx["cnn_logits_train"])
    loss = R.caption_loss(logits, caps[:, 1:])
    close(loss.detach(), fx["cnn_loss_train"])
    loss.backward()
    for k in ("fc1.weight", "batchnorm.weight", "batchnorm.bias", "decoder_embedding.weight", "fc2.weight"):
        close(sdg[k].grad, fx["cnn_grad__" + k], atol=1e-7)
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        close(new["batchnorm." + k], fx["cnn_after__batchnorm." + k])


def test_f6_pair_head_eval_train():
    fx = load_golden("f6_heads")
    H = int(fx["H"])
    caps = t(fx["captions"])
    feats = t(fx["pair_feats"])
    split = int(fx["pair_split"])
    sd = sub_sd(fx, "pair__sd__")
    enc, _ = R.head_pair(sd, feats, "fc_vit", split, training=False)
    close(enc, fx["pair_enc_eval"])
    close(R.decoder_forward(sd, enc, caps[:, :-1], H), fx["pair_logits_eval"])
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and k != "positional_encoding.pe")
           for k, v in sd.items()}
    enc, _ = R.head_pair(sdg, feats, "fc_vit", split, training=True)
    close(enc.detach(), fx["pair_enc_train"])
    logits = R.decoder_forward(sdg, enc, caps[:, :-1], H)
    close(logits.detach(), fx["pair_logits_train"])
    loss = R.caption_loss(logits, caps[:, 1:])
    loss.backward()
    for k in ("fc_vit.weight", "fc_vit.bias", "fc_cnn.weight", "batchnorm.weight"):
        close(sdg[k].grad, fx["pair_grad__" + k], atol=1e-7)


def test_f7_decoder_rnn_forward_grads_step_and_greedy():
    fx = load_golden("f7_decoder_rnn")
    sd = sub_sd(fx, "sd__")
    feats, caps = t(fx["feats"]), t(fx["captions"])
    close(R.decoder_rnn_forward(sd, feats, caps[:, :-1], training=False), fx["logits_eval"])
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    logits = R.decoder_rnn_forward(sdg, feats, caps[:, :-1], training=True)
    close(logits.detach(), fx["logits_train"])
    loss = R.caption_loss(logits, caps[:, 1:])
    close(loss.detach(), fx["loss_train"])
    loss.backward()
    for k, v in sdg.items():
        if "grad__" + k in fx and v.requires_grad:
            close(v.grad if v.grad is not None else torch.zeros_like(v), fx["grad__" + k], atol=1e-7)
    h1, c1 = R.lstm_cell(t(fx["step_x"]), t(fx["step_h0"]), t(fx["step_c0"]),
                         sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"], sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"])
    close(h1, fx["step_h1"])
    close(c1, fx["step_c1"])
    sd2 = dict(sd)
    for k in ("fc2.bias", "batchnorm.running_mean", "batchnorm.running_var"):
        sd2[k] = t(fx["greedy__" + k])
    got = R.cnnrnn_greedy_ids(sd2, feats, 15)
    assert got == unpad(fx["greedy_ids"])


def test_f11_beam_search_transformer_and_rnn():
    """Beam search restatement against the reference's beam_search_inference / caption_images_beam_search:
    default weights (two widths), hypotheses that finish early, all finishing at once, PAD tokens inside
    hypotheses, and the LSTM variant."""
    fx = load_golden("f11_beam")
    H = int(fx["H"])
    sd = sub_sd(fx, "a__sd__")
    for S, W, ml in ((1, 3, 12), (2, 3, 12), (2, 2, 9)):
        got = R.beam_search_ids(sd, t(fx[f"a_enc_S{S}_W{W}"]), H, W, ml)
        assert got == unpad(fx[f"a_ids_S{S}_W{W}"]), (S, W)
    sdb = sub_sd(fx, "b__sd__")
    assert R.beam_search_ids(sdb, t(fx["b_enc"]), H, 3, 20) == unpad(fx["b_ids"])
    sdb2 = dict(sdb)
    sdb2["fc2.bias"] = t(fx["b2__fc2.bias"])
    assert R.beam_search_ids(sdb2, t(fx["b2_enc"]), H, 3, 20) == unpad(fx["b2_ids"])
    assert R.beam_search_ids(sub_sd(fx, "c__sd__"), t(fx["c_enc"]), H, 3, 12) == unpad(fx["c_ids"])
    sdr = sub_sd(fx, "r__sd__")
    assert R.cnnrnn_beam_search_ids(sdr, t(fx["r_feats"]), 3, 15) == unpad(fx["r0_ids"])
    sdr2 = dict(sdr)
    sdr2["fc2.bias"] = t(fx["r__fc2.bias_biased"])
    assert R.cnnrnn_beam_search_ids(sdr2, t(fx["r_feats"]), 3, 15) == unpad(fx["r_ids"])


def _peaked_sd(fx, prefix, scale, eos_bias):
    sd = sub_sd(fx, prefix)
    sd["fc2.weight"] = sd["fc2.weight"] * float(scale)
    b = sd["fc2.bias"].clone()
    b[2] += float(eos_bias)
    sd["fc2.bias"] = b
    return sd


def test_f12_beam_search_after_eos():
    """Peaked output distributions: hypotheses end at every length.  Captions, final hypothesis scores and the number
    of steps the loop ran, all taken from the reference's own run (the scores through a tap on its second topk), for
    the Transformer and the LSTM decoder at widths 2, 3, 5 and 8, plus a one-sample search that the reference leaves
    early because all hypotheses produce <EOS> in the same step."""
    fx = load_golden("f12_beam_peaked")
    H = int(fx["H"])
    enc = t(fx["t_enc"])
    for n, (scale, eos_bias, W) in enumerate(fx["tf_cases"]):
        sd = _peaked_sd(fx, "t__sd__", scale, eos_bias)
        ml = fx[f"t{n}_ids"].shape[1]
        got, seqs, scores = R.beam_search_ids(sd, enc, H, int(W), ml, return_state=True)
        assert got == unpad(fx[f"t{n}_ids"]), n
        assert seqs.shape[2] - 1 == int(fx[f"t{n}_steps"])
        assert torch.allclose(scores, t(fx[f"t{n}_scores"]), rtol=1e-5, atol=1e-5), n
    sd = _peaked_sd(fx, "t__sd__", 6.0, 3.0)
    b, W = int(fx["te_row"]), int(fx["te_W"])
    got, seqs, scores = R.beam_search_ids(sd, enc[b:b + 1], H, W, fx["te_ids"].shape[1], return_state=True)
    assert got == unpad(fx["te_ids"])
    assert seqs.shape[2] - 1 == int(fx["te_steps"]) < fx["te_ids"].shape[1]
    assert torch.allclose(scores, t(fx["te_scores"]), rtol=1e-5, atol=1e-5)
    feats = t(fx["r_feats"])
    for n, (scale, eos_bias, W) in enumerate(fx["rnn_cases"]):
        sd = _peaked_sd(fx, "r__sd__", scale, eos_bias)
        ml = fx[f"r{n}_ids"].shape[1]
        got, seqs, scores = R.cnnrnn_beam_search_ids(sd, feats, int(W), ml, return_state=True)
        assert got == unpad(fx[f"r{n}_ids"]), n
        assert seqs.shape[2] - 1 == int(fx[f"r{n}_steps"])
        assert torch.allclose(scores, t(fx[f"r{n}_scores"]), rtol=1e-5, atol=1e-5), n


def test_topk_order_matches_torch():
    """The restated order of torch.topk among equal values (oracle/topk_order.py) against torch.topk itself on rows
    full of ties: the W*W candidate rows of a beam-search step for every width up to 8."""
    import random
    from oracle.topk_order import topk_cpu_order
    rnd = random.Random(0)
    for W in range(1, 9):
        n = W * W
        for trial in range(400):
            mode = trial % 4
            if mode == 0:
                vals = [rnd.choice([-1.0, -2.0, -3.5, float("-inf")]) for _ in range(n)]
            elif mode == 1:
                vals = [rnd.gauss(0, 1) for _ in range(n)]
            elif mode == 2:      # some hypotheses frozen (W equal proposals), the others with falling log-probs
                vals = []
                for _ in range(W):
                    s = -rnd.random() * 5
                    vals += [s] * W if rnd.random() < 0.5 else sorted((s - rnd.random() * 3 for _ in range(W)), reverse=True)
            else:                # first step: only hypothesis 0 is eligible
                vals = [rnd.gauss(0, 1) for _ in range(W)] + [float("-inf")] * (n - W)
            x = torch.tensor(vals, dtype=torch.float32)
            want_v, want_i = x.topk(W)
            got_v, got_i = topk_cpu_order(x.tolist(), W)
            assert got_i == want_i.tolist(), (W, vals)
            assert got_v == want_v.tolist()
