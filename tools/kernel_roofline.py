#!/usr/bin/env python3
"""Achieved rate of every hot kernel at the benchmark shapes against its roof (run on the GPU box).

MFMA kernels: algorithmic FLOP / time against the dense fp32 matrix peak (157.3 TFLOP/s); the other kernels:
algorithmic bytes (each operand once, as listed per row) / time against the HBM peak (8 TB/s).  Times are HIP-event
averages over back-to-back launches on one stream.  Writes a markdown table to stdout."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
import torch
from capdec import _lib, ops

PEAK_TF, PEAK_GBS = 157.3, 8000.0


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    _lib.load()
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda()
    rows = []

    def mfma(name, shape, flops, fn):
        us = timeit(fn)
        tf = flops / us / 1e6
        rows.append((name, shape, f"{flops / 1e9:.2f} GFLOP", us, f"{tf:.1f} TFLOP/s", tf / PEAK_TF))

    def hbm(name, shape, nbytes, fn):
        us = timeit(fn)
        gbs = nbytes / us / 1e3
        rows.append((name, shape, f"{nbytes / 1e6:.1f} MB", us, f"{gbs:.0f} GB/s", gbs / PEAK_GBS))

    E, H, V = 512, 4, 10000
    # ---- decode step, cfg2: 1024 rows ----
    M = 1024
    x, W, b = rnd(M, E), rnd(E, E), rnd(E)
    out = torch.empty(M, E, device="cuda")
    mfma("gemm_kernel 32x64 (decode projection)", "1024x512x512", 2.0 * M * E * E, lambda: ops.linear(x, W, b, out=out))
    Ws = [rnd(E, E) for _ in range(3)]
    outs = [torch.empty(M, E, device="cuda") for _ in range(3)]
    mfma("gemm_kernel 32x64 BK16 grouped (decode Q,K,V)", "3 x 1024x512x512", 6.0 * M * E * E,
         lambda: ops.gemm_groups([dict(P=x, Q=w, C=o) for w, o in zip(Ws, outs)], E))
    Wv, bv = rnd(V, E), rnd(V)
    mfma("ops.vocab_argmax: vocab_reduce_kernel + the merge launch + two output allocations (the kernel alone: kernel statistics)", "1024x10000x512", 2.0 * M * V * E, lambda: ops.vocab_argmax(x, Wv, bv))
    gam, bet = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda")
    y = torch.empty(M, E, device="cuda")
    hbm("add_layernorm_kernel (decode)", "1024x512, x+add->y", 3.0 * M * E * 4, lambda: ops.add_layernorm(x, gam, bet, add=out, out=y))
    T = 10
    kc, vc = rnd(M * 20, E), rnd(M * 20, E)
    hbm("attention_kernel (decode step, 10 cached keys)", "B=1024,H=4,Tq=1,Tk=10", (2.0 * T + 2) * M * E * 4,
        lambda: ops.attention(x, kc[: M * T], vc[: M * T], M, H, 1, T))
    # ---- training, cfg3: 10240 rows ----
    Mt, Bt, Tt = 10240, 512, 20
    xt, dy = rnd(Mt, E), rnd(Mt, E)
    ot = torch.empty(Mt, E, device="cuda")
    mfma("gemm_kernel 64x64 BK16 (train projection)", "10240x512x512", 2.0 * Mt * E * E, lambda: ops.linear(xt, W, b, out=ot))
    mfma("gemm_kernel 64x64 (dgrad)", "10240x512x512", 2.0 * Mt * E * E, lambda: ops.linear_dgrad(dy, W))
    dW = torch.zeros(E, E, device="cuda")
    mfma("gemm_kernel 64x64 split-K (wgrad)", "512x512x10240", 2.0 * Mt * E * E, lambda: ops.linear_wgrad(dy, xt, out=dW))
    logits = torch.empty(Mt, V, device="cuda")
    mfma("gemm_kernel 64x64 BK16 (train logits)", "10240x10000x512", 2.0 * Mt * V * E, lambda: ops.linear(xt, Wv, bv, out=logits))
    dWv = torch.zeros(V, E, device="cuda")
    mfma("gemm_kernel 128x128 (vocab wgrad)", "10000x512x10240", 2.0 * Mt * V * E, lambda: ops.linear_wgrad(logits, xt, out=dWv))
    tg = torch.randint(1, V, (Mt,), generator=g).cuda()
    hbm("softmax_nll_grad_reg_kernel (in place, row in registers)", "10240x10000", 2.0 * Mt * V * 4, lambda: ops.softmax_nll_grad_(logits, tg))
    hbm("ln_train_fwd_kernel (dropout+add+LN)", "10240x512, x,add->y,s", 4.0 * Mt * E * 4,
        lambda: ops.add_layernorm_train(xt, dy, gam, bet, 0.1, 7))
    _, s_, mean, rstd = ops.add_layernorm_train(xt, dy, gam, bet, 0.1, 7)
    dg, db = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    hbm("ln_bwd_kernel", "10240x512, dy,s->d_add,d_x", 4.0 * Mt * E * 4, lambda: ops.layernorm_bwd(dy, s_, mean, rstd, gam, 0.1, 7, dg, db))
    q, k, v = rnd(Mt, E), rnd(Mt, E), rnd(Mt, E)
    ids32 = torch.randint(1, 9, (Bt, Tt), generator=g).to(torch.int32).cuda()
    hbm("attention_kernel (train, causal)", "B=512,H=4,T=20", 4.0 * Mt * E * 4,
        lambda: ops.attention(q, k, v, Bt, H, Tt, Tt, causal=True, ids32=ids32, want_lse=True))
    o, lse = ops.attention(q, k, v, Bt, H, Tt, Tt, causal=True, ids32=ids32, want_lse=True)
    hbm("attn_bwd_mfma_kernel (dense call: the matrix-core form at this shape)", "B=512,H=4,T=20", 7.0 * Mt * E * 4,
        lambda: ops.attention_bwd(q, k, v, dy, lse, Bt, H, Tt, Tt, causal=True, ids32=ids32))
    # the forms the trainer takes: the same batch as a ragged one (every row live here)
    ids64 = ids32.to(torch.int64)
    rr = ops.RaggedRows(ids64, ids64, 0)
    hbm("attention_mfma_kernel (ragged batch: the trainer's forward, matrix cores)", "B=512,H=4,T=20", 4.0 * Mt * E * 4,
        lambda: ops.attention_ragged(q, k, v, rr, H, want_lse=True))
    _, lse_r = ops.attention_ragged(q, k, v, rr, H, want_lse=True)
    hbm("attn_bwd_mfma_kernel (ragged batch: the trainer's backward, matrix cores)", "B=512,H=4,T=20", 7.0 * Mt * E * 4,
        lambda: ops.attention_bwd_ragged(q, k, v, dy, lse_r, rr, H))
    St = 197
    km, vm = rnd(Bt * St, E), rnd(Bt * St, E)
    o2, lse2 = ops.attention(q, km, vm, Bt, H, Tt, St, want_lse=True)
    hbm("attention_kernel (train, 197 memory rows)", "B=512,H=4,Tq=20,Tk=197", (2.0 * Mt + 2.0 * Bt * St) * E * 4,
        lambda: ops.attention(q, km, vm, Bt, H, Tt, St, want_lse=True))
    hbm("attn_bwd_chunked_kernel (197 memory rows)", "B=512,H=4,Tq=20,Tk=197", (4.0 * Mt + 4.0 * Bt * St) * E * 4,
        lambda: ops.attention_bwd(q, km, vm, dy, lse2, Bt, H, Tt, St, out=o2))
    # the same cross-attention with the key / value projections folded onto the query side: per-sample contractions
    # of the H*T query rows against the raw memory (S=197, 512 samples, all 20 positions live)
    R4 = Mt * H
    qp, mem = rnd(R4, E), rnd(Bt * St, E)
    pbuf = torch.zeros(R4, 224, device="cuda")
    cbuf = torch.zeros(R4, E, device="cuda")
    dmem = torch.zeros(Bt * St, E, device="cuda")
    mfma("bgemm_kernel<NT> (scores = (q W_k) mem^T)", "512 x [80x197x512]", 2.0 * R4 * St * E,
         lambda: ops.bgemm("nt", qp, mem, pbuf, Bt, Tt, H, scale=0.0884))
    hbm("softmax_rows_kernel", "40960x197", 2.0 * R4 * St * 4, lambda: ops.softmax_rows_(pbuf, St))
    mfma("bgemm_kernel<NN> (context = P mem)", "512 x [80x512x197]", 2.0 * R4 * St * E,
         lambda: ops.bgemm("nn", pbuf, mem, cbuf, Bt, Tt, H))
    dpb = torch.zeros(R4, 224, device="cuda")
    hbm("softmax_rows_bwd_kernel", "40960x197", 3.0 * R4 * St * 4, lambda: ops.softmax_rows_bwd_(pbuf, dpb, St, scale=0.0884))
    mfma("bgemm_kernel<TN> (d mem += P^T dC + dS^T q')", "512 x [197x512x160]", 4.0 * R4 * St * E,
         lambda: ops.bgemm("tn", pbuf, cbuf, dmem, Bt, Tt, H, P2=dpb, Q2=qp, accumulate=True))
    bias_g = torch.zeros(E, device="cuda")
    hbm("colsum_kernel (bias gradient)", "10240x512", 1.0 * Mt * E * 4, lambda: ops.colsum(dy, bias_g))
    hbm("group_sum_kernel", "512x20x512", 1.0 * Mt * E * 4, lambda: ops.group_sum(dy, Tt))
    hbm("dropout4_kernel", "10240x512", 2.0 * Mt * E * 4, lambda: ops.dropout(dy, 0.1, 3))
    emb_g = torch.zeros(V, E, device="cuda")
    idl = torch.randint(4, V, (Mt,), generator=g)
    idl[::20] = 1
    idl[15::20] = 0
    idl = idl.cuda()
    hbm("embed_bwd_kernel (scatter-add)", "10240 rows", 2.0 * Mt * E * 4, lambda: ops.embed_bwd(idl, dy, emb_g))
    n = 18652432
    p_, g_, m_, v_ = (torch.zeros(n, device="cuda") for _ in range(4))
    hbm("adam_kernel", "18.65 M parameters", 7.0 * n * 4, lambda: ops.adam_step(p_, g_, m_, v_, 1e-3, 1))
    # ---- CNN-RNN, cfg1 shapes scaled to a full batch ----
    Br, Er, Hd = 1024, 256, 512
    xr, h0, c0 = rnd(Br, Er), rnd(Br, Hd), rnd(Br, Hd)
    wih, whh, bih, bhh = rnd(4 * Hd, Er), rnd(4 * Hd, Hd), rnd(4 * Hd), rnd(4 * Hd)
    mfma("lstm_step_kernel (gates + cell)", "B=1024,E=256,Hd=512", 2.0 * Br * (Er + Hd) * 4 * Hd,
         lambda: ops.lstm_step(xr, h0, c0, wih, whh, bih, bhh))
    x32, h32, c32 = rnd(32, Er), rnd(32, Hd), rnd(32, Hd)
    mfma("lstm_step_small_lds_kernel (gates + cell, batch 32)", "B=32,E=256,Hd=512", 2.0 * 32 * (Er + Hd) * 4 * Hd,
         lambda: ops.lstm_step(x32, h32, c32, wih, whh, bih, bhh))
    Wv5, bv5, x5 = rnd(5000, Hd), rnd(5000), rnd(32, Hd)
    mfma("vocab_argmax_small_kernel + merge (batch 32, V=5000)", "32x5000x512", 2.0 * 32 * 5000 * Hd, lambda: ops.vocab_argmax(x5, Wv5, bv5))
    Bs, Ss, As = 1024, 64, 512
    u, pfm, vv, encm = rnd(Bs, As), rnd(Bs * Ss, As), rnd(As), rnd(Bs * Ss, E)
    ctx = torch.empty(Bs, E, device="cuda")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    hbm("additive_attention_kernel (soft-attention LSTM step)", "B=1024,S=64,A=512,E=512", 1.0 * Bs * Ss * (As + E) * 4,
        lambda: lib.capdec_additive_attention(u.data_ptr(), As, pfm.data_ptr(), vv.data_ptr(), encm.data_ptr(), Bs, Ss, As, E,
                                              ctx.data_ptr(), E, None, st))
    print("| kernel | shape | algorithmic work | avg µs | achieved | of roof |")
    print("|---|---|---|---|---|---|")
    for name, shape, work, us, rate, frac in rows:
        print(f"| `{name}` | {shape} | {work} | {us:.1f} | {rate} | {100 * frac:.0f} % |")


if __name__ == "__main__":
    main()
