"""Explicit forward / backward of the caption decoders on the HIP kernels, plus thin
``torch.autograd.Function`` wrappers so that ``loss.backward()`` in an unmodified training loop
(reference train.py:252-262) lands in these kernels.

No ATen arithmetic is used: torch only allocates tensors.  Every saved activation is a plain tensor held
in a ``Tape``; gradients of parameters are written into ``grads[name]`` tensors that the caller may
pre-allocate (e.g. as views of one flat buffer for a single RCCL all-reduce and a fused Adam step).
"""
import threading

import torch

from . import _lib, ops

_SITE = 0x51ED2701  # odd stride between the dropout seeds of different sites


def draw_seed():
    """A fresh 62-bit dropout seed from torch's CPU generator (so torch.manual_seed makes runs repeatable)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class Tape:
    pass


def _new(grads, name, like):
    """Gradient buffer for parameter ``name``: the caller's (pre-zeroed) tensor, or a fresh zero tensor."""
    g = grads.get(name)
    if g is None:
        g = torch.zeros_like(like)
        grads[name] = g
    return g


_TLS = threading.local()      # per host thread: the optional second stream (see use_side_stream)


def _side():
    return getattr(_TLS, "side", None)


def use_side_stream(stream):
    """Run the weight-gradient contractions (and the bias column sums) of every linear layer on ``stream`` while the
    input-gradient chain continues on the current stream: the two are independent, and a launch that cannot fill the
    chip on its own overlaps with the other stream's.  The caller joins the streams (``join_side_stream``) before it
    reads the gradients.  ``None`` switches the overlap off.  The setting belongs to the calling host thread."""
    _TLS.side = stream


def join_side_stream():
    if _side() is not None:
        torch.cuda.current_stream().wait_stream(_side())


def _lin_bwd(grads, prefix, lin, dy, x, need_dx=True, mask=None, accum=None, bias_done=False, rows=None):
    """Backward of y = x W^T + b: fills grads[prefix.weight/.bias]; returns dX (added onto ``accum`` if given).
    bias_done: the bias gradient was already accumulated by the kernel that produced dy.
    rows: int32 device scalar (a multiple of 32); only the first rows[0] rows of dy and x exist."""
    gw = _new(grads, prefix + ".weight", lin.weight)
    gb = _new(grads, prefix + ".bias", lin.bias) if (lin.bias is not None and not bias_done) else None
    return _lin_bwd_into(gw, gb, lin.weight, dy, x, need_dx, mask, accum, rows)


def _back_to_back(ts):
    """One [n*rows, ...] view over n equally shaped tensors that lie back to back in one allocation (the trainer lays
    the parameters and their gradients out that way), else None."""
    t0 = ts[0]
    step = t0.numel() * t0.element_size()
    for i, t in enumerate(ts):
        if (t.shape != t0.shape or not t.is_contiguous() or t.data_ptr() != t0.data_ptr() + i * step or
                t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr()):
            return None
    return t0.detach().as_strided((len(ts) * t0.shape[0],) + tuple(t0.shape[1:]), t0.stride())


def _stacked_projections(grads, prefix, lins):
    """(W, gW, gb) of several Linear layers of one input as one stacked layer, when their weights, their biases and the
    gradients of both lie back to back; else None."""
    if any(ln.bias is None for ln in lins):
        return None
    names = [prefix + n for n in ("W_q", "W_k", "W_v")]
    W = _back_to_back([ln.weight for ln in lins])
    gW = _back_to_back([_new(grads, n + ".weight", ln.weight) for n, ln in zip(names, lins)])
    gb = _back_to_back([_new(grads, n + ".bias", ln.bias) for n, ln in zip(names, lins)])
    return None if (W is None or gW is None or gb is None) else (W, gW, gb)


def _lin_bwd_into(gw, gb, W, dy, x, need_dx=True, mask=None, accum=None, rows=None):
    """``_lin_bwd`` on explicit gradient buffers (gb None: no bias gradient wanted)."""
    if _side() is not None:
        main = torch.cuda.current_stream()
        _side().wait_stream(main)                        # dy and x are complete on the main stream
        with torch.cuda.stream(_side()):
            ops.linear_wgrad(dy, x, out=gw, rows_dev=rows)
            if gb is not None:
                ops.colsum(dy, gb, rows_dev=rows)
        for t in (dy, x):                              # their memory must outlive the side stream's reads
            t.record_stream(_side())
    else:
        ops.linear_wgrad(dy, x, out=gw, rows_dev=rows)
        if gb is not None:
            ops.colsum(dy, gb, rows_dev=rows)
    if need_dx:
        return ops.linear_dgrad(dy, W, mask=mask, accum=accum, rows_dev=rows)
    return None


# ---------------------------------------------------------------------------------------------
# Transformer decoder stack
# ---------------------------------------------------------------------------------------------
RAGGED_MAX_BATCH = 11946      # the compaction kernel keeps three ints per sample in 144 KiB of LDS


def ragged_supported(dec, S, T, B=1):
    """Whether ``tf_forward`` can run on the live rows only: sequences short enough for the one-workgroup attention
    kernels (T <= 64 and their LDS budgets, memories up to 2048 rows) and a batch whose per-sample bookkeeping fits the
    compaction kernel's LDS."""
    E, H = dec.decoder_embedding.weight.shape[1], dec.num_heads
    dk = E // H
    if not (T <= 64 and 1 <= B <= RAGGED_MAX_BATCH and dk in (16, 32, 64, 128, 256) and
            (4 * T * (dk + 16) + 2 * T * T) * 4 <= 144 * 1024):
        return False
    if 1 <= S <= 64:       # cross-attention with the memory of a sample resident in LDS
        return (2 * (T + S) * (dk + 16) + 2 * T * S) * 4 <= 144 * 1024 and (2 * S * (dk + 16) + 4 * S) * 4 <= 64 * 1024
    # longer memories: keys streamed through LDS in chunks of 32
    return (S <= 2048 and T >= 4 and ((3 * T + 64) * (dk + 16) + 64 * T + 2 * T) * 4 <= 144 * 1024 and
            ((T + 64) * (dk + 16) + 128) * 4 <= 144 * 1024)


def cross_absorbed(S, T, E, H):
    """Whether the cross-attention sub-layer of a teacher-forced pass contracts the memory against the H*T query rows
    of a sample (key / value projections folded onto the query side) instead of projecting its S rows: multiply-adds
    per sample 2*T*(H*S + E)*E against 2*S*E*E + 2*T*S*E, taken when it is at most half."""
    dk = E // H
    return S > 64 and E % 4 == 0 and dk % 4 == 0 and 2 * T * (H * S + E) <= S * E + T * S


def _heads(t, H, width):
    """The H column blocks of ``width`` columns of a 2-D tensor, as views."""
    return [t[:, h * width:(h + 1) * width] for h in range(H)]


def _on_side(fn, keep):
    """Run ``fn`` on the side stream when there is one (after everything enqueued so far on the current stream), else
    here.  ``keep``: tensors read by it, whose memory must outlive the side stream's work."""
    side = _side()
    if side is None:
        fn()
        return
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    for tn in keep:
        tn.record_stream(side)


def _cross_absorbed_fwd(t, c, ca, rows):
    """scores = (q_h W_k,h) mem^T / sqrt(dk), probabilities, context = P mem, out_h = context W_v,h^T + b_v,h.
    Row (r, h) of the per-head tensors is row r*H + h.  Returns the sub-layer's attention output [M, E]."""
    B, S, T, E, H = t.B, t.S, t.T, t.E, t.H
    dk, M = E // H, c.q2.shape[0]
    rg = t.ragged
    seg_off = None if rg is None else rg.seq_off
    dev = c.q2.device
    c.qp = torch.empty((M * H, E), device=dev, dtype=torch.float32)
    qp_heads = _heads(c.qp.view(M, H * E), H, E)
    wk = ca.W_k.weight
    ops.gemm_layout_groups([dict(P=q, Q=wk[h * dk:(h + 1) * dk], C=qp_heads[h])
                            for h, q in enumerate(_heads(c.q2, H, dk))], "nn", rows_dev=rows)
    c.p2 = torch.empty((M * H, (S + 31) // 32 * 32), device=dev, dtype=torch.float32)
    ops.bgemm("nt", c.qp, t.enc, c.p2, B, T, H, seg_off=seg_off, scale=dk ** -0.5)
    ops.softmax_rows_(c.p2, S, live=seg_off, live_idx=B, mult=H)
    c.c2 = torch.zeros((M * H, E), device=dev, dtype=torch.float32)      # rows of no sample stay zero
    ops.bgemm("nn", c.p2, t.enc, c.c2, B, T, H, seg_off=seg_off)
    ao2 = torch.empty((M, E), device=dev, dtype=torch.float32)
    wv, bv = ca.W_v.weight, ca.W_v.bias
    ops.gemm_layout_groups([dict(P=cc, Q=wv[h * dk:(h + 1) * dk], C=o, bias_j=bv[h * dk:(h + 1) * dk])
                            for h, (cc, o) in enumerate(zip(_heads(c.c2.view(M, H * E), H, E), _heads(ao2, H, dk)))],
                           "nt", rows_dev=rows)
    return ao2


def _cross_absorbed_bwd(t, c, ca, pre, d_ao2, grads, rows, denc, first=False):
    """Backward of ``_cross_absorbed_fwd``: fills the W_k / W_v gradients, adds the memory's gradient onto ``denc``
    (when not None; ``first``: stores it instead, the buffer holds nothing yet), returns dq2 [M, E]."""
    B, S, T, E, H = t.B, t.S, t.T, t.E, t.H
    dk, M = E // H, c.q2.shape[0]
    rg = t.ragged
    seg_off = None if rg is None else rg.seq_off
    dev = d_ao2.device
    wk, wv = ca.W_k.weight, ca.W_v.weight
    g_wk, g_wv = _new(grads, pre + "cross_attn.W_k.weight", wk), _new(grads, pre + "cross_attn.W_v.weight", wv)
    _new(grads, pre + "cross_attn.W_k.bias", ca.W_k.bias)       # exactly zero: b_k drops out of the softmax
    g_bv = _new(grads, pre + "cross_attn.W_v.bias", ca.W_v.bias)
    dao_heads = _heads(d_ao2, H, dk)
    c2_heads = _heads(c.c2.view(M, H * E), H, E)

    def wv_grads():
        ops.gemm_layout_groups([dict(P=dao_heads[h], Q=c2_heads[h], C=g_wv[h * dk:(h + 1) * dk]) for h in range(H)],
                               "tn", k_dev=rows)
        ops.colsum(d_ao2, g_bv, rows_dev=rows)                   # the probabilities of a row sum to one

    _on_side(wv_grads, (d_ao2, c.c2))
    dc = torch.empty((M * H, E), device=dev, dtype=torch.float32)
    dc_heads = _heads(dc.view(M, H * E), H, E)
    ops.gemm_layout_groups([dict(P=dao_heads[h], Q=wv[h * dk:(h + 1) * dk], C=dc_heads[h]) for h in range(H)], "nn",
                           rows_dev=rows)
    dp = torch.empty_like(c.p2)
    ops.bgemm("nt", dc, t.enc, dp, B, T, H, seg_off=seg_off)
    ops.softmax_rows_bwd_(c.p2, dp, S, scale=dk ** -0.5, live=seg_off, live_idx=B, mult=H)
    dqp = torch.zeros((M * H, E), device=dev, dtype=torch.float32)   # rows of no sample must not reach the W_k gradient
    ops.bgemm("nn", dp, t.enc, dqp, B, T, H, seg_off=seg_off)
    if denc is not None:
        # d mem = P^T d context + dS^T (q W_k): one pass over the memory's gradient for both products
        ops.bgemm("tn", c.p2, dc, denc, B, T, H, seg_off=seg_off, P2=dp, Q2=c.qp, accumulate=not first)
    dqp_heads = _heads(dqp.view(M, H * E), H, E)
    q2_heads = _heads(c.q2, H, dk)
    _on_side(lambda: ops.gemm_layout_groups([dict(P=q2_heads[h], Q=dqp_heads[h], C=g_wk[h * dk:(h + 1) * dk])
                                             for h in range(H)], "tn", k_dev=rows), (c.q2, dqp))
    dq2 = torch.empty((M, E), device=dev, dtype=torch.float32)
    ops.gemm_layout_groups([dict(P=dqp_heads[h], Q=wk[h * dk:(h + 1) * dk], C=o)
                            for h, o in enumerate(_heads(dq2, H, dk))], "nt", rows_dev=rows)
    return dq2


def _pair_constants(t, ca):
    """The per-sample constants of a cross-attention over two memory rows (capdec_pair_gate_fwd in capdec.h): B rows of
    work that depends on the memory only.  Returns a Tape with kv2 [2, B*2, E], dk, dv [B, E], sb [B, H], G, D
    [B, H, E], c1 [B, E]."""
    B, E, H = t.B, t.E, t.H
    dk = E // H
    dev = t.enc.device
    pc = Tape()
    pc.kv2 = torch.empty((2, B * 2, E), device=dev, dtype=torch.float32)
    ops.gemm_groups([dict(P=t.enc, Q=ca.W_k.weight, C=pc.kv2[0], bias_j=ca.W_k.bias),
                     dict(P=t.enc, Q=ca.W_v.weight, C=pc.kv2[1], bias_j=ca.W_v.bias)], E)
    pc.dk, pc.dv, pc.sb = ops.pair_diff(pc.kv2[0], pc.kv2[1], ca.W_q.bias, H)
    pc.G = torch.empty((B, H, E), device=dev, dtype=torch.float32)
    pc.D = torch.empty((B, H, E), device=dev, dtype=torch.float32)
    g_heads, d_heads = _heads(pc.G.view(B, H * E), H, E), _heads(pc.D.view(B, H * E), H, E)
    wq, wo = ca.W_q.weight, ca.W_o.weight
    for h0 in range(0, H, 4):
        hs = range(h0, min(h0 + 4, H))
        ops.gemm_layout_groups([dict(P=pc.dk[:, h * dk:(h + 1) * dk], Q=wq[h * dk:(h + 1) * dk], C=g_heads[h]) for h in hs],
                               "nn")
        ops.gemm_layout_groups([dict(P=pc.dv[:, h * dk:(h + 1) * dk], Q=wo[:, h * dk:(h + 1) * dk], C=d_heads[h]) for h in hs],
                               "nt")
    pc.v1 = pc.kv2[1].view(B, 2 * E)[:, E:]                       # the second memory row's values, [B, E] with stride 2E
    pc.c1 = ops.linear(pc.v1, wo, ca.W_o.bias)
    return pc


def _pair_constants_bwd(t, c, ca, pre, grads, dG, dD, dc1, dsb, denc):
    """Backward of ``_pair_constants`` given the gradients of G, D, c1, sb: fills the W_q / W_k / W_v gradients and the
    weight part of W_o's, adds the memory's gradient onto ``denc`` (when not None)."""
    B, E, H = t.B, t.E, t.H
    dk = E // H
    pc = c.pc
    dev = dG.device
    wq, wo = ca.W_q.weight, ca.W_o.weight
    g_wq, g_wo = _new(grads, pre + "cross_attn.W_q.weight", wq), _new(grads, pre + "cross_attn.W_o.weight", wo)
    g_bq = _new(grads, pre + "cross_attn.W_q.bias", ca.W_q.bias)
    _new(grads, pre + "cross_attn.W_k.bias", ca.W_k.bias)          # exactly zero: b_k cancels in k0 - k1
    dg_heads, dd_heads = _heads(dG.view(B, H * E), H, E), _heads(dD.view(B, H * E), H, E)
    ops.linear_wgrad(dc1, pc.v1, out=g_wo)
    d_v1 = ops.linear_dgrad(dc1, wo)
    ddk = torch.empty((B, E), device=dev, dtype=torch.float32)
    ddv = torch.empty((B, E), device=dev, dtype=torch.float32)
    for h0 in range(0, H, 4):
        hs = range(h0, min(h0 + 4, H))
        sl = {h: slice(h * dk, (h + 1) * dk) for h in hs}
        ops.gemm_layout_groups([dict(P=dd_heads[h], Q=pc.dv[:, sl[h]], C=g_wo[:, sl[h]]) for h in hs], "tn")
        ops.gemm_layout_groups([dict(P=pc.dk[:, sl[h]], Q=dg_heads[h], C=g_wq[sl[h]]) for h in hs], "tn")
        ops.gemm_layout_groups([dict(P=dd_heads[h], Q=wo[:, sl[h]], C=ddv[:, sl[h]]) for h in hs], "nn")
        ops.gemm_layout_groups([dict(P=dg_heads[h], Q=wq[sl[h]], C=ddk[:, sl[h]]) for h in hs], "nt")
    dk2, dv2, qb = ops.pair_diff_bwd(ddk, ddv, dsb, d_v1, ca.W_q.bias, pc.dk, H)
    ops.colsum(qb, g_bq)
    _lin_bwd(grads, pre + "cross_attn.W_k", ca.W_k, dk2, t.enc, need_dx=denc is not None, accum=denc, bias_done=True)
    _lin_bwd(grads, pre + "cross_attn.W_v", ca.W_v, dv2, t.enc, need_dx=denc is not None, accum=denc)


def tf_forward(dec, enc, ids, p, seed, single_row_shortcut=True, ragged=None, seed_dev=None, absorb_cross=None,
               pair_cross=None):
    """Teacher-forced decoder stack (reference transformer.py:125-132) keeping what the backward needs.
    enc [B,S,E] fp32, ids [B,T] int64.  Returns (x [B*T,E] post-LN output of the last layer, tape).

    ragged (ops.RaggedRows of the same ids and their targets; needs ``ragged_supported``): the stack runs on the live
    rows only (the memories, and with them the K/V projections of the cross-attention, stay dense).  Every [B*T,E] tensor is then indexed by slot, the first ``ragged.rows[0]`` rows exist, and the caller
    takes the loss against ``ragged.targets_c`` with that row limit.  The dropout masks are drawn per slot.
    seed_dev (int64 device tensor, optional): its first element is added to ``seed`` by every kernel that draws a mask
    (a step captured in a HIP graph keeps ``seed`` for ever; the device value changes from replay to replay).
    absorb_cross: None = decide by ``cross_absorbed`` (long memories, few query rows); True / False force the form of
    the cross-attention sub-layer (the two forms differ only in the order of the floating-point sums).
    pair_cross: None = memories of exactly two rows take the logistic closed form (``_pair_constants``); False keeps
    them on the general kernels."""
    B, S, E = enc.shape
    T = ids.shape[1]
    M, H = B * T, dec.num_heads
    t = Tape()
    t.B, t.S, t.T, t.E, t.H, t.p, t.seed, t.seed_dev = B, S, T, E, H, p, seed, seed_dev
    t.shortcut = S == 1 and single_row_shortcut
    t.pair = S == 2 and pair_cross is not False and not absorb_cross and E % (4 * H) == 0 and E <= 1024
    t.absorbed = not (t.shortcut or t.pair) and absorb_cross is not False and (absorb_cross or cross_absorbed(S, T, E, H))
    t.enc = enc.reshape(B * S, E)
    pe = dec.positional_encoding.pe[0]
    if T > pe.shape[0]:
        raise RuntimeError(f"capdec: sequence length {T} exceeds the positional-encoding buffer ({pe.shape[0]})")
    t.ragged = ragged
    rows = None
    # embedding + position + dropout (and, on a ragged batch, the move of the live rows to their slots) as one launch;
    # the ids may be a strided view (captions[:, :-1])
    if ragged is not None:
        if not ragged_supported(dec, S, T, B) or (ragged.B, ragged.T) != (B, T):
            raise RuntimeError("capdec tf_forward: this configuration cannot run on a ragged batch (see ragged_supported)")
        rows = ragged.rows
        t.ids = t.ids32 = None                       # (the dense ids are not needed again)
        x, t.ids_c = ops.embed_train(ids, dec.decoder_embedding.weight, pe, p, seed, seed_dev=seed_dev, order=ragged.order,
                                     want_ids=True)
    else:
        t.ids = ids.contiguous()
        t.ids32 = t.ids.to(torch.int32)
        x = ops.embed_train(t.ids, dec.decoder_embedding.weight, pe, p, seed, seed_dev=seed_dev)
    t.layers = []
    pre_cc = None
    if t.shortcut and _side() is not None:
        # the per-sample cross-attention constants depend on enc only: all layers' worth on the side stream now
        main = torch.cuda.current_stream()
        _side().wait_stream(main)
        pre_cc = []
        with torch.cuda.stream(_side()):
            for ly in dec.decoder_layers:
                ca = ly.cross_attn
                v2 = ops.linear(t.enc, ca.W_v.weight, ca.W_v.bias)
                pre_cc.append((v2, ops.linear(v2, ca.W_o.weight, ca.W_o.bias)))
        t.enc.record_stream(_side())
    pre_pc = None
    if t.pair and _side() is not None:
        # likewise the constants of the two-row closed form
        side = _side()
        side.wait_stream(torch.cuda.current_stream())
        use_side_stream(None)
        try:
            with torch.cuda.stream(side):
                pre_pc = [_pair_constants(t, ly.cross_attn) for ly in dec.decoder_layers]
        finally:
            use_side_stream(side)
        t.enc.record_stream(side)
    for l, ly in enumerate(dec.decoder_layers):
        c = Tape()
        sd = seed + (3 * l + 1) * _SITE
        c.x = x
        sa, ca, ff = ly.self_attn, ly.cross_attn, ly.feed_forward
        qkv = torch.empty((3, M, E), device=x.device, dtype=torch.float32)
        ops.gemm_groups([dict(P=x, Q=sa.W_q.weight, C=qkv[0], bias_j=sa.W_q.bias),
                         dict(P=x, Q=sa.W_k.weight, C=qkv[1], bias_j=sa.W_k.bias),
                         dict(P=x, Q=sa.W_v.weight, C=qkv[2], bias_j=sa.W_v.bias)], E, rows_dev=rows)
        c.qkv = qkv
        if ragged is not None:
            c.ao, c.lse = ops.attention_ragged(qkv[0], qkv[1], qkv[2], ragged, H, want_lse=True)
        else:
            c.ao, c.lse = ops.attention(qkv[0], qkv[1], qkv[2], B, H, T, T, causal=True, ids32=t.ids32, want_lse=True)
        a = ops.linear(c.ao, sa.W_o.weight, sa.W_o.bias, rows_dev=rows)
        c.x1, c.s1, c.m1, c.r1 = ops.add_layernorm_train(a, x, ly.norm1.weight, ly.norm1.bias, p, sd, rows_dev=rows,
                                                         seed_dev=seed_dev)
        if S == 1 and single_row_shortcut:
            # one memory row: the softmax over it is exactly 1, so the sub-layer output is W_o(W_v enc + b_v) + b_o
            # for every position, whatever the query (reference transformer.py:13-22 with Tk = 1)
            if pre_cc is not None:
                if l == 0:
                    torch.cuda.current_stream().wait_stream(_side())
                c.v2, c.cc = pre_cc[l]                                                 # [B,E] each
                c.cc.record_stream(torch.cuda.current_stream())
            else:
                c.v2 = ops.linear(t.enc, ca.W_v.weight, ca.W_v.bias)                   # [B,E]
                c.cc = ops.linear(c.v2, ca.W_o.weight, ca.W_o.bias)                    # [B,E]
            c.x2, c.s2, c.m2, c.r2 = ops.add_layernorm_train(c.cc, c.x1, ly.norm2.weight, ly.norm2.bias, p,
                                                             sd + _SITE, x_div=T, rows_dev=rows,
                                                             x_map=None if ragged is None else ragged.sample,
                                                             seed_dev=seed_dev)
        elif t.pair:
            if pre_pc is not None:
                if l == 0:
                    torch.cuda.current_stream().wait_stream(_side())
                c.pc = pre_pc[l]
                for tn in (c.pc.G, c.pc.D, c.pc.c1, c.pc.sb):
                    tn.record_stream(torch.cuda.current_stream())
            else:
                c.pc = _pair_constants(t, ca)
            a2, c.p2 = ops.pair_gate_fwd(c.x1, c.pc.G, c.pc.D, c.pc.c1, c.pc.sb, H, (E // H) ** -0.5, T,
                                         sample=None if ragged is None else ragged.sample, rows_dev=rows)
            c.x2, c.s2, c.m2, c.r2 = ops.add_layernorm_train(a2, c.x1, ly.norm2.weight, ly.norm2.bias, p, sd + _SITE,
                                                             rows_dev=rows, seed_dev=seed_dev)
        else:
            c.q2 = ops.linear(c.x1, ca.W_q.weight, ca.W_q.bias, rows_dev=rows)
            if t.absorbed:
                c.ao2 = _cross_absorbed_fwd(t, c, ca, rows)
            else:
                kv2 = torch.empty((2, B * S, E), device=x.device, dtype=torch.float32)
                ops.gemm_groups([dict(P=t.enc, Q=ca.W_k.weight, C=kv2[0], bias_j=ca.W_k.bias),
                                 dict(P=t.enc, Q=ca.W_v.weight, C=kv2[1], bias_j=ca.W_v.bias)], E)
                c.kv2 = kv2
                if ragged is not None:
                    c.ao2, c.lse2 = ops.cross_attention_ragged(c.q2, kv2[0], kv2[1], ragged, H, S, want_lse=True)
                else:
                    c.ao2, c.lse2 = ops.attention(c.q2, kv2[0], kv2[1], B, H, T, S, want_lse=True)
            a2 = ops.linear(c.ao2, ca.W_o.weight, ca.W_o.bias, rows_dev=rows)
            c.x2, c.s2, c.m2, c.r2 = ops.add_layernorm_train(a2, c.x1, ly.norm2.weight, ly.norm2.bias, p, sd + _SITE,
                                                             rows_dev=rows, seed_dev=seed_dev)
        c.h = ops.linear(c.x2, ff.fc1.weight, ff.fc1.bias, relu=True, rows_dev=rows)
        f = ops.linear(c.h, ff.fc2.weight, ff.fc2.bias, rows_dev=rows)
        x, c.s3, c.m3, c.r3 = ops.add_layernorm_train(f, c.x2, ly.norm3.weight, ly.norm3.bias, p, sd + 2 * _SITE,
                                                      rows_dev=rows, seed_dev=seed_dev)
        t.layers.append(c)
    t.out = x
    return x, t


def tf_backward(dec, t, dx, grads, need_denc=True, on_layer_done=None):
    """Backward of ``tf_forward``.  dx [B*T,E].  Fills ``grads`` (keys = names under ``dec``); returns denc
    [B,S,E] (or None)."""
    B, S, T, E, H, p = t.B, t.S, t.T, t.E, t.H, t.p
    rg = t.ragged
    rows = None if rg is None else rg.rows
    side_used = False
    denc = None
    if need_denc:
        # the folded cross-attention's first contribution overwrites the buffer: no zero fill of B*S*E floats
        denc = (torch.empty if t.absorbed else torch.zeros)((B * S, E), device=dx.device, dtype=torch.float32)
    for l in range(len(t.layers) - 1, -1, -1):
        ly, c = dec.decoder_layers[l], t.layers[l]
        pre = f"decoder_layers.{l}."
        sd = t.seed + (3 * l + 1) * _SITE
        sa, ca, ff = ly.self_attn, ly.cross_attn, ly.feed_forward
        # x3 = LN3(x2 + drop(f)),  f = relu(x2 W1^T + b1) W2^T + b2
        # (the bias gradient of the layer in front of each LayerNorm is the column sum of that LayerNorm's d_x: the
        #  LayerNorm backward kernel accumulates it next to dgamma / dbeta)
        d_x2, d_f = ops.layernorm_bwd(dx, c.s3, c.m3, c.r3, ly.norm3.weight, p, sd + 2 * _SITE,
                                      _new(grads, pre + "norm3.weight", ly.norm3.weight),
                                      _new(grads, pre + "norm3.bias", ly.norm3.bias),
                                      dxsum=_new(grads, pre + "feed_forward.fc2.bias", ff.fc2.bias), rows_dev=rows,
                                      seed_dev=t.seed_dev)
        dh = _lin_bwd(grads, pre + "feed_forward.fc2", ff.fc2, d_f, c.h, mask=c.h, bias_done=True, rows=rows)   # ReLU backward fused
        _lin_bwd(grads, pre + "feed_forward.fc1", ff.fc1, dh, c.x2, accum=d_x2, rows=rows)
        # x2 = LN2(x1 + drop(a2)),  a2 = attn(x1 Wq, enc Wk, enc Wv) Wo
        d_x1, d_a2 = ops.layernorm_bwd(d_x2, c.s2, c.m2, c.r2, ly.norm2.weight, p, sd + _SITE,
                                       _new(grads, pre + "norm2.weight", ly.norm2.weight),
                                       _new(grads, pre + "norm2.bias", ly.norm2.bias),
                                       dxsum=_new(grads, pre + "cross_attn.W_o.bias", ca.W_o.bias), rows_dev=rows,
                                       seed_dev=t.seed_dev)
        if t.shortcut:
            def cross_constant_bwd():
                d_cc = ops.group_sum(d_a2, T) if rg is None else ops.group_sum_segments(d_a2, rg.seq_off, B)   # [B,E]
                d_v2 = _lin_bwd(grads, pre + "cross_attn.W_o", ca.W_o, d_cc, c.v2, bias_done=True)
                _lin_bwd(grads, pre + "cross_attn.W_v", ca.W_v, d_v2, t.enc, need_dx=need_denc, accum=denc)
            side = _side()
            if side is not None:
                # nothing on the main chain needs these B-row contractions: the whole sub-chain goes to the side stream
                side.wait_stream(torch.cuda.current_stream())
                use_side_stream(None)
                try:
                    with torch.cuda.stream(side):
                        cross_constant_bwd()
                finally:
                    use_side_stream(side)
                for tn in (d_a2, c.v2, t.enc) + ((denc,) if denc is not None else ()):
                    tn.record_stream(side)
                side_used = True
            else:
                cross_constant_bwd()
            for n, prm in (("W_q.weight", ca.W_q.weight), ("W_q.bias", ca.W_q.bias), ("W_k.weight", ca.W_k.weight),
                           ("W_k.bias", ca.W_k.bias)):
                _new(grads, pre + "cross_attn." + n, prm)                              # exactly zero
        elif t.pair:
            dG, dD, dc1, dsb = ops.pair_gate_bwd(d_a2, c.x1, c.p2, c.pc.G, c.pc.D, B, T, H, (E // H) ** -0.5, d_x1,
                                                 seg_off=None if rg is None else rg.seq_off)
            side = _side()
            if side is not None:
                # B rows of work that nothing on the main chain waits for
                side.wait_stream(torch.cuda.current_stream())
                use_side_stream(None)
                try:
                    with torch.cuda.stream(side):
                        _pair_constants_bwd(t, c, ca, pre, grads, dG, dD, dc1, dsb, denc)
                finally:
                    use_side_stream(side)
                for tn in (dG, dD, dc1, dsb, t.enc) + ((denc,) if denc is not None else ()):
                    tn.record_stream(side)
                side_used = True
            else:
                _pair_constants_bwd(t, c, ca, pre, grads, dG, dD, dc1, dsb, denc)
        else:
            d_ao2 = _lin_bwd(grads, pre + "cross_attn.W_o", ca.W_o, d_a2, c.ao2, bias_done=True, rows=rows)
            if t.absorbed:
                dq2 = _cross_absorbed_bwd(t, c, ca, pre, d_ao2, grads, rows, denc, first=l == len(t.layers) - 1)
                _lin_bwd(grads, pre + "cross_attn.W_q", ca.W_q, dq2, c.x1, accum=d_x1, rows=rows)
            else:
                if rg is not None:
                    dq2, dk2, dv2 = ops.cross_attention_bwd_ragged(c.q2, c.kv2[0], c.kv2[1], d_ao2, c.lse2, rg, H, S,
                                                                   out=c.ao2)
                else:
                    dq2, dk2, dv2 = ops.attention_bwd(c.q2, c.kv2[0], c.kv2[1], d_ao2, c.lse2, B, H, T, S, out=c.ao2)
                _lin_bwd(grads, pre + "cross_attn.W_q", ca.W_q, dq2, c.x1, accum=d_x1, rows=rows)
                _lin_bwd(grads, pre + "cross_attn.W_k", ca.W_k, dk2, t.enc, need_dx=need_denc, accum=denc)
                _lin_bwd(grads, pre + "cross_attn.W_v", ca.W_v, dv2, t.enc, need_dx=need_denc, accum=denc)
        # x1 = LN1(x + drop(a)),  a = attn(x Wq, x Wk, x Wv; causal, PAD rows) Wo
        d_x, d_a = ops.layernorm_bwd(d_x1, c.s1, c.m1, c.r1, ly.norm1.weight, p, sd,
                                     _new(grads, pre + "norm1.weight", ly.norm1.weight),
                                     _new(grads, pre + "norm1.bias", ly.norm1.bias),
                                     dxsum=_new(grads, pre + "self_attn.W_o.bias", sa.W_o.bias), rows_dev=rows,
                                     seed_dev=t.seed_dev)
        d_ao = _lin_bwd(grads, pre + "self_attn.W_o", sa.W_o, d_a, c.ao, bias_done=True, rows=rows)
        stk = _stacked_projections(grads, pre + "self_attn.", (sa.W_q, sa.W_k, sa.W_v))
        if rg is not None:
            dqkv = ops.attention_bwd_ragged(c.qkv[0], c.qkv[1], c.qkv[2], d_ao, c.lse, rg, H, packed=stk is not None)
        else:
            dqkv = ops.attention_bwd(c.qkv[0], c.qkv[1], c.qkv[2], d_ao, c.lse, B, H, T, T, causal=True, ids32=t.ids32,
                                     packed=stk is not None)
        if stk is not None:
            # the three projections as one layer of 3E outputs: one contraction each for the weight gradients, the bias
            # gradients and the input gradient
            _lin_bwd_into(stk[1], stk[2], stk[0], dqkv, c.x, accum=d_x, rows=rows)
        else:
            dq, dk, dv = dqkv
            _lin_bwd(grads, pre + "self_attn.W_q", sa.W_q, dq, c.x, accum=d_x, rows=rows)
            _lin_bwd(grads, pre + "self_attn.W_k", sa.W_k, dk, c.x, accum=d_x, rows=rows)
            _lin_bwd(grads, pre + "self_attn.W_v", sa.W_v, dv, c.x, accum=d_x, rows=rows)
        dx = d_x
        if on_layer_done is not None:
            on_layer_done(l)          # every gradient of layer l is now enqueued
    # (the dropout of the embedding rows is undone by the scatter kernel itself: same mask, no launch in between)
    ops.embed_bwd(t.ids.view(-1) if rg is None else t.ids_c, dx,
                  _new(grads, "decoder_embedding.weight", dec.decoder_embedding.weight), rows_dev=rows, p=p, seed=t.seed,
                  seed_dev=t.seed_dev)
    if side_used and need_denc:
        torch.cuda.current_stream().wait_stream(_side())        # denc was accumulated on the side stream
    return denc.view(B, S, E) if need_denc else None


def vocab_logits_bwd(dec, x, dlogits, grads):
    """Backward of logits = x fc2^T + b given dlogits [M,V]; returns dx."""
    return _lin_bwd(grads, "fc2", dec.fc2, dlogits, x)


def vocab_loss(dec, x, targets, ignore_index=0, materialize=True, skip_ignored=True, rows=None):
    """Projection + CrossEntropyLoss(ignore_index) (reference train.py:216,254-256).
    Returns (loss [1] device tensor, saved state for ``vocab_loss_bwd``).

    materialize=True (training): the logits are written once and turned in place into the softmax gradient, which
    the backward pass contracts with fc2 and x; on this machine writing and re-reading 4*M*V bytes is several times
    cheaper than recomputing the M x V x E contraction.  With skip_ignored the rows whose target is ``ignore_index``
    (<PAD>: a quarter to a half of a padded caption batch) are moved behind the counted ones by a device-side
    compaction and take no part in the projection, the softmax or the two backward contractions: their loss term and
    every gradient through them are exactly zero in the reference too.  Nothing is synchronised: the number of
    counted rows stays on the device and bounds the kernels' row and contraction ranges.
    materialize=False: fused log-sum-exp + NLL, no logits (evaluation loss; its backward recomputes the projection
    tile-wise)."""
    tg = targets.reshape(-1).contiguous()
    if rows is not None:
        # x comes from a ragged stack: only its first rows[0] rows exist, the dead rows are gone already
        M, V = tg.numel(), dec.fc2.weight.shape[0]
        dlogits = torch.empty((M, V), device=x.device, dtype=torch.float32)
        ops.gemm_raw(x, dec.fc2.weight, dlogits, M, V, x.shape[1], x.stride(0), dec.fc2.weight.stride(0),
                     bias_j=dec.fc2.bias, rows_dev=rows)
        sums, _ = ops.softmax_nll_grad_(dlogits, tg, ignore_index, rows_dev=rows)
        loss, _ = ops.loss_finish(sums)
        return loss, ("rows", dlogits, rows)
    if materialize and skip_ignored:
        M = tg.numel()
        order, slot, tg_c, count = ops.compact_targets(tg, ignore_index)
        xc = x.index_select(0, order)
        dlogits = torch.empty((M, dec.fc2.weight.shape[0]), device=x.device, dtype=torch.float32)
        ops.gemm_raw(xc, dec.fc2.weight, dlogits, M, dlogits.shape[1], xc.shape[1], xc.stride(0), dec.fc2.weight.stride(0),
                     bias_j=dec.fc2.bias, rows_dev=count)
        sums, _ = ops.softmax_nll_grad_(dlogits, tg_c, ignore_index, rows_dev=count)
        loss, _ = ops.loss_finish(sums)
        return loss, ("compact", dlogits, xc, slot, count)
    if materialize:
        dlogits = ops.linear(x, dec.fc2.weight, dec.fc2.bias)
        sums, _ = ops.softmax_nll_grad_(dlogits, tg, ignore_index)
        loss, _ = ops.loss_finish(sums)
        return loss, ("dense", dlogits)
    lse, nll, sums = ops.vocab_nll(x, dec.fc2.weight, dec.fc2.bias, tg, ignore_index)
    loss, gscale = ops.loss_finish(sums)
    return loss, ("fused", lse, gscale, tg, ignore_index)


def vocab_loss_bwd(dec, x, saved, grads):
    """d(loss)/dx and the fc2 gradients from what ``vocab_loss`` saved."""
    if saved[0] == "rows":
        _, dlogits, rows = saved
        return _lin_bwd(grads, "fc2", dec.fc2, dlogits, x, rows=rows)
    if saved[0] == "compact":
        _, dlogits, xc, slot, count = saved
        M, E = xc.shape
        W = dec.fc2.weight
        ops.linear_wgrad(dlogits, xc, out=_new(grads, "fc2.weight", W), rows_dev=count)
        if dec.fc2.bias is not None:
            ops.colsum(dlogits, _new(grads, "fc2.bias", dec.fc2.bias), rows_dev=count)
        # compacted d(loss)/dx with an all-zero extra row that the ignored rows pick up
        dxc = torch.zeros((M + 1, E), device=x.device, dtype=torch.float32)
        ops.gemm_raw(dlogits, W, dxc[:M], M, E, W.shape[0], dlogits.stride(0), W.stride(0), _lib.GEMM_TQ, rows_dev=count)
        return dxc.index_select(0, slot)
    if saved[0] == "dense":
        dlogits = saved[1]
    else:
        _, lse, gscale, targets, ignore_index = saved
        dlogits = ops.softmax_grad_logits(x, dec.fc2.weight, dec.fc2.bias, lse, targets, gscale, ignore_index)
    return _lin_bwd(grads, "fc2", dec.fc2, dlogits, x)


# ---------------------------------------------------------------------------------------------
# feature heads: Linear -> dropout -> BatchNorm1d (training statistics)
# ---------------------------------------------------------------------------------------------
def head_forward(model, parts, x, dropout_first, seed, seed_dev=None):
    """Training-mode head; returns (y [B, n*E], tape).  Updates the BatchNorm running buffers in place."""
    bn = model.batchnorm
    p = model.dropout.p
    t = Tape()
    t.parts, t.p, t.seed, t.dropout_first, t.seed_dev = parts, p, seed, dropout_first, seed_dev
    xin = x.contiguous()
    if dropout_first and p > 0:
        xin = ops.dropout(xin, p, seed, seed_dev=seed_dev)
    t.xin = xin
    B = x.shape[0]
    E = parts[0][0].weight.shape[0]
    z = torch.empty((B, len(parts) * E), device=x.device, dtype=torch.float32)
    for n, (fc, lo, hi) in enumerate(parts):
        ops.linear(xin[:, lo:hi], fc.weight, fc.bias, out=z[:, n * E:(n + 1) * E], split_k=(hi - lo >= 8192))
    if not dropout_first and p > 0:
        z = ops.dropout(z, p, seed, seed_dev=seed_dev)
    t.z = z
    momentum = 0.1 if bn.momentum is None else bn.momentum
    y, t.mean, t.rstd = ops.bn_train_forward(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps)
    bn.num_batches_tracked += 1
    return y, t


def head_backward(model, names, t, dy, grads):
    """names[n] is the attribute name of parts[n][0] (e.g. 'fc1', 'fc_vit', 'fc_cnn')."""
    bn = model.batchnorm
    dz = ops.bn_train_backward(dy.contiguous(), t.z, bn.weight, t.mean, t.rstd,
                               _new(grads, "batchnorm.weight", bn.weight), _new(grads, "batchnorm.bias", bn.bias))
    if not t.dropout_first and t.p > 0:
        dz = ops.dropout(dz, t.p, t.seed, seed_dev=t.seed_dev)
    E = t.parts[0][0].weight.shape[0]
    for n, (fc, lo, hi) in enumerate(t.parts):
        _lin_bwd(grads, names[n], fc, dz[:, n * E:(n + 1) * E], t.xin[:, lo:hi], need_dx=False)


# ---------------------------------------------------------------------------------------------
# torch.autograd bridges (drop-in use: model(imgs, captions) -> logits; loss.backward())
# ---------------------------------------------------------------------------------------------
def _decoder_params(dec):
    """(name, tensor) of every parameter of the decoder proper, in a fixed order."""
    out = []
    for name, prm in dec.named_parameters():
        if name.startswith("decoder_layers.") or name.startswith("fc2.") or name.startswith("decoder_embedding."):
            out.append((name, prm))
    return out


class _DecoderLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dec, captions, enc, *params):
        p = dec.dropout.p if dec.training else 0.0
        x, tape = tf_forward(dec, enc.detach(), captions, p, draw_seed() if p > 0 else 0)
        B, T = captions.shape
        logits = ops.linear(x, dec.fc2.weight, dec.fc2.bias).view(B, T, -1)
        ctx.dec, ctx.tape = dec, tape
        ctx.need_denc = enc.requires_grad
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        dec, tape = ctx.dec, ctx.tape
        grads = {}
        V = dlogits.shape[-1]
        dx = vocab_logits_bwd(dec, tape.out, dlogits.contiguous().view(-1, V), grads)
        denc = tf_backward(dec, tape, dx, grads, need_denc=ctx.need_denc)
        return (None, None, denc) + tuple(grads.get(n) for n, _ in _decoder_params(dec))


def decoder_forward(dec, enc_output, captions, take_last=False):
    if take_last:
        raise NotImplementedError("take_last is an inference option (greedy decode); call it under torch.no_grad()")
    if not enc_output.is_cuda or not captions.is_cuda:
        raise RuntimeError("capdec: inputs must live on the GPU (HIP); there is no CPU path")
    return _DecoderLogitsFn.apply(dec, captions, enc_output, *[p for _, p in _decoder_params(dec)])


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, parts, names, dropout_first, x, *params):
        p = model.dropout.p
        y, tape = head_forward(model, parts, x.detach(), dropout_first, draw_seed() if p > 0 else 0)
        ctx.model, ctx.tape, ctx.names = model, tape, names
        return y

    @staticmethod
    def backward(ctx, dy):
        grads = {}
        head_backward(ctx.model, ctx.names, ctx.tape, dy, grads)
        order = []
        for n in ctx.names:
            fc = getattr(ctx.model, n)
            order.append(grads.get(n + ".weight"))
            if fc.bias is not None:
                order.append(grads.get(n + ".bias"))
        order += [grads.get("batchnorm.weight"), grads.get("batchnorm.bias")]
        return (None, None, None, None, None) + tuple(order)


def head_train(model, parts, x, dropout_first):
    names = []
    for fc, _, _ in parts:
        names.append(next(n for n, m in model.named_children() if m is fc))
    params = []
    for n in names:
        fc = getattr(model, n)
        params.append(fc.weight)
        if fc.bias is not None:
            params.append(fc.bias)
    params += [model.batchnorm.weight, model.batchnorm.bias]
    if x.requires_grad:
        raise NotImplementedError("capdec: gradients w.r.t. the backbone features are not part of the hot path")
    return _HeadFn.apply(model, parts, names, dropout_first, x, *params)


# ---------------------------------------------------------------------------------------------
# CNN-RNN: image step + teacher-forced LSTM (reference cnnrnn_model.py:20-34)
# ---------------------------------------------------------------------------------------------
def _lstm_w(rnn):
    return (rnn.lstm.weight_ih_l0, rnn.lstm.weight_hh_l0, rnn.lstm.bias_ih_l0, rnn.lstm.bias_hh_l0)


def rnn_forward(rnn, x0, ids, p, seed, seed_dev=None):
    """LSTM over [image projection x0, dropout(embed(ids))] from the zero state, keeping what the backward
    needs.  Everything is stored time-major (row = step * B + sample) so that every step is a contiguous
    [B, .] block and the weight gradients are single contractions over all steps.
    Returns (h [T*B, Hd] time-major outputs of the caption steps, tape)."""
    B, T = ids.shape
    E = x0.shape[1]
    w_ih, w_hh, b_ih, b_hh = _lstm_w(rnn)
    Hd = w_hh.shape[1]
    dev = x0.device
    t = Tape()
    t.B, t.T, t.E, t.Hd, t.p, t.seed, t.seed_dev = B, T, E, Hd, p, seed, seed_dev
    t.ids_tm = ids.t().contiguous()                                   # [T,B]
    n = (T + 1) * B
    t.x = torch.empty((n, E), device=dev, dtype=torch.float32)
    t.h = torch.empty((n, Hd), device=dev, dtype=torch.float32)
    t.c = torch.empty((n, Hd), device=dev, dtype=torch.float32)
    t.g = torch.empty((n, 4 * Hd), device=dev, dtype=torch.float32)
    t.x[:B].copy_(x0)
    if T > 0:
        if p > 0:
            ops.dropout(ops.embed_pe(t.ids_tm, rnn.embed.weight, None), p, seed, out=t.x[B:], seed_dev=seed_dev)
        else:
            ops.embed_pe(t.ids_tm, rnn.embed.weight, None, out=t.x[B:])
    lib_step = ops.lstm_step_into
    lib_step(t.x[:B], None, None, w_ih, w_hh, b_ih, b_hh, t.h[:B], t.c[:B], t.g[:B])
    for s in range(1, T + 1):
        a, b, c = (s - 1) * B, s * B, (s + 1) * B
        lib_step(t.x[b:c], t.h[a:b], t.c[a:b], w_ih, w_hh, b_ih, b_hh, t.h[b:c], t.c[b:c], t.g[b:c])
    return t.h[B:], t


def rnn_backward(rnn, t, dh_out, grads):
    """Backward of ``rnn_forward``: dh_out [T*B, Hd] time-major.  Fills the lstm.* and embed.weight gradients;
    returns d(x0) [B,E]."""
    B, T, E, Hd = t.B, t.T, t.E, t.Hd
    w_ih, w_hh, b_ih, b_hh = _lstm_w(rnn)
    dg = torch.empty_like(t.g)
    dh_rec, dc = None, None
    for s in range(T, -1, -1):
        a, b, c = (s - 1) * B, s * B, (s + 1) * B
        dc = ops.lstm_step_bwd(t.g[b:c], t.c[a:b] if s > 0 else None, t.c[b:c],
                               dh_out[a:b] if s > 0 else None, dh_rec, dc, dg[b:c])
        if s > 0:
            dh_rec = ops.linear_dgrad(dg[b:c], w_hh)
    ops.linear_wgrad(dg, t.x, out=_new(grads, "lstm.weight_ih_l0", w_ih))
    if T > 0:
        ops.linear_wgrad(dg[B:], t.h[:T * B], out=_new(grads, "lstm.weight_hh_l0", w_hh))
    else:
        _new(grads, "lstm.weight_hh_l0", w_hh)
    ops.colsum(dg, _new(grads, "lstm.bias_ih_l0", b_ih))
    ops.colsum(dg, _new(grads, "lstm.bias_hh_l0", b_hh))
    dx = ops.linear_dgrad(dg, w_ih)
    if T > 0:
        demb = dx[B:]
        if t.p > 0:
            demb = ops.dropout(demb, t.p, t.seed, seed_dev=t.seed_dev)
        ops.embed_bwd(t.ids_tm.view(-1), demb, _new(grads, "embed.weight", rnn.embed.weight))
    else:
        _new(grads, "embed.weight", rnn.embed.weight)
    return dx[:B]


_RNN_PARAMS = ("embed.weight", "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
               "fc2.weight", "fc2.bias")


class _RnnLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rnn, captions, x0, *params):
        p = rnn.dropout.p if rnn.training else 0.0
        h, tape = rnn_forward(rnn, x0.detach().contiguous(), captions, p, draw_seed() if p > 0 else 0)
        B, T = captions.shape
        logits_tm = ops.linear(h, rnn.fc2.weight, rnn.fc2.bias)          # [T*B, V]
        ctx.rnn, ctx.tape, ctx.h = rnn, tape, h
        return logits_tm.view(T, B, -1).transpose(0, 1)                  # [B,T,V] view of the time-major rows

    @staticmethod
    def backward(ctx, dlogits):
        rnn, tape = ctx.rnn, ctx.tape
        grads = {}
        V = dlogits.shape[-1]
        d_tm = dlogits.transpose(0, 1).contiguous().view(-1, V)          # back to time-major rows
        dh = _lin_bwd(grads, "fc2", rnn.fc2, d_tm, ctx.h)
        dx0 = rnn_backward(rnn, tape, dh, grads)
        return (None, None, dx0) + tuple(grads.get(n) for n in _RNN_PARAMS)


def decoder_rnn_forward(rnn, features, captions):
    """Autograd-connected ``DecoderRNN.forward`` (reference cnnrnn_model.py:20-34): logits [B,T,V]."""
    if not features.is_cuda or not captions.is_cuda:
        raise RuntimeError("capdec: inputs must live on the GPU (HIP); there is no CPU path")
    x0 = rnn.image_input(features)                                       # head: autograd-connected in train mode
    named = dict(rnn.named_parameters())
    return _RnnLogitsFn.apply(rnn, captions, x0, *[named[n] for n in _RNN_PARAMS])


# ---------------------------------------------------------------------------------------------
# Soft-attention LSTM decoder (models/softattn_model.py; no counterpart in the reference): teacher-forced forward
# and backward.  Built from the same step kernels as the CNN-RNN decoder above -- capdec_lstm_step /
# capdec_lstm_step_bwd over a time-major tape -- plus the additive-attention operand: the LSTM input row is
# [embedding | context] and the context, attended with the previous state, adds one more recurrent path
# (h -> att_h -> scores -> context -> gates).  Without that operand the loops below are rnn_forward / rnn_backward.
# ---------------------------------------------------------------------------------------------
_SATT_PARAMS = ("fc_feat.weight", "fc_feat.bias", "att_f.weight", "att_f.bias", "att_h.weight", "att_h.bias", "att_v.weight",
                "init_h.weight", "init_h.bias", "init_c.weight", "init_c.bias", "embed.weight", "lstm.weight_ih_l0",
                "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "fc2.weight", "fc2.bias")


def satt_prepare(m, feats):
    """Memory rows, attention keys and initial state: (enc [B*S,E], pf [B*S,A], mean [B,E], h0, c0)."""
    B, S, D = feats.shape
    enc = ops.linear(feats.reshape(B * S, D).contiguous(), m.fc_feat.weight, m.fc_feat.bias)
    pf = ops.linear(enc, m.att_f.weight, m.att_f.bias)
    mean = ops.axpy_rows(ops.group_sum(enc, S), 1.0 / S)
    h0 = ops.linear(mean, m.init_h.weight, m.init_h.bias)
    c0 = ops.linear(mean, m.init_c.weight, m.init_c.bias)
    return enc, pf, mean, h0, c0


def satt_forward(m, feats, ids):
    """Teacher-forced pass keeping what the backward needs.  feats [B,S,D], ids [B,T] int64.
    Returns (h [T*B, Hd] time-major outputs, tape)."""
    B, S, D = feats.shape
    T = ids.shape[1]
    t = Tape()
    t.feats2d = feats.reshape(B * S, D).contiguous()
    t.enc, t.pf, t.mean, h0, c0 = satt_prepare(m, feats)
    E, A, Hd = t.enc.shape[1], t.pf.shape[1], h0.shape[1]
    dev = feats.device
    t.B, t.S, t.T, t.E, t.A, t.Hd = B, S, T, E, A, Hd
    t.ids_tm = ids.t().contiguous()
    t.x = torch.empty((T * B, 2 * E), device=dev, dtype=torch.float32)            # [emb | ctx] rows, time-major
    t.h = torch.empty(((T + 1) * B, Hd), device=dev, dtype=torch.float32)         # block 0 = h0, block s+1 = after step s
    t.c = torch.empty(((T + 1) * B, Hd), device=dev, dtype=torch.float32)
    t.g = torch.empty((T * B, 4 * Hd), device=dev, dtype=torch.float32)
    t.u = torch.empty((T * B, A), device=dev, dtype=torch.float32)
    t.alpha = torch.empty((T * B, S), device=dev, dtype=torch.float32)
    t.h[:B].copy_(h0)
    t.c[:B].copy_(c0)
    if T > 0:
        emb = ops.embed_pe(t.ids_tm, m.embed.weight, None)                         # [T*B, E]
        t.x[:, :E].copy_(emb)
    w = (m.lstm.weight_ih_l0, m.lstm.weight_hh_l0, m.lstm.bias_ih_l0, m.lstm.bias_hh_l0)
    v = m.att_v.weight.reshape(-1)
    enc3, pf3 = t.enc.view(B, S, E), t.pf.view(B, S, A)
    for s in range(T):
        a, b, c = s * B, (s + 1) * B, (s + 2) * B
        ops.linear(t.h[a:b], m.att_h.weight, m.att_h.bias, out=t.u[a:b])
        ops.additive_attention(t.u[a:b], pf3, v, enc3, t.x[a:b, E:], t.alpha[a:b])
        ops.lstm_step_into(t.x[a:b], t.h[a:b], t.c[a:b], *w, t.h[b:c], t.c[b:c], t.g[a:b])
    return t.h[B:], t


def satt_backward(m, t, dh_out, grads):
    """Backward of ``satt_forward``: dh_out [T*B, Hd] time-major.  Fills every gradient except fc2's."""
    B, S, T, E, A, Hd = t.B, t.S, t.T, t.E, t.A, t.Hd
    dev = dh_out.device
    w_ih, w_hh, b_ih, b_hh = m.lstm.weight_ih_l0, m.lstm.weight_hh_l0, m.lstm.bias_ih_l0, m.lstm.bias_hh_l0
    v = m.att_v.weight.reshape(-1)
    enc3, pf3 = t.enc.view(B, S, E), t.pf.view(B, S, A)
    dg = torch.empty_like(t.g)
    dx = torch.empty_like(t.x)
    du = torch.empty_like(t.u)
    dpf = torch.zeros_like(t.pf)
    denc = torch.zeros_like(t.enc)
    dvb = torch.zeros((B, A), device=dev, dtype=torch.float32)
    dh_rec, dc = None, None
    for s in range(T - 1, -1, -1):
        a, b, c = s * B, (s + 1) * B, (s + 2) * B
        dc = ops.lstm_step_bwd(t.g[a:b], t.c[a:b], t.c[b:c], dh_out[a:b], dh_rec, dc, dg[a:b])
        ops.gemm_raw(dg[a:b], w_ih, dx[a:b], B, 2 * E, 4 * Hd, 4 * Hd, w_ih.stride(0), _lib.GEMM_TQ)     # d[emb | ctx]
        ops.additive_attention_bwd(t.u[a:b], pf3, v, enc3, t.alpha[a:b], dx[a:b, E:], du[a:b], dpf, denc, dvb)
        dh_rec = ops.linear_dgrad(dg[a:b], w_hh)
        ops.linear_dgrad(du[a:b], m.att_h.weight, accum=dh_rec)
    if dh_rec is None:                                             # no steps: the state received no gradient
        dh_rec = torch.zeros((B, Hd), device=dev, dtype=torch.float32)
        dc = torch.zeros_like(dh_rec)
    # LSTM and attention-query weights: single contractions over all steps of the tape
    ops.linear_wgrad(dg, t.x, out=_new(grads, "lstm.weight_ih_l0", w_ih))
    ops.linear_wgrad(dg, t.h[:T * B], out=_new(grads, "lstm.weight_hh_l0", w_hh))
    ops.colsum(dg, _new(grads, "lstm.bias_ih_l0", b_ih))
    ops.colsum(dg, _new(grads, "lstm.bias_hh_l0", b_hh))
    ops.linear_wgrad(du, t.h[:T * B], out=_new(grads, "att_h.weight", m.att_h.weight))
    ops.colsum(du, _new(grads, "att_h.bias", m.att_h.bias))
    ops.colsum(dvb, _new(grads, "att_v.weight", m.att_v.weight).view(-1))
    demb = _new(grads, "embed.weight", m.embed.weight)
    if T > 0:
        ops.embed_bwd(t.ids_tm.view(-1), dx[:, :E].contiguous(), demb)
    # initial state: h0 = init_h(mean), c0 = init_c(mean), mean = sum_s enc / S
    dmean = _lin_bwd(grads, "init_h", m.init_h, dh_rec, t.mean)
    _lin_bwd(grads, "init_c", m.init_c, dc, t.mean, accum=dmean)
    ops.axpy_rows(dmean, 1.0 / S, x_div=S, out=denc, accumulate=True)
    # attention keys pf = att_f(enc), memory rows enc = fc_feat(feats)
    _lin_bwd(grads, "att_f", m.att_f, dpf, t.enc, accum=denc)
    _lin_bwd(grads, "fc_feat", m.fc_feat, denc, t.feats2d, need_dx=False)


class _SattLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, feats, captions, *params):
        h, tape = satt_forward(m, feats.detach(), captions)
        B, T = captions.shape
        logits_tm = ops.linear(h, m.fc2.weight, m.fc2.bias)
        ctx.m, ctx.tape, ctx.h = m, tape, h
        return logits_tm.view(T, B, -1).transpose(0, 1)

    @staticmethod
    def backward(ctx, dlogits):
        m, tape = ctx.m, ctx.tape
        grads = {}
        V = dlogits.shape[-1]
        d_tm = dlogits.transpose(0, 1).contiguous().view(-1, V)
        dh = _lin_bwd(grads, "fc2", m.fc2, d_tm, ctx.h)
        satt_backward(m, tape, dh, grads)
        return (None, None, None) + tuple(grads.get(n) for n in _SATT_PARAMS)


def satt_logits(m, feats, captions):
    """Autograd-connected teacher-forced logits [B,T,V] of the soft-attention LSTM decoder."""
    if not feats.is_cuda or not captions.is_cuda:
        raise RuntimeError("capdec: inputs must live on the GPU (HIP); there is no CPU path")
    if feats.requires_grad:
        raise NotImplementedError("capdec: gradients w.r.t. the backbone features are not part of the hot path")
    named = dict(m.named_parameters())
    return _SattLogitsFn.apply(m, feats, captions, *[named[n] for n in _SATT_PARAMS])


def satt_train_step(m, feats, captions, grads, pad_idx=0):
    """Forward, CrossEntropyLoss(ignore_index=<PAD>) and backward with the fused loss path (no [B,T,V] autograd
    tensor): captions [B,T+1].  Returns the loss (1-element device tensor); ``grads`` receives every gradient."""
    ids = captions[:, :-1].contiguous()
    tgt = captions[:, 1:].contiguous()
    h, tape = satt_forward(m, feats, ids)
    loss, saved = vocab_loss(m, h, tgt.t().contiguous(), pad_idx)                 # rows of h are time-major
    dh = vocab_loss_bwd(m, h, saved, grads)
    satt_backward(m, tape, dh, grads)
    return loss
