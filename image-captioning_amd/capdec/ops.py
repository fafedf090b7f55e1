"""Tensor-level wrappers over the C ABI (forward kernels).  Every tensor must live on a HIP device.

PyTorch is used for device memory and the stream only; all arithmetic happens in libcapdec.so.
"""
import os
from ctypes import byref

import torch

from . import _lib
from ._lib import GemmGroup


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as an integer handle (every wrapper passes it on; the direct
    C-level query costs a fraction of ``torch.cuda.current_stream()``, which is called ~200 times per training step)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype=torch.float32, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"capdec: {name} must be a tensor on the GPU (HIP); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"capdec: {name} must be {dtype}, got {t.dtype}")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _row_stride(t, name):
    """2-D view with unit column stride: returns (tensor, ld)."""
    if t.dim() != 2 or t.stride(1) != 1 and t.shape[1] != 1:
        raise RuntimeError(f"capdec: {name} must be 2-D with contiguous rows, got shape {tuple(t.shape)} "
                           f"strides {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def gemm_groups(groups, K, relu=False, extra_flags=0, rows_dev=None, split_k=False):
    """groups: list of dicts with P,Q,C (2-D, row-contiguous) and optional bias_j,bias_i,scale_j,shift_j,R.
    rows_dev: int32 device scalar; only the first rows_dev[0] output rows are computed.
    split_k (one group): long contraction split over workgroups with the partial sums added in a fixed order."""
    lib = _lib.load()
    n = len(groups)
    arr = (GemmGroup * n)()
    for g, d in zip(arr, groups):
        P, Q, C = _chk(d["P"], name="P"), _chk(d["Q"], name="Q"), _chk(d["C"], name="C")
        if P.shape[1] != K or Q.shape[1] != K:
            raise RuntimeError(f"capdec gemm: K mismatch {tuple(P.shape)} {tuple(Q.shape)} K={K}")
        if C.shape[0] != P.shape[0] or C.shape[1] != Q.shape[0]:
            raise RuntimeError(f"capdec gemm: C shape {tuple(C.shape)} != ({P.shape[0]}, {Q.shape[0]})")
        g.P, g.Q, g.C = P.data_ptr(), Q.data_ptr(), C.data_ptr()
        g.ldp, g.ldq = _row_stride(P, "P"), _row_stride(Q, "Q")
        g.ldci, g.ldcj = _row_stride(C, "C"), 1
        g.I, g.J = P.shape[0], Q.shape[0]
        for key, dim in (("bias_i", 0), ("bias_j", 1), ("scale_j", 1), ("shift_j", 1)):
            t = d.get(key)
            if t is not None:
                _chk(t, name=key)
                if t.numel() != C.shape[dim] or not t.is_contiguous():
                    raise RuntimeError(f"capdec gemm: {key} has {t.numel()} elements, expected {C.shape[dim]}")
            setattr(g, key, _ptr(t))
        R = d.get("R")
        if R is not None:
            _chk(R, name="R")
            if R.shape != C.shape:
                raise RuntimeError("capdec gemm: residual shape mismatch")
            g.R, g.ldri, g.ldrj = R.data_ptr(), _row_stride(R, "R"), 1
    flags = (_lib.GEMM_RELU if relu else 0) | extra_flags
    if split_k:
        if n != 1 or rows_dev is not None:
            raise RuntimeError("capdec gemm: split_k takes one group and no row limit")
        elems = lib.capdec_gemm_splitk_workspace_elems(arr[0].I, arr[0].J, K)
        ws = torch.empty(max(int(elems), 1), device=groups[0]["C"].device, dtype=torch.float32)
        ex = _lib.GemmExtra()
        ex.splitk_ws, ex.splitk_ws_elems = ws.data_ptr(), ws.numel()
        _lib.check(lib.capdec_gemm_ex(arr, n, K, flags | _lib.GEMM_SPLITK, byref(ex), _stream()), "capdec_gemm_ex")
        return
    if rows_dev is None:
        _lib.check(lib.capdec_gemm(arr, n, K, flags, _stream()), "capdec_gemm")
        return
    ex = _lib.GemmExtra()
    ex.rows_dev = _chk(rows_dev, torch.int32, "rows_dev").data_ptr()
    _lib.check(lib.capdec_gemm_ex(arr, n, K, flags, byref(ex), _stream()), "capdec_gemm_ex")


def linear(x, W, b=None, relu=False, residual=None, scale=None, shift=None, out=None, split_k=False, rows_dev=None):
    """y = x W^T + b with fused epilogue; x [M,K] (row stride free), W [N,K].
    split_k: split the contraction over workgroups (long K, few output tiles: the 86436-wide YOLO head); the partial
    sums go to a scratch tensor and are added in a fixed order, so the result is the same from run to run."""
    _chk(x, name="x")
    _chk(W, name="W")
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    gemm_groups([dict(P=x, Q=W, C=out, bias_j=b, R=residual, scale_j=scale, shift_j=shift)], K, relu,
                rows_dev=rows_dev, split_k=split_k)
    return out


def embed_pe(ids, emb, pe, pos0=0, out=None):
    """ids [B,T] int64/int32 -> [B*T, E] = emb[ids] + pe[pos0 : pos0+T]  (pe may be None)."""
    lib = _lib.load()
    if ids.dtype not in (torch.int64, torch.int32):
        raise RuntimeError("capdec embed_pe: ids must be int64 or int32")
    _chk(ids, ids.dtype, "ids")
    _chk(emb, name="emb")
    B, T = ids.shape
    V, E = emb.shape
    if ids.stride(1) != 1 and T > 1:
        ids = ids.contiguous()
    x = torch.empty((B * T, E), device=emb.device, dtype=torch.float32) if out is None else _chk(out, name="out")
    if x.shape != (B * T, E) or not x.is_contiguous():
        raise RuntimeError(f"capdec embed_pe: out must be a contiguous [{B * T},{E}] tensor")
    if pe is not None:
        _chk(pe, name="pe")
        if pe.dim() != 2 or pe.shape[1] != E or pe.shape[0] < pos0 + T or not pe.is_contiguous():
            raise RuntimeError(f"capdec embed_pe: pe {tuple(pe.shape)} too short for {pos0 + T} positions")
    _lib.check(lib.capdec_embed_pe(ids.data_ptr(), 1 if ids.dtype == torch.int32 else 0, ids.stride(0) if B > 1 else T,
                                   emb.data_ptr(), _ptr(pe), pos0, B, T, E, V, x.data_ptr(), _stream()),
               "capdec_embed_pe")
    return x


def add_layernorm(x, gamma, beta, add=None, add_div=1, eps=1e-5, out=None, want_stats=False):
    lib = _lib.load()
    _chk(x, name="x")
    M, E = x.shape
    if not x.is_contiguous():
        raise RuntimeError("capdec add_layernorm: x must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    mean = rstd = None
    if want_stats:
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    if add is not None:
        _chk(add, name="add")
        if not add.is_contiguous() or add.shape[-1] != E or add.numel() // E * add_div < M:
            raise RuntimeError("capdec add_layernorm: add shape mismatch")
    _lib.check(lib.capdec_add_layernorm(x.data_ptr(), _ptr(add), add_div, _chk(gamma).data_ptr(),
                                        _chk(beta).data_ptr(), eps, M, E, out.data_ptr(), _ptr(mean), _ptr(rstd),
                                        _stream()), "capdec_add_layernorm")
    return (out, mean, rstd) if want_stats else out


def attention(q, k, v, B, H, Tq, Tk, causal=False, ids32=None, out=None, want_lse=False):
    """q [B*Tq, E], k/v [B*Tk, E] row-contiguous 2-D views (row stride free) -> out [B*Tq, E]."""
    lib = _lib.load()
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, name=n)
    E = q.shape[1]
    dk = E // H
    if out is None:
        out = torch.empty((B * Tq, E), device=q.device, dtype=torch.float32)
    lse = torch.empty((B, H, Tq), device=q.device, dtype=torch.float32) if want_lse else None
    ldid = 0
    if ids32 is not None:
        _chk(ids32, torch.int32, "ids32")
        ldid = ids32.stride(0) if B > 1 else max(ids32.shape[1], Tq)
    ldk, ldv = _row_stride(k, "k"), _row_stride(v, "v")
    _lib.check(lib.capdec_attention(q.data_ptr(), _row_stride(q, "q"), k.data_ptr(), ldk, ldk * Tk, v.data_ptr(), ldv,
                                    ldv * Tk, out.data_ptr(), _row_stride(out, "out"), B, H, Tq, Tk, dk,
                                    1 if causal else 0, 0, _ptr(ids32), ldid, _ptr(lse), _stream()),
               "capdec_attention")
    return (out, lse) if want_lse else out


def bn_fold(gamma, beta, running_mean, running_var, eps=1e-5):
    lib = _lib.load()
    C = gamma.numel()
    scale = torch.empty(C, device=gamma.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    _lib.check(lib.capdec_bn_fold(_chk(gamma).data_ptr(), _chk(beta).data_ptr(), _chk(running_mean).data_ptr(),
                                  _chk(running_var).data_ptr(), eps, C, scale.data_ptr(), shift.data_ptr(), _stream()),
               "capdec_bn_fold")
    return scale, shift


def head_eval(x, parts, gamma, beta, running_mean, running_var, eps, split_k):
    """Eval-mode feature head: ``capdec_bn_fold`` of the BatchNorm's running statistics, then for each ``(W [E,hi-lo], b [E], lo, hi)`` of ``parts`` the contraction of ``x[:, lo:hi]`` with W
    whose epilogue applies that part's scale and shift, written into its E columns of the [B, len(parts)*E] result.
    The operands go to the library as addresses and strides (no tensor views on the way: the head is on the host's
    critical path of every synchronous decode call).  ``split_k(K)`` says whether a part takes the fixed-order split
    of a long contraction (``linear(..., split_k=True)``)."""
    lib = _lib.load()
    _chk(x, name="x")
    B, F = x.shape
    ldx = _row_stride(x, "x")
    E = parts[0][0].shape[0]
    C = len(parts) * E
    for t, nm in ((gamma, "batchnorm.weight"), (beta, "batchnorm.bias"), (running_mean, "batchnorm.running_mean"),
                  (running_var, "batchnorm.running_var")):
        if _chk(t, name=nm).numel() != C or not t.is_contiguous():
            raise RuntimeError(f"capdec head: {nm} must hold {C} contiguous values")
    out = torch.empty((B, C), device=x.device, dtype=torch.float32)
    fold = torch.empty((2 * C,), device=x.device, dtype=torch.float32)         # scale, then shift
    fp, xp, op, s = fold.data_ptr(), x.data_ptr(), out.data_ptr(), _stream()
    _lib.check(lib.capdec_bn_fold(gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                                  eps, C, fp, fp + 4 * C, s), "capdec_bn_fold")
    arr = (GemmGroup * 1)()
    g = arr[0]
    for n, (W, b, lo, hi) in enumerate(parts):
        K = hi - lo
        _chk(W, name="fc.weight")
        if b is not None and (_chk(b, name="fc.bias").numel() != E or not b.is_contiguous()):
            raise RuntimeError(f"capdec head: part {n}: bias {tuple(b.shape)} for {E} outputs")
        if not 0 <= lo < hi <= F or tuple(W.shape) != (E, K) or not W.is_contiguous():
            raise RuntimeError(f"capdec head: part {n} does not fit: x {tuple(x.shape)}, columns {lo}:{hi}, "
                               f"weight {tuple(W.shape)}")
        if split_k(K):
            linear(x[:, lo:hi], W, b, scale=fold[n * E:(n + 1) * E], shift=fold[C + n * E:C + (n + 1) * E],
                   out=out[:, n * E:(n + 1) * E], split_k=True)
            continue
        g.P, g.Q, g.C = xp + 4 * lo, W.data_ptr(), op + 4 * n * E
        g.ldp, g.ldq, g.ldci, g.ldcj, g.I, g.J = ldx, K, C, 1, B, E
        g.bias_j, g.scale_j, g.shift_j = _ptr(b), fp + 4 * n * E, fp + 4 * (C + n * E)
        _lib.check(lib.capdec_gemm(arr, 1, K, 0, s), "capdec_gemm")
    return out


def bn_train_forward(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
    lib = _lib.load()
    _chk(x, name="x")
    if not x.is_contiguous():
        raise RuntimeError("capdec bn_train_forward: x must be contiguous")
    B, C = x.shape
    if B < 2:
        raise ValueError("Expected more than 1 value per channel when training")
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    _lib.check(lib.capdec_bn_train_forward(x.data_ptr(), _chk(gamma).data_ptr(), _chk(beta).data_ptr(), eps, momentum,
                                           _ptr(running_mean), _ptr(running_var), B, C, y.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), _stream()), "capdec_bn_train_forward")
    return y, mean, rstd


def lstm_step(x, h0, c0, w_ih, w_hh, b_ih, b_hh, want_gates=False, h_out=None):
    """One fused LSTM step.  h_out (optional) is a [B,Hd] view with unit column stride to write h' into."""
    lib = _lib.load()
    _chk(x, name="x")
    B, E = x.shape
    Hd = w_hh.shape[1]
    h1 = torch.empty((B, Hd), device=x.device, dtype=torch.float32) if h_out is None else _chk(h_out, name="h_out")
    c1 = torch.empty((B, Hd), device=x.device, dtype=torch.float32)
    gates = torch.empty((B, 4 * Hd), device=x.device, dtype=torch.float32) if want_gates else None
    if c0 is not None and not c0.is_contiguous():
        raise RuntimeError("capdec lstm_step: c0 must be contiguous")
    _lib.check(lib.capdec_lstm_step(x.data_ptr(), _row_stride(x, "x"), _ptr(h0),
                                    0 if h0 is None else _row_stride(h0, "h0"), _ptr(c0), _chk(w_ih).data_ptr(),
                                    _chk(w_hh).data_ptr(), _chk(b_ih).data_ptr(), _chk(b_hh).data_ptr(), B, E, Hd,
                                    h1.data_ptr(), _row_stride(h1, "h1"), c1.data_ptr(), _ptr(gates), _stream()),
               "capdec_lstm_step")
    return (h1, c1, gates) if want_gates else (h1, c1)


def lstm_step_into(x, h0, c0, w_ih, w_hh, b_ih, b_hh, h1, c1, gates):
    """One fused LSTM step writing h', c' and the pre-activation gates into caller-provided contiguous blocks."""
    lib = _lib.load()
    B, E = x.shape
    Hd = w_hh.shape[1]
    for name, t in (("x", x), ("h1", h1), ("c1", c1), ("gates", gates), ("h0", h0), ("c0", c0)):
        if t is not None and not _chk(t, name=name).is_contiguous():
            raise RuntimeError(f"capdec lstm_step_into: {name} must be contiguous")
    if h1.shape != (B, Hd) or c1.shape != (B, Hd) or gates.shape != (B, 4 * Hd):
        raise RuntimeError("capdec lstm_step_into: output shapes do not match [B,Hd] / [B,4Hd]")
    _lib.check(lib.capdec_lstm_step(x.data_ptr(), E, _ptr(h0), Hd, _ptr(c0), _chk(w_ih).data_ptr(),
                                    _chk(w_hh).data_ptr(), _chk(b_ih).data_ptr(), _chk(b_hh).data_ptr(), B, E, Hd,
                                    h1.data_ptr(), Hd, c1.data_ptr(), gates.data_ptr(), _stream()), "capdec_lstm_step")


def lstm_step_bwd(gates, c0, c1, dh_out, dh_rec, dc_in, dgates):
    """Backward of one LSTM step (see capdec_lstm_step_bwd).  Writes ``dgates`` [B,4Hd]; returns dc0 [B,Hd]
    (written over ``dc_in`` when that is given)."""
    lib = _lib.load()
    B, Hd = c1.shape
    for name, t in (("gates", gates), ("c1", c1), ("dgates", dgates), ("c0", c0), ("dh_rec", dh_rec), ("dc_in", dc_in)):
        if t is not None and not _chk(t, name=name).is_contiguous():
            raise RuntimeError(f"capdec lstm_step_bwd: {name} must be contiguous")
    dc0 = dc_in if dc_in is not None else torch.empty_like(c1)
    _lib.check(lib.capdec_lstm_step_bwd(gates.data_ptr(), _ptr(c0), c1.data_ptr(), _ptr(dh_out),
                                        0 if dh_out is None else _row_stride(dh_out, "dh_out"), _ptr(dh_rec),
                                        _ptr(dc_in), B, Hd, dgates.data_ptr(), dc0.data_ptr(), _stream()),
               "capdec_lstm_step_bwd")
    return dc0


def vocab_argmax(x, W, b):
    """argmax_v (x W^T + b) per row without writing the logits; returns (idx int32 [M], val fp32 [M])."""
    lib = _lib.load()
    _chk(x, name="x")
    M, K = x.shape
    V = W.shape[0]
    nt = lib.capdec_vocab_tiles(V)
    n = lib.capdec_vocab_partial_elems(M, V)
    pv = torch.empty((n,), device=x.device, dtype=torch.float32)
    pi = torch.empty((n,), device=x.device, dtype=torch.int32)
    _lib.check(lib.capdec_vocab_argmax_partial(x.data_ptr(), _row_stride(x, "x"), _chk(W).data_ptr(),
                                               _chk(b).data_ptr(), M, V, K, pv.data_ptr(), pi.data_ptr(), _stream()),
               "capdec_vocab_argmax_partial")
    idx = torch.empty(M, device=x.device, dtype=torch.int32)
    val = torch.empty(M, device=x.device, dtype=torch.float32)
    _lib.check(lib.capdec_argmax_finalize(pv.data_ptr(), pi.data_ptr(), M, nt, idx.data_ptr(), val.data_ptr(),
                                          _stream()), "capdec_argmax_finalize")
    return idx, val


def vocab_nll(x, W, b, targets, ignore_index=0):
    """Fused projection + log-sum-exp + target pick.  Returns (lse [M], nll [M], sums [2] = (sum nll, n valid))."""
    lib = _lib.load()
    _chk(x, name="x")
    _chk(targets, torch.int64, "targets")
    M, K = x.shape
    V = W.shape[0]
    nt = lib.capdec_vocab_tiles(V)
    n = lib.capdec_vocab_partial_elems(M, V)
    pm = torch.empty((n,), device=x.device, dtype=torch.float32)
    ps = torch.empty((n,), device=x.device, dtype=torch.float32)
    _lib.check(lib.capdec_vocab_lse_partial(x.data_ptr(), _row_stride(x, "x"), _chk(W).data_ptr(), _chk(b).data_ptr(),
                                            M, V, K, pm.data_ptr(), ps.data_ptr(), _stream()),
               "capdec_vocab_lse_partial")
    lse = torch.empty(M, device=x.device, dtype=torch.float32)
    nll = torch.empty(M, device=x.device, dtype=torch.float32)
    sums = torch.zeros(2, device=x.device, dtype=torch.float32)
    _lib.check(lib.capdec_nll_finalize(pm.data_ptr(), ps.data_ptr(), M, V, x.data_ptr(), _row_stride(x, "x"),
                                       W.data_ptr(), b.data_ptr(), K, targets.contiguous().data_ptr(), ignore_index,
                                       lse.data_ptr(), nll.data_ptr(), sums.data_ptr(), _stream()),
               "capdec_nll_finalize")
    return lse, nll, sums


# ---------------------------------------------------------------------------------------------
# training-side ops
# ---------------------------------------------------------------------------------------------
def gemm_raw(P, Q, C, I, J, K, ldp, ldq, flags=0, bias_j=None, R=None, mask=None, lse=None, tgt=None, gscale=None,
             ignore_index=0, rows_dev=None, k_dev=None):
    """Single-group contraction with explicit layout: see capdec_gemm_ex in capdec.h.
    rows_dev / k_dev: int32 device scalars limiting the output rows / the contraction length."""
    lib = _lib.load()
    g = (GemmGroup * 1)()
    g[0].P, g[0].Q, g[0].C = _chk(P).data_ptr(), _chk(Q).data_ptr(), _chk(C).data_ptr()
    g[0].ldp, g[0].ldq, g[0].ldci, g[0].ldcj, g[0].I, g[0].J = ldp, ldq, _row_stride(C, "C"), 1, I, J
    if C.shape[0] != I or C.shape[1] != J:
        raise RuntimeError(f"capdec gemm_raw: C {tuple(C.shape)} != ({I},{J})")
    g[0].bias_j = _ptr(bias_j)
    if R is not None:
        g[0].R, g[0].ldri, g[0].ldrj = _chk(R).data_ptr(), _row_stride(R, "R"), 1
    ex = _lib.GemmExtra()
    if mask is not None:
        ex.mask, ex.ldm = _chk(mask).data_ptr(), _row_stride(mask, "mask")
    if lse is not None:
        ex.lse, ex.tgt, ex.gscale, ex.ignore_index = (_chk(lse).data_ptr(), _chk(tgt, torch.int64).data_ptr(),
                                                       _chk(gscale).data_ptr(), ignore_index)
    if rows_dev is not None:
        ex.rows_dev = _chk(rows_dev, torch.int32, "rows_dev").data_ptr()
    if k_dev is not None:
        ex.k_dev = _chk(k_dev, torch.int32, "k_dev").data_ptr()
    _lib.check(lib.capdec_gemm_ex(g, 1, K, flags, byref(ex), _stream()), "capdec_gemm_ex")
    return C


def gemm_layout_groups(groups, layout, rows_dev=None, k_dev=None):
    """Up to four contractions of one layout in one launch; each group is a dict of 2-D row-contiguous views P, Q, C
    (+ optional bias_j):
      "nt": C[I,J] = P[I,K] Q[J,K]^T + bias_j       "nn": C[I,J] = P[I,K] Q[K,J]
      "tn": C[I,J] += P[K,I]^T Q[K,J]   (split over workgroups, summed with float atomics: C is accumulated into)
    rows_dev limits the output rows (nt, nn), k_dev the contraction length (tn): int32 device scalars."""
    lib = _lib.load()
    n = len(groups)
    arr = (GemmGroup * n)()
    K = None
    for g, d in zip(arr, groups):
        P, Q, C = _chk(d["P"], name="P"), _chk(d["Q"], name="Q"), _chk(d["C"], name="C")
        if layout == "nt":
            (I, k), (J, k2) = P.shape, Q.shape
        elif layout == "nn":
            (I, k), (k2, J) = P.shape, Q.shape
        elif layout == "tn":
            (k, I), (k2, J) = P.shape, Q.shape
        else:
            raise RuntimeError(f"capdec gemm: unknown layout {layout!r}")
        if k != k2 or tuple(C.shape) != (I, J) or (K is not None and k != K):
            raise RuntimeError(f"capdec gemm ({layout}): shapes {tuple(P.shape)} {tuple(Q.shape)} -> {tuple(C.shape)}")
        K = k
        g.P, g.Q, g.C = P.data_ptr(), Q.data_ptr(), C.data_ptr()
        g.ldp, g.ldq, g.ldci, g.ldcj, g.I, g.J = _row_stride(P, "P"), _row_stride(Q, "Q"), _row_stride(C, "C"), 1, I, J
        b = d.get("bias_j")
        if b is not None:
            if layout == "tn" or _chk(b, name="bias_j").numel() != J or not b.is_contiguous():
                raise RuntimeError("capdec gemm: bias_j must hold J contiguous values (nt / nn only)")
            g.bias_j = b.data_ptr()
    flags = {"nt": 0, "nn": _lib.GEMM_TQ, "tn": _lib.GEMM_TP | _lib.GEMM_TQ | _lib.GEMM_ATOMIC}[layout]
    ex = _lib.GemmExtra()
    if rows_dev is not None:
        ex.rows_dev = _chk(rows_dev, torch.int32, "rows_dev").data_ptr()
    if k_dev is not None:
        ex.k_dev = _chk(k_dev, torch.int32, "k_dev").data_ptr()
    _lib.check(lib.capdec_gemm_ex(arr, n, K, flags, byref(ex), _stream()), "capdec_gemm_ex")


def bgemm(form, P, Q, C, B, seg, mult, seg_off=None, scale=1.0, P2=None, Q2=None, accumulate=False):
    """Per-sample contractions over a ragged batch (capdec_bgemm in capdec.h).  form "nt" / "nn": P, C hold the
    samples' rows ([rows, K] / [rows, J]), Q is [B*J, K] / [B*K, J]; form "tn": P, Q hold the rows ([rows, >= I],
    [rows, J]) and C is [B*I, J].  The per-sample sizes come from the dense operand's shape.  Sample b owns the rows
    [seg_off[b]*mult, seg_off[b+1]*mult) (seg_off None: seg*mult rows each); with seg_off the caller guarantees that
    seg_off[B]*mult does not exceed the rows of P (the offsets live on the device and are not read here)."""
    lib = _lib.load()
    d = _lib.BgemmDesc()
    for t, nm in ((P, "P"), (Q, "Q"), (C, "C")):
        _chk(t, name=nm)
    d.P, d.Q, d.C = P.data_ptr(), Q.data_ptr(), C.data_ptr()
    d.ldp, d.ldq, d.ldc = _row_stride(P, "P"), _row_stride(Q, "Q"), _row_stride(C, "C")
    d.B, d.seg, d.mult, d.scale, d.accumulate = B, seg, mult, scale, 1 if accumulate else 0
    d.seg_off = None if seg_off is None else _chk(seg_off, torch.int32, "seg_off").data_ptr()
    if B < 1 or (seg_off is not None and seg_off.numel() != B + 1):
        raise RuntimeError("capdec bgemm: B must be positive and seg_off hold B + 1 offsets")
    rows = P.shape[0]                     # rows of the ragged operands (every sample's rows lie below this)
    if seg_off is None and B * seg * mult > rows:
        raise RuntimeError(f"capdec bgemm: {B} samples of {seg * mult} rows do not fit the {rows} rows of P")
    dense = C if form == "tn" else Q
    if dense.shape[0] % B:
        raise RuntimeError(f"capdec bgemm: the dense operand's {dense.shape[0]} rows are not a multiple of B={B}")
    if form == "nt":
        d.J, d.K, d.I = Q.shape[0] // B, Q.shape[1], 1
        if P.shape[1] != d.K or C.shape[0] != rows or C.shape[1] < d.J:
            raise RuntimeError(f"capdec bgemm nt: P {tuple(P.shape)} Q {tuple(Q.shape)} C {tuple(C.shape)}")
        d.dense_stride = d.J * d.ldq
        code = 0
    elif form == "nn":
        d.K, d.J, d.I = Q.shape[0] // B, Q.shape[1], 1
        if P.shape[1] < (d.K + 3) // 4 * 4 or C.shape[0] != rows or C.shape[1] != d.J:
            raise RuntimeError(f"capdec bgemm nn: P {tuple(P.shape)} Q {tuple(Q.shape)} C {tuple(C.shape)}")
        d.dense_stride = d.K * d.ldq
        code = 1
    elif form == "tn":
        d.I, d.J, d.K = C.shape[0] // B, C.shape[1], 1
        if P.shape[1] < (d.I + 3) // 4 * 4 or Q.shape[0] != rows or Q.shape[1] != d.J:
            raise RuntimeError(f"capdec bgemm tn: P {tuple(P.shape)} Q {tuple(Q.shape)} C {tuple(C.shape)}")
        d.dense_stride = d.I * d.ldc
        code = 2
        if P2 is not None:
            if Q2 is None or P2.shape != P.shape or Q2.shape != Q.shape:
                raise RuntimeError("capdec bgemm tn: P2 / Q2 must have the shapes of P / Q")
            if _row_stride(P2, "P2") != d.ldp or _row_stride(Q2, "Q2") != d.ldq:
                raise RuntimeError("capdec bgemm: P2 / Q2 must have the row strides of P / Q")
            d.P2, d.Q2 = _chk(P2, name="P2").data_ptr(), _chk(Q2, name="Q2").data_ptr()
    else:
        raise RuntimeError(f"capdec bgemm: unknown form {form!r}")
    _lib.check(lib.capdec_bgemm(byref(d), code, _stream()), "capdec_bgemm")
    return C


def pair_diff(k2, v2, bq, H):
    """k2, v2 [B*2, E] (a sample's two projected memory rows) -> dK = k0 - k1, dV = v0 - v1 [B, E], sb [B, H] = b_q,h . dK_h"""
    lib = _lib.load()
    B, E = k2.shape[0] // 2, k2.shape[1]
    dk = torch.empty((B, E), device=k2.device, dtype=torch.float32)
    dv = torch.empty_like(dk)
    sb = torch.empty((B, H), device=k2.device, dtype=torch.float32)
    if not (k2.is_contiguous() and v2.is_contiguous()):
        raise RuntimeError("capdec pair_diff: k2, v2 must be contiguous")
    _lib.check(lib.capdec_pair_diff(_chk(k2).data_ptr(), _chk(v2).data_ptr(), _chk(bq).data_ptr(), B, E, H, dk.data_ptr(),
                                    dv.data_ptr(), sb.data_ptr(), _stream()), "capdec_pair_diff")
    return dk, dv, sb


def pair_diff_bwd(ddk, ddv, dsb, dv1, bq, dk, H):
    """Backward of ``pair_diff``: -> dk2, dv2 [B*2, E] and qb [B, E] (its column sum is the gradient of b_q)."""
    lib = _lib.load()
    B, E = ddk.shape
    dk2 = torch.empty((2 * B, E), device=ddk.device, dtype=torch.float32)
    dv2 = torch.empty_like(dk2)
    qb = torch.empty((B, E), device=ddk.device, dtype=torch.float32)
    for t_ in (ddk, ddv, dsb, dv1, dk):
        if not _chk(t_).is_contiguous():
            raise RuntimeError("capdec pair_diff_bwd: operands must be contiguous")
    _lib.check(lib.capdec_pair_diff_bwd(ddk.data_ptr(), ddv.data_ptr(), dsb.data_ptr(), dv1.data_ptr(), _chk(bq).data_ptr(),
                                        dk.data_ptr(), B, E, H, dk2.data_ptr(), dv2.data_ptr(), qb.data_ptr(), _stream()),
               "capdec_pair_diff_bwd")
    return dk2, dv2, qb


def pair_gate_fwd(x1, G, D, c1, sb, H, scale, T, sample=None, rows_dev=None):
    """a2 = c1[b] + sum_h sigmoid((x1 . G[b,h] + sb[b,h]) * scale) D[b,h] per row; returns (a2 [M,E], p [M,H])."""
    lib = _lib.load()
    M, E = x1.shape
    a2 = torch.empty((M, E), device=x1.device, dtype=torch.float32)
    p = torch.zeros((M, H), device=x1.device, dtype=torch.float32)
    for t_ in (x1, G, D, c1, sb):
        if not _chk(t_).is_contiguous():
            raise RuntimeError("capdec pair_gate_fwd: operands must be contiguous")
    _lib.check(lib.capdec_pair_gate_fwd(x1.data_ptr(), G.data_ptr(), D.data_ptr(), c1.data_ptr(), sb.data_ptr(),
                                        None if sample is None else _chk(sample, torch.int32, "sample").data_ptr(), T, M, E, H,
                                        scale, a2.data_ptr(), p.data_ptr(), _ptr(rows_dev), _stream()), "capdec_pair_gate_fwd")
    return a2, p


def pair_gate_bwd(d_a2, x1, p, G, D, B, T, H, scale, d_x1, seg_off=None):
    """Backward of ``pair_gate_fwd``: adds the rows' input gradient onto d_x1 (in place); returns (dG, dD [B,H,E],
    dc1 [B,E], dsb [B,H])."""
    lib = _lib.load()
    E = x1.shape[1]
    dG = torch.empty((B, H, E), device=x1.device, dtype=torch.float32)
    dD = torch.empty_like(dG)
    dc1 = torch.empty((B, E), device=x1.device, dtype=torch.float32)
    dsb = torch.empty((B, H), device=x1.device, dtype=torch.float32)
    for t_ in (d_a2, x1, p, G, D, d_x1):
        if not _chk(t_).is_contiguous():
            raise RuntimeError("capdec pair_gate_bwd: operands must be contiguous")
    _lib.check(lib.capdec_pair_gate_bwd(d_a2.data_ptr(), x1.data_ptr(), p.data_ptr(), G.data_ptr(), D.data_ptr(),
                                        None if seg_off is None else _chk(seg_off, torch.int32, "seg_off").data_ptr(), B, T, E,
                                        H, scale, d_x1.data_ptr(), dG.data_ptr(), dD.data_ptr(), dc1.data_ptr(),
                                        dsb.data_ptr(), _stream()), "capdec_pair_gate_bwd")
    return dG, dD, dc1, dsb


def softmax_rows_(s, cols, live=None, live_idx=0, mult=1):
    """In place: every row of s [rows, ld] becomes softmax(row[:cols]), zeros from there to the row's end.  live: int32
    device array; only the first live[live_idx]*mult rows are touched."""
    lib = _lib.load()
    rows, ld = s.shape[0], _row_stride(s, "s")
    _lib.check(lib.capdec_softmax_rows(_chk(s).data_ptr(), ld, rows, cols, s.shape[1], _ptr(live), live_idx, mult,
                                       _stream()), "capdec_softmax_rows")
    return s


def softmax_rows_bwd_(p, dp, cols, scale=1.0, live=None, live_idx=0, mult=1):
    """In place on dp: dp <- p * (dp - sum_j p_j dp_j) * scale over the first ``cols`` columns, zeros after them."""
    lib = _lib.load()
    rows, ld = p.shape[0], _row_stride(p, "p")
    if dp.shape != p.shape or _row_stride(dp, "dp") != ld:
        raise RuntimeError("capdec softmax_rows_bwd_: p and dp must have the same shape and row stride")
    _lib.check(lib.capdec_softmax_rows_bwd(_chk(p).data_ptr(), _chk(dp).data_ptr(), ld, rows, cols, p.shape[1], scale,
                                           _ptr(live), live_idx, mult, _stream()), "capdec_softmax_rows_bwd")
    return dp


def linear_dgrad(dy, W, mask=None, accum=None, rows_dev=None):
    """dX = dY W  (dY [M,N], W [N,K]); zeroed where mask <= 0; added onto ``accum`` in place when given."""
    M, N = dy.shape
    K = W.shape[1]
    if accum is None and mask is None and N >= 4096 and ((M + 63) // 64) * ((K + 63) // 64) <= 512:
        # a long contraction into a small output (the vocabulary projection's input gradient at a small batch): too few
        # output tiles to fill the chip, so the contraction is split over workgroups and summed with float atomics,
        # like the weight gradients
        out = torch.zeros((M, K), device=dy.device, dtype=torch.float32)
        return gemm_raw(dy, W, out, M, K, N, _row_stride(dy, "dy"), _row_stride(W, "W"), _lib.GEMM_TQ | _lib.GEMM_ATOMIC,
                        rows_dev=rows_dev)
    out = accum if accum is not None else torch.empty((M, K), device=dy.device, dtype=torch.float32)
    return gemm_raw(dy, W, out, M, K, N, _row_stride(dy, "dy"), _row_stride(W, "W"), _lib.GEMM_TQ, R=accum, mask=mask,
                    rows_dev=rows_dev)


def linear_wgrad(dy, x, out=None, rows_dev=None):
    """dW += dY^T X  (dY [M,N], X [M,K]) -> [N,K].  The contraction over the M rows is split over workgroups and
    summed with float atomics, so ``out`` is accumulated into (a fresh zero tensor when omitted).
    rows_dev: int32 device scalar; only the first rows_dev[0] rows are contracted (dY must hold zeros, X anything
    finite, from there up to the next multiple of 32)."""
    M, N = dy.shape
    K = x.shape[1]
    if out is None:
        out = torch.zeros((N, K), device=dy.device, dtype=torch.float32)
    return gemm_raw(dy, x, out, N, K, M, _row_stride(dy, "dy"), _row_stride(x, "x"),
                    _lib.GEMM_TP | _lib.GEMM_TQ | _lib.GEMM_ATOMIC, k_dev=rows_dev)


def colsum(x, out, rows_dev=None):
    """out[j] += sum_i x[i,j]  (over the first rows_dev[0] rows when that int32 device scalar is given)"""
    lib = _lib.load()
    M, N = x.shape
    _lib.check(lib.capdec_colsum_rows(_chk(x).data_ptr(), _row_stride(x, "x"), M, N, _chk(out).data_ptr(),
                                      _ptr(rows_dev), _stream()), "capdec_colsum_rows")
    return out


def compact_targets(targets, ignore_index=0):
    """Rows with a counted target first (stable).  targets [M] int64 ->
    (order int64 [M]: row of slot s, slot int64 [M]: slot of row r or M if ignored, targets_c int64 [M], count int32 [1])."""
    lib = _lib.load()
    _chk(targets, torch.int64, "targets")
    if not targets.is_contiguous() or targets.dim() != 1:
        raise RuntimeError("capdec compact_targets: targets must be a contiguous [M] tensor")
    M = targets.numel()
    order = torch.empty(M, device=targets.device, dtype=torch.int64)
    slot = torch.empty(M, device=targets.device, dtype=torch.int64)
    tgt_c = torch.empty(M, device=targets.device, dtype=torch.int64)
    count = torch.empty(1, device=targets.device, dtype=torch.int32)
    _lib.check(lib.capdec_compact_targets(targets.data_ptr(), M, ignore_index, order.data_ptr(), slot.data_ptr(),
                                          tgt_c.data_ptr(), count.data_ptr(), _stream()), "capdec_compact_targets")
    return order, slot, tgt_c, count


def group_sum(x, T):
    """[G*T, E] -> [G, E]: sums of T consecutive rows."""
    lib = _lib.load()
    M, E = x.shape
    if M % T or not x.is_contiguous():
        raise RuntimeError("capdec group_sum: x must be a contiguous [G*T, E] tensor")
    out = torch.empty((M // T, E), device=x.device, dtype=torch.float32)
    _lib.check(lib.capdec_group_sum(_chk(x).data_ptr(), M // T, T, E, out.data_ptr(), _stream()), "capdec_group_sum")
    return out


def _seed_ptr(seed_dev):
    """Device address of the per-step seed (an int64 tensor whose first element is added to the by-value seed)."""
    return None if seed_dev is None else _chk(seed_dev, torch.int64, "seed_dev").data_ptr()


def dropout(x, p, seed, out=None, seed_dev=None):
    lib = _lib.load()
    if not x.is_contiguous():
        raise RuntimeError("capdec dropout: x must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.capdec_dropout(_chk(x).data_ptr(), x.numel(), p, seed, _seed_ptr(seed_dev), out.data_ptr(), _stream()),
               "capdec_dropout")
    return out


def add_layernorm_train(x, add, gamma, beta, p, seed, eps=1e-5, x_div=1, x_map=None, rows_dev=None, seed_dev=None):
    """s = add + dropout(x); y = LN(s).  Returns (y, s, mean, rstd).  With x_div > 1, x holds one row per
    ``x_div`` consecutive rows of ``add`` (a per-sample row broadcast over its positions); with x_map (int32 [M]) row r
    takes x[x_map[r]].  rows_dev: int32 device scalar, only the first rows_dev[0] rows are processed."""
    lib = _lib.load()
    M, E = add.shape
    if x_map is not None:
        if x_map.numel() != M or x.shape[1] != E:
            raise RuntimeError("capdec add_layernorm_train: x_map must name one row of x per row of add")
    elif x.shape != ((M + x_div - 1) // x_div, E) or M % x_div:
        raise RuntimeError(f"capdec add_layernorm_train: x {tuple(x.shape)} does not match add {tuple(add.shape)} / {x_div}")
    y = torch.empty((M, E), device=x.device, dtype=torch.float32)
    s = torch.empty_like(y)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    if not x.is_contiguous() or not add.is_contiguous():
        raise RuntimeError("capdec add_layernorm_train: operands must be contiguous")
    _lib.check(lib.capdec_add_layernorm_train_rows(_chk(x).data_ptr(), x_div,
                                                   None if x_map is None else _chk(x_map, torch.int32, "x_map").data_ptr(),
                                                   _chk(add).data_ptr(), _chk(gamma).data_ptr(), _chk(beta).data_ptr(),
                                                   eps, M, E, p, seed, _seed_ptr(seed_dev), y.data_ptr(), s.data_ptr(),
                                                   mean.data_ptr(),
                                                   rstd.data_ptr(), _ptr(rows_dev), _stream()),
               "capdec_add_layernorm_train_rows")
    return y, s, mean, rstd


_LN_BWD_WS = {}       # (device, floats) -> slab for the two-stage column sums of layernorm_bwd
# "auto": batches of 4096 rows and more (below that the second launch costs more than the few float atomics it
# replaces: 1.745 against 1.72 ms per 64-sample training step); "1" / "0": always / never
LN_BWD_STAGED = os.environ.get("CAPDEC_LN_BWD_STAGED", "auto")
LN_BWD_STAGED_MIN_ROWS = 4096


def layernorm_bwd(dy, s, mean, rstd, gamma, p, seed, dgamma, dbeta, dxsum=None, rows_dev=None, seed_dev=None, staged=None):
    """Returns (d_add, d_x); dgamma/dbeta are accumulated, and so are the column sums of d_x into ``dxsum`` [E]
    when given (the bias gradient of the linear layer whose output was normalised).  staged: the column sums go
    through a [workgroups][3E] slab and a second launch instead of float atomics (fixed summation order)."""
    lib = _lib.load()
    M, E = dy.shape
    if not dy.is_contiguous():
        raise RuntimeError("capdec layernorm_bwd: dy must be contiguous")
    d_add = torch.empty((M, E), device=dy.device, dtype=torch.float32)
    # d_x always gets a tensor of its own, also when no mask is drawn (p == 0: the same values twice).  The caller
    # accumulates further gradients into d_add in place on its main stream while the weight-gradient contraction that
    # reads d_x may still be running on a second stream; one shared tensor would be overwritten under that reader.
    d_x = torch.empty_like(d_add)
    ws, nws = None, 0
    if staged is None:
        staged = LN_BWD_STAGED == "1" or (LN_BWD_STAGED == "auto" and M >= LN_BWD_STAGED_MIN_ROWS)
    if staged and M > 0:
        nws = int(lib.capdec_layernorm_bwd_workspace_floats(M, E))
        # one slab per stream: the calls of one stream are ordered, so each may reuse it
        key = (dy.device, nws, torch.cuda.current_stream(dy.device).cuda_stream)
        ws = _LN_BWD_WS.get(key)
        if ws is None:
            ws = _LN_BWD_WS[key] = torch.empty(nws, device=dy.device, dtype=torch.float32)
    _lib.check(lib.capdec_layernorm_bwd_staged(_chk(dy).data_ptr(), s.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               _chk(gamma).data_ptr(), M, E, p, seed, _seed_ptr(seed_dev), d_add.data_ptr(),
                                               _ptr(d_x),
                                               _chk(dgamma).data_ptr(), _chk(dbeta).data_ptr(), _ptr(dxsum),
                                               _ptr(rows_dev), _ptr(ws), nws, _stream()), "capdec_layernorm_bwd_staged")
    return d_add, d_x


def attention_bwd(q, k, v, dout, lse, B, H, Tq, Tk, causal=False, ids32=None, dq=None, dk=None, dv=None, out=None,
                  packed=False):
    """Backward of ``attention``; q/dout [B*Tq,E], k/v [B*Tk,E] 2-D views.  Returns (dq, dk, dv).
    ``out`` (the forward output, optional) enables the one-pass kernel for long unmasked memories.
    packed (self-attention, Tq == Tk): the three gradients are the column blocks of one [M, 3E] tensor, which is returned
    instead (one contraction against the stacked projection weights then gives the input gradient)."""
    lib = _lib.load()
    E = q.shape[1]
    dkh = E // H
    dev = q.device
    pk = None
    if packed:
        if Tq != Tk or dq is not None or dk is not None or dv is not None:
            raise RuntimeError("capdec attention_bwd: packed gradients need Tq == Tk and no caller buffers")
        pk = torch.empty((B * Tq, 3 * E), device=dev, dtype=torch.float32)
        dq, dk, dv = pk[:, :E], pk[:, E:2 * E], pk[:, 2 * E:]
    if dq is None:
        dq = torch.empty((B * Tq, E), device=dev, dtype=torch.float32)
    if dk is None:
        dk = torch.empty((B * Tk, E), device=dev, dtype=torch.float32)
    if dv is None:
        dv = torch.empty((B * Tk, E), device=dev, dtype=torch.float32)
    dsum = torch.empty((B, H, Tq), device=dev, dtype=torch.float32)
    ldid = 0
    if ids32 is not None:
        ldid = ids32.stride(0) if B > 1 else max(ids32.shape[1], Tq)
    ldk, ldv, lddk, lddv = (_row_stride(t, "kv") for t in (k, v, dk, dv))
    _lib.check(lib.capdec_attention_bwd(_chk(q).data_ptr(), _row_stride(q, "q"), _chk(k).data_ptr(), ldk, ldk * Tk,
                                        _chk(v).data_ptr(), ldv, ldv * Tk, _chk(dout).data_ptr(),
                                        _row_stride(dout, "dout"), _ptr(out), 0 if out is None else _row_stride(out, "out"),
                                        _chk(lse).data_ptr(), dsum.data_ptr(),
                                        dq.data_ptr(), _row_stride(dq, "dq"), dk.data_ptr(), lddk, lddk * Tk,
                                        dv.data_ptr(), lddv, lddv * Tk, B, H, Tq, Tk, dkh, 1 if causal else 0, 0,
                                        _ptr(ids32), ldid, _stream()), "capdec_attention_bwd")
    return pk if packed else (dq, dk, dv)


def embed_bwd(ids, dx, demb, rows_dev=None, p=0.0, seed=0, seed_dev=None):
    """demb[ids[r]] += dx[r]; with p > 0 dx is first scaled by the dropout mask of (p, seed, seed_dev) over its elements
    (the backward of ``embed_train``), in the same launch."""
    lib = _lib.load()
    ids = ids.contiguous()
    rows, E = dx.shape
    if not dx.is_contiguous():
        raise RuntimeError("capdec embed_bwd: dx must be contiguous")
    _lib.check(lib.capdec_embed_train_bwd(_chk(ids, torch.int64).data_ptr(), _chk(dx).data_ptr(), rows, E, demb.shape[0],
                                          p, seed, _seed_ptr(seed_dev), _chk(demb).data_ptr(), _ptr(rows_dev), _stream()),
               "capdec_embed_train_bwd")
    return demb


def embed_train(ids, emb, pe, p, seed, seed_dev=None, order=None, want_ids=False):
    """dropout(emb[ids] + pe) of a teacher-forced batch in one launch.  ids [B,T] int64, rows may be strided (a view such
    as captions[:, :-1]); order (int64 [M], optional): slot s takes dense row order[s] (a ragged batch); want_ids: also
    return the slots' ids [M].  Returns x [M,E] (and the ids)."""
    lib = _lib.load()
    _chk(ids, torch.int64, "ids")
    if ids.dim() != 2 or (ids.shape[1] > 1 and ids.stride(1) != 1):
        raise RuntimeError("capdec embed_train: ids must be [B,T] with unit column stride")
    B, T = ids.shape
    V, E = emb.shape
    M = B * T
    if order is not None and (order.numel() != M or order.dtype != torch.int64 or not order.is_contiguous()):
        raise RuntimeError("capdec embed_train: order must be a contiguous int64 [B*T] tensor")
    if pe is not None and (pe.shape[0] < T or pe.shape[1] != E or not pe.is_contiguous()):
        raise RuntimeError(f"capdec embed_train: pe {tuple(pe.shape)} too short for {T} positions")
    x = torch.empty((M, E), device=ids.device, dtype=torch.float32)
    ids_out = torch.empty(M, device=ids.device, dtype=torch.int64) if want_ids else None
    ldid = ids.stride(0) if B > 1 else max(T, ids.stride(0))
    _lib.check(lib.capdec_embed_train_fwd(ids.data_ptr(), ldid, _ptr(order), _chk(emb).data_ptr(), _ptr(pe), M, T, E, V, p,
                                          seed, _seed_ptr(seed_dev), x.data_ptr(), _ptr(ids_out), _stream()),
               "capdec_embed_train_fwd")
    return (x, ids_out) if want_ids else x


def bn_train_backward(dy, x, gamma, mean, rstd, dgamma, dbeta):
    lib = _lib.load()
    B, C = x.shape
    dx = torch.empty_like(x)
    _lib.check(lib.capdec_bn_train_backward(_chk(dy).data_ptr(), _chk(x).data_ptr(), _chk(gamma).data_ptr(),
                                            mean.data_ptr(), rstd.data_ptr(), B, C, dx.data_ptr(), dgamma.data_ptr(),
                                            dbeta.data_ptr(), _stream()), "capdec_bn_train_backward")
    return dx


def softmax_grad_logits(x, W, b, lse, targets, gscale, ignore_index=0, out=None):
    """dlogits = (softmax(x W^T + b) - onehot(targets)) * gscale on counted rows, recomputed from the projection."""
    M, K = x.shape
    V = W.shape[0]
    if out is None:
        out = torch.empty((M, V), device=x.device, dtype=torch.float32)
    return gemm_raw(x, W, out, M, V, K, _row_stride(x, "x"), _row_stride(W, "W"), 0, bias_j=b, lse=lse, tgt=targets,
                    gscale=gscale, ignore_index=ignore_index)


def group_sum_segments(x, seg_off, G):
    """[rows, E] -> [G, E]: out[g] = sum of the rows [seg_off[g], seg_off[g+1]) (seg_off: int32 [G+1] on the device)."""
    lib = _lib.load()
    E = x.shape[1]
    out = torch.empty((G, E), device=x.device, dtype=torch.float32)
    _lib.check(lib.capdec_group_sum_segments(_chk(x).data_ptr(), G, _chk(seg_off, torch.int32, "seg_off").data_ptr(), E,
                                             out.data_ptr(), _stream()), "capdec_group_sum_segments")
    return out


class RaggedRows:
    """The live prefixes of a padded caption batch as a ragged batch (see capdec_compact_prefix)."""

    def __init__(self, ids, targets, pad=0):
        lib = _lib.load()
        _chk(ids, torch.int64, "ids")
        _chk(targets, torch.int64, "targets")
        B, T = ids.shape if ids.dim() == 2 else (0, 0)
        # packed [B,T] tensors, or two views of one wider matrix with the same row stride (captions[:, :-1], [:, 1:])
        ld = ids.stride(0) if B > 1 else T
        if (ids.shape != targets.shape or ids.dim() != 2 or (T > 1 and (ids.stride(1) != 1 or targets.stride(1) != 1)) or
                ld < T or (B > 1 and targets.stride(0) != ld)):
            raise RuntimeError("capdec RaggedRows: ids and targets must be [B,T] tensors with unit column stride and one "
                               "common row stride")
        M, dev = B * T, ids.device
        self.B, self.T, self.M = B, T, M
        self.order = torch.empty(M, device=dev, dtype=torch.int64)       # dense row of slot s
        self.slot = torch.empty(M, device=dev, dtype=torch.int64)        # slot of dense row r (M: dead)
        self.seq_off = torch.empty(B + 1, device=dev, dtype=torch.int32)
        self.sample = torch.empty(M, device=dev, dtype=torch.int32)
        self.ids_c = torch.empty(M, device=dev, dtype=torch.int32)
        self.targets_c = torch.empty(M, device=dev, dtype=torch.int64)
        self.counts = torch.empty(2, device=dev, dtype=torch.int32)
        _lib.check(lib.capdec_compact_prefix_strided(ids.data_ptr(), targets.data_ptr(), ld, B, T, pad,
                                                     self.order.data_ptr(), self.slot.data_ptr(), self.seq_off.data_ptr(),
                                                     self.sample.data_ptr(), self.ids_c.data_ptr(),
                                                     self.targets_c.data_ptr(), self.counts.data_ptr(), _stream()),
                   "capdec_compact_prefix_strided")
        self.rows = self.counts[:1]      # live rows rounded up to a multiple of 32: the row limit of every kernel


def attention_ragged(q, k, v, rr, H, want_lse=False):
    """Causal self-attention over a ragged batch ``rr`` (RaggedRows); q, k, v [M,E] by slot.  The output rows between
    the last sample and ``rr.rows[0]`` are zero, those past it are not written."""
    lib = _lib.load()
    M, E = q.shape
    out = torch.empty((M, E), device=q.device, dtype=torch.float32)
    lse = torch.empty((rr.B, H, rr.T), device=q.device, dtype=torch.float32) if want_lse else None
    _lib.check(lib.capdec_attention_ragged(_chk(q).data_ptr(), _row_stride(q, "q"), _chk(k).data_ptr(), _row_stride(k, "k"),
                                           _chk(v).data_ptr(), _row_stride(v, "v"), out.data_ptr(), E, rr.B, H, rr.T,
                                           E // H, rr.ids_c.data_ptr(), rr.seq_off.data_ptr(), _ptr(lse), _stream()),
               "capdec_attention_ragged")
    return (out, lse) if want_lse else out


def attention_bwd_ragged(q, k, v, dout, lse, rr, H, packed=False):
    """Backward of ``attention_ragged``.  Returns (dq, dk, dv) [M,E]; the rows between the last sample and
    ``rr.rows[0]`` are zero, those past it are not written.  packed: one [M, 3E] tensor whose column blocks they are."""
    lib = _lib.load()
    M, E = q.shape
    if packed:
        pk = torch.empty((M, 3 * E), device=q.device, dtype=torch.float32)
        d, ld = (pk[:, :E], pk[:, E:2 * E], pk[:, 2 * E:]), 3 * E
    else:
        d, ld = torch.empty((3, M, E), device=q.device, dtype=torch.float32), E
    _lib.check(lib.capdec_attention_bwd_ragged(_chk(q).data_ptr(), _row_stride(q, "q"), _chk(k).data_ptr(),
                                               _row_stride(k, "k"), _chk(v).data_ptr(), _row_stride(v, "v"),
                                               _chk(dout).data_ptr(), _row_stride(dout, "dout"), _chk(lse).data_ptr(),
                                               d[0].data_ptr(), ld, d[1].data_ptr(), ld, d[2].data_ptr(), ld, rr.B, H, rr.T,
                                               E // H, rr.ids_c.data_ptr(), rr.seq_off.data_ptr(), _stream()),
               "capdec_attention_bwd_ragged")
    return pk if packed else (d[0], d[1], d[2])


def cross_attention_ragged(q, k, v, rr, H, S, want_lse=False):
    """Unmasked attention of the ragged rows q [M,E] (by slot) over the dense memories k, v [B*S,E]."""
    lib = _lib.load()
    M, E = q.shape
    out = torch.empty((M, E), device=q.device, dtype=torch.float32)
    lse = torch.empty((rr.B, H, rr.T), device=q.device, dtype=torch.float32) if want_lse else None
    _lib.check(lib.capdec_cross_attention_ragged(_chk(q).data_ptr(), _row_stride(q, "q"), _chk(k).data_ptr(),
                                                 _row_stride(k, "k"), _chk(v).data_ptr(), _row_stride(v, "v"),
                                                 out.data_ptr(), E, rr.B, H, rr.T, S, E // H, rr.seq_off.data_ptr(),
                                                 _ptr(lse), _stream()), "capdec_cross_attention_ragged")
    return (out, lse) if want_lse else out


def cross_attention_bwd_ragged(q, k, v, dout, lse, rr, H, S, out=None):
    """Backward of ``cross_attention_ragged``.  Returns (dq [M,E] by slot, dk, dv [B*S,E]).  ``out`` (the forward
    output) is needed for memories longer than 64 rows."""
    lib = _lib.load()
    M, E = q.shape
    dq = torch.empty((M, E), device=q.device, dtype=torch.float32)
    dkv = torch.empty((2, rr.B * S, E), device=q.device, dtype=torch.float32)
    _lib.check(lib.capdec_cross_attention_bwd_ragged(_chk(q).data_ptr(), _row_stride(q, "q"), _chk(k).data_ptr(),
                                                     _row_stride(k, "k"), _chk(v).data_ptr(), _row_stride(v, "v"),
                                                     _chk(dout).data_ptr(), _row_stride(dout, "dout"), _ptr(out),
                                                     0 if out is None else _row_stride(out, "out"),
                                                     _chk(lse).data_ptr(), dq.data_ptr(), E, dkv[0].data_ptr(), E,
                                                     dkv[1].data_ptr(), E, rr.B, H, rr.T, S, E // H,
                                                     rr.seq_off.data_ptr(), _stream()),
               "capdec_cross_attention_bwd_ragged")
    return dq, dkv[0], dkv[1]


def softmax_nll_grad_(logits, targets, ignore_index=0, rows_dev=None):
    """In place: logits [M,V] -> d(mean NLL)/d(logits).  Returns (sums [2] = (sum nll, n counted), gscale [1]).
    rows_dev: int32 device scalar; only the first rows_dev[0] rows hold logits (see capdec_softmax_nll_grad_rows)."""
    lib = _lib.load()
    _chk(logits, name="logits")
    _chk(targets, torch.int64, "targets")
    M, V = logits.shape
    if targets.numel() != M or not targets.is_contiguous():
        raise RuntimeError("capdec softmax_nll_grad_: targets must be a contiguous [M] tensor")
    sums = torch.zeros(2, device=logits.device, dtype=torch.float32)
    gscale = torch.empty(1, device=logits.device, dtype=torch.float32)
    _lib.check(lib.capdec_softmax_nll_grad_rows(logits.data_ptr(), _row_stride(logits, "logits"), targets.data_ptr(), M,
                                                V, ignore_index, sums.data_ptr(), gscale.data_ptr(), None,
                                                _ptr(rows_dev), _stream()), "capdec_softmax_nll_grad_rows")
    return sums, gscale


def loss_finish(sums):
    lib = _lib.load()
    loss = torch.empty(1, device=sums.device, dtype=torch.float32)
    gscale = torch.empty(1, device=sums.device, dtype=torch.float32)
    _lib.check(lib.capdec_loss_finish(_chk(sums).data_ptr(), loss.data_ptr(), gscale.data_ptr(), _stream()),
               "capdec_loss_finish")
    return loss, gscale


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    lib = _lib.load()
    for t in (p, g, m, v):
        if not t.is_contiguous() or t.numel() != p.numel():
            raise RuntimeError("capdec adam_step: flat contiguous buffers of equal length expected")
    _lib.check(lib.capdec_adam_step(_chk(p).data_ptr(), _chk(g).data_ptr(), _chk(m).data_ptr(), _chk(v).data_ptr(),
                                    p.numel(), lr, beta1, beta2, eps, step, grad_scale, _stream()), "capdec_adam_step")


class TrainState:
    """capdec_train_state in device memory (32 bytes): dropout seed of the step, Adam's step count and bias corrections,
    learning rate.  ``seed_dev`` is the tensor to hand to the dropout-drawing wrappers."""

    def __init__(self, device, lr, seed, step=0):
        self.buf = torch.zeros(4, dtype=torch.int64, device=device)
        self.seed_dev = self.buf[:1]
        self._i32 = self.buf.view(torch.int32)        # [seed lo, seed hi, step, lr bits, bias1, bias2_sqrt, -, -]
        self._f32 = self.buf.view(torch.float32)
        self.set(seed=seed, step=step, lr=lr)

    def set(self, seed=None, step=None, lr=None):
        """Host-side writes (outside any captured graph): tiny device copies on the current stream."""
        if seed is not None:
            self.buf[0] = int(seed) & (2 ** 62 - 1)
        if step is not None:
            self._i32[2] = int(step)
            # bias corrections of `step` itself, for a caller that reads them before the next advance
        if lr is not None:
            self._f32[3] = float(lr)

    def advance(self, beta1, beta2):
        lib = _lib.load()
        _lib.check(lib.capdec_train_state_advance(self.buf.data_ptr(), beta1, beta2, _stream()),
                   "capdec_train_state_advance")

    def read(self):
        """(seed, step, lr, bias1, bias2_sqrt) -- synchronises; for tests and checkpoints."""
        h = self.buf.cpu()
        return (int(h[0]), int(h.view(torch.int32)[2]), float(h.view(torch.float32)[3]),
                float(h.view(torch.float32)[4]), float(h.view(torch.float32)[5]))


def adam_step_dev(p, g, m, v, state, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """Adam over flat buffers with lr / step count / bias corrections read from ``state`` (TrainState)."""
    lib = _lib.load()
    for t in (p, g, m, v):
        if not t.is_contiguous() or t.numel() != p.numel():
            raise RuntimeError("capdec adam_step_dev: flat contiguous buffers of equal length expected")
    _lib.check(lib.capdec_adam_step_dev(_chk(p).data_ptr(), _chk(g).data_ptr(), _chk(m).data_ptr(), _chk(v).data_ptr(),
                                        p.numel(), beta1, beta2, eps, state.buf.data_ptr(), grad_scale, _stream()),
               "capdec_adam_step_dev")


# ---------------------------------------------------------------------------------------------
# additive attention (soft-attention LSTM decoder; no counterpart in the reference)
# ---------------------------------------------------------------------------------------------
def additive_attention(u, pf, v, enc, ctx_out, alpha_out=None):
    """u [B,A] (row stride free), pf [B,S,A], v [A], enc [B,S,E] -> ctx_out [B,E] (a view with unit column stride, e.g.
    the right half of the LSTM's [emb | ctx] input rows); alpha_out [B,S] optional."""
    lib = _lib.load()
    B, S, A = pf.shape
    E = enc.shape[2]
    for name, t in (("pf", pf), ("enc", enc), ("v", v)):
        if not _chk(t, name=name).is_contiguous():
            raise RuntimeError(f"capdec additive_attention: {name} must be contiguous")
    if alpha_out is not None and (alpha_out.shape != (B, S) or not alpha_out.is_contiguous()):
        raise RuntimeError("capdec additive_attention: alpha_out must be a contiguous [B,S] tensor")
    _lib.check(lib.capdec_additive_attention(_chk(u, name="u").data_ptr(), _row_stride(u, "u"), pf.data_ptr(), v.data_ptr(),
                                             enc.data_ptr(), B, S, A, E, _chk(ctx_out, name="ctx_out").data_ptr(),
                                             _row_stride(ctx_out, "ctx_out"), _ptr(alpha_out), _stream()),
               "capdec_additive_attention")
    return ctx_out


def additive_attention_bwd(u, pf, v, enc, alpha, dctx, du_out, dpf_acc, denc_acc, dvb_acc):
    """Backward of one ``additive_attention`` step: writes du_out [B,A]; accumulates into dpf_acc [B,S,A], denc_acc
    [B,S,E] and dvb_acc [B,A] (per-sample partial of the gradient of v)."""
    lib = _lib.load()
    B, S, A = pf.shape
    E = enc.shape[2]
    for name, t in (("pf", pf), ("enc", enc), ("v", v), ("alpha", alpha), ("dpf", dpf_acc), ("denc", denc_acc), ("dvb", dvb_acc)):
        if not _chk(t, name=name).is_contiguous():
            raise RuntimeError(f"capdec additive_attention_bwd: {name} must be contiguous")
    if dpf_acc.numel() != B * S * A or denc_acc.numel() != B * S * E or dvb_acc.numel() != B * A or alpha.numel() != B * S:
        raise RuntimeError("capdec additive_attention_bwd: accumulator shapes do not match [B,S,A] / [B,S,E] / [B,A]")
    _lib.check(lib.capdec_additive_attention_bwd(_chk(u, name="u").data_ptr(), _row_stride(u, "u"), pf.data_ptr(),
                                                 v.data_ptr(), enc.data_ptr(), alpha.data_ptr(),
                                                 _chk(dctx, name="dctx").data_ptr(), _row_stride(dctx, "dctx"), B, S, A, E,
                                                 _chk(du_out, name="du").data_ptr(), _row_stride(du_out, "du"),
                                                 dpf_acc.data_ptr(), denc_acc.data_ptr(), dvb_acc.data_ptr(), _stream()),
               "capdec_additive_attention_bwd")
    return du_out


def axpy_rows(x, scale, x_div=1, out=None, rows=None, accumulate=False):
    """out[r,:] = (accumulate ? out[r,:] : 0) + scale * x[r // x_div, :] for r < rows (default: x rows * x_div)."""
    lib = _lib.load()
    _chk(x, name="x")
    E = x.shape[1]
    M = x.shape[0] * x_div if rows is None else rows
    if out is None:
        if accumulate:
            raise RuntimeError("capdec axpy_rows: accumulate needs an output tensor")
        out = torch.empty((M, E), device=x.device, dtype=torch.float32)
    if not x.is_contiguous() or not out.is_contiguous() or out.shape != (M, E):
        raise RuntimeError("capdec axpy_rows: contiguous [rows,E] tensors expected")
    _lib.check(lib.capdec_axpy_rows(x.data_ptr(), x_div, float(scale), M, E, _chk(out, name="out").data_ptr(),
                                    1 if accumulate else 0, _stream()), "capdec_axpy_rows")
    return out
