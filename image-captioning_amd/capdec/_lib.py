"""ctypes binding of libcapdec.so (the C ABI declared in include/capdec.h).

There is no CPU fallback: if the library is missing or a tensor is not on a HIP device the call
raises.  Build the library first with ``python image-captioning_amd/capdec/_build.py``.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcapdec.so")
LIB_TUNE_PATH = os.path.join(HERE, "libcapdec_tune.so")

ABI_VERSION = 2
GEMM_RELU = 1
GEMM_TP = 2
GEMM_TQ = 4
GEMM_ATOMIC = 8
GEMM_SPLITK = 16
DECODE_NO_S1_SHORTCUT = 1
DECODE_FOLD_MEMORY = 2
DECODE_NO_FUSED_LAYERS = 4
VOCAB_STATIC_MAX_ELEMS = 1 << 20
E2BIG = -7


class GemmGroup(Structure):
    _fields_ = [("P", c_void_p), ("Q", c_void_p), ("C", c_void_p), ("bias_i", c_void_p), ("bias_j", c_void_p),
                ("scale_j", c_void_p), ("shift_j", c_void_p), ("R", c_void_p),
                ("ldp", c_int64), ("ldq", c_int64), ("ldci", c_int64), ("ldcj", c_int64), ("ldri", c_int64),
                ("ldrj", c_int64), ("I", c_int32), ("J", c_int32), ("c_rows", c_void_p)]


class BgemmDesc(Structure):
    _fields_ = [("P", c_void_p), ("Q", c_void_p), ("C", c_void_p), ("P2", c_void_p), ("Q2", c_void_p),
                ("ldp", c_int64), ("ldq", c_int64), ("ldc", c_int64), ("dense_stride", c_int64), ("seg_off", c_void_p),
                ("mult", c_int32), ("seg", c_int32), ("B", c_int32), ("I", c_int32), ("J", c_int32), ("K", c_int32),
                ("scale", c_float), ("accumulate", c_int32)]


class GemmExtra(Structure):
    _fields_ = [("mask", c_void_p), ("ldm", c_int64), ("lse", c_void_p), ("tgt", c_void_p), ("gscale", c_void_p),
                ("ignore_index", c_int64), ("rows_dev", c_void_p), ("k_dev", c_void_p), ("splitk_ws", c_void_p),
                ("splitk_ws_elems", c_int64)]


class AttnW(Structure):
    _fields_ = [(n, c_void_p) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo")]


class TfLayer(Structure):
    _fields_ = [("self_attn", AttnW), ("cross_attn", AttnW)] + \
               [(n, c_void_p) for n in ("ff_w1", "ff_b1", "ff_w2", "ff_b2", "n1_g", "n1_b", "n2_g", "n2_b", "n3_g", "n3_b")]


class TfModel(Structure):
    _fields_ = [("E", c_int32), ("H", c_int32), ("L", c_int32), ("V", c_int32), ("max_pos", c_int32),
                ("emb", c_void_p), ("pe", c_void_p), ("fc2_w", c_void_p), ("fc2_b", c_void_p),
                ("layers", POINTER(TfLayer))]


class RnnModel(Structure):
    _fields_ = [("E", c_int32), ("Hd", c_int32), ("V", c_int32), ("emb", c_void_p), ("w_ih", c_void_p),
                ("w_hh", c_void_p), ("b_ih", c_void_p), ("b_hh", c_void_p), ("fc2_w", c_void_p), ("fc2_b", c_void_p)]


class SattModel(Structure):
    _fields_ = [("E", c_int32), ("Hd", c_int32), ("A", c_int32), ("V", c_int32)] + \
               [(n, c_void_p) for n in ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "att_h_w", "att_h_b", "att_v", "fc2_w", "fc2_b")]


class Probe(Structure):
    _fields_ = [("begin", POINTER(c_void_p)), ("end", POINTER(c_void_p)), ("n", c_int32)]


_P = c_void_p
_SIGNATURES = {
    "capdec_abi_version": (c_int, []),
    "capdec_strerror": (c_char_p, [c_int]),
    "capdec_gemm": (c_int, [POINTER(GemmGroup), c_int, c_int, c_int, _P]),
    "capdec_gemm_splitk_workspace_elems": (c_size_t, [c_int, c_int, c_int]),
    "capdec_vocab_tiles": (c_int, [c_int]),
    "capdec_vocab_partial_elems": (c_size_t, [c_int, c_int]),
    "capdec_vocab_argmax_partial": (c_int, [_P, c_int64, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "capdec_argmax_finalize": (c_int, [_P, _P, c_int, c_int, _P, _P, _P]),
    "capdec_vocab_lse_partial": (c_int, [_P, c_int64, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "capdec_nll_finalize": (c_int, [_P, _P, c_int, c_int, _P, c_int64, _P, _P, c_int, _P, c_int64, _P, _P, _P, _P]),
    "capdec_embed_pe": (c_int, [_P, c_int, c_int64, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "capdec_add_layernorm": (c_int, [_P, _P, c_int, _P, _P, c_float, c_int, c_int, _P, _P, _P, _P]),
    "capdec_attention": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64,
                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int64, _P, _P]),
    "capdec_lstm_step": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_int64, _P, _P, _P]),
    "capdec_lstm_step_bwd": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, c_int, c_int, _P, _P, _P]),
    "capdec_gemm_ex": (c_int, [POINTER(GemmGroup), c_int, c_int, c_int, POINTER(GemmExtra), _P]),
    "capdec_dropout": (c_int, [_P, c_int64, c_float, c_uint64, _P, _P, _P]),
    "capdec_add_layernorm_train": (c_int, [_P, c_int, _P, _P, _P, c_float, c_int, c_int, c_float, c_uint64, _P, _P, _P, _P, _P]),
    "capdec_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, c_uint64, _P, _P, _P, _P, _P, _P]),
    "capdec_colsum": (c_int, [_P, c_int64, c_int, c_int, _P, _P]),
    "capdec_colsum_rows": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P]),
    "capdec_bgemm": (c_int, [POINTER(BgemmDesc), c_int, _P]),
    "capdec_pair_diff": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "capdec_pair_diff_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "capdec_pair_gate_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    "capdec_pair_gate_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P]),
    "capdec_softmax_rows": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "capdec_softmax_rows_bwd": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, c_float, _P, c_int, c_int, _P]),
    "capdec_softmax_nll_grad": (c_int, [_P, c_int64, _P, c_int, c_int, c_int64, _P, _P, _P, _P]),
    "capdec_softmax_nll_grad_rows": (c_int, [_P, c_int64, _P, c_int, c_int, c_int64, _P, _P, _P, _P, _P]),
    "capdec_compact_targets": (c_int, [_P, c_int, c_int64, _P, _P, _P, _P, _P]),
    "capdec_group_sum": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "capdec_embed_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    "capdec_bn_train_backward": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    "capdec_attention_bwd": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, _P,
                                     _P, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int64, _P]),
    "capdec_loss_finish": (c_int, [_P, _P, _P, _P]),
    "capdec_add_layernorm_train_rows": (c_int, [_P, c_int, _P, _P, _P, _P, c_float, c_int, c_int, c_float, c_uint64, _P, _P,
                                                _P, _P, _P, _P, _P]),
    "capdec_layernorm_bwd_rows": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, c_uint64, _P, _P, _P, _P, _P, _P, _P,
                                          _P]),
    "capdec_layernorm_bwd_workspace_floats": (c_size_t, [c_int, c_int]),
    "capdec_layernorm_bwd_staged": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, c_uint64, _P, _P, _P, _P, _P, _P, _P,
                                            _P, c_size_t, _P]),
    "capdec_group_sum_segments": (c_int, [_P, c_int, _P, c_int, _P, _P]),
    "capdec_embed_bwd_rows": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "capdec_embed_train_fwd": (c_int, [_P, c_int64, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_uint64, _P, _P, _P, _P]),
    "capdec_embed_train_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_float, c_uint64, _P, _P, _P, _P]),
    "capdec_compact_prefix_strided": (c_int, [_P, _P, c_int64, c_int, c_int, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "capdec_compact_prefix": (c_int, [_P, _P, c_int, c_int, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "capdec_attention_ragged": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int, c_int, c_int, _P,
                                        _P, _P, _P]),
    "capdec_cross_attention_ragged": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int, c_int, c_int,
                                              c_int, _P, _P, _P]),
    "capdec_cross_attention_bwd_ragged": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, _P,
                                                  c_int64, _P, c_int64, _P, c_int64, c_int, c_int, c_int, c_int, c_int,
                                                  _P, _P]),
    "capdec_attention_bwd_ragged": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int64,
                                            _P, c_int64, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "capdec_adam_step": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int, c_float, _P]),
    "capdec_bn_fold": (c_int, [_P, _P, _P, _P, c_float, c_int, _P, _P, _P]),
    "capdec_bn_train_forward": (c_int, [_P, _P, _P, c_float, c_float, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    "capdec_tf_decode_workspace_bytes": (c_size_t, [POINTER(TfModel), c_int, c_int, c_int, c_int]),
    "capdec_tf_greedy_decode": (c_int, [POINTER(TfModel), _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                        c_size_t, _P, _P, _P, POINTER(Probe), _P]),
    "capdec_event_create": (c_int, [POINTER(c_void_p)]),
    "capdec_event_destroy": (c_int, [_P]),
    "capdec_event_record": (c_int, [_P, _P]),
    "capdec_event_elapsed_ms": (c_int, [_P, _P, POINTER(c_float)]),
    "capdec_tf_exact_workspace_bytes": (c_size_t, [POINTER(TfModel), c_int, c_int, c_int]),
    "capdec_tf_greedy_decode_exact": (c_int, [POINTER(TfModel), _P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t,
                                              _P, _P, _P]),
    "capdec_tf_forward_workspace_bytes": (c_size_t, [POINTER(TfModel), c_int, c_int, c_int]),
    "capdec_tf_forward_logits": (c_int, [POINTER(TfModel), _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "capdec_rnn_decode_workspace_bytes": (c_size_t, [POINTER(RnnModel), c_int, c_int]),
    "capdec_rnn_greedy_decode": (c_int, [POINTER(RnnModel), _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "capdec_additive_attention": (c_int, [_P, c_int64, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int64, _P, _P]),
    "capdec_satt_decode_workspace_bytes": (c_size_t, [POINTER(SattModel), c_int, c_int]),
    "capdec_satt_greedy_decode": (c_int, [POINTER(SattModel), _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P,
                                          c_size_t, _P, _P, _P]),
    "capdec_tf_beam_workspace_bytes": (c_size_t, [POINTER(TfModel), c_int, c_int, c_int, c_int]),
    "capdec_tf_beam_search": (c_int, [POINTER(TfModel), _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                      c_size_t, _P, _P, _P, _P, _P]),
    "capdec_tf_beam_cached_workspace_bytes": (c_size_t, [POINTER(TfModel), c_int, c_int, c_int, c_int]),
    "capdec_tf_beam_search_cached": (c_int, [POINTER(TfModel), _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                             _P, c_size_t, _P, _P, _P, _P, _P, _P]),
    "capdec_rnn_beam_workspace_bytes": (c_size_t, [POINTER(RnnModel), c_int, c_int, c_int]),
    "capdec_rnn_beam_search": (c_int, [POINTER(RnnModel), _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                       c_size_t, _P, _P, _P, _P, _P]),
    "capdec_train_state_advance": (c_int, [_P, c_float, c_float, _P]),
    "capdec_adam_step_dev": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, _P, c_float, _P]),
    "capdec_additive_attention_bwd": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P, c_int64,
                                              _P, _P, _P, _P]),
    "capdec_axpy_rows": (c_int, [_P, c_int, c_float, c_int, c_int, _P, c_int, _P]),
    "capdec_topk_rows": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "capdec_vocab_argmax_static": (c_int, [_P, c_int64, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
}

# include/capdec_tuning.h: exported by the tuning build only
_TUNING_SIGNATURES = {
    "capdec_gemm_force_config": (c_int, [c_int]),
    "capdec_gemm_force_staging": (c_int, [c_int]),
    "capdec_lstm_force_staging": (c_int, [c_int]),
    "capdec_attention_bwd_force_two_pass": (c_int, [c_int]),
    "capdec_attention_bwd_force_row_form": (c_int, [c_int]),
    "capdec_attention_ragged_force_row_form": (c_int, [c_int]),
    "capdec_attention_force_row_kernel": (c_int, [c_int]),
    "capdec_beam_force_unfused": (c_int, [c_int]),
}

_lib = None          # the library every wrapper of this package calls
_product = None
_tuning = None


def _open(path, signatures):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{os.path.basename(path)} not found at {path}: the HIP kernels are the only implementation of this package "
            "(no CPU fallback). Build them with `python image-captioning_amd/capdec/_build.py` (needs hipcc).")
    # PyTorch ships its own HIP runtime; it must be the one already mapped when the library resolves its HIP symbols,
    # or the process ends up with two runtimes and this library's copy finds no device.
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (res, args) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{os.path.basename(path)} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.capdec_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{os.path.basename(path)} ABI version mismatch; rebuild it")
    return lib


def load():
    """The library in use: libcapdec.so, loaded once (or the tuning build inside ``tuning_build()``); raises loudly if
    it is missing or has the wrong ABI."""
    global _lib, _product
    if _lib is not None:
        return _lib
    _product = _open(LIB_PATH, _SIGNATURES)
    _lib = _product
    return _lib


class tuning_build:
    """Context manager for tests and benchmark tools: inside it every wrapper of this package calls libcapdec_tune.so,
    the same kernels compiled with the process-wide kernel-selection overrides of include/capdec_tuning.h; yields that
    library.  The overrides are reset on exit and the product library is put back."""

    def __enter__(self):
        global _lib, _tuning
        load()
        if _tuning is None:
            _tuning = _open(LIB_TUNE_PATH, {**_SIGNATURES, **_TUNING_SIGNATURES})
        self._saved = _lib
        _lib = _tuning
        return _tuning

    def __exit__(self, *exc):
        global _lib
        for name in _TUNING_SIGNATURES:
            getattr(_tuning, name)(0)
        _lib = self._saved
        return False


def exported_symbols():
    return sorted(_SIGNATURES)


def tuning_symbols():
    return sorted(_TUNING_SIGNATURES)


def check(rc, what):
    if rc != 0:
        msg = load().capdec_strerror(int(rc)).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")
