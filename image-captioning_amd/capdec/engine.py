"""Host side of the whole-loop greedy decoders: packs a module's parameters into the C structs,
owns the workspace tensor, and resolves rows that generated <PAD> through the exact path.

The decode loops themselves live in libcapdec.so (csrc/engine.hip); one C call enqueues every step.
"""
import ctypes
import warnings

import torch

from . import _lib
from ._lib import AttnW, RnnModel, TfLayer, TfModel


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("capdec: decoder parameters must be contiguous fp32 tensors on the GPU (HIP)")
    return t.data_ptr()


def _attn(m):
    return AttnW(_p(m.W_q.weight), _p(m.W_q.bias), _p(m.W_k.weight), _p(m.W_k.bias), _p(m.W_v.weight),
                 _p(m.W_v.bias), _p(m.W_o.weight), _p(m.W_o.bias))


class TfModelRef:
    """capdec_tf_model built from a DecoderTransformer; keeps the ctypes arrays alive."""

    def __init__(self, dec):
        L = len(dec.decoder_layers)
        self.layers = (TfLayer * max(L, 1))()
        for l, ly in enumerate(dec.decoder_layers):
            c = self.layers[l]
            c.self_attn = _attn(ly.self_attn)
            c.cross_attn = _attn(ly.cross_attn)
            c.ff_w1, c.ff_b1 = _p(ly.feed_forward.fc1.weight), _p(ly.feed_forward.fc1.bias)
            c.ff_w2, c.ff_b2 = _p(ly.feed_forward.fc2.weight), _p(ly.feed_forward.fc2.bias)
            c.n1_g, c.n1_b = _p(ly.norm1.weight), _p(ly.norm1.bias)
            c.n2_g, c.n2_b = _p(ly.norm2.weight), _p(ly.norm2.bias)
            c.n3_g, c.n3_b = _p(ly.norm3.weight), _p(ly.norm3.bias)
        emb = dec.decoder_embedding.weight
        pe = dec.positional_encoding.pe
        self.V, self.E = emb.shape
        self.H = dec.decoder_layers[0].self_attn.num_heads if L else dec.num_heads
        self.L = L
        self.max_pos = pe.shape[1]
        self.c = TfModel(self.E, self.H, L, self.V, self.max_pos, _p(emb), _p(pe), _p(dec.fc2.weight),
                         _p(dec.fc2.bias), ctypes.cast(self.layers, ctypes.POINTER(TfLayer)))
        self.device = emb.device


def _workspace(nbytes, device):
    return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)


class VocabProbe:
    """HIP events around the vocabulary-projection kernel of decode steps ``first, first + every, ...`` below ``n``
    (benchmark plumbing; every record takes about 3 us of stream time, which is why a caller may sample)."""

    def __init__(self, n, every=1, first=0):
        lib = _lib.load()
        self.n = n
        self.steps = [i for i in range(n) if i >= first and (i - first) % every == 0]
        self.begin = (ctypes.c_void_p * n)()          # entries left null: steps that are not timed
        self.end = (ctypes.c_void_p * n)()
        for arr in (self.begin, self.end):
            for i in self.steps:
                ev = ctypes.c_void_p()
                _lib.check(lib.capdec_event_create(ctypes.byref(ev)), "capdec_event_create")
                arr[i] = ev
        self.c = _lib.Probe(ctypes.cast(self.begin, ctypes.POINTER(ctypes.c_void_p)),
                            ctypes.cast(self.end, ctypes.POINTER(ctypes.c_void_p)), n)

    def elapsed_ms(self):
        """Per-step kernel durations; call after the stream has been synchronised."""
        lib = _lib.load()
        out = []
        for i in self.steps:
            ms = ctypes.c_float()
            _lib.check(lib.capdec_event_elapsed_ms(self.begin[i], self.end[i], ctypes.byref(ms)), "capdec_event_elapsed_ms")
            out.append(ms.value)
        return out

    def close(self):
        lib = _lib.load()
        for arr in (self.begin, self.end):
            for i in self.steps:
                lib.capdec_event_destroy(arr[i])
        self.n, self.steps = 0, []


def tf_greedy_ids(dec, enc, max_length, sos, eos, exact=False, no_s1_shortcut=False, resolve_dirty=True, probe=None,
                  to_cpu=True, extra_cap=0, fold_memory=False, fused_layers=True, keep_workspace=None, keep_pack=None):
    """Greedy decode on the GPU.  enc [B,S,E] fp32.  Returns (ids int64 [B,max_length] padded with -1, lengths
    int32 [B], number of rows that went through the exact path).  ids/lengths are CPU tensors unless
    ``to_cpu=False`` (then they stay on the device, e.g. for a cross-rank gather).  ``fused_layers=False`` keeps the
    launch-per-operation form of a step where the library would run all decoder layers as one launch (capdec.h,
    CAPDEC_DECODE_NO_FUSED_LAYERS).  ``keep_workspace``: a list that receives the workspace tensor (tests);
    ``keep_pack``: a list that receives the device allocation holding ids, lengths and flags back to back."""
    lib = _lib.load()
    if not enc.is_cuda or enc.dtype != torch.float32:
        raise RuntimeError("capdec: enc must be an fp32 tensor on the GPU (HIP); there is no CPU path")
    enc = enc.contiguous()
    B, S, E = enc.shape
    ref = dec if isinstance(dec, TfModelRef) else TfModelRef(dec)     # a serving loop builds the struct once
    if E != ref.E:
        raise RuntimeError(f"capdec: encoder width {E} != embed size {ref.E}")
    dev = enc.device
    # ids, lengths and the dirty flags share one allocation, so that one copy brings all three to the host
    n_ids = 2 * B * max_length                                   # int64 ids counted in int32 words
    pack = torch.empty((n_ids + 2 * B,), dtype=torch.int32, device=dev)
    out_ids = pack[:n_ids].view(torch.int64).view(B, max_length)
    out_len = pack[n_ids:n_ids + B]
    dirty = pack[n_ids + B:]
    if keep_pack is not None:
        keep_pack.append(pack)

    def to_host():
        host = torch.empty(pack.shape, dtype=torch.int32, pin_memory=True)
        host.copy_(pack, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        return host[:n_ids].view(torch.int64).view(B, max_length), host[n_ids:n_ids + B], host[n_ids + B:]

    n_dirty = 0
    host = None
    if B > 0:
        with torch.cuda.device(dev):
            if exact:
                _exact(lib, ref, enc, max_length, sos, eos, out_ids, out_len)
                n_dirty = B
            else:
                nbytes = lib.capdec_tf_decode_workspace_bytes(ctypes.byref(ref.c), B, S, max_length, extra_cap)
                if nbytes == 0:
                    raise RuntimeError("capdec: unsupported decoder configuration (see capdec.h)")
                ws = _workspace(nbytes, dev)
                # no_s1_shortcut: the general cross-attention kernels whatever the number of memory rows;
                # fold_memory: the folded form (taken by itself for long memories) forced for any S >= 3
                flags = (_lib.DECODE_NO_S1_SHORTCUT if no_s1_shortcut else 0) | (_lib.DECODE_FOLD_MEMORY if fold_memory else 0)
                if not fused_layers:
                    flags |= _lib.DECODE_NO_FUSED_LAYERS
                if keep_workspace is not None:
                    keep_workspace.append(ws)
                _lib.check(lib.capdec_tf_greedy_decode(ctypes.byref(ref.c), enc.data_ptr(), B, S, max_length, sos, eos,
                                                       flags, extra_cap, ws.data_ptr(), ws.numel(), out_ids.data_ptr(),
                                                       out_len.data_ptr(), dirty.data_ptr(),
                                                       ctypes.byref(probe.c) if probe else None, _stream()),
                           "capdec_tf_greedy_decode")
                if resolve_dirty:
                    # the one host sync of the decode
                    if to_cpu:
                        host = to_host()
                    else:
                        # the flags alone, through pinned memory (a pageable copy is staged and waited for twice)
                        hflags = torch.empty(dirty.shape, dtype=torch.int32, pin_memory=True)
                        hflags.copy_(dirty, non_blocking=True)
                        torch.cuda.current_stream(dev).synchronize()
                        host = (None, None, hflags)
                    # (almost always no flag is set: one reduction instead of building an index list)
                    rows = torch.nonzero(host[2], as_tuple=False).flatten() if bool(host[2].any()) else None
                    n_dirty = 0 if rows is None else int(rows.numel())
                    if n_dirty and bool((host[2] == 2).any()):
                        warnings.warn("capdec: a workgroup of the one-launch decoder layers gave up waiting for its team; "
                                      "the batch is re-decoded on the exact path", RuntimeWarning)
                    if n_dirty:
                        # samples whose <PAD>-induced re-run rows did not fit the step's row list (see capdec.h):
                        # redo those samples on the exact full-prefix path
                        rows = rows.to(dev)
                        sub = enc.index_select(0, rows).contiguous()
                        ids2 = torch.empty((n_dirty, max_length), dtype=torch.int64, device=dev)
                        len2 = torch.empty((n_dirty,), dtype=torch.int32, device=dev)
                        _exact(lib, ref, sub, max_length, sos, eos, ids2, len2)
                        out_ids.index_copy_(0, rows, ids2)
                        out_len.index_copy_(0, rows, len2)
                        host = None
    if to_cpu:
        if host is None:
            with torch.cuda.device(dev):
                host = to_host()
        return host[0], host[1], n_dirty
    return out_ids, out_len, n_dirty


class PendingCaptions:
    """A greedy decode that has been enqueued but not waited for (``tf_greedy_ids_async``): the ids, lengths and flags
    are on their way to pinned host memory behind an event on the decode's stream.  ``result()`` waits for that event
    only, re-decodes flagged samples on the exact path if there are any, and returns what ``tf_greedy_ids`` returns.
    A serving loop enqueues batch k+1 before it asks for batch k, so the host's work between two batches (feature
    head, argument set-up, the wait itself) no longer leaves the GPU idle."""

    def __init__(self, dec, enc, max_length, sos, eos, pack, host, event):
        self._args = (dec, enc, max_length, sos, eos)
        self._pack, self._host, self._event = pack, host, event

    def result(self):
        dec, enc, max_length, sos, eos = self._args
        B = enc.shape[0]
        n_ids = 2 * B * max_length
        self._event.synchronize()
        host = self._host
        ids = host[:n_ids].view(torch.int64).view(B, max_length)
        lens, dirty = host[n_ids:n_ids + B], host[n_ids + B:]
        if not bool(dirty.any()):
            return ids, lens, 0
        # rare: samples the cached decode could not follow (capdec.h, out_dirty): the synchronous wrapper redoes them
        return tf_greedy_ids(dec, enc, max_length, sos, eos)


def tf_greedy_ids_async(dec, enc, max_length, sos, eos):
    """``tf_greedy_ids`` without the wait: enqueues the decode and the copy of its results to pinned host memory on the
    current stream and returns a ``PendingCaptions``."""
    dev = enc.device
    B = enc.shape[0]
    kept = []
    tf_greedy_ids(dec, enc, max_length, sos, eos, resolve_dirty=False, to_cpu=False, keep_pack=kept)
    pack = kept[0]                                  # ids, lengths and flags in one allocation: one copy brings all three
    host = torch.empty(pack.shape, dtype=torch.int32, pin_memory=True)
    with torch.cuda.device(dev):
        host.copy_(pack, non_blocking=True)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(dev))
    return PendingCaptions(dec, enc, max_length, sos, eos, pack, host, event)


def _exact(lib, ref, enc, max_length, sos, eos, out_ids, out_len):
    B, S, _ = enc.shape
    nbytes = lib.capdec_tf_exact_workspace_bytes(ctypes.byref(ref.c), B, S, max_length)
    if nbytes == 0 and B > 0:
        raise RuntimeError("capdec: unsupported decoder configuration (see capdec.h)")
    ws = _workspace(nbytes, enc.device)
    _lib.check(lib.capdec_tf_greedy_decode_exact(ctypes.byref(ref.c), enc.data_ptr(), B, S, max_length, sos, eos,
                                                 ws.data_ptr(), ws.numel(), out_ids.data_ptr(), out_len.data_ptr(),
                                                 _stream()), "capdec_tf_greedy_decode_exact")


def tf_forward_logits(dec, enc, ids, take_last=False):
    """Inference-only teacher-forced forward (no autograd graph): logits [B,T,V] or [B,V]."""
    lib = _lib.load()
    if not enc.is_cuda or not ids.is_cuda:
        raise RuntimeError("capdec: inputs must live on the GPU (HIP); there is no CPU path")
    if enc.dtype != torch.float32:
        raise RuntimeError(f"capdec: enc must be fp32 (got {enc.dtype}); the kernels read it as float*")
    if ids.dtype not in (torch.int64, torch.int32):
        raise RuntimeError(f"capdec: ids must be an integer tensor (got {ids.dtype})")
    enc = enc.contiguous()
    ids = ids.contiguous().to(torch.int64)
    B, S, _ = enc.shape
    T = ids.shape[1]
    ref = TfModelRef(dec)
    with torch.cuda.device(enc.device):
        nbytes = lib.capdec_tf_forward_workspace_bytes(ctypes.byref(ref.c), B, S, T)
        ws = _workspace(nbytes, enc.device)
        shape = (B, ref.V) if take_last else (B, T, ref.V)
        logits = torch.empty(shape, dtype=torch.float32, device=enc.device)
        _lib.check(lib.capdec_tf_forward_logits(ctypes.byref(ref.c), enc.data_ptr(), ids.data_ptr(), B, S, T,
                                                1 if take_last else 0, ws.data_ptr(), ws.numel(), logits.data_ptr(),
                                                _stream()), "capdec_tf_forward_logits")
    return logits


def rnn_greedy_ids(rnn, x0, max_length, sos, eos):
    """CNN-RNN greedy decode.  x0 [B,E] is the BatchNorm'ed image projection.  Returns CPU (ids, lengths)."""
    lib = _lib.load()
    if not x0.is_cuda or x0.dtype != torch.float32:
        raise RuntimeError("capdec: x0 must be an fp32 tensor on the GPU (HIP); there is no CPU path")
    x0 = x0.contiguous()
    B, E = x0.shape
    V = rnn.embed.weight.shape[0]
    Hd = rnn.lstm.weight_hh_l0.shape[1]
    m = RnnModel(E, Hd, V, _p(rnn.embed.weight), _p(rnn.lstm.weight_ih_l0), _p(rnn.lstm.weight_hh_l0),
                 _p(rnn.lstm.bias_ih_l0), _p(rnn.lstm.bias_hh_l0), _p(rnn.fc2.weight), _p(rnn.fc2.bias))
    dev = x0.device
    if B == 0:
        return torch.empty((0, max_length), dtype=torch.int64), torch.empty((0,), dtype=torch.int32)
    out_ids = torch.empty((B, max_length), dtype=torch.int64, device=dev)
    out_len = torch.empty((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        nbytes = lib.capdec_rnn_decode_workspace_bytes(ctypes.byref(m), B, max_length)
        ws = _workspace(nbytes, dev)
        _lib.check(lib.capdec_rnn_greedy_decode(ctypes.byref(m), x0.data_ptr(), B, max_length, sos, eos, ws.data_ptr(),
                                                ws.numel(), out_ids.data_ptr(), out_len.data_ptr(), _stream()),
                   "capdec_rnn_greedy_decode")
    return out_ids.cpu(), out_len.cpu()


def _rnn_model(rnn, E):
    V = rnn.embed.weight.shape[0]
    Hd = rnn.lstm.weight_hh_l0.shape[1]
    return RnnModel(E, Hd, V, _p(rnn.embed.weight), _p(rnn.lstm.weight_ih_l0), _p(rnn.lstm.weight_hh_l0),
                    _p(rnn.lstm.bias_ih_l0), _p(rnn.lstm.bias_hh_l0), _p(rnn.fc2.weight), _p(rnn.fc2.bias))


BEAM_CHUNK = 8      # decode steps enqueued between two looks at the device-side "unfinished" counter


def _beam_loop(run, B, W, max_length, dev, return_scores):
    """Drive a beam-search entry point in chunks of steps; stop once the device reports that all B*W hypotheses
    produced <EOS> in one step (the reference's ``if done.all(): break``; the device froze the state at that step, so
    the steps of the chunk that were enqueued past it changed nothing).  One 4-byte device-to-host read per chunk."""
    out_ids = torch.empty((B, max_length), dtype=torch.int64, device=dev)
    out_len = torch.empty((B,), dtype=torch.int32, device=dev)
    scores = torch.empty((B, W), dtype=torch.float32, device=dev)
    left = torch.ones(1, dtype=torch.int32, device=dev)
    t = 0
    while t < max_length:
        t1 = min(max_length, t + BEAM_CHUNK)
        run(t, t1, out_ids, out_len, scores, left)
        t = t1
        if t < max_length and int(left.item()) == 0:
            break
    if return_scores:
        return out_ids.cpu(), out_len.cpu(), scores.cpu()
    return out_ids.cpu(), out_len.cpu()


def _tf_beam_exact(lib, ref, enc, W, max_length, sos, eos, return_scores):
    B, S, _ = enc.shape
    dev = enc.device
    nbytes = lib.capdec_tf_beam_workspace_bytes(ctypes.byref(ref.c), B, W, S, max_length)
    if nbytes == 0:
        raise RuntimeError("capdec: unsupported decoder configuration (see capdec.h)")
    ws = _workspace(nbytes, dev)

    def run(t0, t1, out_ids, out_len, scores, left):
        _lib.check(lib.capdec_tf_beam_search(ctypes.byref(ref.c), enc.data_ptr(), B, W, S, max_length, sos, eos, t0, t1,
                                             ws.data_ptr(), ws.numel(), out_ids.data_ptr(), out_len.data_ptr(),
                                             scores.data_ptr(), left.data_ptr(), _stream()), "capdec_tf_beam_search")

    return _beam_loop(run, B, W, max_length, dev, return_scores)


def tf_beam_ids(dec, enc, beam_width, max_length, sos, eos, return_scores=False, exact=False):
    """Beam search of a DecoderTransformer (reference transformer.py:177-244).  enc [B,S,E] fp32 on the GPU, one
    memory per sample.  Returns CPU (ids int64 [B,max_length] padded with -1, lengths int32 [B][, scores [B,W]]):
    the best hypothesis of every sample, without <SOS>, cut at its first <EOS>.

    By default the KV-cached form runs first; the samples it reports (capdec.h: those whose <PAD> hypotheses it could not
    follow - the reference lets a <PAD> query row attend uniformly over the current length) are then redone with the
    full-prefix form, so the captions are the reference's in every case.  ``exact=True`` uses the full-prefix form for
    every sample.  (The reference leaves its loop when ALL hypotheses of the batch produce <EOS> in one step; a redone
    subset is a batch of its own in that respect: its loop can end earlier than the full batch's did, which changes the
    scores and, only where two hypotheses tie exactly at that step, could change which of them is returned.)"""
    lib = _lib.load()
    if not enc.is_cuda or enc.dtype != torch.float32:
        raise RuntimeError("capdec: enc must be an fp32 tensor on the GPU (HIP); there is no CPU path")
    enc = enc.contiguous()
    B, S, _ = enc.shape
    ref = TfModelRef(dec)
    W = int(beam_width)
    if not 1 <= W <= 8:
        raise RuntimeError("capdec: beam_width must be between 1 and 8")
    if max_length > ref.max_pos:
        raise RuntimeError(f"capdec: max_length {max_length} exceeds the positional-encoding buffer ({ref.max_pos})")
    if B == 0:
        empty = (torch.empty((0, max_length), dtype=torch.int64), torch.empty((0,), dtype=torch.int32))
        return empty + (torch.empty((0, W), dtype=torch.float32),) if return_scores else empty
    dev = enc.device
    with torch.cuda.device(dev):
        if exact:
            return _tf_beam_exact(lib, ref, enc, W, max_length, sos, eos, return_scores)
        nbytes = lib.capdec_tf_beam_cached_workspace_bytes(ctypes.byref(ref.c), B, W, S, max_length)
        if nbytes == 0:
            raise RuntimeError("capdec: unsupported decoder configuration (see capdec.h)")
        ws = _workspace(nbytes, dev)
        pad_seen = torch.zeros((B,), dtype=torch.int32, device=dev)

        def run(t0, t1, out_ids, out_len, scores, left):
            _lib.check(lib.capdec_tf_beam_search_cached(ctypes.byref(ref.c), enc.data_ptr(), B, W, S, max_length, sos, eos,
                                                        t0, t1, ws.data_ptr(), ws.numel(), out_ids.data_ptr(),
                                                        out_len.data_ptr(), scores.data_ptr(), left.data_ptr(),
                                                        pad_seen.data_ptr(), _stream()), "capdec_tf_beam_search_cached")

        res = _beam_loop(run, B, W, max_length, dev, True)
        rows = torch.nonzero(pad_seen.cpu(), as_tuple=False).flatten()
        if rows.numel():
            sub = enc.index_select(0, rows.to(dev)).contiguous()
            fix = _tf_beam_exact(lib, ref, sub, W, max_length, sos, eos, True)
            for dst, src in zip(res, fix):
                dst[rows] = src
        return res if return_scores else res[:2]


def rnn_beam_ids(rnn, x0, beam_width, max_length, sos, eos, return_scores=False):
    """Beam search of the CNN-RNN decoder (reference cnnrnn_model.py:112-209).  x0 [B,E]: BatchNorm'ed image
    projection.  Returns as ``tf_beam_ids``."""
    lib = _lib.load()
    if not x0.is_cuda or x0.dtype != torch.float32:
        raise RuntimeError("capdec: x0 must be an fp32 tensor on the GPU (HIP); there is no CPU path")
    x0 = x0.contiguous()
    B, E = x0.shape
    m = _rnn_model(rnn, E)
    W = int(beam_width)
    if not 1 <= W <= 8:
        raise RuntimeError("capdec: beam_width must be between 1 and 8")
    if B == 0:
        empty = (torch.empty((0, max_length), dtype=torch.int64), torch.empty((0,), dtype=torch.int32))
        return empty + (torch.empty((0, W), dtype=torch.float32),) if return_scores else empty
    dev = x0.device
    with torch.cuda.device(dev):
        nbytes = lib.capdec_rnn_beam_workspace_bytes(ctypes.byref(m), B, W, max_length)
        if nbytes == 0:
            raise RuntimeError("capdec: unsupported decoder configuration (see capdec.h)")
        ws = _workspace(nbytes, dev)

        def run(t0, t1, out_ids, out_len, scores, left):
            _lib.check(lib.capdec_rnn_beam_search(ctypes.byref(m), x0.data_ptr(), B, W, max_length, sos, eos, t0, t1,
                                                  ws.data_ptr(), ws.numel(), out_ids.data_ptr(), out_len.data_ptr(),
                                                  scores.data_ptr(), left.data_ptr(), _stream()), "capdec_rnn_beam_search")

        return _beam_loop(run, B, W, max_length, dev, return_scores)


def ids_to_lists(ids, lengths):
    ids = ids.tolist()
    return [row[:n] for row, n in zip(ids, lengths.tolist())]
