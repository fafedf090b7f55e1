"""The allocation-free vocabulary arg-max (capdec_vocab_argmax_static, fixed device array inside the library) against
the two-call form with a caller workspace (capdec_vocab_argmax_partial + capdec_argmax_finalize), through the C ABI.
Bit-exact: both run the same kernels.  Two calls in a row, the second with a smaller input, because all calls on a GPU
share one array that keeps the previous call's contents."""
import ctypes

import pytest
import torch

from capdec import _lib

pytestmark = pytest.mark.gpu


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _two_call(lib, x, W, b):
    M, K = x.shape
    V = W.shape[0]
    n = lib.capdec_vocab_partial_elems(M, V)
    pv = torch.empty(n, dtype=torch.float32, device=x.device)
    pi = torch.empty(n, dtype=torch.int32, device=x.device)
    idx = torch.empty(M, dtype=torch.int32, device=x.device)
    val = torch.empty(M, dtype=torch.float32, device=x.device)
    _lib.check(lib.capdec_vocab_argmax_partial(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), M, V, K, pv.data_ptr(),
                                               pi.data_ptr(), _stream()), "capdec_vocab_argmax_partial")
    _lib.check(lib.capdec_argmax_finalize(pv.data_ptr(), pi.data_ptr(), M, lib.capdec_vocab_tiles(V), idx.data_ptr(),
                                          val.data_ptr(), _stream()), "capdec_argmax_finalize")
    return idx, val


def _static(lib, x, W, b, idx, val):
    M, K = x.shape
    return lib.capdec_vocab_argmax_static(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), M, W.shape[0], K, idx.data_ptr(),
                                          val.data_ptr() if val is not None else None, _stream())


def test_static_vocab_argmax_matches_the_workspace_form_on_two_calls_in_a_row():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    # (rows, vocabulary, depth): the benchmark's decode step first, then smaller and ragged ones that reuse the array
    cases = [(1024, 10000, 512), (200, 5000, 512), (33, 777, 96), (1, 64, 16), (640, 10000, 512)]
    want = []
    inputs = []
    for M, V, K in cases:
        x = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(V, K, generator=g) / K ** 0.5).to(dev)
        b = (0.1 * torch.randn(V, generator=g)).to(dev)
        W[V // 3] = W[V // 2]                       # an exact tie between two vocabulary rows: lowest index must win
        b[V // 3] = b[V // 2]
        inputs.append((x, W, b))
        want.append(_two_call(lib, x, W, b))
    # outputs are allocated before the calls; the calls themselves are back to back on one stream
    outs = [(torch.full((M,), -7, dtype=torch.int32, device=dev), torch.zeros(M, dtype=torch.float32, device=dev))
            for M, _, _ in cases]
    for (x, W, b), (idx, val) in zip(inputs, outs):
        _lib.check(_static(lib, x, W, b, idx, val), "capdec_vocab_argmax_static")
    torch.cuda.synchronize()
    for n, ((idx, val), (widx, wval)) in enumerate(zip(outs, want)):
        assert torch.equal(idx, widx), cases[n]
        assert torch.equal(val, wval), cases[n]
    # and against plain arithmetic, so that the pair of entry points is not merely self-consistent
    x, W, b = inputs[1]
    ref = (x.double() @ W.double().t() + b.double()).cpu()
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * ref.abs().max()
    assert torch.equal(outs[1][0].cpu().long()[clear], ref.argmax(dim=1)[clear])
    # value output is optional
    idx2 = torch.empty(cases[2][0], dtype=torch.int32, device=dev)
    _lib.check(_static(lib, *inputs[2], idx2, None), "capdec_vocab_argmax_static")
    assert torch.equal(idx2, want[2][0])


def test_static_vocab_argmax_rejects_sizes_beyond_its_array():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M, V, K = 8192, 10000, 16
    assert lib.capdec_vocab_partial_elems(M, V) > _lib.VOCAB_STATIC_MAX_ELEMS
    x = torch.zeros(M, K, device=dev)
    W = torch.zeros(V, K, device=dev)
    b = torch.zeros(V, device=dev)
    idx = torch.full((M,), -7, dtype=torch.int32, device=dev)
    rc = _static(lib, x, W, b, idx, None)
    assert rc == _lib.E2BIG
    assert b"capacity" in lib.capdec_strerror(rc)
    torch.cuda.synchronize()
    assert bool((idx == -7).all())          # nothing was launched
    # the largest size that fits still runs
    M = _lib.VOCAB_STATIC_MAX_ELEMS // (64 * lib.capdec_vocab_tiles(V)) * 64
    assert lib.capdec_vocab_partial_elems(M, V) <= _lib.VOCAB_STATIC_MAX_ELEMS
    x = torch.randn(M, K, device=dev)
    W = torch.randn(V, K, device=dev)
    idx = torch.empty(M, dtype=torch.int32, device=dev)
    val = torch.empty(M, dtype=torch.float32, device=dev)
    _lib.check(_static(lib, x, W, b, idx, val), "capdec_vocab_argmax_static")
    widx, wval = _two_call(lib, x, W, b)
    assert torch.equal(idx, widx) and torch.equal(val, wval)


@pytest.mark.parametrize("M", [1024, 32])
def test_static_vocab_argmax_under_stream_capture(M):
    """What the entry point exists for: after one call outside capture (it resolves the array's address on the device)
    the call is captured into a HIP graph - an allocation inside the captured region would end the capture with an
    error - and replayed with changed inputs; every replay must equal the two-call form on the inputs of that replay."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    V, K = 5000, 512
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(V, K, generator=g) / K ** 0.5).to(dev)
    b = (0.1 * torch.randn(V, generator=g)).to(dev)
    idx = torch.full((M,), -7, dtype=torch.int32, device=dev)
    val = torch.zeros(M, dtype=torch.float32, device=dev)
    _lib.check(_static(lib, x, W, b, idx, val), "capdec_vocab_argmax_static")          # warm-up outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _lib.check(_static(lib, x, W, b, idx, val), "capdec_vocab_argmax_static (captured)")
    for seed in (1, 2):
        g2 = torch.Generator().manual_seed(100 * M + seed)
        x.copy_(torch.randn(M, K, generator=g2))
        idx.fill_(-7)
        graph.replay()
        torch.cuda.synchronize()
        widx, wval = _two_call(lib, x, W, b)
        assert torch.equal(idx, widx) and torch.equal(val, wval), seed
