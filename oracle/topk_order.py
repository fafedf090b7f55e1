"""TEST INFRASTRUCTURE -- not part of the product path.

Order in which ``torch.topk(values, k)`` (largest, sorted) returns EQUAL values on the CPU, for short rows.

Why this exists.  The reference's beam search selects the best W of W*W candidates with ``new_scores.topk(beam_width)``
(transformer.py:211, cnnrnn_model.py:172).  A hypothesis that has just produced <EOS> contributes W candidates with
exactly the same score (transformer.py:194-196 adds ``(1 - done) * log_prob``), so exact ties occur at every such step
and which of the tied candidates survive decides the later course of the search.  ``torch.topk`` does not define an
order among equal values; what the reference *does* on the CPU is what ATen's kernel does there:
``std::nth_element(begin, begin + k - 1, end, greater)`` followed by ``std::sort(begin, begin + k - 1, greater)`` over
(value, index) pairs whenever ``k * 64 > n`` (ATen/native/TopKImpl.h, ``topk_impl_loop``), with GNU libstdc++'s
introselect (median-of-three pivot moved to the front, unguarded Hoare partition, insertion sort below four elements,
heap select when the depth limit ``2 * floor(log2 n)`` runs out).  This module restates that procedure so that the
HIP selection kernel can be checked against ``torch.topk`` itself on rows full of ties.

Pinned by tests/test_oracle_golden.py::test_topk_order_matches_torch (thousands of random rows with ties).
"""
import math
from typing import List, Sequence, Tuple


def _gt(x: Tuple[float, int], y: Tuple[float, int]) -> bool:
    """ATen's comparator for ``largest=True``: NaN sorts first, otherwise strictly greater value."""
    xn, yn = x[0] != x[0], y[0] != y[0]
    return (xn and not yn) or (x[0] > y[0])


def _median_to_first(q, r, a, b, c):
    if _gt(q[a], q[b]):
        if _gt(q[b], q[c]):
            m = b
        elif _gt(q[a], q[c]):
            m = c
        else:
            m = a
    elif _gt(q[a], q[c]):
        m = a
    elif _gt(q[b], q[c]):
        m = c
    else:
        m = b
    q[r], q[m] = q[m], q[r]


def _partition_pivot(q, first, last):
    mid = first + (last - first) // 2
    _median_to_first(q, first, first + 1, mid, last - 1)
    lo, hi, piv = first + 1, last, first
    while True:
        while _gt(q[lo], q[piv]):
            lo += 1
        hi -= 1
        while _gt(q[piv], q[hi]):
            hi -= 1
        if not lo < hi:
            return lo
        q[lo], q[hi] = q[hi], q[lo]
        lo += 1


def _insertion_sort(q, first, last):
    for i in range(first + 1, last):
        val = q[i]
        if _gt(val, q[first]):
            q[first + 1:i + 1] = q[first:i]
            q[first] = val
        else:
            j = i
            while _gt(val, q[j - 1]):
                q[j] = q[j - 1]
                j -= 1
            q[j] = val


def _adjust_heap(q, first, hole, length, value):
    top = hole
    child = hole
    while child < (length - 1) // 2:
        child = 2 * (child + 1)
        if _gt(q[first + child], q[first + child - 1]):
            child -= 1
        q[first + hole] = q[first + child]
        hole = child
    if length % 2 == 0 and child == (length - 2) // 2:
        child = 2 * (child + 1)
        q[first + hole] = q[first + child - 1]
        hole = child - 1
    parent = (hole - 1) // 2
    while hole > top and _gt(q[first + parent], value):
        q[first + hole] = q[first + parent]
        hole = parent
        parent = (hole - 1) // 2
    q[first + hole] = value


def _heap_select(q, first, middle, last):
    length = middle - first
    if length >= 2:
        parent = (length - 2) // 2
        while True:
            _adjust_heap(q, first, parent, length, q[first + parent])
            if parent == 0:
                break
            parent -= 1
    for i in range(middle, last):
        if _gt(q[i], q[first]):
            value = q[i]
            q[i] = q[first]
            _adjust_heap(q, first, 0, length, value)


def topk_cpu_order(values: Sequence[float], k: int) -> Tuple[List[float], List[int]]:
    """(values, indices) exactly as ``torch.topk(torch.tensor(values), k)`` returns them on the CPU when
    ``k * 64 > len(values)`` (the nth_element branch), including the order among equal values."""
    n = len(values)
    assert 1 <= k <= n and k * 64 > n
    q = [(float(v), i) for i, v in enumerate(values)]
    first, last, nth = 0, n, k - 1
    depth = 2 * int(math.floor(math.log2(n))) if n > 0 else 0
    done = False
    while last - first > 3:
        if depth == 0:
            _heap_select(q, first, nth + 1, last)
            q[first], q[nth] = q[nth], q[first]
            done = True
            break
        depth -= 1
        cut = _partition_pivot(q, first, last)
        if cut <= nth:
            first = cut
        else:
            last = cut
    if not done:
        _insertion_sort(q, first, last)
    _insertion_sort(q, 0, k - 1)          # std::sort of fewer than 16 elements is one insertion sort
    assert k - 1 <= 16
    return [v for v, _ in q[:k]], [i for _, i in q[:k]]
