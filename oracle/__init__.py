"""CPU oracle for the caption-decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under this package is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker.  The product path
(``image-captioning_amd/``) never imports this package and fails loudly when
its HIP library is missing.

Parity status: PINNED.  Every function in ``decoder_ref`` is checked in
``tests/test_oracle_golden.py`` against vectors in ``tests/golden/`` that were
produced by running the reference's own classes (imported from the reference
checkout) through ``tools/gen_golden.py``.
"""
