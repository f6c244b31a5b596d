"""CPU oracle for the doubly-contrastive segmentation train step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / CPU baseline.
The product (``doubly-contrastive-semseg_amd/``) never imports this package
and fails loudly when its HIP library is missing.

Parity pin: the oracle is checked against golden vectors produced by running
the reference itself in the build container (``tests/golden/make_golden.py``,
fixtures under ``tests/golden/*.npz``).  The reference ships no tests or
fixtures of its own.
"""
