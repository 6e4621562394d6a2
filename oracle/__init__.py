"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the reference's gate-application path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import, link or execute it, and only as the checker / the CPU
baseline — never as the thing measured or shipped.  The product (``quantum_computations_amd``) must not
import this package; ``tests/test_no_oracle_in_product.py`` enforces that.

Parity status: **pinned**.  Every function here is checked against golden vectors produced by importing the
reference (``/root/reference/simulators``) in the build container with ``tests/golden/generate_golden.py``;
the vectors are committed under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.
Exceptions (no reference counterpart exists): Fock-basis squeezing / beam-splitter matrices
and n>3 Grover are "parity unpinned"; only the contraction that applies them is pinned.
"""
