"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's hot path (3-D multi-scale deformable
attention and the Python glue around it).  Nothing under ``transoar_amd/`` may
import this package: it is the *checker* used by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.

Parity status: PINNED.  Both restatements are checked against golden vectors
produced by importing the reference's own Python path
(``transoar/models/ops/functions/ms_deform_attn_func.py:41-65``) in the build
container -- see ``tests/golden/make_golden.py`` and ``tests/test_oracle.py``.
"""
