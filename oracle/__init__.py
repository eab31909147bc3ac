"""CPU oracle for the qpmpc hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``qpmpc_amd``) never does.

* ``oracle.condense_np``  NumPy restatement of qpmpc/mpc_qp.py (pinned by
  tests/golden, captured from the real reference).
* ``oracle.mpc_oracle.c`` plain-C restatement of the same build plus the
  published Goldfarb-Idnani dual active-set method that the reference reaches
  through qpsolvers/quadprog. Parity at the solver boundary is UNPINNED by the
  reference (no numeric golden solution exists upstream); see the C header.
"""
from .capi import (  # noqa: F401
    build,
    build_solve_batch,
    condense_one,
    gi_solve,
    rollout_one,
    solve_mpc_like_reference,
    solve_workload,
)
from .condense_np import condense, constraint_vector, cost_vector, integrate  # noqa: F401
