"""MI355X-native batched linear MPC: a drop-in for the hot path of qpmpc.

Same public names as the reference (qpmpc/__init__.py:9-21) -- ``MPCProblem``,
``MPCQP``, ``Plan``, ``solve_mpc`` -- plus the batched entry points that are the
reason this package exists: ``BatchMPCProblem``, ``solve_mpc_batch``,
``BatchMPCQP``, ``solve_qp_batch``, ``rollout_batch``.

All arithmetic runs in hand-written HIP kernels behind the C ABI of
``include/mpcqp.h``; without the compiled library or without a GPU every
compute entry point raises ``BackendError`` (no CPU fallback).
"""
from .batch import (  # noqa: F401
    BatchMPCProblem,
    BatchMPCQP,
    BatchPlan,
    PreparedModelSolve,
    PreparedSolve,
    WarmState,
    SharedModel,
    rollout_batch,
    pairing_order,
    solve_mpc_batch,
    solve_qp_batch,
)
from .exceptions import (  # noqa: F401
    BackendError,
    PlanError,
    ProblemDefinitionError,
    QPMPCException,
    StateError,
)
from .mpc_problem import MPCProblem  # noqa: F401
from .mpc_qp import MPCQP  # noqa: F401
from .plan import Plan, Solution  # noqa: F401
from .solve_mpc import available_solvers, solve_mpc  # noqa: F401

__all__ = [
    "MPCProblem",
    "MPCQP",
    "Plan",
    "solve_mpc",
    "BatchMPCProblem",
    "BatchMPCQP",
    "BatchPlan",
    "PreparedSolve",
    "WarmState",
    "SharedModel",
    "PreparedModelSolve",
    "pairing_order",
    "solve_mpc_batch",
    "solve_qp_batch",
    "rollout_batch",
]

__version__ = "0.1.0"
REFERENCE_VERSION = "3.1.0"  # qpmpc release whose API surface is mirrored
