"""MI355X-native batched linear MPC: a drop-in for the hot path of qpmpc.

Same public names as the reference (qpmpc/__init__.py:9-21) plus the batched
entry points that are the reason this package exists.
"""
from .exceptions import (  # noqa: F401
    BackendError,
    PlanError,
    ProblemDefinitionError,
    QPMPCException,
    StateError,
)
from .mpc_problem import MPCProblem  # noqa: F401

__version__ = "0.1.0"
REFERENCE_VERSION = "3.1.0"  # qpmpc release whose API surface is mirrored
