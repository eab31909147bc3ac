"""Error taxonomy of the boundary (same names as reference qpmpc/exceptions.py:10-23).

Definition errors are raised on the host *before* any kernel launch; a solver
that does not converge is not an exception (it yields an empty ``Plan``).
"""


class QPMPCException(Exception):
    """Root of every exception raised by this package."""


class ProblemDefinitionError(QPMPCException):
    """The MPC problem is ill-defined (missing state, bad weights, ...)."""


class PlanError(QPMPCException):
    """A plan is inconsistent with its problem."""


class StateError(QPMPCException):
    """A state vector does not have the dimension the problem expects."""


class BackendError(QPMPCException):
    """The HIP library is missing, failed to load, or a launch returned an error.

    There is no CPU fallback: the product path fails loudly instead.
    """
