"""Host container for a linear time-variant MPC problem.

Drop-in for the reference's ``qpmpc.MPCProblem`` (qpmpc/mpc_problem.py:16-335):
same constructor signature, attributes, ``get_*(k)`` accessors, ``update_*``
setters with their dimension checks, ``has_*`` properties and ``integrate``.
It holds NumPy arrays exactly as the caller passed them; device tensors are
made by :mod:`qpmpc_amd.batch` when a kernel is launched.

Behaviours kept on purpose:
  * the ``target_states`` constructor argument is accepted and ignored
    (mpc_problem.py:129,136-139) -- use ``update_target_states``;
  * a field is time-varying iff it is a ``list`` (mpc_problem.py:177-245);
  * setters flatten their argument (mpc_problem.py:261,277,295).
"""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np

from .exceptions import ProblemDefinitionError, StateError

MatrixOrList = Union[np.ndarray, List[np.ndarray]]
OptMatrixOrList = Union[None, np.ndarray, List[np.ndarray]]

_COST_EPS = 1e-10  # weights at or below this do not enter q (mpc_problem.py:146,159)

# Error texts are part of the boundary (callers and tests match on them): identical to
# the reference's (mpc_problem.py:104-111,146-165,256-294), pinned by tests/golden.
_MSG = {
    "weight": "stage non-negative control weight needed for regularization",
    "no_cost": "either terminal or stage state cost should be set",
    "goal": "MPC problem has terminal cost but the goal state is undefined",
    "targets": "MPC problem has a stage state cost but the reference trajectory is undefined",
}
_DIM_TAIL = "does not match state dimension ({nx})"


def _at(field, k: int):
    """Per-step view of a field: lists are indexed, anything else is shared."""
    return field[k] if isinstance(field, list) else field


def _ncols(field) -> int:
    first = field if isinstance(field, np.ndarray) else field[0]
    return first.shape[1]


class MPCProblem:
    """x_{k+1} = A_k x_k + B_k u_k,  C_k x_k + D_k u_k <= e_k,  k = 0..N-1.

    Cost: ``w_t |x_N - goal|^2 + w_x sum_k |x_k - target_k|^2 + w_u sum_k |u_k|^2``.
    """

    def __init__(self, transition_state_matrix: MatrixOrList, transition_input_matrix: MatrixOrList,
                 ineq_state_matrix: OptMatrixOrList, ineq_input_matrix: OptMatrixOrList,
                 ineq_vector: MatrixOrList, nb_timesteps: int, terminal_cost_weight: Optional[float],
                 stage_state_cost_weight: Optional[float], stage_input_cost_weight: float,
                 initial_state: Optional[np.ndarray] = None, goal_state: Optional[np.ndarray] = None,
                 target_states: Optional[np.ndarray] = None) -> None:
        # w_u > 0 makes P >= w_u I positive definite: the QP has ONE minimiser,
        # which is what solver-independent parity rests on (mpc_problem.py:104-107).
        if stage_input_cost_weight <= 0.0:
            raise ProblemDefinitionError(_MSG["weight"])
        if terminal_cost_weight is None and stage_state_cost_weight is None:
            raise ProblemDefinitionError(_MSG["no_cost"])
        self.transition_state_matrix = transition_state_matrix
        self.transition_input_matrix = transition_input_matrix
        self.ineq_state_matrix = ineq_state_matrix
        self.ineq_input_matrix = ineq_input_matrix
        self.ineq_vector = ineq_vector
        self.nb_timesteps = nb_timesteps
        self.state_dim = _ncols(transition_state_matrix)
        self.input_dim = _ncols(transition_input_matrix)
        self.terminal_cost_weight = terminal_cost_weight
        self.stage_state_cost_weight = stage_state_cost_weight
        self.stage_input_cost_weight = stage_input_cost_weight
        self.goal_state = None
        self.initial_state = None
        self.target_states = None  # constructor argument deliberately unused (quirk 1)
        del target_states
        if goal_state is not None:
            self.update_goal_state(goal_state)
        if initial_state is not None:
            self.update_initial_state(initial_state)

    # ------------------------------------------------------------ cost flags
    @property
    def has_terminal_cost(self) -> bool:
        """True when the terminal term enters q; raises if it should but no goal is set."""
        w = self.terminal_cost_weight
        active = w is not None and w > _COST_EPS
        if active and self.goal_state is None:
            raise ProblemDefinitionError(_MSG["goal"])
        return active

    @property
    def has_stage_state_cost(self) -> bool:
        """True when the stage term enters q; raises if it should but no targets are set."""
        w = self.stage_state_cost_weight
        active = w is not None and w > _COST_EPS
        if active and self.target_states is None:
            raise ProblemDefinitionError(_MSG["targets"])
        return active

    # -------------------------------------------------------------- accessors
    def get_transition_state_matrix(self, k) -> np.ndarray:
        """A_k."""
        return _at(self.transition_state_matrix, k)

    def get_transition_input_matrix(self, k) -> np.ndarray:
        """B_k."""
        return _at(self.transition_input_matrix, k)

    def get_ineq_state_matrix(self, k):
        """C_k, or None for the null matrix."""
        return _at(self.ineq_state_matrix, k)

    def get_ineq_input_matrix(self, k):
        """D_k, or None for the null matrix."""
        return _at(self.ineq_input_matrix, k)

    def get_ineq_vector(self, k) -> np.ndarray:
        """e_k."""
        return _at(self.ineq_vector, k)

    # ---------------------------------------------------------------- setters
    def _checked(self, value: np.ndarray, size: int, template: str) -> np.ndarray:
        """Flattened copy of a state-like array, or ``StateError`` if its size is wrong."""
        if value.size != size:
            raise StateError(template.format(shape=value.shape, nx=self.state_dim, N=self.nb_timesteps, size=size))
        return value.flatten()

    def update_goal_state(self, goal_state: np.ndarray) -> None:
        """Set x_goal; ``StateError`` unless it has ``state_dim`` entries."""
        self.goal_state = self._checked(goal_state, self.state_dim, "goal state of shape {shape} " + _DIM_TAIL)

    def update_initial_state(self, initial_state: np.ndarray) -> None:
        """Set x_0; ``StateError`` unless it has ``state_dim`` entries."""
        self.initial_state = self._checked(initial_state, self.state_dim, "Initial state of shape {shape} " + _DIM_TAIL)

    def update_target_states(self, target_states: np.ndarray) -> None:
        """Set the N stacked stage targets; ``StateError`` on a size mismatch."""
        self.target_states = self._checked(
            target_states, self.state_dim * self.nb_timesteps,
            "Reference state trajectory of shape {shape} does not match nb_timesteps * state dimension = "
            "{N} * {nx} = {size}")

    def __repr__(self) -> str:
        names = (
            "goal_state", "ineq_input_matrix", "ineq_state_matrix", "ineq_vector",
            "initial_state", "input_dim", "nb_timesteps", "stage_input_cost_weight",
            "stage_state_cost_weight", "state_dim", "terminal_cost_weight",
            "transition_input_matrix", "transition_state_matrix",
        )
        return "MPCProblem(" + ", ".join(f"{a}={getattr(self, a)}" for a in names) + ")"

    # ---------------------------------------------------------------- rollout
    def integrate(self, initial_state: np.ndarray, inputs: np.ndarray) -> np.ndarray:
        """State trajectory X (N+1, nx) under ``inputs`` (N, nu) from ``initial_state``.

        Replaces the Python loop of mpc_problem.py:316-335 by the HIP rollout
        kernel (``mpcqp_rollout_batch``); raises ``BackendError`` without a GPU.
        """
        from .batch import rollout_single

        return rollout_single(self, initial_state, inputs)
