"""``Plan``: the input and state trajectories that solve an MPC problem.

Drop-in for the reference's ``qpmpc.Plan`` (qpmpc/plan.py:18-109): built from a
problem and a QP solution object exposing ``found`` and ``x``; ``inputs`` is
``x.reshape(N, nu)``; ``states`` is rolled out lazily (by the HIP rollout kernel
through ``MPCProblem.integrate``) and memoised; everything is ``None`` for an
empty plan.
"""
from __future__ import annotations

import logging
from typing import Any, Optional

import numpy as np

from .mpc_problem import MPCProblem


class Solution:
    """Attribute-compatible with ``qpsolvers.Solution`` for what Plan and users read.

    ``x`` primal solution, ``z`` multipliers of ``G x <= h``, ``found``, ``obj``,
    ``extras`` = {"status", "iters"} as reported by the kernel.
    """

    def __init__(self, problem: Any, x=None, z=None, found: bool = False, obj=None, extras=None):
        self.problem = problem
        self.x = x
        self.y = None
        self.z = z
        self.z_box = None
        self.found = found
        self.obj = obj
        self.extras = extras if extras is not None else {}


class Plan:
    """State and input trajectories that optimize an MPC problem."""

    problem: MPCProblem

    def __init__(self, problem: MPCProblem, qpsol):
        self.problem = problem
        self.qpsol = qpsol
        self._inputs: Optional[np.ndarray] = None
        self._states: Optional[np.ndarray] = None
        if qpsol.found:
            self._inputs = np.asarray(qpsol.x).reshape((problem.nb_timesteps, problem.input_dim))

    @property
    def is_empty(self) -> bool:
        """True when the solver found no solution."""
        return self._inputs is None

    @property
    def first_input(self) -> Optional[np.ndarray]:
        """u_0, the input a receding-horizon controller applies; ``None`` if empty."""
        return None if self._inputs is None else self._inputs[0]

    @property
    def inputs(self) -> Optional[np.ndarray]:
        """Stacked inputs (N, nu), or ``None`` if the plan is empty."""
        return self._inputs

    @property
    def states(self) -> Optional[np.ndarray]:
        """Stacked states (N+1, nx), computed on first access; ``None`` if empty."""
        if self._inputs is None:
            return None
        if self._states is None:
            x_init = self.problem.initial_state
            if x_init is None:
                logging.warning("Problem has undefined initial state")
                return None
            pre = getattr(self, "_precomputed_rollout", None)  # (x0, X) from the fused single-problem path
            if pre is not None and np.array_equal(pre[0], np.asarray(x_init, dtype=float).ravel()):
                self._states = pre[1]
            else:
                self._states = self.problem.integrate(x_init, self._inputs)
        return self._states
