"""``MPCQP``: an MPC problem condensed into a dense QP -- computed on the GPU.

Drop-in for the reference's ``qpmpc.MPCQP`` (qpmpc/mpc_qp.py:21-163): same
constructor, same attributes (``P q G h Phi Psi phi_last psi_last e C``), same
``problem`` property and ``update_*`` methods. The arithmetic of
``__init__`` (mpc_qp.py:53-114) runs in ``mpcqp_condense_batch`` with a batch
of one; the two update methods run in ``mpcqp_update_vectors_batch``. Arrays
are copied back to NumPy because that is what callers of the reference expect;
the batched, device-resident form is :class:`qpmpc_amd.batch.BatchMPCQP`.

Reference behaviours kept: P includes a cost term when its
weight ``is not None`` while q needs ``> 1e-10``; an undefined goal/targets
leaves q zero / partially filled instead of raising; outputs are float64.
Deliberate deviation: when every ``C_k`` is ``None`` the reference's ``C`` is a
meaningless object array and ``update_constraint_vector`` crashes; here ``C`` is
``None`` and the update leaves ``h = e``.
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np

from .batch import BatchMPCProblem, BatchMPCQP
from .exceptions import ProblemDefinitionError
from .mpc_problem import MPCProblem


class QPData:
    """Minimal stand-in for ``qpsolvers.Problem`` (P, q, G, h; no equalities)."""

    def __init__(self, P, q, G, h):
        self.P, self.q, self.G, self.h = P, q, G, h
        self.A = self.b = self.lb = self.ub = None

    def unpack(self):
        return self.P, self.q, self.G, self.h, None, None, None, None


def _block_diag(blocks):
    rows = sum(b.shape[0] for b in blocks)
    cols = sum(b.shape[1] for b in blocks)
    out = np.zeros((rows, cols))
    r = c = 0
    for b in blocks:
        out[r: r + b.shape[0], c: c + b.shape[1]] = b
        r += b.shape[0]
        c += b.shape[1]
    return out


class MPCQP:
    """MPC problem represented as a quadratic program (built by HIP kernels)."""

    G: np.ndarray
    P: np.ndarray
    Phi: np.ndarray
    Psi: np.ndarray
    h: np.ndarray
    phi_last: np.ndarray
    psi_last: np.ndarray
    q: np.ndarray
    e: np.ndarray
    C: Optional[np.ndarray]

    def __init__(self, mpc_problem: MPCProblem, sparse: bool = False) -> None:
        if mpc_problem.initial_state is None:
            raise ProblemDefinitionError("initial state is undefined")
        bp = BatchMPCProblem.from_problems([mpc_problem])
        self._batch_problem = bp
        self._dev = BatchMPCQP(bp, keep_propagators=True)
        self._rows = bp.valid_rows
        self._sparse = sparse
        N, nx = mpc_problem.nb_timesteps, mpc_problem.state_dim
        # ONE device-to-host copy for everything the reference exposes as NumPy attributes
        import torch

        d = self._dev
        parts = [d.Phi_all[0], d.Psi_all[0], d.P[0], d.G[0], d.q[0], d.h[0]]
        flat = torch.cat([t.reshape(-1) for t in parts]).cpu().numpy()
        views, o = [], 0
        for t in parts:
            views.append(flat[o: o + t.numel()].reshape(tuple(t.shape)))
            o += t.numel()
        Phi_all, Psi_all, P, G, q, h = views
        self.Phi, self.phi_last = Phi_all[: N * nx], Phi_all[N * nx:]
        self.Psi, self.psi_last = Psi_all[: N * nx], Psi_all[N * nx:]
        G = G[self._rows]
        if sparse:  # mpc_qp.py:108-109
            from scipy.sparse import csc_matrix

            P, G = csc_matrix(P), csc_matrix(G)
        self.P, self.G = P, G
        self.q = q.copy()
        self.h = h[self._rows].copy()
        self.e = np.hstack(
            [np.asarray(mpc_problem.get_ineq_vector(k), dtype=float).ravel() for k in range(N)]
        )
        C_list = [mpc_problem.get_ineq_state_matrix(k) for k in range(N)]
        if all(c is None for c in C_list):
            self.C = None
        else:
            mks = [len(np.asarray(mpc_problem.get_ineq_vector(k)).ravel()) for k in range(N)]
            self.C = _block_diag(
                [np.zeros((mks[k], nx)) if c is None else np.asarray(c, dtype=float).reshape(mks[k], nx)
                 for k, c in enumerate(C_list)]
            )
        # mpc_qp.py:79-85: x0 violates a state-only constraint at k = 0
        m0 = len(np.asarray(mpc_problem.get_ineq_vector(0)).ravel())
        if mpc_problem.get_ineq_input_matrix(0) is None and m0 and np.any(self.h[:m0] < 0.0):
            logging.warning(
                "initial state is unfeasible: "
                f"G_0 * x <= h_0 with G_0 == 0 and min(h_0) == {min(self.h[:m0])}"
            )

    @property
    def problem(self):
        """(P, q, G, h) for a QP solver: ``qpsolvers.Problem`` when that package
        is installed (mpc_qp.py:124-127), else an attribute-compatible ``QPData``."""
        try:
            import qpsolvers

            return qpsolvers.Problem(self.P, self.q, self.G, self.h)
        except ImportError:
            return QPData(self.P, self.q, self.G, self.h)

    def _refresh_states(self, mpc_problem: MPCProblem) -> None:
        if mpc_problem.initial_state is None:
            raise ProblemDefinitionError("initial state is undefined")
        bp = self._batch_problem
        bp.update_initial_state(mpc_problem.initial_state)
        bp.terminal_cost_weight = mpc_problem.terminal_cost_weight
        bp.stage_state_cost_weight = mpc_problem.stage_state_cost_weight
        bp.goal_state = None
        bp.target_states = None
        if mpc_problem.goal_state is not None:
            bp.update_goal_state(mpc_problem.goal_state)
        if mpc_problem.target_states is not None:
            bp.update_target_states(mpc_problem.target_states)

    def update_cost_vector(self, mpc_problem: MPCProblem) -> None:
        """Recompute q for new x0 / goal / targets (mpc_qp.py:129-149)."""
        self._refresh_states(mpc_problem)
        self._dev.update_cost_vector(self._batch_problem)
        self.q[:] = self._dev.q[0].cpu().numpy()
        # the reference raises AFTER accumulating the terminal term when targets
        # are missing (mpc_qp.py:145 -> mpc_problem.py:161-165); same here
        mpc_problem.has_terminal_cost
        mpc_problem.has_stage_state_cost

    def update_constraint_vector(self, mpc_problem: MPCProblem) -> None:
        """Recompute h = e - C Phi x0 for a new x0 (mpc_qp.py:151-163)."""
        if mpc_problem.initial_state is None:
            raise ProblemDefinitionError("initial state is undefined")
        if self.C is not None:
            self._refresh_states(mpc_problem)
            self._dev.update_constraint_vector(self._batch_problem)
            self.h = self._dev.h[0].cpu().numpy()[self._rows]
