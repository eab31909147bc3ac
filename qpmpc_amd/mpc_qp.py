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
        N, nx = mpc_problem.nb_timesteps, mpc_problem.state_dim
        self._sparse = sparse
        self._upd = None  # staging buffers of the q / h updates, created on first use
        from .single import condense_single

        fast = condense_single(mpc_problem)  # one upload, one launch (+ Phi), one download
        if fast is not None:
            (Phi_all, Psi_all, P, G, q, h), self._ctx = fast
            self._rows = np.arange(G.shape[0])
        else:  # ragged per-step row counts: the batch containers pad them
            bp = BatchMPCProblem.from_problems([mpc_problem])
            dev = BatchMPCQP(bp, keep_propagators=True)
            self._rows = bp.valid_rows
            import torch

            parts = [dev.Phi_all[0], dev.Psi_all[0], dev.P[0], dev.G[0], dev.q[0], dev.h[0]]
            flat = torch.cat([t.reshape(-1) for t in parts]).cpu().numpy()
            views, o = [], 0
            for t in parts:
                views.append(flat[o: o + t.numel()].reshape(tuple(t.shape)))
                o += t.numel()
            Phi_all, Psi_all, P, G, q, h = views
            self._ctx = {"out": dev, "in": bp, "cp": bp.c_problem(), "dims": bp.dims(), "Phi": dev.Phi_all.data_ptr(),
                         "Psi": dev.Psi_all.data_ptr(), "nx": nx, "N": N, "n": bp.nb_variables,
                         "m": bp.nb_constraints, "device": bp.device}
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

    def _update_vectors(self, mpc_problem: MPCProblem):
        """q and h for the problem's current x0 / goal / targets in ONE round trip: the three states are
        packed into one pinned buffer (one upload), ``mpcqp_update_vectors_batch`` computes both vectors in
        one launch from the Phi/Psi kept on the device, and ``[q | h]`` comes back in one copy. The result is
        memoised on the states, so ``update_cost_vector`` followed by ``update_constraint_vector`` for the same
        problem -- the reference's usage, doc/src/developer-notes.rst:10 -- costs a single launch."""
        import ctypes as C

        import torch

        from . import _capi

        if mpc_problem.initial_state is None:
            raise ProblemDefinitionError("initial state is undefined")
        ctx = self._ctx
        nx, N, n, m = ctx["nx"], ctx["N"], ctx["n"], ctx["m"]
        u = self._upd
        if u is None:
            u = self._upd = {}
            # Pinned host buffers that the kernel reads and writes DIRECTLY (ROCm maps pinned host memory
            # into the device's address space at the same address): a q / h update is one launch and one
            # stream synchronisation, no copy commands -- 2 N nx + n + m doubles cross PCIe either way.
            u["h_in"] = torch.empty((2 * nx + N * nx,), dtype=torch.float64).pin_memory()
            u["h_out"] = torch.empty((n + m,), dtype=torch.float64).pin_memory()
            u["key"] = None
            base = u["h_in"].data_ptr()
            cp = ctx["cp"]
            cp.x0 = _capi.Operand(base, 0, 0)
            u["goal_op"] = _capi.Operand(base + 8 * nx, 0, 0)
            u["tgt_op"] = _capi.Operand(base + 16 * nx, 0, 0)
            u["cp"] = cp
        x0 = np.asarray(mpc_problem.initial_state, dtype=np.float64).ravel()
        goal, tgt = mpc_problem.goal_state, mpc_problem.target_states
        wt, wx = mpc_problem.terminal_cost_weight, mpc_problem.stage_state_cost_weight
        key = (x0.tobytes(), None if goal is None else np.asarray(goal, dtype=np.float64).tobytes(),
               None if tgt is None else np.asarray(tgt, dtype=np.float64).tobytes(), wt, wx)
        if u["key"] == key:
            return u["q"], u["h"]
        buf = u["h_in"].numpy()
        buf[:nx] = x0
        if goal is not None:
            buf[nx:2 * nx] = np.asarray(goal, dtype=np.float64).ravel()
        if tgt is not None:
            buf[2 * nx:] = np.asarray(tgt, dtype=np.float64).ravel()
        # cost flags as BatchMPCProblem.cost_flags (mpc_qp.py:119-122, mpc_problem.py:141-166)
        flags = (_capi.P_TERMINAL if wt is not None else 0) | (_capi.P_STAGE if wx is not None else 0)
        t_on, s_on = wt is not None and wt > 1e-10, wx is not None and wx > 1e-10
        if not (t_on and goal is None):
            if t_on:
                flags |= _capi.Q_TERMINAL
            if s_on and tgt is not None:
                flags |= _capi.Q_STAGE
        dims = ctx["dims"]
        dims.flags = flags
        dims.w_terminal = 0.0 if wt is None else float(wt)
        dims.w_stage = 0.0 if wx is None else float(wx)
        cp = u["cp"]
        cp.goal = u["goal_op"] if goal is not None else _capi.Operand(None, 0, 0)
        cp.targets = u["tgt_op"] if tgt is not None else _capi.Operand(None, 0, 0)
        stream, raw_stream = _capi.current_stream()
        out = u["h_out"].data_ptr()
        rc = _capi.load().mpcqp_update_vectors_batch(
            C.byref(dims), C.byref(cp), ctx["Phi"], 0, ctx["Psi"], 0, 1, out,
            (out + 8 * n) if m else None, C.c_void_p(raw_stream))
        _capi.check(rc, "mpcqp_update_vectors_batch")
        stream.synchronize()
        res = u["h_out"].numpy()
        u["q"], u["h"], u["key"] = res[:n].copy(), res[n:].copy(), key
        return u["q"], u["h"]

    def update_cost_vector(self, mpc_problem: MPCProblem) -> None:
        """Recompute q for new x0 / goal / targets (mpc_qp.py:129-149)."""
        q, _ = self._update_vectors(mpc_problem)
        self.q[:] = q
        # the reference raises AFTER accumulating the terminal term when targets
        # are missing (mpc_qp.py:145 -> mpc_problem.py:161-165); same here
        mpc_problem.has_terminal_cost
        mpc_problem.has_stage_state_cost

    def update_constraint_vector(self, mpc_problem: MPCProblem) -> None:
        """Recompute h = e - C Phi x0 for a new x0 (mpc_qp.py:151-163)."""
        if mpc_problem.initial_state is None:
            raise ProblemDefinitionError("initial state is undefined")
        if self.C is not None:
            _, h = self._update_vectors(mpc_problem)
            self.h = h[self._rows].copy()
