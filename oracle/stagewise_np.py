"""ORACLE (test infrastructure): stage-wise (uncondensed) solve of the MPC QP in O(N) per iteration.

The reference has no sparse formulation: ``sparse=True`` only wraps the dense condensed matrices in
CSC (qpmpc/mpc_qp.py:39,108-109; qpmpc/solve_mpc.py:31-32), so its cost grows as O(N^3) and the
condensed Hessian of a long horizon does not even fit. This module restates, in NumPy, the
stage-wise method that ``qpmpc_amd/csrc/mpcqp_stage.hip`` implements, and is pinned
on the dense path: same problem -> same minimiser (the QP is strictly convex, mpc_problem.py:104-107).

Problem (what MPCQP condenses, mpc_qp.py:53-149; cost = (J - const)/2 of doc/src/index.rst:15):

    min  1/2 sum_{k<N} w_u |u_k|^2 + 1/2 sum_{1<=k<N} w_x |x_k - xref_k|^2 + 1/2 w_t |x_N - x_goal|^2
    s.t. x_{k+1} = A_k x_k + B_k u_k,   C_k x_k + D_k u_k <= e_k  (k = 0..N-1)

Method: Goldfarb-Idnani's dual active set in the metric of the condensed Hessian P, with P never
formed: a product v -> P^-1 v is one LQR solve (Riccati gains computed once per problem, then one
backward and one forward sweep over the horizon), a row of G applied to a vector is a read of that
vector's state trajectory. Per active row a the method keeps V_a = P^-1 g_a' and its trajectory
X_a = Psi V_a, and the inverse W = (G_A P^-1 G_A')^-1 of the small Gram matrix (bordered / deflated by
rank-one updates). One iteration = one LQR solve + O(|A| N) vector work.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np


class StageProblem:
    """Per-step operands as arrays [N, ...] (LTI fields broadcast), states and weights."""

    def __init__(self, A, B, C, D, e, x0, goal, targets, wt, wx, wu):
        self.A, self.B, self.C, self.D, self.e = A, B, C, D, e
        self.x0, self.goal, self.targets = x0, goal, targets
        self.wt, self.wx, self.wu = wt, wx, wu
        self.N, self.nx, self.nu = A.shape[0], A.shape[1], B.shape[2]
        self.mk = e.shape[1]


def from_mpc_problem(p) -> StageProblem:
    """Stack the getters of an ``MPCProblem`` (mpc_problem.py:168-245); absent C_k / D_k become zero rows."""
    N, nx, nu = p.nb_timesteps, p.state_dim, p.input_dim
    A = np.stack([np.asarray(p.get_transition_state_matrix(k), dtype=float).reshape(nx, nx) for k in range(N)])
    B = np.stack([np.asarray(p.get_transition_input_matrix(k), dtype=float).reshape(nx, nu) for k in range(N)])
    mks = [len(np.asarray(p.get_ineq_vector(k)).ravel()) for k in range(N)]
    mk = max(mks)
    C = np.zeros((N, mk, nx))
    D = np.zeros((N, mk, nu))
    e = np.full((N, mk), 1e30)
    for k in range(N):
        e[k, : mks[k]] = np.asarray(p.get_ineq_vector(k), dtype=float).ravel()
        Ck, Dk = p.get_ineq_state_matrix(k), p.get_ineq_input_matrix(k)
        if Ck is not None:
            C[k, : mks[k]] = np.asarray(Ck, dtype=float).reshape(mks[k], nx)
        if Dk is not None:
            D[k, : mks[k]] = np.asarray(Dk, dtype=float).reshape(mks[k], nu)
    wt, wx = p.terminal_cost_weight, p.stage_state_cost_weight
    # q follows mpc_problem.py:141-166 (term present only above 1e-10 and with its state defined);
    # P follows "weight is not None" (mpc_qp.py:102,104)
    goal = None if p.goal_state is None else np.asarray(p.goal_state, dtype=float)
    tg = None if p.target_states is None else np.asarray(p.target_states, dtype=float).reshape(N, nx)
    return StageProblem(A, B, C, D, e, np.asarray(p.initial_state, dtype=float), goal, tg,
                        wt, wx, float(p.stage_input_cost_weight))


class Riccati:
    """Backward Riccati recursion of the LQR problem whose Hessian (in U) is the condensed P:
    stage cost 1/2 w_u |u|^2 + 1/2 Q_k |x|^2 with Q_k = w_x I (1 <= k < N, if P has the stage term),
    terminal Q_N = w_t I (if P has the terminal term). ``solve`` applies P^-1."""

    def __init__(self, sp: StageProblem):
        N, nx, nu = sp.N, sp.nx, sp.nu
        self.sp = sp
        qt = 0.0 if sp.wt is None else sp.wt
        qs = 0.0 if sp.wx is None else sp.wx
        P = qt * np.eye(nx)
        self.Acl = np.zeros((N, nx, nx))
        self.K = np.zeros((N, nu, nx))
        self.Sinv = np.zeros((N, nu, nu))
        for k in range(N - 1, -1, -1):
            A, B = sp.A[k], sp.B[k]
            S = sp.wu * np.eye(nu) + B.T @ P @ B
            Sinv = np.linalg.inv(S)
            K = Sinv @ (B.T @ P @ A)
            Acl = A - B @ K
            self.Acl[k], self.K[k], self.Sinv[k] = Acl, K, Sinv
            Qk = qs * np.eye(nx) if k >= 1 else np.zeros((nx, nx))  # x_0 is data: its cost is a constant
            Pn = Qk + A.T @ P @ Acl
            P = 0.5 * (Pn + Pn.T)

    def solve(self, qlin, rlin, x0=None, pN=None) -> Tuple[np.ndarray, np.ndarray]:
        """argmin_U of  1/2 U'PU-like LQR cost + sum_k (qlin_k . x_k + rlin_k . u_k) + pN . x_N  from x0:
        returns (U [N, nu], X [N+1, nx]). With x0 = 0 and pN = 0 this is U = -P^-1 (G-row-like vector)."""
        sp = self.sp
        N, nx, nu = sp.N, sp.nx, sp.nu
        p = np.zeros(nx) if pN is None else pN.copy()
        ff = np.zeros((N, nu))
        for k in range(N - 1, -1, -1):
            t = sp.B[k].T @ p + rlin[k]
            ff[k] = -self.Sinv[k] @ t
            p = qlin[k] + self.Acl[k].T @ p - self.K[k].T @ rlin[k]
        X = np.zeros((N + 1, nx))
        U = np.zeros((N, nu))
        if x0 is not None:
            X[0] = x0
        for k in range(N):
            U[k] = -self.K[k] @ X[k] + ff[k]
            X[k + 1] = sp.A[k] @ X[k] + sp.B[k] @ U[k]
        return U, X


def solve_stagewise(sp: StageProblem, max_iter: int = 10000, tol: float = 1e-12):
    """(U [N*nu], lam [N*mk], status, iters): status 0 solved, 1 iteration limit, 2 infeasible."""
    N, nx, nu, mk = sp.N, sp.nx, sp.nu, sp.mk
    m = N * mk
    ric = Riccati(sp)
    # unconstrained minimiser: tracking terms as linear costs (q of mpc_qp.py:129-149)
    t_on = sp.wt is not None and sp.wt > 1e-10 and sp.goal is not None
    s_on = sp.wx is not None and sp.wx > 1e-10 and sp.targets is not None
    qlin = np.zeros((N, nx))
    if s_on:
        qlin[1:] = -sp.wx * sp.targets[1:]
    if sp.wt is not None and sp.wt > 1e-10 and sp.goal is None:
        qlin[:] = 0.0  # mpc_qp.py:119-122: the exception leaves q untouched (zero)
        s_on = False
    pN = -sp.wt * sp.goal if t_on else np.zeros(nx)
    # the quadratic state terms of J enter P only when their weight "is not None" -- already in Riccati
    U0, X0 = ric.solve(qlin, np.zeros((N, nu)), x0=sp.x0, pN=pN)
    # when a tracking term is off in q but on in P (weight set, state undefined) the cost still pulls to 0
    s = sp.e - np.einsum("kri,ki->kr", sp.C, X0[:N]) - np.einsum("kri,ki->kr", sp.D, U0)  # [N, mk]
    selectable = sp.e < 1e29  # padded rows (C = D = 0, e = 1e30) can never be active
    tolh = tol * (1.0 + np.abs(sp.e))  # a row is violated when its slack is below -tol (1 + |e_i|)

    act: List[Tuple[int, int]] = []   # (k, r) of every active row, in slot order
    V: List[np.ndarray] = []          # V_a = P^-1 g_a'   [N, nu]
    XV: List[np.ndarray] = []         # its trajectory    [N+1, nx]
    lam: List[float] = []
    W = np.zeros((0, 0))
    gdot = lambda k, r, Uv, Xv: float(sp.C[k, r] @ Xv[k] + sp.D[k, r] @ Uv[k])  # noqa: E731
    invn = _row_inv_norms(sp)
    iters = 0
    status = 1
    n = N * nu
    while True:
        viol = selectable & (s < -tolh)
        for (k, r) in act:
            viol[k, r] = False
        if not viol.any():
            status = 0
            break
        score = np.where(viol, s * invn, np.inf)
        kp, rp = np.unravel_index(np.argmin(score), score.shape)
        up = 0.0
        added = False
        while not added:
            if iters >= max_iter:
                return _finish(sp, U0, V, lam, act, 1, iters)
            iters += 1
            ql = np.zeros((N, nx))
            rl = np.zeros((N, nu))
            ql[kp] = -sp.C[kp, rp]
            rl[kp] = -sp.D[kp, rp]
            Vp, Xp = ric.solve(ql, rl)  # = P^-1 g_p'
            c = np.array([gdot(k, r, Vp, Xp) for (k, r) in act])
            dpp = gdot(kp, rp, Vp, Xp)
            r_ = W @ c if len(act) else np.zeros(0)
            d2 = dpp - float(c @ r_) if len(act) else dpp
            if len(act) and not d2 > 1e-3 * dpp:
                # A row that looks nearly dependent on the active ones is judged on a refined r: one step of iterative
                # refinement with the Gram matrix of the active rows (entry (a, b) = g_a . V_b). |z|^2 = dpp - c' W c is a
                # difference of two nearly equal numbers there, and W is an explicit inverse kept by rank-one updates: without
                # this a feasible, nearly fully active problem (tools/stress_general.py, seed 7) came back 'infeasible'
                # after 389 iterations; with it, solved in 519 like the dense oracle (515). mpcqp_stageg.hip does the same.
                Gm = np.array([[gdot(ka_, ra_, V[b_], XV[b_]) for b_ in range(len(act))] for (ka_, ra_) in act])
                r_ = r_ + W @ (c - Gm @ r_)
                d2 = dpp - float(c @ r_)
            can_move = len(act) < n and d2 > 1e-13 * dpp and d2 > 0.0
            cand = [a for a in range(len(act)) if r_[a] > 0.0]
            t1, l = (np.inf, -1)
            for a in cand:
                if lam[a] / r_[a] < t1:
                    t1, l = lam[a] / r_[a], a
            t2 = -s[kp, rp] / d2 if can_move else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return _finish(sp, U0, V, lam, act, 2, iters)
            # z = -(V_p - sum r_a V_a): primal step; slacks move by -t g_i . z
            Zu = -Vp
            Zx = -Xp
            for a in range(len(act)):
                Zu = Zu + r_[a] * V[a]
                Zx = Zx + r_[a] * XV[a]
            gz = np.einsum("kri,ki->kr", sp.C, Zx[:N]) + np.einsum("kri,ki->kr", sp.D, Zu)
            s = s - t * gz
            for a in range(len(act)):
                lam[a] = max(lam[a] - t * r_[a], 0.0)
                s[act[a]] = 0.0
            up += t
            if t2 <= t1:  # full step: p becomes active
                q = len(act)
                Wn = np.zeros((q + 1, q + 1))
                Wn[:q, :q] = W + np.outer(r_, r_) / d2
                Wn[:q, q] = -r_ / d2
                Wn[q, :q] = -r_ / d2
                Wn[q, q] = 1.0 / d2
                W = Wn
                act.append((kp, rp))
                V.append(Vp)
                XV.append(Xp)
                lam.append(up)
                s[kp, rp] = 0.0
                added = True
            else:  # partial step: slot l leaves
                w_l = W[:, l].copy()
                W = W - np.outer(w_l, w_l) / W[l, l]
                keep = [a for a in range(len(act)) if a != l]
                W = W[np.ix_(keep, keep)]
                for lst in (act, V, XV, lam):
                    del lst[l]
    return _finish(sp, U0, V, lam, act, status, iters)


def _row_inv_norms(sp: StageProblem) -> np.ndarray:
    """Selection metric: the violated row farthest from its hyperplane in the Euclidean norm of its own
    stage block [C_k[r] D_k[r]] (the dense kernels use the P^-1 metric, which costs one more O(N) recursion
    here and only changes the iteration count, never the minimiser)."""
    nn = np.sqrt((sp.C ** 2).sum(axis=2) + (sp.D ** 2).sum(axis=2))
    return np.where(nn > 0.0, 1.0 / np.where(nn > 0.0, nn, 1.0), 1.0)


def _finish(sp, U0, V, lam, act, status, iters):
    U = U0.copy()
    for a in range(len(act)):
        U = U - lam[a] * V[a]
    lam_full = np.zeros((sp.N, sp.mk))
    for a, (k, r) in enumerate(act):
        lam_full[k, r] = lam[a]
    if status != 0:
        return np.zeros(sp.N * sp.nu), np.zeros(sp.N * sp.mk), status, iters
    return U.reshape(-1), lam_full.reshape(-1), status, iters


def kkt_residuals_stagewise(sp: StageProblem, U: np.ndarray, lam: np.ndarray):
    """KKT residuals of (U, lam) WITHOUT the condensed matrices (usable at any horizon): stationarity
    through the adjoint recursion, primal / dual feasibility, complementarity."""
    N, nx, nu, mk = sp.N, sp.nx, sp.nu, sp.mk
    U = U.reshape(N, nu)
    lam = lam.reshape(N, mk)
    X = np.zeros((N + 1, nx))
    X[0] = sp.x0
    for k in range(N):
        X[k + 1] = sp.A[k] @ X[k] + sp.B[k] @ U[k]
    t_on = sp.wt is not None and sp.wt > 1e-10 and sp.goal is not None
    s_on = sp.wx is not None and sp.wx > 1e-10 and sp.targets is not None
    qt = 0.0 if sp.wt is None else sp.wt
    qs = 0.0 if sp.wx is None else sp.wx
    # costate: mu_N = dJ/dx_N ; mu_k = dJ/dx_k + A_k' mu_{k+1} + C_k' lam_k
    mu = qt * X[N] - (sp.wt * sp.goal if t_on else 0.0)
    stat = 0.0
    for k in range(N - 1, -1, -1):
        gu = sp.wu * U[k] + sp.B[k].T @ mu + sp.D[k].T @ lam[k]
        stat = max(stat, float(np.abs(gu).max()))
        dx = (qs * X[k] - (sp.wx * sp.targets[k] if s_on else 0.0)) if k >= 1 else np.zeros(nx)
        mu = dx + sp.A[k].T @ mu + sp.C[k].T @ lam[k]
    slack = sp.e - np.einsum("kri,ki->kr", sp.C, X[:N]) - np.einsum("kri,ki->kr", sp.D, U)
    real = sp.e < 1e29
    prim = float(np.maximum(-slack[real], 0.0).max()) if real.any() else 0.0
    dual = float(np.maximum(-lam, 0.0).max())
    comp = float(np.abs(lam[real] * slack[real]).max()) if real.any() else 0.0
    return {"stationarity": stat, "primal": prim, "dual": dual, "complementarity": comp}
