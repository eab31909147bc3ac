"""ORACLE (test infrastructure): the stage-wise solve with a thin-QR active-set operator, as
``qpmpc_amd/csrc/mpcqp_stagew.hip`` / ``mpcqp_stage.hip`` run it since round 6.

Same problem and same outer method as ``oracle/stagewise_np.py`` (Goldfarb-Idnani's dual active set in the
metric of the condensed Hessian P of qpmpc/mpc_qp.py:99-105, P never formed). What differs is the operator
kept for the active rows. ``stagewise_np`` keeps V_a = P^-1 g_a' per active row and the explicit inverse
W = (G_A P^-1 G_A')^-1; |z|^2 = g_p V_p - c' W c is then a difference of nearly equal numbers on nearly fully
active problems. Here:

* The Riccati recursion is a block Cholesky factorisation of P: with S_k = w_u I + B_k' P_{k+1} B_k = Ls_k Ls_k'
  the backward sweep of a row g_a yields its WHITENED vector y_a (y_k = -Ls_k^-1 (B_k' p_{k+1} + r_k)), and
  g_a P^-1 g_b' = y_a . y_b. The kernels store the sweeps' per-step records in whitened coordinates
  (backward [Acl' ; -Ls^-1 B'], forward [[Acl, B Ls^-T], [-K, Ls^-T]]), so a sweep works on y directly.
* Thin QR of the active rows' whitened vectors, Y_A = Q R: a candidate is orthogonalised against Q (classical
  Gram-Schmidt, a second pass when the first one cancelled), |z|^2 is a SUM OF SQUARES, r = R^-1 Q' y_p.
* No per-row V_a / h_a storage and NO incremental slack update: the point is carried in whitened coordinates (v -= t z) and
  evaluated FROM SCRATCH when needed -- one forward sweep of v from x0 gives the inputs, every row's slack and the most
  violated row. In between the loop works on the rows whose whitened vectors the latest backward sweep cached (the most
  violated row and the next ones): such a row's slack gains t y_c . z with a step, a dot product; the most violated cached row
  is taken next (any violated row is a valid Goldfarb-Idnani choice), and when none is violated the point is evaluated again.
* A leaving row deletes its column of R; Givens rotations on rows of R / vectors of Q restore the triangle.
* Acceptance is such an evaluation (closed loop: stable whatever the spectrum of A, unlike a roll-out of rounded inputs)
  with no inactive row violated; while an active row sits off its bound, a polish step (z = Q R^-T rho,
  lam -= R^-1 R^-T rho) and the evaluation again.

``dtype=np.float32`` runs every operation in float32 (what config 5's instantiation computes).
"""
from __future__ import annotations

import numpy as np

from .stagewise_np import StageProblem, _row_inv_norms


class WhitenedRiccati:
    """Riccati recursion with the stage Hessians' Cholesky factors: records in whitened coordinates."""

    def __init__(self, sp: StageProblem, dt=np.float64):
        N, nx, nu = sp.N, sp.nx, sp.nu
        self.sp, self.dt = sp, dt
        A, B = sp.A.astype(dt), sp.B.astype(dt)
        qt = dt(0.0 if sp.wt is None else sp.wt)
        qs = dt(0.0 if sp.wx is None else sp.wx)
        wu = dt(sp.wu)
        P = qt * np.eye(nx, dtype=dt)
        self.Acl = np.zeros((N, nx, nx), dt)
        self.K = np.zeros((N, nu, nx), dt)
        self.Fw = np.zeros((N, nu, nx), dt)    # -Ls^-1 B'   (backward: y_k = Fw p_{k+1} - Ls^-1 r_k)
        self.Bw = np.zeros((N, nx, nu), dt)    # B Ls^-T     (forward: x+ = Acl x + Bw y)
        self.LiT = np.zeros((N, nu, nu), dt)   # Ls^-T       (forward: u = -K x + LiT y)
        self.pd = True
        for k in range(N - 1, -1, -1):
            S = wu * np.eye(nu, dtype=dt) + B[k].T @ P @ B[k]
            try:
                Ls = np.linalg.cholesky(S.astype(np.float64)).astype(dt)
            except np.linalg.LinAlgError:
                self.pd = False
                return
            Li = np.linalg.inv(Ls.astype(np.float64)).astype(dt)
            K = (Li.T @ (Li @ (B[k].T @ P @ A[k]))).astype(dt)
            Acl = A[k] - B[k] @ K
            self.Acl[k], self.K[k] = Acl, K
            self.Fw[k] = -(Li @ B[k].T)
            self.Bw[k] = B[k] @ Li.T
            self.LiT[k] = Li.T
            Qk = qs * np.eye(nx, dtype=dt) if k >= 1 else np.zeros((nx, nx), dt)
            Pn = Qk + A[k].T @ P @ Acl
            P = dt(0.5) * (Pn + Pn.T)

    def backward(self, qlin, rlin, pN=None, ktop=None):
        """whitened feed-forward terms y [N, nu] of the linear costs qlin [N, nx], rlin [N, nu] (+ pN on x_N)"""
        sp, dt = self.sp, self.dt
        N, nx, nu = sp.N, sp.nx, sp.nu
        p = np.zeros(nx, dt) if pN is None else pN.astype(dt)
        y = np.zeros((N, nu), dt)
        for k in range(N - 1 if ktop is None else ktop, -1, -1):
            y[k] = self.Fw[k] @ p - self.LiT[k].T @ rlin[k]
            p = qlin[k] + self.Acl[k].T @ p - self.K[k].T @ rlin[k]
        return y

    def forward(self, y, x0=None):
        """(U [N, nu], X [N+1, nx]) of the whitened feed-forward terms y from x0"""
        sp, dt = self.sp, self.dt
        N, nx, nu = sp.N, sp.nx, sp.nu
        X = np.zeros((N + 1, nx), dt)
        U = np.zeros((N, nu), dt)
        if x0 is not None:
            X[0] = x0
        for k in range(N):
            U[k] = -self.K[k] @ X[k] + self.LiT[k] @ y[k]
            X[k + 1] = self.Acl[k] @ X[k] + self.Bw[k] @ y[k]
        return U, X


def _rollout(sp, U, dt):
    N, nx = sp.N, sp.nx
    X = np.zeros((N + 1, nx), dt)
    X[0] = sp.x0
    A, B = sp.A.astype(dt), sp.B.astype(dt)
    for k in range(N):
        X[k + 1] = A[k] @ X[k] + B[k] @ U[k]
    return X


def solve_stagewise_qr(sp: StageProblem, max_iter: int = 10000, tol: float = 1e-12, dtype=np.float64, stats=None,
                       cached_rows: int = 8):
    """(U [N*nu], lam [N*mk], status, iters): status 0 solved, 1 iteration limit, 2 infeasible, 3 P not PD.

    ``cached_rows``: right-hand sides of a backward sweep (the kernels' R: 8 in the general constraint layout, 7 / 12 when the
    constraint matrices are fixed along the horizon). It shapes the iterates: between two evaluations the loop only takes rows whose
    whitened vectors the latest backward sweep cached."""
    dt = dtype
    f32 = dt == np.float32
    N, nx, nu, mk = sp.N, sp.nx, sp.nu, sp.mk
    n = N * nu
    ric = WhitenedRiccati(sp, dt)
    if not ric.pd:
        return np.zeros(n), np.zeros(N * mk), 3, 0
    C, D, e = sp.C.astype(dt), sp.D.astype(dt), sp.e.astype(dt)
    t_on = sp.wt is not None and sp.wt > 1e-10 and sp.goal is not None
    s_on = sp.wx is not None and sp.wx > 1e-10 and sp.targets is not None
    qlin = np.zeros((N, nx), dt)
    if s_on:
        qlin[1:] = -dt(sp.wx) * sp.targets[1:].astype(dt)
    pN = (-dt(sp.wt) * sp.goal).astype(dt) if t_on else np.zeros(nx, dt)
    gmul = lambda Xv, Uv: np.einsum("kri,ki->kr", C, Xv[:N]) + np.einsum("kri,ki->kr", D, Uv)  # noqa: E731
    selectable = sp.e < 1e29
    tolh = (dt(tol) * (1 + np.abs(e))).astype(dt)
    invn = _row_inv_norms(sp).astype(dt)
    dep = dt(1e-10 if f32 else 1e-26)   # |z|^2 <= dep |y|^2: the row depends on the active ones
    x0 = sp.x0.astype(dt)

    act, lam = [], []
    Q = np.zeros((0, n), dt)            # vectors of Q by rows
    R = np.zeros((0, 0), dt)
    active = np.zeros((N, mk), bool)
    iters = 0
    v = ric.backward(qlin, np.zeros((N, nu), dt), pN=pN).reshape(-1)  # the point in whitened coordinates: u = forward(v, x0)

    def evaluate():
        """the point from scratch: inputs, every row's slack (active rows: their residual)"""
        U, X = ric.forward(v.reshape(N, nu), x0=x0)
        return U, e - gmul(X, U)

    def row_y(kp, rp):
        ql = np.zeros((N, nx), dt)
        rl = np.zeros((N, nu), dt)
        ql[kp] = -C[kp, rp]
        rl[kp] = -D[kp, rp]
        return ric.backward(ql, rl, ktop=kp).reshape(-1)

    def drop(l):
        nonlocal Q, R
        k = len(act) - 1
        R = np.delete(R, l, axis=1)          # rows 0..k, columns 0..k-1: upper Hessenberg behind column l
        for j in range(l, k):
            a, b = R[j, j], R[j + 1, j]
            hh = dt(np.sqrt(a * a + b * b))
            c, sn = (a / hh, b / hh) if hh > 0 else (dt(1), dt(0))
            rj, rj1 = R[j].copy(), R[j + 1].copy()
            R[j], R[j + 1] = c * rj + sn * rj1, c * rj1 - sn * rj
            qj, qj1 = Q[j].copy(), Q[j + 1].copy()
            Q[j], Q[j + 1] = c * qj + sn * qj1, c * qj1 - sn * qj
        R = R[:k]
        Q = Q[:k]
        active[act[l]] = False
        del act[l], lam[l]

    U, s = evaluate()
    polish, vpass = 0, 3
    while True:
        viol = selectable & ~active & (s < -tolh)
        if not viol.any():
            # ---- no inactive row is violated: the point is the answer once its ACTIVE rows sit on their bounds
            rho = np.array([s[a] for a in act], dt)
            trig = (8 if f32 else 10) if polish < vpass - 1 else (16 if f32 else max(100.0, 1e-7 / tol))
            lim = np.array([trig * tolh[a] for a in act])
            if f32 and len(act):  # plus what a float32 evaluation cannot resolve
                xmax = max(float(np.abs(U).max()), 1.0)
                lim = lim + 16 * 6e-8 * np.array([abs(e[a]) + (np.abs(C[a[0], a[1]]).sum() + np.abs(D[a[0], a[1]]).sum()) * xmax for a in act])
            if stats is not None:
                stats.setdefault("rho", []).append(float(np.abs(rho).max()) if len(act) else 0.0)
            if not (len(act) and bool((np.abs(rho) > lim).any())):
                break
            polish += 1
            if polish >= vpass:
                return np.zeros(n), np.zeros(N * mk), 1, iters
            # polish: S dlam = rho with S = R'R; the point moves along Q R^-T rho (whitened)
            nq = len(act)
            w = np.zeros(nq, dt)
            for i in range(nq):
                w[i] = (rho[i] - R[:i, i] @ w[:i]) / R[i, i]
            dl = np.zeros(nq, dt)
            for b in range(nq - 1, -1, -1):
                dl[b] = (w[b] - R[b, b + 1:] @ dl[b + 1:]) / R[b, b]
            v = v + w @ Q
            for a in range(nq):
                lam[a] = max(lam[a] - dl[a], dt(0))
            U, s = evaluate()
            continue
        # ---- the most violated row and the next ones: their whitened vectors (one backward sweep), their slacks
        # (the most violated row first, then the inactive rows of smallest scaled slack, violated or not yet: the rows next in line)
        score = np.where(selectable & ~active, s * invn, np.inf).reshape(-1)
        first = int(np.argmin(np.where(viol, s * invn, np.inf).reshape(-1)))
        order = [first] + [int(i) for i in np.argsort(score, kind="stable") if int(i) != first]
        rows = [i for i in order[:cached_rows] if np.isfinite(score[i])]
        ys = [row_y(i // mk, i % mk) for i in rows]
        crs = [s.reshape(-1)[i] for i in rows]
        cact = [False] * len(rows)
        while True:
            # the most violated cached row (ties: the lowest cache slot)
            best, hit = np.inf, -1
            for j, i in enumerate(rows):
                if not cact[j] and crs[j] < -tolh.reshape(-1)[i] and crs[j] * invn.reshape(-1)[i] < best:
                    best, hit = crs[j] * invn.reshape(-1)[i], j
            if hit < 0:
                break
            kp, rp = rows[hit] // mk, rows[hit] % mk
            y = ys[hit]
            yy = dt(y @ y)
            up = dt(0)
            added = False
            while not added:
                if iters >= max_iter:
                    return np.zeros(n), np.zeros(N * mk), 1, iters
                iters += 1
                nq = len(act)
                d = Q @ y
                z = y - d @ Q
                zz = dt(z @ z)
                if nq and zz < dt(0.25) * yy:   # cancellation: once more ("twice is enough")
                    d2 = Q @ z
                    z = z - d2 @ Q
                    d = d + d2
                    zz = dt(z @ z)
                r = np.zeros(nq, dt)
                for b in range(nq - 1, -1, -1):
                    r[b] = (d[b] - R[b, b + 1:] @ r[b + 1:]) / R[b, b]
                can_move = nq < n and zz > dep * yy and zz > 0
                t1, l = np.inf, -1
                for a in range(nq):
                    if r[a] > 0 and lam[a] / r[a] < t1:
                        t1, l = lam[a] / r[a], a
                t2 = -crs[hit] / zz if can_move else np.inf
                t = min(t1, t2)
                if not np.isfinite(t):
                    return np.zeros(n), np.zeros(N * mk), 2, iters
                full = t2 <= t1
                if can_move:  # the point moves against z; a cached row's slack gains t y_c . z
                    v = v - dt(t) * z
                    for j in range(len(rows)):
                        if not cact[j]:
                            crs[j] = crs[j] + dt(t) * dt(ys[j] @ z)
                for a in range(nq):
                    lam[a] = max(lam[a] - dt(t) * r[a], dt(0))
                up = up + dt(t)
                if full:
                    zn = dt(np.sqrt(zz))
                    Q = np.vstack([Q, (z / zn)[None]])
                    Rn = np.zeros((nq + 1, nq + 1), dt)
                    Rn[:nq, :nq] = R
                    Rn[:nq, nq] = d
                    Rn[nq, nq] = zn
                    R = Rn
                    act.append((kp, rp))
                    active[kp, rp] = True
                    lam.append(up)
                    cact[hit], crs[hit] = True, dt(0)
                    added = True
                else:
                    rowl = act[l][0] * mk + act[l][1]
                    drop(l)
                    if rowl in rows:  # a cached row that leaves is tracked again, from its bound
                        j = rows.index(rowl)
                        cact[j], crs[j] = False, dt(0)
        U, s = evaluate()
    lam_full = np.zeros((N, mk))
    for a, (k, r_) in enumerate(act):
        lam_full[k, r_] = lam[a]
    return U.astype(np.float64).reshape(-1), lam_full.reshape(-1), 0, iters
