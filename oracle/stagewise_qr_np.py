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
* The projected vector z goes through the forward sweep, which returns the step in the inputs and G z: the primal
  point and the slacks are carried explicitly (u -= t z_u, s += t G z_u). No per-row V_a / h_a storage.
* A leaving row deletes its column of R; Givens rotations on rows of R / vectors of Q restore the triangle.
* The point is carried in whitened coordinates as well (v = y0 - sum t z). Acceptance evaluates it FROM SCRATCH: one forward
  sweep of v from x0 (closed loop: stable whatever the spectrum of A, unlike a roll-out of rounded inputs) gives the inputs
  that are returned and their rows; while an active row sits off its bound, a polish step (z = Q R^-T rho,
  lam -= R^-1 R^-T rho).

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
                       evaluate: str = "closed"):
    """(U [N*nu], lam [N*mk], status, iters): status 0 solved, 1 iteration limit, 2 infeasible, 3 P not PD."""
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
    y0 = ric.backward(qlin, np.zeros((N, nu), dt), pN=pN)
    U, X = ric.forward(y0, x0=sp.x0.astype(dt))
    v = y0.reshape(-1).copy()          # the point in whitened coordinates: u = forward(v, x0)
    gmul = lambda Xv, Uv: np.einsum("kri,ki->kr", C, Xv[:N]) + np.einsum("kri,ki->kr", D, Uv)  # noqa: E731
    s = e - gmul(X, U)
    selectable = sp.e < 1e29
    tolh = (dt(tol) * (1 + np.abs(e))).astype(dt)
    invn = _row_inv_norms(sp).astype(dt)
    dep = dt(1e-10 if f32 else 1e-26)   # |z|^2 <= dep |y|^2: the row depends on the active ones

    act, lam = [], []
    Q = np.zeros((0, n), dt)            # vectors of Q by rows
    R = np.zeros((0, 0), dt)
    active = np.zeros((N, mk), bool)
    iters, status = 0, 1

    def row_y(kp, rp):
        ql = np.zeros((N, nx), dt)
        rl = np.zeros((N, nu), dt)
        ql[kp] = -C[kp, rp]
        rl[kp] = -D[kp, rp]
        return ric.backward(ql, rl, ktop=kp).reshape(-1)

    def move(z, t):
        """the step along the whitened vector z: u -= t z_u, s += t G z_u (active rows stay on their bounds)"""
        nonlocal U, s, v
        Zu, Zx = ric.forward(z.reshape(N, nu))
        U = U - dt(t) * Zu
        v = v - dt(t) * z
        s = np.where(active, dt(0), s + dt(t) * gmul(Zx, Zu))

    def drop(l):
        nonlocal Q, R
        k = len(act) - 1
        R = np.delete(R, l, axis=1)          # rows 0..k, columns 0..k-1: upper Hessenberg behind column l
        for j in range(l, k):
            a, b = R[j, j], R[j + 1, j]
            hh = dt(np.hypot(a, b))
            c, sn = (a / hh, b / hh) if hh > 0 else (dt(1), dt(0))
            rj, rj1 = R[j].copy(), R[j + 1].copy()
            R[j], R[j + 1] = c * rj + sn * rj1, c * rj1 - sn * rj
            qj, qj1 = Q[j].copy(), Q[j + 1].copy()
            Q[j], Q[j + 1] = c * qj + sn * qj1, c * qj1 - sn * qj
        R = R[:k]
        Q = Q[:k]
        active[act[l]] = False
        del act[l], lam[l]

    for rnd in range(4):
        fail = False
        while True:
            viol = selectable & ~active & (s < -tolh)
            if not viol.any():
                status = 0
                break
            score = np.where(viol, s * invn, np.inf)
            kp, rp = np.unravel_index(np.argmin(score), score.shape)
            y = row_y(kp, rp)
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
                t2 = -s[kp, rp] / zz if can_move else np.inf
                t = min(t1, t2)
                if not np.isfinite(t):
                    return np.zeros(n), np.zeros(N * mk), 2, iters
                full = t2 <= t1
                if can_move:
                    if full:
                        active[kp, rp] = True  # (lands on its bound exactly)
                    move(z, t)
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
                    lam.append(up)
                    added = True
                else:
                    drop(l)
        # ---- acceptance on a roll-out of the inputs through the original dynamics; polish while an active row is off its bound
        vpass = 3
        for vp in range(vpass):
            if evaluate == "closed":   # the point from scratch through the closed-loop sweep (what the kernels do)
                U, Xr = ric.forward(v.reshape(N, nu), x0=sp.x0.astype(dt))
            else:                      # a roll-out of the carried inputs through the original dynamics
                Xr = _rollout(sp, U, dt)
            sr = e - gmul(Xr, U)
            rho = np.array([sr[a] for a in act], dt)
            offa = any(not (lv >= 0) for lv in lam)
            trig = (8 if f32 else 10) if vp < vpass - 1 else (16 if f32 else max(100.0, 1e-7 / tol))
            lim = np.array([trig * tolh[a] for a in act])
            if f32:  # plus what a float32 evaluation cannot resolve
                lim = lim + 16 * 6e-8 * np.array([np.abs(C[a[0], a[1]]) @ np.abs(Xr[a[0]]) + np.abs(D[a[0], a[1]]) @ np.abs(U[a[0]]) + abs(e[a]) for a in act])
            if len(act):
                offa = offa or bool((np.abs(rho) > lim).any())
            if stats is not None:
                stats.setdefault("rho", []).append(float(np.abs(rho).max()) if len(act) else 0.0)
            if not offa:
                break
            if vp == vpass - 1:
                fail = True
                break
            # polish: S dlam = rho with S = R'R; the point moves along z = Q R^-T rho (whitened), t = 1 in the loop's sign
            nq = len(act)
            w = np.zeros(nq, dt)
            for i in range(nq):
                w[i] = (rho[i] - R[:i, i] @ w[:i]) / R[i, i]
            dl = np.zeros(nq, dt)
            for b in range(nq - 1, -1, -1):
                dl[b] = (w[b] - R[b, b + 1:] @ dl[b + 1:]) / R[b, b]
            Zu, Zx = ric.forward((w @ Q).reshape(N, nu))
            U = U + Zu                      # s = e - G u must drop by rho on the active rows: G_A dU = rho
            v = v + w @ Q
            for a in range(nq):
                lam[a] = max(lam[a] - dl[a], dt(0))
        if fail:
            status = 1
            break
        s = np.where(active, dt(0), sr)
        if not (selectable & ~active & (sr < -4 * tolh)).any():
            status = 0
            break
        status = 1  # an inactive row came out violated: continue from the re-evaluated slacks
    lam_full = np.zeros((N, mk))
    for a, (k, r_) in enumerate(act):
        lam_full[k, r_] = lam[a]
    if status != 0:
        return np.zeros(n), np.zeros(N * mk), status, iters
    return U.astype(np.float64).reshape(-1), lam_full.reshape(-1), status, iters
