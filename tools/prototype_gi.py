#!/usr/bin/env python3
"""Development prototype (NOT product, NOT oracle): the wave-parallel dual
active-set variant implemented by qpmpc_amd/csrc, written with numpy so the
math can be checked against tests/golden before it is written in HIP.

Differences from textbook Goldfarb-Idnani (which the oracle restates):
  * works in y = L^T x coordinates (P = L L^T), so J is an orthogonal Q and the
    constraint matrix is M = G L^{-T};
  * adding a constraint uses ONE Householder reflector on Q2 (a rank-1 update,
    lane-parallel) instead of a chain of n-q Givens rotations;
  * keeps S = R^{-1} explicitly (r = S d1 is a mat-vec, no back-substitution);
    the new column of S on an add is [-r; 1] / R_qq;
  * dropping constraint l uses rotations whose coefficients come from prefix
    norms of row l of S (a scan), applied row-wise to S and Q.
"""
import glob
import os
import sys

import numpy as np


def solve(P, q, G, h, max_iter=1000, tol=1e-12, verbose=False):
    n, m = P.shape[0], G.shape[0]
    L = np.linalg.cholesky(P)
    M = np.linalg.solve(L, G.T).T  # M = G L^{-T}
    y = -np.linalg.solve(L, q)
    Q = np.eye(n)
    S = np.zeros((n, n))
    act = []  # active constraint indices, in order
    u = np.zeros(n)  # multipliers of active constraints (by position)
    where = -np.ones(m, dtype=int)
    iters = 0
    status = 1
    scale = 1.0 + np.abs(h)
    while iters < max_iter:
        s = h - M @ y
        key = np.where(where >= 0, np.inf, s / scale)
        p = int(np.argmin(key)) if m else -1
        if m == 0 or key[p] >= -tol:
            status = 0
            break
        sp = s[p]
        npv = -M[p]
        d = Q.T @ npv
        up = 0.0
        while True:
            iters += 1
            nq = len(act)
            d2 = d[nq:]
            delta2 = float(d2 @ d2)
            z = Q[:, nq:] @ d2
            r = S[:nq, :nq] @ d[:nq]
            t1, l = np.inf, -1
            for i in range(nq):
                if r[i] > 0 and u[i] / r[i] < t1:
                    t1, l = u[i] / r[i], i
            nrm2 = float(npv @ npv)
            if delta2 > 1e-28 * max(nrm2, 1e-300) and nq < n:
                t2 = -sp / delta2
            else:
                t2 = np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return None, None, 2, iters
            if np.isfinite(t2):
                y = y + t * z
            u[:nq] -= t * r
            up += t
            if t2 <= t1:
                # full step: add p with one Householder reflector on Q2
                delta = np.sqrt(delta2)
                dq = d[nq]
                sigma = 1.0 if dq >= 0 else -1.0
                v = d2.copy()
                v[0] += sigma * delta
                beta = 1.0 / (delta * (delta + abs(dq)))
                w = z + sigma * delta * Q[:, nq]
                Q[:, nq:] -= beta * np.outer(w, v)
                rqq = -sigma * delta
                S[:nq, nq] = -r / rqq
                S[nq, nq] = 1.0 / rqq
                u[nq] = up
                where[p] = nq
                act.append(p)
                break
            # partial (or dual-only) step: drop active constraint at position l
            nq = len(act)
            a = S[l, l:nq].copy()
            rho = np.sqrt(np.cumsum(a * a))  # prefix norms
            for jj in range(nq - l - 1):
                j = l + jj
                den = rho[jj + 1]
                if den == 0.0:
                    c_, s_ = 1.0, 0.0
                else:
                    c_, s_ = a[jj + 1] / den, rho[jj] / den if jj > 0 else a[0] / den
                # column rotation on (j, j+1): zero row-l entry in column j
                for Mat in (S[: nq, :], Q, d[None, :]):
                    cj, cj1 = Mat[:, j].copy(), Mat[:, j + 1].copy()
                    Mat[:, j] = c_ * cj - s_ * cj1
                    Mat[:, j + 1] = s_ * cj + c_ * cj1
            # remove row l and the last column of S
            S[l: nq - 1, :] = S[l + 1: nq, :]
            S[nq - 1, :] = 0.0
            S[:, nq - 1] = 0.0
            where[act[l]] = -1
            del act[l]
            u[l: nq - 1] = u[l + 1: nq]
            u[nq - 1] = 0.0
            for i, c in enumerate(act):
                where[c] = i
            if np.isfinite(t2):
                sp = sp * (1.0 - t / t2)
    x = np.linalg.solve(L.T, y)
    lam = np.zeros(m)
    for i, c in enumerate(act):
        lam[c] = u[i]
    return x, lam, status, iters


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    worst = 0.0
    for f in sorted(glob.glob(os.path.join(here, "..", "tests", "golden", "*.npz"))):
        z = np.load(f)
        if "U_star" not in z:
            continue
        P, q, G, h = z["out_P"], z["out_q"], z["out_G"], z["out_h"]
        x, lam, status, iters = solve(P, q, G, h)
        err = np.abs(x - z["U_star"]).max()
        lerr = np.abs(lam - z["lambda_star"]).max()
        stat = np.abs(P @ x + q + G.T @ lam).max()
        print(f"{os.path.basename(f):36s} status={status} iters={iters:3d} |u-u*|={err:.2e} "
              f"|lam-lam*|={lerr:.2e} stat={stat:.1e} viol={np.maximum(G @ x - h, 0).max():.1e}")
        worst = max(worst, err)
    print("worst", worst)
    # random dense QPs incl. degenerate / many-active cases vs brute KKT check
    rng = np.random.default_rng(0)
    bad = 0
    for trial in range(300):
        n = int(rng.integers(2, 12))
        m = int(rng.integers(1, 30))
        A = rng.standard_normal((n, n))
        P = A @ A.T + 1e-3 * np.eye(n)
        q = rng.standard_normal(n) * 3
        G = rng.standard_normal((m, n))
        h = rng.standard_normal(m) + 0.5
        x, lam, status, iters = solve(P, q, G, h)
        if status == 2:
            from scipy.optimize import linprog
            res = linprog(np.zeros(n), A_ub=G, b_ub=h, bounds=[(None, None)] * n)
            if res.status != 2:
                bad += 1
                print("trial", trial, "claimed infeasible but LP says", res.status)
            continue
        stat = np.abs(P @ x + q + G.T @ lam).max()
        viol = np.maximum(G @ x - h, 0).max()
        comp = np.abs(lam * (G @ x - h)).max()
        if stat > 1e-8 or viol > 1e-9 or lam.min() < -1e-10 or comp > 1e-8:
            bad += 1
            print("trial", trial, n, m, "stat", stat, "viol", viol, "lam", lam.min(), "comp", comp, "iters", iters)
    print("random bad:", bad)


if __name__ == "__main__":
    sys.exit(main())
