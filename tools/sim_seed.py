#!/usr/bin/env python3
"""Development simulation (NOT product, NOT oracle): trip counts of the pair kernel's dual active-set loop with
and without a SEEDED active set (the rows violated at the unconstrained minimiser are added by light steps --
no selection, no ratio test --, multipliers that come out negative leave, then the ordinary iterations go on).

    python tools/sim_seed.py [config2|config4] [count]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import condense_np  # noqa: E402  (development tool: the oracle is the builder of P, q, G, h here)
from qpmpc_amd import workloads  # noqa: E402
from qpmpc_amd.mpc_problem import MPCProblem  # noqa: E402


def gi(M, h, y0, seed=None, tol=1e-9, ratio_in_seed=False):
    """T/H/K formulation as csrc/mpcqp_pair.hip; returns (y, active, trips, light, drops)."""
    m, n = M.shape
    T = np.zeros((0, n))
    act = []
    lam = np.zeros(0)
    H = np.eye(n)
    y = y0.copy()
    nrm = np.linalg.norm(M, axis=1)
    nrm[nrm == 0] = 1
    tolh = tol * (1 + np.abs(h))
    light = drops = trips = 0

    def add(p, t):
        nonlocal T, H, lam, y
        z = -H @ M[p]
        d2 = z @ z
        r = T @ M[p]
        y = y + t * z
        lam = np.append(lam - t * r, t)
        T = np.vstack([T + np.outer(r / d2, z), -z / d2])
        H = H - np.outer(z, z) / d2
        act.append(p)

    def drop(l):
        nonlocal T, H, lam, y
        tl = T[l]
        w = tl @ tl
        y = y + (lam[l] / w) * tl
        lam = lam - lam[l] * (T @ tl) / w
        T = T - np.outer(T @ tl / w, tl)
        H = H + np.outer(tl, tl) / w
        T = np.delete(T, l, 0)
        lam = np.delete(lam, l)
        act.pop(l)

    if seed is not None:
        for p in seed:
            if len(act) >= n:
                break
            z = -H @ M[p]
            d2 = z @ z
            if d2 <= 1e-6 * nrm[p] ** 2:
                continue
            sp = h[p] - M[p] @ y
            add(p, -sp / d2)
            light += 1
        while len(lam) and lam.min() < 0:
            drop(int(np.argmin(lam)))
            drops += 1
    while True:
        s = h - M @ y
        s[act] = 0
        key = np.where(s < -tolh, s / nrm, 0.0)
        p = int(np.argmin(key))
        if key[p] >= 0:
            return y, act, trips, light, drops
        while True:
            trips += 1
            if trips > 200:
                return y, act, trips, light, drops
            z = -H @ M[p]
            d2 = z @ z
            r = T @ M[p]
            sp = h[p] - M[p] @ y
            t2 = -sp / d2 if (d2 > 1e-14 * nrm[p] ** 2 and len(act) < n) else np.inf
            cand = r > 0
            t1 = np.inf
            l = -1
            if cand.any():
                ratio = np.where(cand, lam / np.where(cand, r, 1), np.inf)
                l = int(np.argmin(ratio))
                t1 = ratio[l]
            if not np.isfinite(min(t1, t2)):
                return None, act, trips, light, drops
            if t2 <= t1:
                add(p, t2)
                lam = np.maximum(lam, 0)
                break
            y = y + t1 * z
            lam = lam - t1 * r
            lam[l] = 0.0
            drop(l)
            drops += 1


def problems(name, count):
    if name == "config4":
        w = workloads.humanoid_batch(count)
    else:
        w = workloads.triple_integrator_batch(count, heterogeneous=False)
    for b in range(count):
        pr = MPCProblem(w["A"], w["B"], w["C"], w["D"], w["e"] if w["e"].ndim == 1 else list(w["e"]), w["N"], w["wt"], w["wx"], w["wu"],
                        initial_state=w["x0"][b], goal_state=w["goal"][b] if w["goal"].ndim == 2 else w["goal"])
        yield condense_np.condense(pr)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config2"
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rows = []
    for cq in problems(name, count):
        L = np.linalg.cholesky(cq.P)
        M = np.linalg.solve(L, cq.G.T).T
        y0 = -np.linalg.solve(L, cq.q)
        yc, ac, tc, _, dc = gi(M, cq.h, y0)
        s0 = cq.h - M @ y0
        seed = [int(i) for i in np.nonzero((s0 < -1e-9 * (1 + np.abs(cq.h))) & (cq.h < 1e29))[0]]
        ys, as_, ts, ls, ds = gi(M, cq.h, y0, seed=seed)
        if yc is None or ys is None:
            rows.append((tc, len(seed), ls, ds, ts, -1.0, len(ac)))
            continue
        rows.append((tc, len(seed), ls, ds, ts, float(np.abs(yc - ys).max()), len(ac)))
    r = np.array(rows)
    print(f"{name}: {count} problems")
    print("cold trips   mean %.2f max %d" % (r[:, 0].mean(), r[:, 0].max()))
    print("final active mean %.2f" % r[:, 6].mean())
    print("seeds        mean %.2f max %d (light adds %.2f)" % (r[:, 1].mean(), r[:, 1].max(), r[:, 2].mean()))
    print("seed drops+later drops mean %.2f max %d ; problems with any %.1f %%" % (r[:, 3].mean(), r[:, 3].max(), 100 * (r[:, 3] > 0).mean()))
    print("trips after seeding mean %.2f max %d" % (r[:, 4].mean(), r[:, 4].max()))
    pr = r[: len(r) // 2 * 2].reshape(-1, 2, r.shape[1])
    print("per PAIR: cold max-trips mean %.2f ; seeded: light max %.2f, (drops+trips) max %.2f" % (
        pr[:, :, 0].max(1).mean(), pr[:, :, 2].max(1).mean(), (pr[:, :, 3] + pr[:, :, 4]).max(1).mean()))
    print("max |dy| seeded vs cold: %.2e" % r[:, 5].max())


if __name__ == "__main__":
    main()
