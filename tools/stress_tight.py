#!/usr/bin/env python3
"""Dev stress run: NEARLY FULLY ACTIVE problems (many tight rows per step: most of the n variables end up pinned) through the
default dispatch against the C oracle, for the system widths of the narrow (nx <= 4), wide (nx <= 16) and general stage-wise
kernels. usage: stress_tight.py [narrow|wide|widef|general] [rounds] [batch] (widef: constraint matrices fixed along the horizon); STRESS_RETRY=1: solve_mpc's setting (items that end
MPCQP_MAX_ITER go through the other formulations of the solver, retry_unsolved=True); STRESS_FORMULATION=stagewise: through
mpcqp_stagewise_solve_batch (the narrow kernel where it applies, the wide one behind it) instead of the default entry point"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv


def row_residuals(w, b, U):
    """G u - h of problem b of a random_ltv workload at the inputs U (<= 0: feasible, 0: on the bound), by a roll-out in extended
    precision: a check that needs no solver."""
    LD = np.longdouble
    A, B, C, D, e, x0 = (np.asarray(w[k]) for k in ("A", "B", "C", "D", "e", "x0"))
    N, nu = A.shape[1], B.shape[-1]
    x, u, out = x0[b].astype(LD), np.asarray(U).reshape(N, nu).astype(LD), []
    for k in range(N):
        out.append(C[b, k].astype(LD) @ x + D[b, k].astype(LD) @ u[k] - e[b, k].astype(LD))
        x = A[b, k].astype(LD) @ x + B[b, k].astype(LD) @ u[k]
    return np.concatenate(out)


def run(kind, rounds, batch, seed, verbose=True):
    rng = np.random.default_rng(seed)
    worst, bad = 0.0, 0
    for it in range(rounds):
        if kind == "narrow":
            nx, nu = int(rng.integers(2, 5)), int(rng.integers(1, 3))
        elif kind in ("wide", "widef"):
            nx, nu = int(rng.integers(5, 17)), int(rng.integers(1, 5))
        else:
            nx, nu = int(rng.integers(17, 33)), int(rng.integers(1, 9))
        N = int(rng.integers(20, 41)); mk = int(rng.integers(4, 7))
        if kind == "widef":  # constraint matrices FIXED along the horizon, 4 / 8 / 12 / 16 rows per step: the wide kernel's layout in
            mk = 4 * int(rng.integers(1, 5))  # which the forward sweep forms the rows itself (config 5's)
        w = random_ltv(rng, batch, nx, nu, N, mk, float(os.environ.get("STRESS_TIGHT", "0.5")))  # (smaller: tighter rows, more of them active)
        w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
        if kind == "widef":
            w["C"], w["D"] = w["C"][:, :1].copy(), w["D"][:, :1].copy()
            for b in range(batch):  # bounds around the free response, as random_ltv makes them (with the fixed C)
                x = w["x0"][b].copy()
                for k in range(N):
                    w["e"][b, k] = w["C"][b, 0] @ x + float(os.environ.get("STRESS_TIGHT", "0.5")) * (0.05 + 0.5 * np.abs(rng.standard_normal(mk)))
                    x = w["A"][b, k] @ x
        kw = {}
        if os.environ.get("STRESS_FORMULATION") == "stagewise":  # (mpcqp_stagewise_solve_batch instead of the default entry point)
            kw = dict(formulation="stagewise", max_active=min(N * nu, N * mk))
        plan = solve_mpc_batch(W.to_batch_problem(w), retry_unsolved=os.environ.get("STRESS_RETRY", "0") == "1", **kw); torch.cuda.synchronize()
        st = plan.status.cpu().numpy()
        Uo, lamo, sto, ito = oracle.solve_workload(w)
        agree = np.array_equal(st == 0, sto == 0)
        ok = (st == 0) & (sto == 0)
        err = float((np.abs(plan.U.cpu().numpy()[ok] - Uo[ok]).max(axis=1) / np.maximum(1.0, np.abs(Uo[ok]).max(axis=1))).max()) if ok.any() else 0.0
        nact = (lamo[ok] > 0).sum(axis=1).mean() if ok.any() else 0.0
        worst = max(worst, err)
        flag = (not agree) or err > 1e-6
        note = ""
        if flag and agree and os.environ.get("STRESS_RESIDUALS", "0") == "1":
            # (plans apart although the statuses agree: which one sits on its active rows? On a fully pinned vertex the ORACLE's rows are
            # up to 1e-8 off -- its rule accepts 1e-6 (1 + |e|) -- and that is 2e-6 in the plan; the round stays flagged only if the
            # GPU plan's active rows are worse than 1e-9 AND worse than the oracle's)
            U = plan.U.cpu().numpy()
            worst_g, worst_o = 0.0, 0.0
            for b in np.flatnonzero(ok):
                act = lamo[b] > 0
                if not act.any():
                    continue
                worst_g = max(worst_g, float(np.abs(row_residuals(w, b, U[b])[act]).max()))
                worst_o = max(worst_o, float(np.abs(row_residuals(w, b, Uo[b])[act]).max()))
            note = f" [active rows off their bounds: gpu {worst_g:.1e}, oracle {worst_o:.1e}]"
            flag = worst_g > 1e-9 and worst_g > worst_o
        bad += flag
        if verbose or flag or note:
            print(f"round {it}: nx={nx} nu={nu} N={N} mk={mk}: n={N*nu} active {nact:.0f}; solved {int(ok.sum())}/{batch}, statuses agree {agree}"
                  f"{'' if agree else ' gpu ' + str(st.tolist()) + ' oracle ' + str(sto.tolist())}, max rel err {err:.1e}, iters mean {plan.iters.float().mean().item():.0f}" + note + ("  <-- CHECK" if flag else ""))
    return worst, bad


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "wide"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    # STRESS_SEEDS="1,2,3": several campaigns in one process (tests/test_gpu_stress.py), one line per seed and the total
    seeds = os.environ.get("STRESS_SEEDS")
    if seeds:
        worst, bad = 0.0, 0
        for sd in (int(x) for x in seeds.split(",")):
            w1, b1 = run(kind, rounds, batch, sd, verbose=False)
            print(f"seed {sd}: worst rel diff {w1} rounds flagged {b1}")
            worst, bad = max(worst, w1), bad + b1
    else:
        worst, bad = run(kind, rounds, batch, int(os.environ.get("STRESS_SEED", "1")))
    print(f"worst rel diff {worst} rounds flagged {bad}")
