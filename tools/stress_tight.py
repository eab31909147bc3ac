#!/usr/bin/env python3
"""Dev stress run: NEARLY FULLY ACTIVE problems (many tight rows per step: most of the n variables end up pinned) through the
default dispatch against the C oracle, for the system widths of the narrow (nx <= 4), wide (nx <= 16) and general stage-wise
kernels. usage: stress_tight.py [narrow|wide|general] [rounds] [batch]; STRESS_RETRY=1: solve_mpc's setting (items that end
MPCQP_MAX_ITER go through the other formulations of the solver, retry_unsolved=True)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv


def run(kind, rounds, batch, seed, verbose=True):
    rng = np.random.default_rng(seed)
    worst, bad = 0.0, 0
    for it in range(rounds):
        if kind == "narrow":
            nx, nu = int(rng.integers(2, 5)), int(rng.integers(1, 3))
        elif kind == "wide":
            nx, nu = int(rng.integers(5, 17)), int(rng.integers(1, 5))
        else:
            nx, nu = int(rng.integers(17, 33)), int(rng.integers(1, 9))
        N = int(rng.integers(20, 41)); mk = int(rng.integers(4, 7))
        w = random_ltv(rng, batch, nx, nu, N, mk, float(os.environ.get("STRESS_TIGHT", "0.5")))  # (smaller: tighter rows, more of them active)
        w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
        plan = solve_mpc_batch(W.to_batch_problem(w), retry_unsolved=os.environ.get("STRESS_RETRY", "0") == "1"); torch.cuda.synchronize()
        st = plan.status.cpu().numpy()
        Uo, lamo, sto, ito = oracle.solve_workload(w)
        agree = np.array_equal(st == 0, sto == 0)
        ok = (st == 0) & (sto == 0)
        err = float((np.abs(plan.U.cpu().numpy()[ok] - Uo[ok]).max(axis=1) / np.maximum(1.0, np.abs(Uo[ok]).max(axis=1))).max()) if ok.any() else 0.0
        nact = (lamo[ok] > 0).sum(axis=1).mean() if ok.any() else 0.0
        worst = max(worst, err)
        flag = (not agree) or err > 1e-6
        bad += flag
        if verbose or flag:
            print(f"round {it}: nx={nx} nu={nu} N={N} mk={mk}: n={N*nu} active {nact:.0f}; solved {int(ok.sum())}/{batch}, statuses agree {agree}"
                  f"{'' if agree else ' gpu ' + str(st.tolist()) + ' oracle ' + str(sto.tolist())}, max rel err {err:.1e}, iters mean {plan.iters.float().mean().item():.0f}" + ("  <-- CHECK" if flag else ""))
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
