#!/usr/bin/env python3
"""Dev check: where the default dispatch SOLVES a problem of tools/stress_tight.py's families that the C oracle calls infeasible (or the
other way round), who is right? A roll-out of the returned inputs in extended precision says whether they are feasible -- no solver
involved. usage: check_disagreement.py [general|wide|narrow] tight seeds(comma)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv
from stress_tight import row_residuals
kind, tight, seeds = sys.argv[1], float(sys.argv[2]), [int(x) for x in sys.argv[3].split(",")]
for seed in seeds:
    rng = np.random.default_rng(seed)
    for it in range(8):
        if kind == "narrow": nx, nu = int(rng.integers(2, 5)), int(rng.integers(1, 3))
        elif kind == "wide": nx, nu = int(rng.integers(5, 17)), int(rng.integers(1, 5))
        else: nx, nu = int(rng.integers(17, 33)), int(rng.integers(1, 9))
        N = int(rng.integers(20, 41)); mk = int(rng.integers(4, 7))
        w = random_ltv(rng, 8, nx, nu, N, mk, tight)
        w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
        plan = solve_mpc_batch(W.to_batch_problem(w)); torch.cuda.synchronize()
        st = plan.status.cpu().numpy(); U = plan.U.cpu().numpy()
        Uo, lamo, sto, ito = oracle.solve_workload(w)
        for b in np.flatnonzero((st == 0) != (sto == 0)):
            if st[b] == 0:
                res = row_residuals(w, b, U[b]); e = np.asarray(w["e"])[b].reshape(-1)
                print(f"seed {seed} round {it} problem {b} (nx={nx} nu={nu} N={N} mk={mk}): GPU solved, oracle status {sto[b]} after {ito[b]} iterations; GPU plan's worst row violation {float((res / (1 + np.abs(e))).max()):.2e} (relative to 1 + |e|), |U|max {np.abs(U[b]).max():.2e}")
            else:
                res = row_residuals(w, b, Uo[b]); e = np.asarray(w["e"])[b].reshape(-1)
                print(f"seed {seed} round {it} problem {b}: oracle solved, GPU status {st[b]}; oracle plan's worst row violation {float((res / (1 + np.abs(e))).max()):.2e}")
print("done")
