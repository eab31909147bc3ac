#!/usr/bin/env python3
"""Dev stress run: random wide LTV systems (17 <= nx <= 32 or 5 <= nu <= 8) through the default dispatch -- the general stage-wise
kernel for every nx > 16 -- against the C oracle: statuses must agree, plans within 1e-7 relative. usage: stress_general.py [rounds] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv

def run(rounds, batch, seed, verbose=True):
    rng = np.random.default_rng(seed)
    worst, bad = 0.0, 0
    for it in range(rounds):
        nx = int(rng.integers(17, 33)); nu = int(rng.integers(1, 9)); N = int(rng.integers(2, 41)); mk = int(rng.integers(1, 7))
        tight = float(rng.choice([0.5, 1.0, 3.0]))
        w = random_ltv(rng, batch, nx, nu, N, mk, tight)
        w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
        plan = solve_mpc_batch(W.to_batch_problem(w)); torch.cuda.synchronize()
        st = plan.status.cpu().numpy()
        Uo, _, sto, _ = oracle.solve_workload(w)
        agree = np.array_equal(st == 0, sto == 0)
        ok = (st == 0) & (sto == 0)
        err = float((np.abs(plan.U.cpu().numpy()[ok] - Uo[ok]).max(axis=1) / np.maximum(1.0, np.abs(Uo[ok]).max(axis=1))).max()) if ok.any() else 0.0
        worst = max(worst, err)
        flag = (not agree) or err > 1e-7
        bad += flag
        if verbose or flag:
            print(f"round {it}: nx={nx} nu={nu} N={N} mk={mk} tight={tight}: solved {int(ok.sum())}/{batch}, statuses agree {agree}, max rel err {err:.1e}, iters mean {plan.iters.float().mean().item():.1f}" + ("  <-- CHECK" if flag else ""))
    return worst, bad

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    # STRESS_SEEDS="1,2,3": several campaigns in one process (tests/test_gpu_stress.py), one line per seed and the total
    seeds = os.environ.get("STRESS_SEEDS")
    if seeds:
        worst, bad = 0.0, 0
        for sd in (int(x) for x in seeds.split(",")):
            w1, b1 = run(rounds, batch, sd, verbose=False)
            print(f"seed {sd}: worst rel diff {w1} rounds flagged {b1}")
            worst, bad = max(worst, w1), bad + b1
    else:
        worst, bad = run(rounds, batch, int(os.environ.get("STRESS_SEED", "2026")))
    print(f"worst rel diff {worst} rounds flagged {bad}")
