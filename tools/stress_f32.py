#!/usr/bin/env python3
"""Dev stress run: random LTV problems of many shapes in FLOAT32 through the default dispatch (small batches take the wide
stage-wise kernel's small-batch instantiation when their constraint matrices are fixed along the horizon) against the float64 C
oracle: statuses must agree except on borderline problems, plans within 1e-3 relative. usage: stress_f32.py [rounds] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, workloads as W
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_stagewise import random_ltv  # noqa

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "31")))
    worst, bad = 0.0, 0
    for it in range(rounds):
        nx, nu = int(rng.integers(2, 17)), int(rng.integers(1, 5))
        N = int(rng.integers(4, 48)); mk = int(rng.choice([1, 2, 3, 4, 8]))
        w = random_ltv(rng, batch, nx, nu, N, mk, float(rng.choice([1.0, 3.0])))
        w["A"] = np.eye(nx) + 0.3 * (w["A"] - np.eye(nx))
        if rng.random() < 0.5:  # constraint matrices fixed along the horizon (the FUSE layout when mk is a multiple of 4)
            w["C"] = np.repeat(w["C"][:, :1], N, axis=1); w["D"] = np.repeat(w["D"][:, :1], N, axis=1)
            w["e"] = w["e"] + 0.5  # (keep them feasible after the change)
        plan = solve_mpc_batch(W.to_batch_problem(w, dtype=torch.float32))
        torch.cuda.synchronize()
        U, st = plan.U.double().cpu().numpy(), plan.status.cpu().numpy()
        Uo, _, sto, _ = oracle.solve_workload(w)
        ok = (st == 0) & (sto == 0)
        agree = float(((st == 0) == (sto == 0)).mean())
        scale = np.maximum(1.0, np.abs(Uo).max(axis=1))
        err = float(((np.abs(U - Uo).max(axis=1) / scale)[ok]).max()) if ok.any() else 0.0
        worst = max(worst, err)
        flag = "" if (agree >= 0.98 and err < 1e-3 and not np.isnan(U).any()) else "   <-- CHECK"
        bad += flag != ""
        print(f"nx={nx:2d} nu={nu} N={N:2d} mk={mk} n={N*nu:3d} m={N*mk:3d}: solved gpu {float((st==0).mean()):.3f} oracle {float((sto==0).mean()):.3f} agreement {agree:.4f} max rel diff {err:.2e} iters max {int(plan.iters.max())}{flag}", flush=True)
    print("worst rel diff", worst, "rounds flagged", bad)
