#!/usr/bin/env python3
"""Dev probe: one round of tools/stress_dense.py (same random stream) through the default dispatch, the LDS workgroup kernel and
the stage-wise kernels, against the oracle. usage: STRESS_SEED=7 probe_round.py nx nu N mk"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, _capi, workloads as W
from stress_stagewise import random_ltv
want = tuple(int(a) for a in sys.argv[1:5])
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "777")))
for it in range(400):
    nx, nu = int(rng.integers(1, 9)), int(rng.integers(1, 4))
    N = int(rng.integers(2, max(3, 64 // nu))); mk = int(rng.integers(1, 5)); tight = float(rng.choice([0.2, 1.0, 3.0]))
    w = random_ltv(rng, 64, nx, nu, N, mk, tight)
    if rng.random() < 0.3:
        w["wx"] = None; w["targets"] = None
    if (nx, nu, N, mk) == want:
        break
print("round", it, want, "tight", tight)
Uo, _, sto, ito = oracle.solve_workload(w)
for name, kw in (("default", {}), ("force LDS", {"flags": _capi.OPT_FORCE_LDS}), ("stagewise", {"formulation": "stagewise"})):
    try:
        plan = solve_mpc_batch(W.to_batch_problem(w), **kw)
    except Exception as e:
        print(name, "n/a", str(e)[:80]); continue
    torch.cuda.synchronize()
    st = plan.status.cpu().numpy(); U = plan.U.cpu().numpy(); itg = plan.iters.cpu().numpy()
    ok = (st == 0) & (sto == 0)
    d = np.abs(U - Uo).max(axis=1) / np.maximum(1, np.abs(Uo).max(axis=1))
    worst = np.argsort(-np.where(ok, d, -1))[:3]
    print(f"{name:10s} unsolved {np.flatnonzero(st != 0).tolist()} worst rel diffs {[(int(b), float(d[b]), int(itg[b]), int(ito[b])) for b in worst]}")
