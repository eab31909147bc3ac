#!/usr/bin/env python3
"""Dev probe: tools/stress_tight.py general, seed 8, round 0 (nx 28, nu 3, N 24, six tight rows per step): statuses / iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv
rng = np.random.default_rng(8)
nx, nu = int(rng.integers(17, 33)), int(rng.integers(1, 9))
N = int(rng.integers(20, 41)); mk = int(rng.integers(4, 7))
w = random_ltv(rng, 8, nx, nu, N, mk, 0.5)
w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
buf = torch.zeros(8 * 16, dtype=torch.int64, device="cuda")
plan = solve_mpc_batch(W.to_batch_problem(w), max_iter=int(os.environ.get("MAXIT", "0")) or None, probe=buf); torch.cuda.synchronize()
print("stop reasons (1 iteration limit, 2 verification passes, 3 verification failed, 4 re-entered four times, 5 no step)", buf.view(8, 16)[:, 8].cpu().tolist())
print("slots / worst |rho|/thr / zero multipliers at the pass limit (or 100+pass: rebuild failed)", buf.view(8, 16)[:, 9:12].cpu().tolist())
Uo, _, sto, ito = oracle.solve_workload(w)
print("dims", nx, nu, N, mk)
print("gpu status", plan.status.cpu().tolist(), "iters", plan.iters.cpu().tolist())
print("ora status", sto.tolist(), "iters", ito.tolist())
plan2 = solve_mpc_batch(W.to_batch_problem(w), retry_unsolved=True); torch.cuda.synchronize()
ok = (plan2.status.cpu().numpy() == 0) & (sto == 0)
print("with retry_unsolved: status", plan2.status.cpu().tolist(), "iters", plan2.iters.cpu().tolist(),
      "max rel err", float((np.abs(plan2.U.cpu().numpy()[ok] - Uo[ok]).max(axis=1) / np.maximum(1.0, np.abs(Uo[ok]).max(axis=1))).max()))
from qpmpc_amd import _capi
bp = W.to_batch_problem(w).select(torch.tensor([3], device="cuda"))
for name, kw in (("FORCE_LDS", {"flags": _capi.OPT_FORCE_LDS}), ("FORCE_CONDENSED", {"flags": _capi.OPT_FORCE_CONDENSED}),
                 ("FORCE_GWS", {"flags": _capi.OPT_FORCE_GWS}), ("stagewise", {"formulation": "stagewise"})):
    try:
        p = solve_mpc_batch(bp, **kw); torch.cuda.synchronize()
        print(name, "status", p.status.cpu().tolist(), "iters", p.iters.cpu().tolist(), "err", float(np.abs(p.U.cpu().numpy()[0] - Uo[3]).max()))
    except Exception as e:
        print(name, "raised", type(e).__name__, e)
