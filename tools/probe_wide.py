#!/usr/bin/env python3
"""Dev probe: the general stage-wise kernel (wide systems) on a random LTV batch: statuses, iterations, error vs the oracle, time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv
nx, nu, N, mk = (int(a) for a in sys.argv[1:5])
tight = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
batch = int(sys.argv[6]) if len(sys.argv) > 6 else 12
rng = np.random.default_rng(1000 * nx + N)
w = random_ltv(rng, batch, nx, nu, N, mk, tight)
w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
bp = W.to_batch_problem(w)
plan = solve_mpc_batch(bp); torch.cuda.synchronize()
t0 = time.perf_counter(); plan = solve_mpc_batch(bp, return_multipliers=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
st, it = plan.status.cpu().numpy(), plan.iters.cpu().numpy()
Uo, lamo, sto, ito = oracle.solve_workload(w)
ok = (st == 0) & (sto == 0)
err = (np.abs(plan.U.cpu().numpy() - Uo).max(axis=1) / np.maximum(1, np.abs(Uo).max(axis=1)))
print(f"nx={nx} nu={nu} N={N} mk={mk} tight={tight}: {dt*1e3:.1f} ms per batch of {batch}")
print(" status gpu", st.tolist()); print(" status ora", sto.tolist()); print(" iters gpu", it.tolist()); print(" iters ora", ito.tolist())
print(" active (oracle)", (lamo > 0).sum(axis=1).tolist()); print(" err", [f"{e:.1e}" for e in err])
