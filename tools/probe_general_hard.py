#!/usr/bin/env python3
"""Dev probe: the one problem family of tools/stress_general.py (seed 7, round 4: nx 28, nu 3, N 39, six tight rows per step) on
which the general stage-wise kernel stops at max_iter where the oracle solves: statuses and iteration counts of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv
rng = np.random.default_rng(7)
for it in range(5):
    nx = int(rng.integers(17, 33)); nu = int(rng.integers(1, 9)); N = int(rng.integers(2, 41)); mk = int(rng.integers(1, 7))
    tight = float(rng.choice([0.5, 1.0, 3.0]))
    w = random_ltv(rng, 8, nx, nu, N, mk, tight)
    w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
plan = solve_mpc_batch(W.to_batch_problem(w), max_iter=int(os.environ.get("MAXIT", "0")) or None); torch.cuda.synchronize()
Uo, _, sto, ito = oracle.solve_workload(w)
print("dims", nx, nu, N, mk, tight)
print("gpu status", plan.status.cpu().tolist(), "iters", plan.iters.cpu().tolist())
print("ora status", sto.tolist(), "iters", ito.tolist())
