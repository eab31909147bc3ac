#!/usr/bin/env python3
"""Dev probe (argument f64 for float64): float32 mid-size problems through the default dispatch (on-chip condensed kernels) and through the stage-wise
kernel: error vs the float64 oracle and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import PreparedSolve, workloads as W
from stress_stagewise import random_ltv
rng = np.random.default_rng(5)
dt = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == 'f64') else torch.float32
for (nx, nu, N, mk) in ((3, 4, 32, 2), (2, 3, 46, 3), (4, 4, 39, 3), (3, 3, 43, 8), (8, 2, 20, 4), (6, 1, 40, 2), (12, 4, 16, 4), (8, 1, 37, 3), (5, 2, 8, 2), (3, 1, 16, 2), (6, 3, 5, 3)):
    w = random_ltv(rng, 512, nx, nu, N, mk, 1.0)
    w["A"] = np.eye(nx) + 0.3 * (w["A"] - np.eye(nx))
    Uo, _, sto, _ = oracle.solve_workload(w)
    scale = np.maximum(1.0, np.abs(Uo).max(axis=1))
    bp = W.to_batch_problem(w, dtype=dt)
    line = f"nx={nx} nu={nu} N={N} mk={mk} n={N*nu} m={N*mk}:"
    for name, kw in (("default", {}), ("stagewise", {"formulation": "stagewise"})):
        run = PreparedSolve(bp, **kw); run.launch(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run.launch()
        e1.record(); torch.cuda.synchronize()
        st = run.status.cpu().numpy(); ok = (st == 0) & (sto == 0)
        err = float((np.abs(run.U.double().cpu().numpy() - Uo).max(axis=1) / scale)[ok].max())
        line += f"  {name}: {e0.elapsed_time(e1)/5*1e3:7.1f} us err {err:.1e} solved {float((st==0).mean()):.3f}"
    print(line)
