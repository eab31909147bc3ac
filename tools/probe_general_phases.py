#!/usr/bin/env python3
"""Dev probe: where a problem of the general stage-wise kernel spends its cycles (MpcqpSolveOpts.probe, csrc/mpcqp_stageg.hip).
usage: probe_general_phases.py nx nu N mk [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv
nx, nu, N, mk = (int(a) for a in sys.argv[1:5])
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 512
rng = np.random.default_rng(7)
w = random_ltv(rng, batch, nx, nu, N, mk, 1.0)
w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
bp = W.to_batch_problem(w)
solve_mpc_batch(bp); torch.cuda.synchronize()
buf = torch.zeros(batch * 16, dtype=torch.int64, device="cuda")
plan = solve_mpc_batch(bp, probe=buf); torch.cuda.synchronize()
t = buf.view(batch, 16).cpu().double()
names = {0: "total", 1: "recursion", 2: "sweeps (all)", 4: "  backward", 5: "  forward", 6: "  rows of G", 7: "  serial wavefront busy",
         12: "orthogonalise", 13: "R solve", 14: "drop", 15: "whiten"}
print(f"nx={nx} nu={nu} N={N} mk={mk} batch {batch}: iters mean {plan.iters.float().mean().item():.1f}, sweeps mean {t[:, 3].mean():.1f}, solved {float((plan.status == 0).float().mean()):.2f}")
for k, nm in names.items():
    print(f"  {nm:26s} mean {t[:, k].mean():12.0f} cyc  ({100 * t[:, k].mean() / t[:, 0].mean():5.1f} %)   max {t[:, k].max():12.0f}")
