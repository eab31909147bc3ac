#!/usr/bin/env python3
"""Dev: HIP-event time per launch of the general stage-wise kernel on a prepared solve (no allocation inside the timed launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, workloads as W
from stress_stagewise import random_ltv
nx, nu, N, mk = (int(a) for a in sys.argv[1:5])
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 512
rng = np.random.default_rng(7)
w = random_ltv(rng, batch, nx, nu, N, mk, 1.0)
w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
run = PreparedSolve(W.to_batch_problem(w))
for _ in range(2): run.launch()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run.launch(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(f"nx={nx} nu={nu} N={N} mk={mk} batch {batch}: {min(ts):.2f} ms per launch (median {sorted(ts)[2]:.2f}), solved {float((run.status == 0).float().mean()):.2f}, iters {run.iters.float().mean().item():.1f}")
