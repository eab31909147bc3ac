#!/usr/bin/env python3
"""Dev tool: time the default dispatch of config 5 (wide stage-wise kernel, float32) at a few batch sizes with the library
named by MPCQP_LIB (A/B runs of build variants). usage: MPCQP_LIB=... ab_c5.py [batch ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, workloads as W

batches = [int(a) for a in sys.argv[1:]] or [1024, 8192]
dt = torch.float64 if os.environ.get("AB_F64") else torch.float32
for B in batches:
    bp = W.to_batch_problem(W.synthetic_ltv_batch_slice(0, B), dtype=dt)
    run = PreparedSolve(bp)
    for _ in range(3): run.launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run.launch(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    U = run.U.double().cpu().numpy()
    st = run.status.cpu().numpy()
    print(f"{os.path.basename(os.environ.get('MPCQP_LIB', 'default')):>18s} batch {B:6d}: min {ts[0]:.3f} ms  median {ts[len(ts)//2]:.3f} ms  "
          f"solved {float((st == 0).mean()):.4f} iters {run.iters.float().mean().item():.3f} |U|sum {np.abs(U).sum():.6e}", flush=True)
