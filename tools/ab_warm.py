#!/usr/bin/env python3
"""Dev: back-to-back launches of the config-2 batch, cold vs warm-started from its own solution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, WarmState, workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = W.triple_integrator_batch(B); bp = W.to_batch_problem(w)
ws = WarmState(bp)
runs = {"cold": PreparedSolve(bp), "cold+store": PreparedSolve(bp, warm_state=WarmState(bp)), "warm": PreparedSolve(bp, warm_state=ws)}
runs["warm"].launch(); runs["warm"].set_warm_start(True)
for name, run in runs.items():
    for _ in range(300): run.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): run.launch()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:11s} {e0.elapsed_time(e1):8.2f} us/launch   mean iters {run.iters.float().mean().item():.2f}")
