#!/usr/bin/env python3
"""Dev probe: what the host-side end of a 20-step timed region costs: torch.cuda.synchronize() alone against polling the end
event first (config 2, 4096 problems, one fused launch per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W
bp = W.to_batch_problem(W.triple_integrator_batch(4096))
run = PreparedSolve(bp)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.25:
    for _ in range(20): run.launch()
torch.cuda.synchronize()
def region(K, poll):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0.record()
    for _ in range(K): run.launch()
    e1.record()
    if poll:
        while not e1.query(): pass
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall / K * 1e6, e0.elapsed_time(e1) / K * 1e3
for K in (20, 200):
    for poll in (False, True, False, True):
        r = [region(K, poll) for _ in range(5)]
        print(f"K={K:4d} poll={poll}: wall per step {min(a for a, _ in r):.2f} us (median {sorted(a for a, _ in r)[2]:.2f}), events {min(b for _, b in r):.2f} us")
