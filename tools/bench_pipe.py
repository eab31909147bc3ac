#!/usr/bin/env python3
"""Dev tool: config-3 closed loop (1024 loops, N = 50), microseconds per period: rebuilding / factor pipelined / factor reused."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd.closed_loop import WIPClosedLoop
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(1)
x0 = rng.standard_normal((B, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # periods per launch
for name, kw in (("rebuild", {}), ("pipeline_factor", {"pipeline_factor": True}), ("reuse_factor", {"reuse_factor": True})):
    kw = dict(kw, periods_per_launch=P)
    loop = WIPClosedLoop(x0.copy(), **kw)
    loop.step(20); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        loop.reset(x0); loop.step(1); torch.cuda.synchronize()
        t0 = time.perf_counter(); loop.step(100); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
    print(f"{name:16s} {best:7.2f} us per period = {B / best:6.2f} M builds+solves/s   stats {loop.stats()}")
