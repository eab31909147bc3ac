#!/usr/bin/env python3
"""Dev/bench: config-3 closed loop rebuilding the Riccati factor every period vs reusing the first period's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd.closed_loop import WIPClosedLoop
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(1)
x0 = rng.standard_normal((B, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
for name, kw in (("rebuild every period", {}), ("factor reused", {"reuse_factor": True})):
    loop = WIPClosedLoop(x0, **kw)
    loop.step(20); torch.cuda.synchronize()
    loop.reset(x0)
    torch.cuda.synchronize(); t0 = time.perf_counter(); loop.step(100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt/100*1e6:.1f} us per period, {B*100/dt/1e6:.2f} M solves/s, {loop.stats()}")
