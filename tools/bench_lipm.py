#!/usr/bin/env python3
"""Dev/bench: batched LIPM walking loops on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd.closed_loop import LIPMWalkingLoop

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(1)
strides = np.stack([-rng.uniform(0.12, 0.2, B), rng.uniform(0.12, 0.2, B)], axis=1)
shared = len(sys.argv) > 3 and sys.argv[3] == "shared"  # matrices factored once, bounds per period
loop = LIPMWalkingLoop(B, strides=strides, foot_size=rng.uniform(0.05, 0.08, B), index=rng.integers(0, 8, B), shared_model=shared)
loop.step(5); torch.cuda.synchronize()
t0 = time.perf_counter(); loop.step(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
s = loop.stats()
print(f"{B} walkers x {steps} MPC periods: {dt*1e3/steps:.3f} ms per period -> {B*steps/dt/1e6:.2f} M builds+solves/s; "
      f"failed {s['failed']}, mean iters {s['mean_iters']:.1f}, CoM range {loop.states[:,0].min().item():.3f}..{loop.states[:,0].max().item():.3f}")
