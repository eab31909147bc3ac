#!/usr/bin/env python3
"""Dev/bench: fused per-problem build vs the shared-model (build once) path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, SharedModel, workloads as W
from qpmpc_amd.closed_loop import WIPClosedLoop

def timeit(run, steps=100, warm=10):
    for _ in range(warm): run.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): run.launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3

for name, w in (("config 2 shared-LTI 4096", W.triple_integrator_batch(4096, heterogeneous=False)),
                ("config 4 humanoid 8192", W.humanoid_batch(8192)), ("config 4 humanoid 65536", W.humanoid_batch(65536))):
    bp = W.to_batch_problem(w)
    fused = PreparedSolve(bp); model = SharedModel(bp).prepare(bp)
    tf, tm = timeit(fused), timeit(model)
    B = w["x0"].shape[0]
    print(f"{name}: fused {tf:.1f} us ({B/tf:.1f} M/s) | shared model {tm:.1f} us ({B/tm:.1f} M/s)", flush=True)
rng = np.random.default_rng(1); x0 = rng.standard_normal((1024, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
for shared in (False, True):
    loop = WIPClosedLoop(x0.copy(), shared_model=shared); loop.step(5); torch.cuda.synchronize()
    t = time.perf_counter(); loop.step(100); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"config 3 closed loop (1024 loops, N=50) shared_model={shared}: {1024*100/dt:.0f} builds+solves/s", loop.stats()["failed"], flush=True)
    bp = loop.problem
    print("   solver launch alone: %.1f us per 1024 problems" % timeit(loop.solver, 30, 5))
