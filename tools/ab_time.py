#!/usr/bin/env python3
"""Dev: time one library build (MPCQP_LIB) on the bench workload; prints us per step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = W.triple_integrator_batch(batch); run = PreparedSolve(W.to_batch_problem(w))
best = []
for rep in range(5):
    for _ in range(20): run.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run.launch()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 200 * 1e3)
print(os.path.basename(os.environ.get("MPCQP_LIB", "default")), "batch", batch, "us/step: min %.1f median %.1f" % (min(best), sorted(best)[2]))
