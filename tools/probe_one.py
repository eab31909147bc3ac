#!/usr/bin/env python3
"""Dev probe: launch the fused kernel a few times on one batch size (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = W.triple_integrator_batch(batch); bp = W.to_batch_problem(w); run = PreparedSolve(bp)
for _ in range(reps): run.launch()
torch.cuda.synchronize()
print("ok", run.iters.float().mean().item())
