#!/usr/bin/env python3
"""Dev/profiling driver: config 5's problems (nx=12 nu=4 N=64) through the stage-wise kernel only.
usage: run_config5_stagewise.py [batch] [launches] [f32|f64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dt = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
bp = W.to_batch_problem(W.synthetic_ltv_batch_slice(0, batch), dtype=dt)
ps = PreparedSolve(bp, formulation="stagewise")
for _ in range(launches): ps.launch()
torch.cuda.synchronize()
p = ps.plan
print("solved", float((p.status == 0).float().mean()), "iters", float(p.iters.float().mean()))
