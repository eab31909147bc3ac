#!/usr/bin/env python3
"""Dev probe: config 3 (WIP N=50, LTV lists) -- condense-only, solve-only and fused launch times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import BatchMPCQP, PreparedSolve, solve_qp_batch, workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = W.wip_batch(B)
bp = W.to_batch_problem(w)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
qp = BatchMPCQP(bp, keep_propagators=False)
t_c = timeit(lambda: BatchMPCQP(bp, keep_propagators=False))
t_s = timeit(lambda: solve_qp_batch(qp.P, qp.q, qp.G, qp.h))
run = PreparedSolve(bp)
t_f = timeit(run.launch)
it = run.iters.float()
print(f"config 3, batch {B}: condense {t_c:.3f} ms, solve {t_s:.3f} ms, fused {t_f:.3f} ms; iters mean {it.mean().item():.1f} max {it.max().item():.0f}; solved {(run.status==0).float().mean().item():.3f}")
