#!/usr/bin/env python3
"""Dev/bench: MPCQP_WARM_ACTIVE_SET in the wide stage-wise kernel at config 5's dimensions: cold launch against a launch
whose warm rows are the previous solve's active rows (perturbed states), per batch size. usage: bench_stagew_warm.py [batches...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, WarmState, workloads as W

def timed(run, reps=10):
    for _ in range(3): run.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run.launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for B in [int(a) for a in sys.argv[1:]] or [1024, 8192]:
    w = W.synthetic_ltv_batch(B)
    bc, bw = W.to_batch_problem(w, dtype=torch.float32), W.to_batch_problem(w, dtype=torch.float32)
    ws = WarmState(bw)
    cold, warm = PreparedSolve(bc), PreparedSolve(bw, warm_state=ws)
    warm.launch(); torch.cuda.synchronize()
    warm.set_warm_start("active_set", warm_shift=0)
    tc, tw = timed(cold), timed(warm)
    same = bool(torch.equal(cold.iters, warm.iters)) and float((cold.U - warm.U).abs().max()) < 1e-4
    print(f"config-5 problems, batch {B}: cold {tc:.3f} ms, warm rows {tw:.3f} ms per launch; mean iterations {float(cold.iters.float().mean()):.2f}; same iterates {same}")
