#!/usr/bin/env python3
"""Dev probe: launch time in natural order / paired by the previous launch's counts (+ the sort), fused build+solve.
usage: probe_pairing_gain.py [triple|humanoid] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, pairing_order, workloads as W
kind = sys.argv[1] if len(sys.argv) > 1 else "triple"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
bp = W.to_batch_problem(W.triple_integrator_batch(B, seed=5) if kind == "triple" else W.humanoid_batch(B, seed=2))
run = PreparedSolve(bp)
def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
nat = timed(run.launch)
order = pairing_order(run.iters)
run.set_order(order)
paired = timed(run.launch)
def period():
    run.launch(); pairing_order(run.iters, out=order)
both = timed(period)
it = run.iters.float()
print(f"{kind} x {B}: natural {nat:.1f} us, paired {paired:.1f} us ({paired / nat - 1:+.1%}), paired + sort {both:.1f} us ({both / nat - 1:+.1%}); "
      f"iters mean {it.mean().item():.2f} std {it.std().item():.2f}")
