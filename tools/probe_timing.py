#!/usr/bin/env python3
"""Dev probe: kernel time vs batch size and vs iteration cap (separates the
build+factor cost from the per-iteration cost). Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W

def timeit(run, steps=100, warm=10):
    for _ in range(warm): run.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): run.launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3  # us

for batch in (256, 1024, 2048, 4096, 8192, 16384, 65536):
    w = W.triple_integrator_batch(batch)
    bp = W.to_batch_problem(w)
    row = [f"batch {batch:6d}"]
    for mi in (1, 4, 8, None):
        run = PreparedSolve(bp, max_iter=mi)
        row.append(f"max_iter={mi}: {timeit(run):8.1f} us")
    it = run.iters.float()
    row.append(f"iters mean {it.mean().item():.1f} max {it.max().item():.0f}")
    print(" | ".join(row), flush=True)
w = W.humanoid_batch(8192); bp = W.to_batch_problem(w); run = PreparedSolve(bp)
print("humanoid 8192 shared-LTI:", f"{timeit(run):.1f} us", "iters mean", run.iters.float().mean().item(), "max", run.iters.max().item())
w = W.triple_integrator_batch(4096, heterogeneous=False); bp = W.to_batch_problem(w); run = PreparedSolve(bp)
print("triple 4096 shared-LTI:", f"{timeit(run):.1f} us")
