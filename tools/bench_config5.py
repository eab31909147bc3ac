#!/usr/bin/env python3
"""Dev/bench: config 5 (synthetic LTV nx=12 nu=4 N=64, n=256, m=1024, f32) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import BatchMPCQP, PreparedSolve, workloads as W

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.time(); w = W.synthetic_ltv_batch(batch); print("generated", batch, "problems in %.1f s" % (time.time() - t0), flush=True)
bp = W.to_batch_problem(w, dtype=torch.float32)
# condense only
qp = BatchMPCQP(bp, keep_propagators=False); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps): qp = BatchMPCQP(bp, keep_propagators=False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
flops = W.algorithmic_build_flops(12, 4, 64, 16, True, True)
print(f"condense: {ms:.3f} ms per batch of {batch} -> {batch/ms*1e3:.0f} builds/s, {flops*batch/ms/1e9:.2f} TFLOP/s algorithmic", flush=True)
run = PreparedSolve(bp); run.launch(); torch.cuda.synchronize()
e0.record()
for _ in range(steps): run.launch()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
it = run.iters.float()
print(f"build+solve: {ms:.2f} ms per batch of {batch} -> {batch/ms*1e3:.0f} problems/s; solved {(run.status==0).float().mean().item():.3f}, iters mean {it.mean().item():.1f} max {it.max().item():.0f}")
