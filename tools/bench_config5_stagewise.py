#!/usr/bin/env python3
"""Dev/bench: config 5 (nx=12 nu=4 N=64, n=256 m=1024) through the stage-wise formulation, f32 and f64, next to the
condensed f32 path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import solve_mpc_batch, PreparedSolve, workloads as W
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = W.synthetic_ltv_batch_slice(0, batch)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {}
for name, dt, form in (("stagewise f32", torch.float32, "stagewise"), ("stagewise f64", torch.float64, "stagewise"), ("condensed f32", torch.float32, "condensed")):
    bp = W.to_batch_problem(w, dtype=dt)
    ps = PreparedSolve(bp, formulation=form)  # (one workspace, no allocation in the timed loop)
    ps.launch(); torch.cuda.synchronize()
    e0.record()
    for _ in range(3): ps.launch()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    p = ps.plan
    res[name] = p.U.double()
    del ps
    print(f"{name}: {ms:8.2f} ms per {batch} -> {batch/ms:8.1f} k problems/s; solved {(p.status==0).float().mean().item():.4f}, iters {p.iters.float().mean().item():.2f} max {p.iters.max().item()}", flush=True)
ref = res["stagewise f64"]
sc = ref.abs().max(dim=1).values.clamp(min=1.0)
for k in ("stagewise f32", "condensed f32"):
    print(f"  max rel |U - U_f64 stagewise| of {k}: {float(((res[k]-ref).abs().max(dim=1).values/sc).max()):.2e}")
