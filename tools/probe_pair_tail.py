#!/usr/bin/env python3
"""Dev probe: what the END of a pair-kernel launch is made of -- start skew of the wavefronts, and the phases of the
wavefronts that finish last (shader-clock ticks of the kernel's probe build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = W.triple_integrator_batch(batch); bp = W.to_batch_problem(w)
SL = 16
buf = torch.zeros(batch * SL, dtype=torch.int64, device="cuda")
run = PreparedSolve(bp, probe=buf)
for rep in range(4):
    buf.zero_()
    run.launch()
    torch.cuda.synchronize()
    t = buf.view(batch, SL).cpu().double()
    it = run.iters.cpu().double()
    t = t[0::2]  # one record per wavefront (both halves tick the same clock)
    pm = torch.maximum(it[0::2], it[1::2])
    # the shader clock's base differs from CU to CU: start / end of a wavefront on the launch's time line come from the
    # 100 MHz real-time counter (slots 12, 13; 24 shader cycles per tick at 2.4 GHz), durations from the shader clock
    CYC = 24.0
    r0 = t[:, 12].min()
    start, end = (t[:, 12] - r0) * CYC, (t[:, 13] - r0) * CYC
    print(f"launch {rep}: span {end.max().item():.0f} cyc; starts: median {start.median().item():.0f}, 90% {torch.quantile(start, 0.9).item():.0f}, max {start.max().item():.0f};"
          f" ends: median {end.median().item():.0f}, 99% {torch.quantile(end, 0.99).item():.0f}")
    order = torch.argsort(end, descending=True)[:6]
    names = ["build", "chol", "fwd", "init", "loop", "refine"]
    for i in order.tolist():
        d = (t[i, 1:7] - t[i, 0:6]).tolist()
        print(f"   wave {i:5d}: start {start[i].item():6.0f} end {end[i].item():6.0f} trips {int(pm[i].item()):2d} | " + " ".join(f"{n} {v:.0f}" for n, v in zip(names, d)))
    # how the wave duration depends on the trip count
    dur = end - start
    for k in sorted(set(pm.tolist()))[-4:]:
        sel = pm == k
        print(f"   trips {int(k):2d}: {int(sel.sum())} wavefronts, duration mean {dur[sel].mean().item():.0f} max {dur[sel].max().item():.0f}, end max {end[sel].max().item():.0f}")
