#!/usr/bin/env python3
"""Dev probe: shader-clock stamps of one period of the config-3 closed loop (1024 loops, N = 50) in its three modes
(rebuilding / factor pipelined by a second wavefront / factor reused). usage: probe_config3_loop.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd.closed_loop import WIPClosedLoop
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(1)
x0 = rng.standard_normal((B, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
names = ["factor (or its load)", "chunk matrices", "u0 backward", "u0 forward", "slacks", "active set", "verification", "outputs + plant epilogue"]
for name, kw in (("rebuild", {}), ("pipeline_factor", {"pipeline_factor": True}), ("reuse_factor", {"reuse_factor": True})):
    loop = WIPClosedLoop(x0.copy(), **kw)
    loop.step(20)
    buf = torch.zeros(B * 16, dtype=torch.int64, device="cuda")
    loop.solver._opts.probe = buf.data_ptr()
    loop._period_args = None
    loop.step(2)
    torch.cuda.synchronize()
    t = buf.view(B, 16).cpu().double()
    print(name)
    for i, nme in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"  {nme:26s} mean {d.mean().item():9.0f} cyc   max {d.max().item():9.0f}")
    tot = t[:, 8] - t[:, 0]
    print(f"  {'solving wavefront, total':26s} mean {tot.mean().item():9.0f} cyc   max {tot.max().item():9.0f}")
    if kw.get("pipeline_factor"):
        f = t[:, 10] - t[:, 9]
        print(f"  {'factor wavefront':26s} mean {f.mean().item():9.0f} cyc   max {f.max().item():9.0f}; ends {(t[:, 10] - t[:, 8]).mean().item():.0f} cyc after the solving one (mean)")
    # (the shader clocks of different CUs are not synchronised: stamps of different wavefronts cannot be compared;
    # every stamp costs the wavefront ~0.9 k cycles, so the phases above are inflated by that much each)
