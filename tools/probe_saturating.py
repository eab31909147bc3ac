#!/usr/bin/env python3
"""Dev probe: config 3's problems with states that saturate the input box (the active-set half of the kernels): the narrow
stage-wise kernel (default dispatch), the wide one and the mid-size condensed kernel on the same batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, _capi, workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = W.wip_batch(B, seed=9)
w["x0"][:, 1] += 0.3
w["x0"][:, 3] += 1.0
pend = w["pendulum"]
ts = np.stack([pend.target_states(x, 0.5) for x in w["x0"]])
w["goal"], w["targets"] = ts[:, -4:], ts[:, :-4]
bp = W.to_batch_problem(w)
for name, kw in (("narrow stage-wise (default)", {}), ("wide stage-wise", {"formulation": "stagewise", "flags": _capi.OPT_STAGE_WIDE}),
                 ("mid-size condensed", {"flags": _capi.OPT_FORCE_CONDENSED})):
    try:
        run = PreparedSolve(bp, **kw)
    except Exception as exn:
        print(name, "unavailable:", exn); continue
    run.launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run.launch()
    e1.record(); torch.cuda.synchronize()
    it = run.iters.float()
    print(f"{name:30s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per {B} problems; solved {(run.status == 0).float().mean().item():.3f}, iterations mean {it.mean().item():.2f} max {it.max().item():.0f}")
