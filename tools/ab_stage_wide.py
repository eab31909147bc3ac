#!/usr/bin/env python3
"""Dev A/B: narrow vs wide stage-wise kernel where both apply (float64, nx <= 4, nu <= 2). usage: ab_stage_wide.py [wip|long] [batch] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, _capi, workloads as W
kind = sys.argv[1] if len(sys.argv) > 1 else "wip"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if kind == "wip":
    bp = W.to_batch_problem(W.wip_batch(batch))
else:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_stagewise import long_batch  # noqa
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    bp = long_batch(batch, N, 1.0 / max(16, N // 16))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
for name, fl in (("narrow", 0), ("wide", _capi.OPT_STAGE_WIDE)):
    ps = PreparedSolve(bp, formulation="stagewise", flags=fl)
    ps.launch(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): ps.launch()
    e1.record(); torch.cuda.synchronize()
    p = ps.plan
    out[name] = p.U.clone()
    print(f"{name}: {e0.elapsed_time(e1)/5:8.3f} ms per {batch} (N={bp.nb_timesteps}); solved {(p.status==0).float().mean().item():.4f} iters {p.iters.float().mean().item():.2f}")
print("max |U_wide - U_narrow|", float((out["wide"] - out["narrow"]).abs().max()))
