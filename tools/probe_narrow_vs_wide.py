import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, _capi
from bench_stagewise import long_batch
for B, N, dt in ((4096, 64, 1 / 16), (4096, 256, 1 / 32), (1024, 1024, 1 / 64), (256, 4096, 1 / 128)):
    bp = long_batch(B, N, dt)
    line = f"triple integrator, boxes on acceleration and velocity, N={N} batch {B}:"
    for name, kw in (("narrow", {"formulation": "stagewise"}), ("wide", {"formulation": "stagewise", "flags": _capi.OPT_STAGE_WIDE})):
        run = PreparedSolve(bp, **kw); run.launch(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run.launch()
        e1.record(); torch.cuda.synchronize()
        line += f"  {name}: {e0.elapsed_time(e1)/3:.2f} ms (it {run.iters.float().mean().item():.1f}, solved {(run.status==0).float().mean().item():.3f})"
    print(line)
