import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, _capi, workloads as W
from stress_stagewise import random_ltv
rng = np.random.default_rng(2)
for (nx, nu, N, mk) in ((4, 1, 160, 1), (3, 1, 200, 2), (4, 2, 100, 2), (2, 1, 256, 2)):
    w = random_ltv(rng, 1024, nx, nu, N, mk, 1.0)
    w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
    bp = W.to_batch_problem(w)
    line = f"nx={nx} nu={nu} N={N} mk={mk} n={N*nu} m={N*mk} f64:"
    for name, kw in (("default", {}), ("narrow (stagewise entry)", {"formulation": "stagewise"}), ("wide", {"formulation": "stagewise", "flags": _capi.OPT_STAGE_WIDE}), ("condensed", {"flags": _capi.OPT_FORCE_CONDENSED})):
        try:
            run = PreparedSolve(bp, **kw); run.launch(); torch.cuda.synchronize()
        except Exception as e:
            line += f"  {name}: n/a"; continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run.launch()
        e1.record(); torch.cuda.synchronize()
        line += f"  {name}: {e0.elapsed_time(e1)/3*1e3:.0f} us (it {run.iters.float().mean().item():.1f})"
    print(line)
