import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from qpmpc_amd import _capi, PreparedSolve, workloads as W
sys.path.insert(0, "tests")
rng = np.random.default_rng(5)
w = W.synthetic_ltv_batch(256, nx=12, nu=4, N=32)   # n = 128, m = 512
bp = W.to_batch_problem(w, dtype=torch.float64)
for name, fl in (("struct (default condensed)", _capi.OPT_FORCE_CONDENSED), ("dense G", _capi.OPT_FORCE_CONDENSED | _capi.OPT_FORCE_DENSE_G)):
    run = PreparedSolve(bp, flags=fl)
    run.launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): run.launch()
    e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1)/3:.3f} ms per 256 problems; solved {(run.status==0).float().mean().item():.3f} iters {run.iters.float().mean().item():.2f} |U| {run.U.abs().sum().item():.6e}")
