import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from qpmpc_amd import solve_mpc_batch, PreparedSolve, workloads as W
for B in (1024, 4096):
    w = W.wip_batch(B); bp = W.to_batch_problem(w)
    # make it harder too: saturating states
    for name, fn in (("condensed", lambda: solve_mpc_batch(bp)), ("stagewise", lambda: solve_mpc_batch(bp, formulation="stagewise"))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): p = fn()
        e1.record(); torch.cuda.synchronize()
        print(B, name, e0.elapsed_time(e1) / 10, "ms", "iters", p.iters.float().mean().item(), "solved", (p.status == 0).float().mean().item())
    run = PreparedSolve(bp); run.launch(); torch.cuda.synchronize()
    e0.record()
    for _ in range(20): run.launch()
    e1.record(); torch.cuda.synchronize(); print(B, "condensed prepared", e0.elapsed_time(e1) / 20, "ms")
