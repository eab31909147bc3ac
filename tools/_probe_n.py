import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from qpmpc_amd import solve_mpc_batch, workloads as W
for N in (20, 40, 60, 80):
    bp = W.to_batch_problem(W.wip_batch(1024, N=N))
    buf = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
    for _ in range(2): plan = solve_mpc_batch(bp, formulation="stagewise", probe=buf)
    torch.cuda.synchronize()
    t = buf.view(1024, 16).cpu().double()
    d = [(t[:, i + 1] - t[:, i]).mean().item() for i in range(7)]
    print(N, "riccati %.0f chunk %.0f backward %.0f forward %.0f slacks %.0f" % tuple(d[:5]))
