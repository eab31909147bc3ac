#!/usr/bin/env python3
"""Dev: acceptance-test reasons of the four-per-wavefront kernel on one family of tools/probe_quad_inconsistent.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import solve_mpc_batch, PreparedSolve, _capi, workloads as W
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_stagewise import random_ltv  # noqa
rng = np.random.default_rng(99)
target = int(sys.argv[1]) if len(sys.argv) > 1 else 17
for rnd in range(24):
    nx, N = int(rng.choice([3, 4])), int(rng.integers(4, 16))
    w = random_ltv(rng, 128, nx, 1, N, 2, float(rng.choice([0.05, 0.2])))
    w["wx"] = w["targets"] = w["D"] = None
    w["A"] = np.ascontiguousarray(w["A"][:, :1]); w["C"] = np.ascontiguousarray(w["C"][:, :1])
    if rnd != target: continue
    bp = W.to_batch_problem(w)
    buf = torch.zeros(128 * 16, dtype=torch.int64, device="cuda")
    run = PreparedSolve(bp, probe=buf, return_multipliers=True); run.launch()
    ref = PreparedSolve(bp, flags=_capi.OPT_TWO_PER_WAVE, return_multipliers=True); ref.launch()
    torch.cuda.synchronize()
    t = buf.view(128, 16).cpu()
    sa, sb = run.status.cpu().numpy(), ref.status.cpu().numpy()
    for b in np.flatnonzero((sa == 0) != (sb == 0)):
        v = int(t[b, 14])
        print(f"problem {b}: status quad/pair {sa[b]}/{sb[b]} iters {run.iters[b].item()}/{ref.iters[b].item()} reasons inactive-violated {v&1} off {v>>1&1} neg {v>>2&1} fails {v>>8&255} nq {v>>16&255} pair max lam {ref.lam[b].abs().max().item():.3e} max|U| {ref.U[b].abs().max().item():.3e}")
