#!/usr/bin/env python3
"""Dev: statuses of the four-per-wavefront kernel on inconsistent problems against the pair / one-per-wavefront kernels and the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, _capi, workloads as W
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_stagewise import random_ltv  # noqa
rng = np.random.default_rng(99)
for rnd in range(24):
    nx, N = int(rng.choice([3, 4])), int(rng.integers(4, 16))
    w = random_ltv(rng, 128, nx, 1, N, 2, float(rng.choice([0.05, 0.2])))
    w["wx"] = w["targets"] = w["D"] = None
    w["A"] = np.ascontiguousarray(w["A"][:, :1]); w["C"] = np.ascontiguousarray(w["C"][:, :1])
    bp = W.to_batch_problem(w)
    plans = [solve_mpc_batch(bp, flags=f) for f in (0, _capi.OPT_TWO_PER_WAVE, _capi.OPT_ONE_PER_WAVE)]
    torch.cuda.synchronize()
    st = [p.status.cpu().numpy() for p in plans]
    it = [p.iters.cpu().numpy() for p in plans]
    Uo, _, sto, ito = oracle.solve_workload(w)
    bad = np.flatnonzero(((st[0] == 0) != (st[2] == 0)) | ((st[0] == 0) != (sto == 0)))
    print(f"round {rnd} nx={nx} N={N}: unsolved quad/pair/w64/oracle {int((st[0]!=0).sum())}/{int((st[1]!=0).sum())}/{int((st[2]!=0).sum())}/{int((sto!=0).sum())} mismatches {len(bad)}")
    for b in bad[:6]:
        print(f"    problem {b}: status quad/pair/w64/oracle {st[0][b]}/{st[1][b]}/{st[2][b]}/{sto[b]} iters {it[0][b]}/{it[1][b]}/{it[2][b]}/{ito[b]} max|U| quad {np.abs(plans[0].U[b].cpu().numpy()).max():.3e} oracle {np.abs(Uo[b]).max():.3e}")
