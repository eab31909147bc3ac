#!/usr/bin/env python3
"""Fixture generator (no reference code involved): the two degenerate problems of tools/stress_dense.py's round 87 (seed 777:
nx = 8, nu = 1, N = 37, mk = 3, rows nearly conflicting, 150+ active-set iterations) on which the mid-size dense kernel's
verification rounds end unsolved while the other formulations and the oracle solve them, with the oracle's plans."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle
from stress_stagewise import random_ltv
rng = np.random.default_rng(777)
for it in range(120):
    nx, nu = int(rng.integers(1, 9)), int(rng.integers(1, 4))
    N = int(rng.integers(2, max(3, 64 // nu))); mk = int(rng.integers(1, 5)); tight = float(rng.choice([0.2, 1.0, 3.0]))
    w = random_ltv(rng, 64, nx, nu, N, mk, tight)
    if rng.random() < 0.3:
        w["wx"] = None; w["targets"] = None
    if (nx, nu, N, mk) == (8, 1, 37, 3):
        break
idx = np.array([4, 30, 0, 1])  # the two hard ones and two ordinary ones of the same round
Uo, _, sto, ito = oracle.solve_workload(w)
out = {k: (v[idx] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 64 else v) for k, v in w.items() if isinstance(v, np.ndarray)}
out.update(N=np.int64(w["N"]), wt=np.float64(w["wt"]), wu=np.float64(w["wu"]), wx=np.float64(-1.0 if w["wx"] is None else w["wx"]),
           U_oracle=Uo[idx], status_oracle=sto[idx], iters_oracle=ito[idx])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "degenerate_nx8_n37.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()}, "oracle iterations", ito[idx])
