#!/usr/bin/env python3
"""Dev probe: one round of tools/stress_f32.py (same random stream) in float32 through the default dispatch, the LDS workgroup
kernel, the condensed kernels and the stage-wise kernels, against the float64 oracle; per-problem error distribution.
usage: STRESS_SEED=40 probe_round_f32.py nx nu N mk [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, _capi, workloads as W
from stress_stagewise import random_ltv
want = tuple(int(a) for a in sys.argv[1:5])
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 128
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "31")))
for it in range(400):
    nx, nu = int(rng.integers(2, 17)), int(rng.integers(1, 5))
    N = int(rng.integers(4, 48)); mk = int(rng.choice([1, 2, 3, 4, 8]))
    w = random_ltv(rng, batch, nx, nu, N, mk, float(rng.choice([1.0, 3.0])))
    w["A"] = np.eye(nx) + 0.3 * (w["A"] - np.eye(nx))
    if rng.random() < 0.5:
        w["C"] = np.repeat(w["C"][:, :1], N, axis=1); w["D"] = np.repeat(w["D"][:, :1], N, axis=1)
        w["e"] = w["e"] + 0.5
    if (nx, nu, N, mk) == want:
        break
else:
    sys.exit("round not found")
print("round", it, want)
Uo, lamo, sto, ito = oracle.solve_workload(w)
for name, dt, kw in (("f32 default", torch.float32, {}), ("f32 force LDS", torch.float32, {"flags": _capi.OPT_FORCE_LDS}),
                     ("f32 condensed", torch.float32, {"flags": _capi.OPT_FORCE_CONDENSED}), ("f32 stagewise", torch.float32, {"formulation": "stagewise"}),
                     ("f64 default", torch.float64, {})):
    try:
        plan = solve_mpc_batch(W.to_batch_problem(w, dtype=dt), **kw)
    except Exception as e:
        print(name, "n/a", str(e)[:80]); continue
    torch.cuda.synchronize()
    st = plan.status.cpu().numpy(); U = plan.U.double().cpu().numpy(); itg = plan.iters.cpu().numpy()
    ok = (st == 0) & (sto == 0)
    d = np.abs(U - Uo).max(axis=1) / np.maximum(1, np.abs(Uo).max(axis=1))
    dd = np.sort(d[ok])[::-1]
    worst = np.argsort(-np.where(ok, d, -1))[:3]
    print(f"{name:14s} solved {int((st==0).sum())} (oracle {int((sto==0).sum())}), disagree {np.flatnonzero((st==0)!=(sto==0)).tolist()[:8]}; rel diff top5 {[f'{x:.1e}' for x in dd[:5]]} median {np.median(dd):.1e};"
          f" worst (b, diff, iters, oracle iters, |U|max) {[(int(b), float(d[b]), int(itg[b]), int(ito[b]), float(np.abs(Uo[b]).max())) for b in worst]}")
