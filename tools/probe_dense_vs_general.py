#!/usr/bin/env python3
"""Dev probe: a wide system that fits the dense path (nx > 16 or nu > 4, n = N nu <= 256): default dispatch (dense
large-problem path) against the general stage-wise kernel. usage: probe_dense_vs_general.py nx nu N mk [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv
nx, nu, N, mk = (int(a) for a in sys.argv[1:5])
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 512
dt = torch.float32 if (len(sys.argv) > 6 and sys.argv[6] == "f32") else torch.float64
rng = np.random.default_rng(7)
w = random_ltv(rng, batch, nx, nu, N, mk, 1.0)
w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
bp = W.to_batch_problem(w, dtype=dt)
def timed(**kw):
    p = solve_mpc_batch(bp, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); p = solve_mpc_batch(bp, **kw); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, p
td, pd = timed()
ts, ps = timed(formulation="stagewise")
ok = (pd.status == 0) & (ps.status == 0)
err = float(((pd.U - ps.U).abs().max(dim=1).values / pd.U.abs().max(dim=1).values.clamp(min=1.0))[ok].max()) if ok.any() else float("nan")
buf = torch.zeros(batch * 16, dtype=torch.int64, device="cuda")
solve_mpc_batch(bp, formulation="stagewise", probe=buf); torch.cuda.synchronize()
t = buf.view(batch, 16).cpu().double()
print(f"  general kernel, cycles per problem: total {t[:,0].mean():.0f}, recursion {t[:,1].mean():.0f}, sweeps {t[:,2].mean():.0f} in {t[:,3].mean():.1f} sweep pairs = {(t[:,2]/t[:,3].clamp(min=1)).mean():.0f} each (backward {(t[:,4]/t[:,3].clamp(min=1)).mean():.0f}, forward {(t[:,5]/t[:,3].clamp(min=1)).mean():.0f}, rows of G {(t[:,6]/t[:,3].clamp(min=1)).mean():.0f}; first wavefront computing {(t[:,7]/t[:,3].clamp(min=1)).mean():.0f})")
print(f"  active-set operator, cycles per problem: orthogonalisation {t[:,12].mean():.0f}, R solve {t[:,13].mean():.0f}, drops {t[:,14].mean():.0f}, whitening {t[:,15].mean():.0f}")
print(f"nx={nx} nu={nu} N={N} mk={mk} batch {batch}: default {td:.2f} ms (solved {float((pd.status==0).float().mean()):.2f}, iters {pd.iters.float().mean().item():.1f}) | stage-wise {ts:.2f} ms"
      f" (solved {float((ps.status==0).float().mean()):.2f}, iters {ps.iters.float().mean().item():.1f}) | max rel diff {err:.1e}")
