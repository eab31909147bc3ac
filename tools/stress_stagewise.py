#!/usr/bin/env python3
"""Dev stress run: random LTV problems of many shapes through the stage-wise kernels (float64) against the condensed
HIP path (float64) -- statuses must agree, plans within 1e-6 relative. usage: stress_stagewise.py [rounds] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import solve_mpc_batch, _capi, workloads as W

def random_ltv(rng, B, nx, nu, N, mk, tight):
    A = np.eye(nx) + 0.08 * rng.standard_normal((B, N, nx, nx))
    Bm = rng.standard_normal((B, N, nx, nu))
    Cm = rng.standard_normal((B, N, mk, nx))
    D = rng.standard_normal((B, N, mk, nu))
    x0 = 0.1 * rng.standard_normal((B, nx))
    e = np.zeros((B, N, mk))
    for b in range(B):
        x = x0[b].copy()
        for k in range(N):
            e[b, k] = Cm[b, k] @ x + tight * (0.05 + 0.5 * np.abs(rng.standard_normal(mk)))
            x = A[b, k] @ x
    return dict(A=A, B=Bm, C=Cm, D=D, e=e, N=N, wt=2.0, wx=0.5, wu=1e-2, x0=x0,
                goal=rng.standard_normal((B, nx)), targets=rng.standard_normal((B, N * nx)))

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    narrow = len(sys.argv) > 3 and sys.argv[3] == "narrow"  # small systems through mpcqp_stage.hip
    rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "12345")))
    worst, bad = 0.0, 0
    for it in range(rounds):
        nx, nu = (int(rng.integers(2, 5)), int(rng.integers(1, 3))) if narrow else (int(rng.integers(2, 17)), int(rng.integers(1, 5)))
        N = int(rng.integers(3, max(4, 256 // nu)))
        N = min(N, 60)
        mk = int(rng.integers(1, 7))
        w = random_ltv(rng, batch, nx, nu, N, mk, tight=float(rng.choice([0.2, 1.0, 3.0])))
        bp = W.to_batch_problem(w)
        dense = solve_mpc_batch(bp, flags=_capi.OPT_FORCE_CONDENSED)
        wide = solve_mpc_batch(bp, formulation="stagewise", flags=0 if narrow else _capi.OPT_STAGE_WIDE, max_active=min(N * nu, N * mk))
        torch.cuda.synchronize()
        sd, sw = dense.status.cpu().numpy(), wide.status.cpu().numpy()
        ok = (sd == 0) & (sw == 0)
        agree = float(((sd == 0) == (sw == 0)).mean())
        scale = dense.U.abs().amax(dim=1).clamp(min=1.0)
        err = float((((dense.U - wide.U).abs().amax(dim=1) / scale)[torch.from_numpy(ok).cuda()]).max()) if ok.any() else 0.0
        nan = bool(torch.isnan(wide.U).any())
        worst = max(worst, err)
        flag = "" if (agree == 1.0 and err < 1e-6 and not nan) else "   <-- CHECK"
        bad += flag != ""
        print(f"nx={nx:2d} nu={nu} N={N:3d} mk={mk}: solved dense {float((sd==0).mean()):.3f} wide {float((sw==0).mean()):.3f} status agreement {agree:.4f} max rel diff {err:.2e} iters max {int(wide.iters.max())}{flag}", flush=True)
    print("worst rel diff", worst, "rounds flagged", bad)
