#!/usr/bin/env python3
"""Dev stress run: random LTV problems of many small shapes through mpcqp_build_solve_batch (automatic dispatch: pair,
one-per-wavefront, LDS, mid-size and stage kernels) against the C oracle -- statuses must agree, plans within 1e-7
relative. usage: stress_dense.py [rounds] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, workloads as W
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_stagewise import random_ltv  # noqa

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "777")))
    worst, bad = 0.0, 0
    for it in range(rounds):
        nx, nu = int(rng.integers(1, 9)), int(rng.integers(1, 4))
        N = int(rng.integers(2, max(3, 64 // nu)))
        mk = int(rng.integers(1, 5))
        tight = float(rng.choice([0.2, 1.0, 3.0]))
        w = random_ltv(rng, batch, nx, nu, N, mk, tight)
        if rng.random() < 0.3:
            w["wx"] = None
            w["targets"] = None
        if os.environ.get("STRESS_INCONSISTENT") and rng.random() < 0.5:
            # rows that are no longer consistent with their bounds: infeasible and borderline problems
            w["A"] = np.ascontiguousarray(w["A"][:, :1])
            w["C"] = np.ascontiguousarray(w["C"][:, :1])
        plan = solve_mpc_batch(W.to_batch_problem(w))
        torch.cuda.synchronize()
        U, st = plan.U.cpu().numpy(), plan.status.cpu().numpy()
        Uo, _, sto, _ = oracle.solve_workload(w)
        st = np.where((st == 0) & (np.abs(U).max(axis=1) > 1e8), 9, st)
        ok = (st == 0) & (sto == 0)
        agree = float(((st == 0) == (sto == 0)).mean())
        scale = np.maximum(1.0, np.abs(Uo).max(axis=1))
        err = float(((np.abs(U - Uo).max(axis=1) / scale)[ok]).max()) if ok.any() else 0.0
        worst = max(worst, err)
        if err >= 1e-7 and os.environ.get("STRESS_DUMP"):
            # who is right? objective and worst row violation of both plans, by roll-out
            def judge(b, Uv):
                N, nu_ = w["N"], w["B"].shape[-1]
                A, Bm, Cm, Dm, e = w["A"][b], w["B"][b], w["C"][b], w["D"][b], w["e"][b]
                x = w["x0"][b].copy(); u = Uv.reshape(N, nu_); J = 0.0; viol = -1e300
                for k in range(N):
                    Ak = A[k if A.shape[0] > 1 else 0]; Ck = Cm[k if Cm.shape[0] > 1 else 0]
                    viol = max(viol, float((Ck @ x + Dm[k] @ u[k] - e[k]).max()))
                    J += 0.5 * w["wu"] * float(u[k] @ u[k])
                    x = Ak @ x + Bm[k] @ u[k]
                    if w["wx"] is not None:
                        dx = x - w["targets"][b].reshape(N, -1)[k]; J += 0.5 * w["wx"] * float(dx @ dx)
                dx = x - w["goal"][b]; J += 0.5 * w["wt"] * float(dx @ dx)
                return J, viol
            d = np.abs(U - Uo).max(axis=1) / scale
            for b in sorted(np.nonzero(ok & (d >= 1e-7))[0], key=lambda i: -d[i])[:3]:
                jg, vg = judge(b, U[b]); jo, vo = judge(b, Uo[b])
                print(f"    problem {b}: rel diff {d[b]:.2e}; gpu J {jg:.12e} viol {vg:.2e} iters {int(plan.iters[b])} | oracle J {jo:.12e} viol {vo:.2e}")
        flag = "" if (agree == 1.0 and err < 1e-7 and not np.isnan(U).any()) else "   <-- CHECK"
        bad += flag != ""
        print(f"nx={nx} nu={nu} N={N:2d} mk={mk} n={N*nu:3d} m={N*mk:3d}: solved gpu {float((st==0).mean()):.3f} oracle {float((sto==0).mean()):.3f} agreement {agree:.4f} max rel diff {err:.2e} iters max {int(plan.iters.max())}{flag}", flush=True)
    print("worst rel diff", worst, "rounds flagged", bad)
