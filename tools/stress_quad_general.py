#!/usr/bin/env python3
"""Dev stress run for the four-per-wavefront kernel's general build (csrc/mpcqp_quad.hip, GEN): random LTV problems over its whole
envelope -- nx 2..16, nu 1..4, every horizon with n <= 16, 1..4 rows per step (m <= 32), state rows / input rows / both, with and
without a stage cost, time-invariant or per-step operands, loose to very tight bounds -- forced through the kernel
(MPCQP_OPT_FOUR_PER_WAVE) against the C oracle: statuses equal, plans within 1e-7 relative.
usage: stress_quad_general.py [rounds] [batch]   (STRESS_SEED; STRESS_ROWS64=1: also problems of 33 .. 64 rows, the four-rows-per-lane
copy of the kernel in csrc/mpcqp_quad4.hip)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, _capi, workloads as W
from stress_stagewise import random_ltv


def run(rounds, batch, seed, verbose=True):
    rng = np.random.default_rng(seed)
    worst, bad, solved, drops = 0.0, 0, 0, 0
    for it in range(rounds):
        nx, nu = int(rng.integers(2, 17)), int(rng.integers(1, 5))
        N = int(rng.integers(1, 16 // nu + 1))
        rows64 = bool(os.environ.get("STRESS_ROWS64"))  # (up to 64 rows and up to eight rows per step: mpcqp_quad4.hip)
        mk = int(rng.integers(1, min(8 if rows64 else 4, (64 if rows64 else 32) // N) + 1))
        if (N * mk > 32 or mk > 4) and nx > 8:
            nx = int(rng.integers(2, 9))  # (more than 32 rows: that kernel serves nx <= 8)
        tight = float(rng.choice([0.05, 0.2, 1.0, 3.0]))
        rows = str(rng.choice(["c", "d", "cd"]))
        stage = bool(rng.integers(0, 2))
        w = random_ltv(rng, batch, nx, nu, N, mk, tight)
        if rows == "c":
            w["D"] = None
        elif rows == "d":
            w["C"] = None
            w["e"] = tight * (0.05 + 0.5 * np.abs(rng.standard_normal(w["e"].shape)))
        if not stage:
            w["wx"] = w["targets"] = None
        if rng.random() < 0.3:  # time-invariant operands (stride 0 along the horizon); bounds around the free response of THOSE
            for k in ("A", "B", "C", "D"):
                if w[k] is not None:
                    w[k] = np.ascontiguousarray(w[k][:, :1])
            if w["C"] is not None:
                for b in range(batch):
                    x = w["x0"][b].copy()
                    for k in range(N):
                        w["e"][b, k] = w["C"][b, 0] @ x + tight * (0.05 + 0.5 * np.abs(rng.standard_normal(mk)))
                        x = w["A"][b, 0] @ x
        plan = solve_mpc_batch(W.to_batch_problem(w), flags=_capi.OPT_FOUR_PER_WAVE)
        torch.cuda.synchronize()
        U, st, iters = plan.U.cpu().numpy(), plan.status.cpu().numpy(), plan.iters.cpu().numpy()
        Uo, _, sto, _ = oracle.solve_workload(w)
        agree = np.array_equal(st == 0, sto == 0)
        ok = (st == 0) & (sto == 0)
        err = float((np.abs(U[ok] - Uo[ok]).max(axis=1) / np.maximum(1.0, np.abs(Uo[ok]).max(axis=1))).max()) if ok.any() else 0.0
        worst = max(worst, err)
        solved += int(ok.sum())
        drops += int(((iters > N * nu) & ok).sum())
        flag = (not agree) or err > 1e-7 or bool(np.isnan(U).any())
        bad += flag
        if flag and os.environ.get("STRESS_DUMP"):
            two = solve_mpc_batch(W.to_batch_problem(w), flags=_capi.OPT_TWO_PER_WAVE if nx in (3, 4) else _capi.OPT_ONE_PER_WAVE)
            torch.cuda.synchronize()
            s2, i2 = two.status.cpu().numpy(), two.iters.cpu().numpy()
            for b in np.flatnonzero((st == 0) != (sto == 0))[:6]:
                print(f"    problem {b}: status four/other kernel/oracle {st[b]}/{s2[b]}/{sto[b]}, iters {iters[b]}/{i2[b]}, |U|max four {np.abs(U[b]).max():.2e} oracle {np.abs(Uo[b]).max():.2e}")
        if verbose or flag:
            print(f"nx={nx} nu={nu} N={N:2d} mk={mk} rows={rows:2s} stage={int(stage)} tight={tight}: solved {int(ok.sum())}/{batch} statuses agree {agree} max rel err {err:.1e}"
                  f"{'   <-- CHECK' if flag else ''}", flush=True)
    return worst, bad, solved, drops


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    worst, bad, solved, drops = run(rounds, batch, int(os.environ.get("STRESS_SEED", "1")), verbose=not os.environ.get("STRESS_QUIET"))
    print(f"worst rel diff {worst} rounds flagged {bad} (solved {solved}, with more trips than variables {drops})")
