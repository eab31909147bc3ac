#!/usr/bin/env python3
"""Dev stress run for the pair kernel (mpcqp_pair.hip): random LTV problems inside its envelope (nx in {3, 4},
n <= 16, m <= 32, float64), tight enough that partial steps, drops and inconsistent rows occur, against the C oracle
and against the one-problem-per-wavefront kernel and the LDS workgroup kernel (same solver, other formulations). One
round in four runs the lean instantiations, half of those with A, C replaced by their first step (rows that are no longer
consistent with e: infeasible and borderline problems).
Statuses must agree and plans must match to 1e-7 relative. usage: stress_pair.py [rounds] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from qpmpc_amd import solve_mpc_batch, _capi, workloads as W
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_stagewise import random_ltv  # noqa


def run(rounds, batch, seed=4242, verbose=True, lean_only=False, flags_lean=0, flags_other=0):
    """lean_only: every round draws the lean family; flags_lean: MpcqpSolveOpts.flags for those rounds (MPCQP_OPT_FOUR_PER_WAVE forces
    the four-per-wavefront kernel, which the dispatch would only take for thousands of problems); flags_other: for the other rounds
    (stage costs, input rows, one to four rows per step: that kernel's general build)."""
    rng = np.random.default_rng(seed)
    worst, bad, drops = 0.0, 0, 0
    for it in range(rounds):
        nx = int(rng.choice([3, 4]))
        nu = int(rng.integers(1, 3))
        N = int(rng.integers(2, 16 // nu + 1))
        mk = int(rng.integers(1, min(4, 32 // N) + 1))
        tight = float(rng.choice([0.05, 0.2, 1.0, 3.0]))
        w = random_ltv(rng, batch, nx, nu, N, mk, tight)
        mode = 3 if lean_only else int(rng.integers(0, 4))
        if mode == 3:  # the lean instantiations <NX, 2>: terminal cost only, state rows only, two rows per step
            mk = 2
            N = min(N, 16)
            w = random_ltv(rng, batch, nx, nu, N, mk, tight)
            w["wx"] = None
            w["targets"] = None
            w["D"] = None
            if rng.random() < 0.5:  # time-invariant A, C (stride 0 along the horizon)
                w["A"] = np.ascontiguousarray(w["A"][:, :1])
                w["C"] = np.ascontiguousarray(w["C"][:, :1])
        if mode == 0:  # terminal cost only
            w["wx"] = None
            w["targets"] = None
        if mode == 1 and rng.random() < 0.5:  # state rows only
            w["D"] = None
        bp = W.to_batch_problem(w)
        fl = _capi.OPT_SEED_VIOLATED if os.environ.get("STRESS_SEEDED") else (flags_lean if mode == 3 else flags_other)  # (STRESS_SEEDED: the seeded start)
        plan = solve_mpc_batch(bp, flags=fl)
        one = solve_mpc_batch(bp, flags=_capi.OPT_ONE_PER_WAVE)
        lds = solve_mpc_batch(bp, flags=_capi.OPT_FORCE_LDS)
        torch.cuda.synchronize()
        U, st, iters = plan.U.cpu().numpy(), plan.status.cpu().numpy(), plan.iters.cpu().numpy()
        Uo, _, sto, _ = oracle.solve_workload(w)
        ok = (st == 0) & (sto == 0)
        agree = float(((st == 0) == (sto == 0)).mean())
        scale = np.maximum(1.0, np.abs(Uo).max(axis=1))
        err = float(((np.abs(U - Uo).max(axis=1) / scale)[ok]).max()) if ok.any() else 0.0
        errs = [err]
        for other in (one, lds):
            so = other.status.cpu().numpy()
            both = (st == 0) & (so == 0) & (sto == 0)
            if other is one:  # (the workgroup kernel, a Q-based variant, calls a few borderline-inconsistent problems solved)
                agree = min(agree, float(((st == 0) == (so == 0)).mean()))
            d = np.abs(U - other.U.cpu().numpy()).max(axis=1) / scale
            errs.append(float(d[both].max()) if both.any() else 0.0)
        same_iters = float((iters == one.iters.cpu().numpy())[st == 0].mean()) if (st == 0).any() else 1.0
        n = N * nu
        # a solved problem that made more trips than it can hold active rows has dropped at least one
        drops += int(((iters > n) & (st == 0)).sum())
        worst = max(worst, max(errs))
        good = agree == 1.0 and max(errs) < 1e-7 and not np.isnan(U).any()
        bad += not good
        if verbose and not good and os.environ.get("STRESS_DUMP"):
            so, sl_ = one.status.cpu().numpy(), lds.status.cpu().numpy()
            io = one.iters.cpu().numpy()
            idx = np.nonzero(((st == 0) != (sto == 0)) | ((st == 0) != (so == 0)) | ((st == 0) != (sl_ == 0)))[0]
            for b in idx[:12]:
                print(f"    problem {b}: status pair/oracle/w64/lds {st[b]}/{sto[b]}/{so[b]}/{sl_[b]} iters pair/w64 {iters[b]}/{io[b]} "
                      f"max|U| pair {np.abs(U[b]).max():.3e} w64 {np.abs(one.U[b].cpu().numpy()).max():.3e} oracle {np.abs(Uo[b]).max():.3e}")
        if verbose:
            print(f"nx={nx} nu={nu} N={N:2d} mk={mk} n={n:2d} m={N*mk:2d} tight={tight}: solved {float((st==0).mean()):.3f} "
                  f"(oracle {float((sto==0).mean()):.3f}) agreement {agree:.4f} rel diff oracle/w64/lds "
                  f"{errs[0]:.1e}/{errs[1]:.1e}/{errs[2]:.1e} iters max {int(iters.max())} same-iters-as-w64 {same_iters:.4f}"
                  f"{'' if good else '   <-- CHECK'}", flush=True)
    return worst, bad, drops


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    worst, bad, drops = run(rounds, batch, seed=int(os.environ.get("STRESS_SEED", "4242")), lean_only=bool(os.environ.get("STRESS_LEAN")),
                            flags_lean=_capi.OPT_FOUR_PER_WAVE if os.environ.get("STRESS_FOUR") else 0,
                            flags_other=_capi.OPT_FOUR_PER_WAVE if os.environ.get("STRESS_FOUR") else 0)
    print("worst rel diff", worst, "rounds flagged", bad, "problems with more trips than variables", drops)
