#!/usr/bin/env python3
"""Dev: the pair kernel's seeded start against plain iterations (default; seeding: MPCQP_OPT_SEED_VIOLATED) on configs 2 and 4 and the LIPM-like
humanoid sweep: plans against the oracle, statuses, iteration counts, microseconds per launch.
usage: ab_seed.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from qpmpc_amd import PreparedSolve, _capi, solve_mpc_batch, workloads as W


def timed(run, steps=400, reps=5):
    best = []
    for _ in range(reps):
        for _ in range(40): run.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): run.launch()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / steps * 1e3)
    return min(best), sorted(best)[len(best) // 2]


def case(name, w, check=4096):
    bp = W.to_batch_problem(w)
    res = {}
    for tag, fl in (("seed", _capi.OPT_SEED_VIOLATED), ("plain", 0)):
        plan = solve_mpc_batch(bp, flags=fl)
        torch.cuda.synchronize()
        res[tag] = (plan.U.cpu().numpy(), plan.status.cpu().numpy(), plan.iters.cpu().numpy())
    Uo, _, sto, _ = oracle.solve_workload(w, count=min(check, bp.batch_size))
    line = [name]
    for tag in ("seed", "plain"):
        U, st, it = res[tag]
        U, st = U[:check], st[:check]
        ok = (st == 0) & (sto == 0)
        scale = np.maximum(1.0, np.abs(Uo).max(axis=1))
        err = float((np.abs(U - Uo).max(axis=1) / scale)[ok].max()) if ok.any() else 0.0
        line.append(f"{tag}: status==oracle {float((st == sto).mean()):.4f} solved {float((st == 0).mean()):.4f} err {err:.1e} iters {it.mean():.2f}/{it.max()}")
    d = np.abs(res["seed"][0] - res["plain"][0]).max()
    line.append(f"seed-vs-plain dU {d:.1e} statuses equal {bool((res['seed'][1] == res['plain'][1]).all())}")
    for tag, fl in (("seed", _capi.OPT_SEED_VIOLATED), ("plain", 0)):
        run = PreparedSolve(bp, flags=fl)
        lo, med = timed(run)
        line.append(f"{tag} {lo:.1f}/{med:.1f} us")
    print(" | ".join(line), flush=True)


quick = len(sys.argv) > 1
case("config2 4096", W.triple_integrator_batch(4096))
case("config2 2048", W.triple_integrator_batch(2048))
case("config2 shared 4096", W.triple_integrator_batch(4096, heterogeneous=False))
case("config4 8192", W.humanoid_batch(8192))
if not quick:
    case("config4 65536", W.humanoid_batch(65536))
    case("config2 65536", W.triple_integrator_batch(65536))
