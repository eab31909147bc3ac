#!/usr/bin/env python3
"""Dev: four-per-wavefront kernel (default dispatch) against the two-per-wavefront one (MPCQP_OPT_TWO_PER_WAVE) on the config-2
and config-4 workloads: plans, statuses, iteration counts, time per launch. usage: ab_quad.py [batch ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W, _capi

def time_it(run, reps=5, n=200):
    best = []
    for _ in range(reps):
        for _ in range(20): run.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): run.launch()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / n * 1e3)
    return min(best), sorted(best)[len(best) // 2]

def one(name, w, batch):
    bp = W.to_batch_problem(w)
    a = PreparedSolve(bp, return_multipliers=True)
    b = PreparedSolve(bp, return_multipliers=True, flags=_capi.OPT_TWO_PER_WAVE)
    a.launch(); b.launch(); torch.cuda.synchronize()
    sa, sb = a.status.cpu(), b.status.cpu()
    ok = (sa == 0) & (sb == 0)
    du = (a.U - b.U).abs().max(dim=1).values.cpu()
    dl = (a.lam - b.lam).abs().max(dim=1).values.cpu()
    print(f"{name} batch {batch}: status equal {bool((sa == sb).all())} (solved {int((sa == 0).sum())}/{batch}, quad statuses {sorted(set(sa.tolist()))}),"
          f" iters equal {int((a.iters.cpu() == b.iters.cpu())[ok].sum())}/{int(ok.sum())}, max|dU| {du[ok].max().item() if ok.any() else float('nan'):.3e}, max|dlam| {dl[ok].max().item() if ok.any() else float('nan'):.3e}")
    ta, tb = time_it(a), time_it(b)
    print(f"   quad us/launch min {ta[0]:.2f} med {ta[1]:.2f} | pair min {tb[0]:.2f} med {tb[1]:.2f} | {batch / ta[0]:.1f} vs {batch / tb[0]:.1f} M/s")

if __name__ == "__main__":
    batches = [int(x) for x in sys.argv[1:]] or [4096]
    for bsz in batches:
        one("config2", W.triple_integrator_batch(bsz), bsz)
    for bsz in ([65536] if len(sys.argv) <= 1 else []):
        one("config4", W.humanoid_batch(bsz), bsz)
