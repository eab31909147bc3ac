#!/usr/bin/env python3
"""Dev: the four-per-wavefront kernel's GENERAL build (input rows, stage cost, nx = 2) against the two-per-wavefront kernel's generic
build on the reference's WIP example (N = 12) and a random family with state + input rows and a stage cost: plans, iteration counts,
time per launch. usage: ab_quad_general.py [batch ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, workloads as W, _capi
from ab_quad import time_it
from stress_stagewise import random_ltv

def one(name, w, batch):
    bp = W.to_batch_problem(w)
    a = PreparedSolve(bp, return_multipliers=True, flags=_capi.OPT_FOUR_PER_WAVE)
    b = PreparedSolve(bp, return_multipliers=True, flags=_capi.OPT_TWO_PER_WAVE)
    a.launch(); b.launch(); torch.cuda.synchronize()
    sa, sb = a.status.cpu(), b.status.cpu()
    ok = (sa == 0) & (sb == 0)
    du = (a.U - b.U).abs().max(dim=1).values.cpu()
    print(f"{name} batch {batch}: status equal {bool((sa == sb).all())} (solved {int((sa == 0).sum())}/{batch}), iters equal {int((a.iters.cpu() == b.iters.cpu())[ok].sum())}/{int(ok.sum())},"
          f" mean iters {a.iters.float().mean().item():.2f}, max|dU| {du[ok].max().item() if ok.any() else float('nan'):.3e}")
    ta, tb = time_it(a), time_it(b)
    print(f"   four per wavefront us/launch min {ta[0]:.2f} med {ta[1]:.2f} | two per wavefront min {tb[0]:.2f} med {tb[1]:.2f} | {batch / ta[0]:.1f} vs {batch / tb[0]:.1f} M/s")

if __name__ == "__main__":
    for bsz in [int(x) for x in sys.argv[1:]] or [4096]:
        w = W.wip_batch(bsz, N=12, sampling_period=0.1, seed=5)
        w["x0"][: bsz // 8, 1] += 0.25
        ts = np.stack([w["pendulum"].target_states(x, 0.5) for x in w["x0"]])
        w["goal"], w["targets"] = ts[:, -4:], ts[:, :-4]
        one("wip N=12 (input box, stage + terminal cost)", w, bsz)
        rng = np.random.default_rng(3)
        one("random LTV nx=3 nu=1 N=16, C + D rows, stage cost", random_ltv(rng, bsz, 3, 1, 16, 2, 0.3), bsz)
        one("config 2 (lean build)", W.triple_integrator_batch(bsz), bsz)
