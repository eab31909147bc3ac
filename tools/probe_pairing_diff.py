#!/usr/bin/env python3
"""Dev probe: how far plans move when problems change their wavefront partner / half (MpcqpSolveOpts.order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import pairing_order, solve_mpc_batch, workloads as W
for name, w in (("triple", W.triple_integrator_batch(9000, seed=5)), ("humanoid", W.humanoid_batch(5001, seed=7))):
    bp = W.to_batch_problem(w)
    ref = solve_mpc_batch(bp, return_multipliers=True)
    got = solve_mpc_batch(bp, return_multipliers=True, order=pairing_order(ref.iters))
    torch.cuda.synchronize()
    ok = ref.status == 0
    scale = ref.U[ok].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
    d = ((got.U[ok] - ref.U[ok]).abs() / scale).amax(dim=1)
    lscale = ref.multipliers[ok].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
    dl = ((got.multipliers[ok] - ref.multipliers[ok]).abs() / lscale).amax(dim=1)
    print(name, "solved", int(ok.sum()), "status equal", torch.equal(got.status, ref.status), "iters equal", torch.equal(got.iters, ref.iters),
          "U max rel", float(d.max()), "count > 1e-12", int((d > 1e-12).sum()), "lam max rel", float(dl.max()), "|U| max", float(ref.U[ok].abs().max()))
