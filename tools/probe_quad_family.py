#!/usr/bin/env python3
"""Dev: the four-per-wavefront kernel against the two-per-wavefront one on random lean families (terminal cost only, two
state rows per step), by horizon, tightness and time-invariant / per-step A, C."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import solve_mpc_batch, _capi, workloads as W
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_stagewise import random_ltv  # noqa

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
for nx, nu in ((3, 1), (4, 1), (3, 2)):
    for N in range(2, 16 // nu + 1):
        tot = bad_tot = 0
        for tight in (3.0, 0.2, 0.05):
            for lti in (False, True):
                w = random_ltv(rng, 96, nx, nu, N, 2, tight)
                w["wx"] = None; w["targets"] = None; w["D"] = None
                if lti:
                    w["A"] = np.ascontiguousarray(w["A"][:, :1]); w["C"] = np.ascontiguousarray(w["C"][:, :1])
                bp = W.to_batch_problem(w)
                a = solve_mpc_batch(bp, return_multipliers=True)
                b = solve_mpc_batch(bp, flags=_capi.OPT_TWO_PER_WAVE, return_multipliers=True)
                torch.cuda.synchronize()
                sa, sb = a.status.cpu().numpy(), b.status.cpu().numpy()
                du = (a.U - b.U).abs().max(dim=1).values.cpu().numpy()
                dl = (a.multipliers - b.multipliers).abs().max(dim=1).values.cpu().numpy() if hasattr(a, "multipliers") and a.multipliers is not None else du * 0
                both = (sa == 0) & (sb == 0)
                bad = (sa != sb) | (both & ((du > 1e-7) | (dl > 1e-6)))
                tot += 96; bad_tot += int(bad.sum())
                if bad.any() and bad_tot < 4:
                    for i in np.nonzero(bad)[0][:4]:
                        print("    ", (nx, nu, N, tight, lti), i, "status quad/pair", sa[i], sb[i], "iters", a.iters[i].item(), b.iters[i].item(), "dU", du[i], "dl", dl[i])
        print(f"nx={nx} nu={nu} N={N:2d}: bad {bad_tot} of {tot}")
