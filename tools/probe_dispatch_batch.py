#!/usr/bin/env python3
"""Dev probe: a mid-size problem family at several batch sizes through the default dispatch (wide stage-wise kernel) and with the
condensed kernels forced (MPCQP_OPT_FORCE_CONDENSED): microseconds per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, _capi, workloads as W
from stress_stagewise import random_ltv
rng = np.random.default_rng(5)
for (nx, nu, N, mk) in ((8, 2, 20, 4), (12, 4, 16, 4), (6, 1, 40, 2)):
    for dt in (torch.float64, torch.float32):
        w = random_ltv(rng, 4096, nx, nu, N, mk, 1.0)
        w["A"] = np.eye(nx) + 0.3 * (w["A"] - np.eye(nx))
        line = f"nx={nx} nu={nu} N={N} mk={mk} {'f64' if dt == torch.float64 else 'f32'}:"
        for B in (1, 16, 128, 1024, 4096):
            wb = {k: (v[:B] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 4096 else v) for k, v in w.items()}
            bp = W.to_batch_problem(wb, dtype=dt)
            ts = []
            for kw in ({}, {"flags": _capi.OPT_FORCE_CONDENSED}):
                run = PreparedSolve(bp, **kw); run.launch(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): run.launch()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5 * 1e3)
            line += f"  B={B}: {ts[0]:.0f} / {ts[1]:.0f}"
        print(line + "   (us: default / condensed)")
