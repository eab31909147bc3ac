#!/usr/bin/env python3
"""Dev: shared-model launches, four problems per wavefront against two, on the config-4 workload at several batch sizes; the two
must agree (statuses equal, plans to 1e-9)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from qpmpc_amd import SharedModel, workloads as W, _capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab_quad import time_it
for bsz in [int(x) for x in sys.argv[1:]] or [1024, 2048, 2304, 3072, 4096, 8192, 16384, 65536]:
    bp = W.to_batch_problem(W.humanoid_batch(bsz))
    model = SharedModel(bp)
    a = model.prepare(bp, flags=_capi.OPT_FOUR_PER_WAVE)
    b = model.prepare(bp, flags=_capi.OPT_TWO_PER_WAVE)
    a.launch(), b.launch()
    torch.cuda.synchronize()
    sa, sb = a.status.cpu().numpy(), b.status.cpu().numpy()
    ok = sb == 0
    dU = float((a.U - b.U).abs().cpu().numpy()[ok].max())
    ta, tb = time_it(a, 3, 50), time_it(b, 3, 50)
    print(f"model batch {bsz}: quad {ta[0]:.1f} us ({bsz/ta[0]:.1f} M/s) | pair {tb[0]:.1f} us ({bsz/tb[0]:.1f} M/s) | "
          f"status equal {np.array_equal(sa, sb)} dU {dU:.2e} solved {ok.mean():.3f}", flush=True)
