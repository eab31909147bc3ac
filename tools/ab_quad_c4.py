#!/usr/bin/env python3
"""Dev: four-per-wavefront against two-per-wavefront on the config-4 workload at several batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W, _capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab_quad import time_it
for bsz in [int(x) for x in sys.argv[1:]] or [8192, 16384, 32768, 65536]:
    bp = W.to_batch_problem(W.humanoid_batch(bsz))
    a, b = PreparedSolve(bp, flags=_capi.OPT_FOUR_PER_WAVE), PreparedSolve(bp, flags=_capi.OPT_TWO_PER_WAVE)
    ta, tb = time_it(a, 3, 50), time_it(b, 3, 50)
    print(f"config4 batch {bsz}: quad {ta[0]:.1f} us ({bsz/ta[0]:.1f} M/s) | pair {tb[0]:.1f} us ({bsz/tb[0]:.1f} M/s)")
