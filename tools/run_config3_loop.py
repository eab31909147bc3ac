#!/usr/bin/env python3
"""Dev/profiling driver: the config-3 closed loop (1024 loops, N = 50) with the factor pipelined by the second wavefront and
`ppl` periods per launch: one single-period launch (the episode's first period), then `launches` launches of `ppl` periods.
usage: run_config3_loop.py [ppl] [launches] [rebuild|pipeline|reuse]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd.closed_loop import WIPClosedLoop
ppl = int(sys.argv[1]) if len(sys.argv) > 1 else 20
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "pipeline"
kw = {"pipeline": {"pipeline_factor": True}, "reuse": {"reuse_factor": True}, "rebuild": {}}[mode]
rng = np.random.default_rng(1)
x0 = rng.standard_normal((1024, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
loop = WIPClosedLoop(x0, periods_per_launch=ppl, **kw)
loop.step(1)
for _ in range(launches):
    loop.step(ppl)
torch.cuda.synchronize()
print(mode, "periods per launch", ppl, loop.stats())
