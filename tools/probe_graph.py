#!/usr/bin/env python3
"""Dev probe: config-3 closed loop, eager launches vs the periods captured in a HIP graph (torch.cuda.CUDAGraph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd.closed_loop import WIPClosedLoop
from qpmpc_amd import workloads as W
B = 1024
rng = np.random.default_rng(1)
x0 = rng.standard_normal((B, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
loop = WIPClosedLoop(x0, nb_timesteps=50, sampling_period=0.024)
loop.step(5); torch.cuda.synchronize()
def timed(fn, periods):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / periods * 1e6
loop.reset(x0); loop.step(1)
print("eager : %.1f us per period" % timed(lambda: loop.step(100), 100))
for K in (1, 10, 25):
    loop.reset(x0); loop.step(1); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        loop.step(1)  # warm on the side stream
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        loop.step(K)
    torch.cuda.synchronize()
    n = 100 // K
    print("graph of %2d periods: %.1f us per period" % (K, timed(lambda: [g.replay() for _ in range(n)], n * K)))
st = loop.stats(); print(st)
