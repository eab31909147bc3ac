#!/usr/bin/env python3
"""Dev probe: config 2 (4096 triple-integrator problems, one fused launch per step): K eager launches vs the same K
launches captured in a HIP graph (torch.cuda.CUDAGraph) -- what the inter-kernel gaps of a launch-bound step cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W
bp = W.to_batch_problem(W.triple_integrator_batch(4096))
run = PreparedSolve(bp)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.25:
    for _ in range(20): run.launch()
torch.cuda.synchronize()
def timed(fn, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6
for K in (20, 200):
    print("eager %4d steps: %.2f us per step" % (K, timed(lambda: [run.launch() for _ in range(K)], K)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run.launch()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(K): run.launch()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("graph %4d steps: %.2f us per step" % (K, timed(g.replay, K)))
print("solved", float((run.status == 0).float().mean()))
