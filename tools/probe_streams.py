#!/usr/bin/env python3
"""Dev probe: throughput with S independent batches in flight on S streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W
batch = 4096
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    runs = [PreparedSolve(W.to_batch_problem(W.triple_integrator_batch(batch, seed=100 + i))) for i in range(S)]
    torch.cuda.synchronize()
    K = 240
    for rep in range(2):
        t0 = time.perf_counter()
        for k in range(K):
            runs[k % S].launch(stream=streams[k % S])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"streams={S}: {dt/K*1e6:.1f} us per 4096-problem step -> {batch*K/dt/1e6:.1f} M problems/s", flush=True)
