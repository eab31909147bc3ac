#!/usr/bin/env python3
"""Dev probe: what pairing problems of similar trip counts would buy the pair kernel (a wavefront takes max(iters_a, iters_b) trips).
The batch is solved once, its problems are re-ordered by their iteration counts, and both orders are timed. usage: probe_pairing.py [config] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, workloads as W
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (65536 if cfg == 4 else 4096)
w = W.humanoid_batch(batch) if cfg == 4 else W.triple_integrator_batch(batch)

def timeit(w):
    run = PreparedSolve(W.to_batch_problem(w))
    best = []
    for rep in range(5):
        for _ in range(10): run.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run.launch()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 50 * 1e3)
    it = run.iters.cpu().numpy()
    return min(best), it

t0, it = timeit(w)
order = np.argsort(it, kind="stable")
ws = {k: (v[order] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == batch else v) for k, v in w.items()}
t1, it1 = timeit(ws)
pm = lambda a: np.maximum(a[0::2], a[1::2]).mean()
print(f"config {cfg} batch {batch}: as given {t0:.1f} us (mean iters {it.mean():.2f}, per wavefront {pm(it):.2f}) | sorted by iteration count {t1:.1f} us (per wavefront {pm(it1):.2f})")
