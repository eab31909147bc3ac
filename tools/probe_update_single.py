import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from qpmpc_amd import MPCQP
from qpmpc_amd import workloads as W
w = W.triple_integrator_batch(4)
p = W.problem_from_workload(w, 0)
qp = MPCQP(p)
x = np.asarray(p.initial_state, dtype=float).copy()
for rep in range(3):
    t0 = time.perf_counter()
    n = 2000
    for i in range(n):
        x[0] += 1e-9
        p.update_initial_state(x)
        qp.update_cost_vector(p)
        qp.update_constraint_vector(p)
    dt = (time.perf_counter() - t0) / n
    print(f"update_cost_vector + update_constraint_vector: {dt*1e6:.1f} us per pair")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(2000):
    x[0] += 1e-9; p.update_initial_state(x); qp.update_cost_vector(p); qp.update_constraint_vector(p)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
