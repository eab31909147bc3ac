#!/usr/bin/env python3
"""Dev probe: latency of ONE problem through the drop-in call ``solve_mpc(problem, solver="hip_gi")`` (config 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import MPCProblem, solve_mpc
from qpmpc_amd import workloads as W

N = 16
A, B, C, e = W.triple_integrator_matrices(N)
problem = MPCProblem(A, B, C, None, e, N, 1.0, None, 1e-6, initial_state=np.zeros(3), goal_state=np.array([1.0, 0.0, 0.0]))
for _ in range(5): plan = solve_mpc(problem, solver="hip_gi")
t0 = time.perf_counter(); reps = 200
for i in range(reps):
    problem.update_initial_state(np.array([0.001 * i, 0.0, 0.0]))
    plan = solve_mpc(problem, solver="hip_gi")
    u = plan.first_input
dt = (time.perf_counter() - t0) / reps
print(f"solve_mpc(problem, solver='hip_gi'), one problem per call: {dt*1e6:.0f} us per call (build + solve + Plan + first_input on the host)")
t0 = time.perf_counter()
for i in range(reps):
    plan = solve_mpc(problem, solver="hip_gi"); X = plan.states
print(f"  ... including plan.states: {(time.perf_counter()-t0)/reps*1e6:.0f} us per call")
from qpmpc_amd import MPCQP
for _ in range(3): qp = MPCQP(problem)
t0 = time.perf_counter()
for i in range(100):
    qp = MPCQP(problem); P = qp.P
print(f"MPCQP(problem) + .P on the host: {(time.perf_counter()-t0)/100*1e6:.0f} us per call")
t0 = time.perf_counter()
for i in range(100):
    problem.update_initial_state(np.array([0.001 * i, 0.0, 0.0])); qp.update_cost_vector(problem); qp.update_constraint_vector(problem); q = qp.q
print(f"update_cost_vector + update_constraint_vector + .q: {(time.perf_counter()-t0)/100*1e6:.0f} us per call")
