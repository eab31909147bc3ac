#!/usr/bin/env python3
"""Config 1 of BASELINE.json through the drop-in API, then the same model as a 4096-problem sweep.

The single problem is the one of the reference's examples/triple_integrator.py (data restated in
qpmpc_amd/workloads.py::triple_integrator_matrices); here it is solved on the GPU by
``solve_mpc(problem, solver="hip_gi")`` and by the batched entry point.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout

import numpy as np
import torch

from qpmpc_amd import MPCProblem, solve_mpc, solve_mpc_batch
from qpmpc_amd import workloads as W

N = 16
A, B, C, e = W.triple_integrator_matrices(N)
problem = MPCProblem(A, B, C, None, e, N, terminal_cost_weight=1.0, stage_state_cost_weight=None,
                     stage_input_cost_weight=1e-6, initial_state=np.zeros(3), goal_state=np.array([1.0, 0.0, 0.0]))
plan = solve_mpc(problem, solver="hip_gi")
print("single problem: first input %.3f, final state" % plan.first_input[0], np.round(plan.states[-1], 4))

w = W.triple_integrator_batch(4096)  # per-problem x0 and goal, operands stacked per problem and per step
batch_plan = solve_mpc_batch(W.to_batch_problem(w))
torch.cuda.synchronize()
print("sweep: %d of %d solved, inputs tensor %s, mean |x_N - goal| = %.2e" % (
    int(batch_plan.found.sum()), w["x0"].shape[0], tuple(batch_plan.inputs.shape),
    float((batch_plan.states[:, -1, 0].cpu() - torch.tensor(w["goal"][:, 0])).abs().mean())))
