#!/usr/bin/env python3
"""One humanoid step (ZMP-constrained CoM transfer) for 65,536 initial states at once (config 4).

A parameter sweep over ONE model: the model-dependent half of the work (P, its Cholesky factor,
M = G L^-T) is done once by ``SharedModel``; every state then only needs the active-set loop.
Infeasible states come back with ``found = False`` (status 2), exactly like an empty ``Plan``.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout

import torch

from qpmpc_amd import SharedModel
from qpmpc_amd import workloads as W

w = W.humanoid_batch(65536)
problem = W.to_batch_problem(w)
model = SharedModel(problem)
plan = model.solve(problem.initial_state, problem.goal_state)
torch.cuda.synchronize()
found = plan.found
end = plan.states[found][:, -1, 0]
q = torch.quantile(end, torch.tensor([0.05, 0.5, 0.95], dtype=end.dtype, device=end.device))
print("%d states, %d feasible; CoM at the end of the plans (5 %% / median / 95 %%): %.3f / %.3f / %.3f m (goal 0.3)" % (
    found.numel(), int(found.sum()), float(q[0]), float(q[1]), float(q[2])))
# states close to the edge of feasibility are still solved: their plans are legal but violent
print("largest jerk in a feasible plan: %.0f m/s^3" % float(plan.inputs[found].abs().max()))
