import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from qpmpc_amd.closed_loop import WIPClosedLoop
rng = np.random.default_rng(123)
x0 = rng.standard_normal((1024, 4)) * np.array([0.1, 0.25, 0.3, 0.9])   # many loops saturate the input box
for mode in ({}, {"pipeline_factor": True}, {"reuse_factor": True}):
    a = WIPClosedLoop(x0.copy(), periods_per_launch=7, **mode)
    b = WIPClosedLoop(x0.copy(), **mode)
    for n in (3, 20, 13, 14):
        a.step(n); b.step(n)
    torch.cuda.synchronize()
    same = torch.equal(a.states, b.states) and torch.equal(a.solver.U, b.solver.U) and a.stats() == b.stats()
    print(mode, "bitwise equal:", same, a.stats())
    assert same
