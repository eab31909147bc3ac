#!/usr/bin/env python3
"""4096 humanoids walking: the model-predictive part of the LIPM walking controller, batched.

Every walker has its own stride lengths, foot size and footstep phase; per MPC period the ZMP bounds
of its 16-step horizon are rebuilt from that phase (a *list* of inequality vectors in the reference,
one [B, N, 2] operand here), the QPs are built and solved in one launch and the plants integrated in
another."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout

import numpy as np
import torch

from qpmpc_amd.closed_loop import LIPMWalkingLoop

B = 4096
rng = np.random.default_rng(0)
strides = np.stack([-rng.uniform(0.12, 0.2, B), rng.uniform(0.12, 0.2, B)], axis=1)
loop = LIPMWalkingLoop(B, strides=strides, foot_size=rng.uniform(0.05, 0.08, B), index=rng.integers(0, 8, B))
for period in range(1, 301):
    loop.step()
    if period % 100 == 0:
        zmp = loop.zmp()
        print("t = %4.1f s: CoM %.3f .. %.3f m, ZMP-to-support distance max %.3f m, failed solves so far %d" % (
            period * loop.sampling_period, float(loop.states[:, 0].min()), float(loop.states[:, 0].max()),
            float((zmp - loop.support).abs().max()), loop.stats()["failed"]))
torch.cuda.synchronize()
