#!/usr/bin/env python3
"""1024 wheeled inverted pendulums balancing and driving at 0.5 m/s, closed loop on the device
(config 3: N = 50, sampling period 24 ms, 15 plant sub-steps per MPC period)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout

import numpy as np
import torch

from qpmpc_amd.closed_loop import WIPClosedLoop

rng = np.random.default_rng(1)
x0 = rng.normal(size=(1024, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
loop = WIPClosedLoop(x0, nb_timesteps=50, sampling_period=0.024, target_vel=0.5, shared_model=True)
loop.step(200)
torch.cuda.synchronize()
s = loop.stats()
print("%d loops x %d periods, %d failed solves, mean %.1f active-set iterations" % (
    s["loops"], s["mpc_steps"], s["failed"], s["mean_iters"]))
print("pitch |theta| max %.4f rad, ground velocity %.3f +- %.3f m/s" % (
    float(loop.states[:, 1].abs().max()), float(loop.states[:, 2].mean()), float(loop.states[:, 2].std())))
