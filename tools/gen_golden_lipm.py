#!/usr/bin/env python3
"""Golden fixtures for the LIPM walking loop, from the REAL reference example.

Runs only in the build container (needs /root/reference). The example module
``examples/lipm_walking_controller.py`` is imported (not copied) with two data-only stand-ins
on ``sys.path``: ``qpsolvers`` (see tools/gen_golden.py) and ``loop_rate_limiters`` (a
``RateLimiter`` that does nothing: the example uses it to pace its live plot only).
What is recorded is DATA produced by the reference's own functions:

* the receding-horizon schedule of its ``PhaseStepper`` + ``update_goal_and_constraints``
  (examples/lipm_walking_controller.py:104-205): for 80 consecutive MPC steps the per-step
  ZMP bounds ``e_k`` [16, 2], the goal state, phase index, stride index and support foot;
* its plant ``integrate`` (``:207-227``) on random (state, jerk, dt) triples;
* the QP the reference builds (``MPCQP``) at three of those steps for given states, with the
  SLSQP+KKT-certified minimiser (independent of this repository's solvers).

Usage: python tools/gen_golden_lipm.py   ->  tests/golden/lipm_schedule.npz, lipm_step_*.npz
"""
from __future__ import annotations

import importlib.util
import os
import sys
import tempfile

import matplotlib

matplotlib.use("Agg")
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  installs the qpsolvers stand-in and imports the reference

sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402  (only to move the state along; see main)

d = tempfile.mkdtemp(prefix="rate_stub_")
with open(os.path.join(d, "loop_rate_limiters.py"), "w") as f:
    f.write('"""Data-only stand-in."""\nclass RateLimiter:\n    def __init__(self, *a, **k): pass\n    def sleep(self): pass\n')
sys.path.insert(0, d)

spec = importlib.util.spec_from_file_location("lipm_ref", os.path.join(gg.REFERENCE, "examples", "lipm_walking_controller.py"))
lipm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lipm)  # the __main__ block does not run


def main():
    params = lipm.Parameters()
    lipm.params = params  # the example reads these two as module globals (set in its __main__ block)
    lipm.T = params.sampling_period
    problem = lipm.build_mpc_problem(params)
    phase = lipm.PhaseStepper(params)
    support = params.init_support_foot_pos
    steps = 80
    e = np.zeros((steps, params.nb_timesteps, 2))
    goal = np.zeros((steps, 3))
    index = np.zeros(steps, dtype=np.int64)
    stride_index = np.zeros(steps, dtype=np.int64)
    support_pos = np.zeros(steps)
    rng = np.random.default_rng(42)
    state = np.array([0.0, 0.5 * params.omega * support, -params.omega**2 * support])  # example :300-302
    state0 = state.copy()
    for s in range(steps):
        lipm.update_goal_and_constraints(problem, phase, support)
        e[s] = np.stack(problem.ineq_vector)
        goal[s] = problem.goal_state
        index[s], stride_index[s], support_pos[s] = phase.index, phase.stride_index, support
        # The state at which each QP is posed is an INPUT of the fixture: it comes from driving the
        # loop with this repository's CPU oracle (any feasible state would do); what is pinned is what
        # the reference builds from it and the independently certified minimiser.
        problem.update_initial_state(state)
        if s in (0, 7, 23):
            gg.record(f"lipm_step_{s:02d}", problem)
        U, status, _ = oracle.solve_mpc_like_reference(problem)
        assert status == 0, (s, status)
        for _ in range(15):
            state = lipm.integrate(state, float(U[0]), params.sampling_period / 15)
        phase.advance()
        if phase.index == 0:
            support = phase.get_next_foot_pos(support)
            phase.advance_stride()
    # plant
    S = rng.normal(size=(32, 3))
    J = rng.normal(size=32) * 5.0
    DT = rng.uniform(0.001, 0.1, size=32)
    Xn = np.stack([lipm.integrate(S[i], J[i], DT[i]) for i in range(32)])
    np.savez_compressed(
        os.path.join(gg.OUT, "lipm_schedule.npz"),
        e=e, goal=goal, index=index, stride_index=stride_index, support_pos=support_pos,
        com_height=params.com_height, dsp_duration=params.dsp_duration, foot_size=params.foot_size,
        gravity=params.gravity, init_support_foot_pos=params.init_support_foot_pos,
        nb_timesteps=params.nb_timesteps, sampling_period=params.sampling_period,
        ssp_duration=params.ssp_duration, strides=np.array(params.strides), omega=params.omega,
        dcm_from_state=params.dcm_from_state, zmp_from_state=params.zmp_from_state,
        max_zmp_dist=lipm.MAX_ZMP_DIST, initial_index=5, init_state=state0,
        A=problem.transition_state_matrix, B=problem.transition_input_matrix, C=problem.ineq_state_matrix,
        wt=problem.terminal_cost_weight, wu=problem.stage_input_cost_weight,
        plant_state=S, plant_jerk=J, plant_dt=DT, plant_next=Xn,
    )
    print("wrote lipm_schedule.npz:", e.shape, "phase indices", index[:12])


if __name__ == "__main__":
    main()
