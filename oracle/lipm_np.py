"""CPU restatement of the LIPM walking example's receding-horizon logic -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/examples/lipm_walking_controller.py; pinned against
tests/golden/lipm_schedule.npz, which tools/gen_golden_lipm.py captured from the reference's
own functions. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

One walker = a dict of scalars; nothing here is vectorised (the batched device version lives in
qpmpc_amd/closed_loop.py::LIPMWalkingLoop and is checked against this).
"""
from __future__ import annotations

import numpy as np

MAX_ZMP_DIST = 100.0  # lipm_walking_controller.py:29


def parameters(**over):
    """``Parameters`` dataclass (lipm_walking_controller.py:32-56)."""
    p = dict(com_height=0.84, dsp_duration=0.1, foot_size=0.065, gravity=9.81, init_support_foot_pos=0.09,
             nb_timesteps=16, sampling_period=0.1, ssp_duration=0.7, strides=(-0.18, 0.18))
    p.update(over)
    p["omega"] = np.sqrt(p["gravity"] / p["com_height"])
    p["dcm_from_state"] = np.array([1.0, 1.0 / p["omega"], 0.0])
    p["zmp_from_state"] = np.array([1.0, 0.0, -1.0 / p["omega"] ** 2])
    return p


def model(p):
    """A, B, C of ``build_mpc_problem`` (lipm_walking_controller.py:59-101)."""
    T = p["sampling_period"]
    A = np.array([[1.0, T, T**2 / 2.0], [0.0, 1.0, T], [0.0, 0.0, 1.0]])
    B = np.array([T**3 / 6.0, T**2 / 2.0, T]).reshape((3, 1))
    C = np.array([+p["zmp_from_state"], -p["zmp_from_state"]])
    return A, B, C


def phase_counts(p):
    """nb_dsp_steps, nb_ssp_steps (``PhaseStepper.__init__``, :104-123)."""
    T = p["sampling_period"]
    nb_dsp, nb_ssp = int(round(p["dsp_duration"] / T)), int(round(p["ssp_duration"] / T))
    if 2 * (nb_dsp + nb_ssp) < p["nb_timesteps"]:
        raise ValueError("there are more than two steps in the receding horizon")
    return nb_dsp, nb_ssp


def new_walker(p, index=5, stride_index=0, support=None):
    """Phase state of one walker; ``index = 5`` is the reference's initial index (:117)."""
    return dict(index=int(index), stride_index=int(stride_index),
                support=p["init_support_foot_pos"] if support is None else float(support))


def nb_steps(p, index):
    """``PhaseStepper.get_nb_steps`` (:134-165)."""
    nb_dsp, nb_ssp = phase_counts(p)
    offset = index
    init_dsp = max(0, nb_dsp - offset)
    offset = max(0, offset - nb_dsp)
    init_ssp = max(0, nb_ssp - offset)
    remaining = p["nb_timesteps"] - init_dsp - init_ssp
    next_dsp = min(nb_dsp, remaining)
    remaining = max(0, remaining - nb_dsp)
    next_ssp = min(nb_ssp, remaining)
    remaining = max(0, remaining - nb_ssp)
    last_dsp = min(nb_dsp, remaining)
    remaining = max(0, remaining - nb_dsp)
    last_ssp = min(nb_ssp, remaining)
    remaining = max(0, remaining - nb_ssp)
    if remaining > 0:
        raise ValueError("there are more than two steps in the receding horizon")
    return init_dsp, init_ssp, next_dsp, next_ssp, last_dsp, last_ssp


def goal_and_constraints(p, w):
    """``update_goal_and_constraints`` (:179-213): e [N, 2] and the goal state."""
    strides = p["strides"]
    c = nb_steps(p, w["index"])
    cur = w["support"]
    nxt = cur + strides[w["stride_index"]]                      # get_next_foot_pos (:167-168)
    last = nxt + strides[(w["stride_index"] + 1) % len(strides)]  # get_last_foot_pos (:170-176)
    half = 0.5 * p["foot_size"]
    free = np.array([+MAX_ZMP_DIST, +MAX_ZMP_DIST])
    rows = ([free] * c[0] + [np.array([cur + half, -(cur - half)])] * c[1]
            + [free] * c[2] + [np.array([nxt + half, -(nxt - half)])] * c[3]
            + [free] * c[4] + [np.array([last + half, -(last - half)])] * c[5])
    goal_pos = last if c[4] > 0 else nxt
    return np.stack(rows), np.array([goal_pos, 0.0, 0.0])


def advance(p, w):
    """End of one MPC period (:329-332 of the main loop, ``advance`` / ``advance_stride`` :125-132)."""
    nb_dsp, nb_ssp = phase_counts(p)
    w["index"] += 1
    if w["index"] >= nb_dsp + nb_ssp:
        w["index"] = 0
    if w["index"] == 0:
        w["support"] = w["support"] + p["strides"][w["stride_index"]]
        w["stride_index"] = (w["stride_index"] + 1) % len(p["strides"])


def integrate(state, jerk, dt):
    """Constant-jerk plant (:216-236)."""
    p0, v0, a0 = state
    return np.array([p0 + dt * (v0 + dt * (a0 / 2 + dt * jerk / 6)), v0 + dt * (a0 + dt * (jerk / 2)), a0 + dt * jerk])


def initial_state(p):
    """ZMP at the centre of the first foothold, DCM halfway (:300-302)."""
    s = p["init_support_foot_pos"]
    return np.array([0.0, 0.5 * p["omega"] * s, -p["omega"] ** 2 * s])


def closed_loop(p, walker, state, steps, substeps=15, solve=None):
    """The example's main loop (:304-333) for one walker, with ``solve(problem) -> U | None``.
    Returns states [steps+1, 3], inputs [steps], statuses [steps]."""
    from qpmpc_amd import MPCProblem

    A, B, C = model(p)
    problem = MPCProblem(A, B, C, None, None, p["nb_timesteps"], 1.0, None, 1e-3)
    X, U0, S = [np.array(state, dtype=float)], [], []
    dt = p["sampling_period"] / substeps
    for _ in range(steps):
        problem.update_initial_state(X[-1])
        e, goal = goal_and_constraints(p, walker)
        problem.ineq_vector = [e[k] for k in range(e.shape[0])]
        problem.update_goal_state(goal)
        U = solve(problem)
        S.append(0 if U is not None else 1)
        u0 = float(U[0]) if U is not None else 0.0
        x = X[-1]
        for _s in range(substeps):
            x = integrate(x, u0, dt)
        X.append(x)
        U0.append(u0)
        advance(p, walker)
    return np.stack(X), np.array(U0), np.array(S)
