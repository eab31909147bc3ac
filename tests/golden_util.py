"""Load tests/golden/*.npz (captured from the real reference by
tools/gen_golden.py) back into host MPCProblem containers."""
import glob
import os

import numpy as np

from qpmpc_amd import MPCProblem

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _field(z, key):
    kind = str(z[f"{key}_kind"])
    if kind == "none":
        return None
    if kind == "array":
        return z[key]
    out = []
    for k in range(int(z[f"{key}_nsteps"])):
        out.append(None if f"{key}_{k}_none" in z else z[f"{key}_{k}"])
    return out


def _weight(z, name):
    return None if bool(z[name + "_is_none"]) else float(z[name])


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = MPCProblem(
        transition_state_matrix=_field(z, "A"),
        transition_input_matrix=_field(z, "B"),
        ineq_state_matrix=_field(z, "C"),
        ineq_input_matrix=_field(z, "D"),
        ineq_vector=_field(z, "e"),
        nb_timesteps=int(z["nb_timesteps"]),
        terminal_cost_weight=_weight(z, "terminal_cost_weight"),
        stage_state_cost_weight=_weight(z, "stage_state_cost_weight"),
        stage_input_cost_weight=float(z["stage_input_cost_weight"]),
    )
    if not bool(z["initial_state_is_none"]):
        p.update_initial_state(z["initial_state"])
    if not bool(z["goal_state_is_none"]):
        p.update_goal_state(z["goal_state"])
    if not bool(z["target_states_is_none"]):
        p.update_target_states(z["target_states"])
    return p, z


def all_cases(solved_only=False):
    names = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        z = np.load(f)
        if "out_P" not in z:
            continue
        if solved_only and "U_star" not in z:
            continue
        names.append(os.path.splitext(os.path.basename(f))[0])
    return names


def kkt_residuals(P, q, G, h, x, lam):
    stat = np.abs(P @ x + q + G.T @ lam).max()
    prim = np.maximum(G @ x - h, 0.0).max() if len(h) else 0.0
    dual = np.maximum(-lam, 0.0).max() if len(h) else 0.0
    comp = np.abs(lam * (G @ x - h)).max() if len(h) else 0.0
    return stat, prim, dual, comp
