#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by importing the REAL reference.

Runs only in the build container (needs /root/reference). The reference's
third-party dependency ``qpsolvers`` is absent there, so a data-only stand-in
module (``Problem`` / ``Solution`` containers, no arithmetic) is written to a
temp dir and put on ``sys.path`` for the duration of this script; it is never
committed and never ships. Everything numeric in the fixtures that concerns the
*build half* (P, q, G, h, Phi, Psi, phi_last, psi_last, e, Plan.states) is
therefore produced by the reference's own code (qpmpc/mpc_qp.py,
qpmpc/mpc_problem.py, qpmpc/plan.py).

The *solve half* of the reference lives in qpsolvers + a backend, none of which
is installed here ("parity unpinned" at that boundary). The
``U_star`` stored here is obtained, independently of this repository's solvers,
by SciPy SLSQP on the reference-built (P, q, G, h), then polished by solving
the equality-constrained KKT system of SLSQP's active set with numpy.linalg and
certified: stationarity, primal feasibility, dual feasibility and
complementarity residuals are stored next to it. Because P >= w_u I > 0 the
minimiser is unique, so any exact backend (quadprog) returns this same vector.

Only data (inputs + expected outputs) is written; no reference source text.

Usage: python tools/gen_golden.py
"""
from __future__ import annotations

import os
import sys
import tempfile
import textwrap

import numpy as np
from scipy.optimize import minimize

REFERENCE = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_qpsolvers_stub() -> str:
    d = tempfile.mkdtemp(prefix="qpsolvers_stub_")
    os.makedirs(os.path.join(d, "qpsolvers"))
    with open(os.path.join(d, "qpsolvers", "__init__.py"), "w") as f:
        f.write(
            textwrap.dedent(
                '''
                """Data-only stand-in for qpsolvers (containers, no solver)."""
                available_solvers = []
                class Problem:
                    def __init__(self, P, q, G=None, h=None, A=None, b=None, lb=None, ub=None):
                        self.P, self.q, self.G, self.h = P, q, G, h
                class Solution:
                    def __init__(self, problem, x=None, found=False):
                        self.problem, self.x, self.found = problem, x, found
                def solve_problem(problem, solver, **kwargs):
                    raise RuntimeError("no QP backend in this container")
                '''
            )
        )
    sys.path.insert(0, d)
    return d


_install_qpsolvers_stub()
sys.path.insert(0, REFERENCE)
import qpmpc  # noqa: E402  (the real reference, v3.1.0)
import qpsolvers  # noqa: E402  (the stub)
from qpmpc import MPCQP, MPCProblem, Plan  # noqa: E402
from qpmpc.exceptions import ProblemDefinitionError, StateError  # noqa: E402
from qpmpc.systems import WheeledInvertedPendulum  # noqa: E402

assert qpmpc.__version__ == "3.1.0"


# --------------------------------------------------------------------------
# exact solution of the reference-built QP, independent of this repo's solvers
# --------------------------------------------------------------------------
def certified_solution(P, q, G, h):
    n = P.shape[0]
    res = minimize(
        lambda x: 0.5 * x @ P @ x + q @ x,
        np.zeros(n),
        jac=lambda x: P @ x + q,
        constraints=[{"type": "ineq", "fun": lambda x: h - G @ x, "jac": lambda x: -G}],
        method="SLSQP",
        options={"ftol": 1e-16, "maxiter": 2000},
    )
    x = res.x
    # active-set guess from SLSQP, then exact KKT solve + a few primal/dual fix-ups
    slack = h - G @ x
    scale = 1.0 + np.abs(h)
    act = list(np.where(slack < 1e-7 * scale)[0])
    for _ in range(200):
        # drop linearly dependent rows (keep a well-conditioned subset)
        Ga = G[act]
        if len(act):
            K = np.block([[P, Ga.T], [Ga, np.zeros((len(act), len(act)))]])
            rhs = np.concatenate([-q, h[act]])
            sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
            x, lam = sol[:n], sol[n:]
        else:
            x, lam = np.linalg.solve(P, -q), np.zeros(0)
        if len(lam) and lam.min() < -1e-12:
            act.pop(int(np.argmin(lam)))
            continue
        viol = G @ x - h
        worst = int(np.argmax(viol / scale))
        if viol[worst] > 1e-11 * scale[worst] and worst not in act:
            act.append(worst)
            continue
        break
    lam_full = np.zeros(G.shape[0])
    lam_full[act] = lam
    kkt = {
        "stationarity": float(np.abs(P @ x + q + G.T @ lam_full).max()),
        "primal": float(np.maximum(G @ x - h, 0.0).max()) if G.shape[0] else 0.0,
        "dual": float(np.maximum(-lam_full, 0.0).max()) if G.shape[0] else 0.0,
        "complementarity": float(np.abs(lam_full * (G @ x - h)).max()) if G.shape[0] else 0.0,
    }
    return x, lam_full, np.array(sorted(act), dtype=np.int64), kkt


# --------------------------------------------------------------------------
# fixture record: inputs as stacked arrays + presence flags, outputs verbatim
# --------------------------------------------------------------------------
def _stack(field, N):
    """Return (array[N,...], is_list) for an LTI array or per-step list."""
    if field is None:
        return None, False
    if isinstance(field, list):
        return field, True
    return field, False


def record(name, problem: MPCProblem, solve=True, extra=None):
    qp = MPCQP(problem)
    rec = {}
    N = problem.nb_timesteps
    for key, field in (
        ("A", problem.transition_state_matrix),
        ("B", problem.transition_input_matrix),
        ("C", problem.ineq_state_matrix),
        ("D", problem.ineq_input_matrix),
        ("e", problem.ineq_vector),
    ):
        if field is None:
            rec[f"{key}_kind"] = np.array("none")
        elif isinstance(field, list):
            rec[f"{key}_kind"] = np.array("list")
            rec[f"{key}_nsteps"] = np.array(len(field))
            for k, blk in enumerate(field):
                if blk is None:
                    rec[f"{key}_{k}_none"] = np.array(True)
                else:
                    rec[f"{key}_{k}"] = np.asarray(blk)
        else:
            rec[f"{key}_kind"] = np.array("array")
            rec[key] = np.asarray(field)
    rec["nb_timesteps"] = np.array(N)
    for wname in ("terminal_cost_weight", "stage_state_cost_weight", "stage_input_cost_weight"):
        w = getattr(problem, wname)
        rec[wname] = np.array(np.nan if w is None else float(w))
        rec[wname + "_is_none"] = np.array(w is None)
    for sname in ("initial_state", "goal_state", "target_states"):
        s = getattr(problem, sname)
        rec[sname + "_is_none"] = np.array(s is None)
        if s is not None:
            rec[sname] = np.asarray(s, dtype=float)
    for oname in ("P", "q", "G", "h", "Phi", "Psi", "phi_last", "psi_last", "e"):
        rec["out_" + oname] = np.asarray(getattr(qp, oname))
    # qp.C is block_diag(*C_list); an object array when every C_k is None (quirk 4)
    rec["out_C_is_object"] = np.array(getattr(qp.C, "dtype", None) == object)
    if qp.C is not None and qp.C.dtype != object:
        rec["out_C"] = np.asarray(qp.C, dtype=float)
    if solve:
        x, lam, act, kkt = certified_solution(qp.P, qp.q, qp.G, qp.h)
        assert kkt["stationarity"] < 1e-8 * (1 + np.abs(qp.q).max()), (name, kkt)
        assert kkt["primal"] < 1e-9 and kkt["dual"] < 1e-9, (name, kkt)
        rec["U_star"] = x
        rec["lambda_star"] = lam
        rec["active_set"] = act
        rec["obj_star"] = np.array(0.5 * x @ qp.P @ x + qp.q @ x)
        for k, v in kkt.items():
            rec["kkt_" + k] = np.array(v)
        # Plan / integrate through the reference's own code (plan.py, mpc_problem.py)
        sol = qpsolvers.Solution(qp.problem, x=x, found=True)
        plan = Plan(problem, sol)
        rec["plan_inputs"] = np.asarray(plan.inputs)
        rec["plan_states"] = np.asarray(plan.states)
        rec["plan_first_input"] = np.asarray(plan.first_input)
        print(
            f"{name:34s} n={qp.P.shape[0]:4d} m={qp.G.shape[0]:5d} active={len(act):3d} "
            f"obj={float(rec['obj_star']):+.10f} stat={kkt['stationarity']:.1e} "
            f"cond(P)={np.linalg.cond(qp.P):.2e}"
        )
    else:
        print(f"{name:34s} n={qp.P.shape[0]:4d} m={qp.G.shape[0]:5d} (build only)")
    if extra:
        rec.update(extra)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    return qp


# --------------------------------------------------------------------------
# problem definitions (data of the reference's examples, restated as numbers)
# --------------------------------------------------------------------------
def triple_integrator(x0=(0.0, 0.0, 0.0), goal=(1.0, 0.0, 0.0)):
    N = 16
    T = 1.0 / N
    A = np.array([[1.0, T, T**2 / 2.0], [0.0, 1.0, T], [0.0, 0.0, 1.0]])
    B = np.array([T**3 / 6.0, T**2 / 2.0, T]).reshape((3, 1))
    C = np.vstack([[0.0, 0.0, 1.0], [0.0, 0.0, -1.0]])
    e = np.array([3.0, 3.0])
    return MPCProblem(
        transition_state_matrix=A,
        transition_input_matrix=B,
        ineq_state_matrix=C,
        ineq_input_matrix=None,
        ineq_vector=e,
        initial_state=np.array(x0),
        goal_state=np.array(goal),
        nb_timesteps=N,
        terminal_cost_weight=1.0,
        stage_state_cost_weight=None,
        stage_input_cost_weight=1e-6,
    )


def humanoid(x0=(0.0, 0.0, 0.0)):
    N, horizon = 16, 2.5
    T = horizon / N
    dsp, ssp, foot, start, end = 0.1, 0.7, 0.1, 0.0, 0.3
    n_dsp0 = int(round(dsp / T))
    n_ssp0 = int(round(ssp / T))
    n_dsp = int(round(dsp / T))
    A = np.array([[1.0, T, T**2 / 2.0], [0.0, 1.0, T], [0.0, 0.0, 1.0]])
    B = np.array([T**3 / 6.0, T**2 / 2.0, T]).reshape((3, 1))
    zmp = np.array([1.0, 0.0, -0.8 / 9.81])
    C = np.array([+zmp, -zmp])
    cur_max, cur_min = start + 0.5 * foot, start - 0.5 * foot
    next_max, next_min = end + 0.5 * foot, end - 0.5 * foot
    e = [
        np.array([+1000.0, +1000.0])
        if i < n_dsp0
        else np.array([+cur_max, -cur_min])
        if i - n_dsp0 <= n_ssp0
        else np.array([+1000.0, +1000.0])
        if i - n_dsp0 - n_ssp0 < n_dsp
        else np.array([+next_max, -next_min])
        for i in range(N)
    ]
    return MPCProblem(
        transition_state_matrix=A,
        transition_input_matrix=B,
        ineq_state_matrix=C,
        ineq_input_matrix=None,
        ineq_vector=e,
        initial_state=np.array(x0),
        goal_state=np.array([end, 0.0, 0.0]),
        nb_timesteps=N,
        terminal_cost_weight=1.0,
        stage_state_cost_weight=None,
        stage_input_cost_weight=1e-3,
    )


def wip_targets(pendulum, state, target_vel):
    nx, T = pendulum.STATE_DIM, pendulum.sampling_period
    ts = np.zeros((pendulum.nb_timesteps + 1) * nx)
    for k in range(pendulum.nb_timesteps + 1):
        ts[k * nx] = state[0] + (k * T) * target_vel
        ts[k * nx + 2] = target_vel
    return ts


def wip(N, T, state, target_vel, ltv_lists=False):
    pend = WheeledInvertedPendulum(nb_timesteps=N, sampling_period=T)
    p = pend.build_mpc_problem(
        terminal_cost_weight=10.0, stage_state_cost_weight=1.0, stage_input_cost_weight=1e-3
    )
    if ltv_lists:
        p.transition_state_matrix = [p.transition_state_matrix.copy() for _ in range(N)]
        p.transition_input_matrix = [p.transition_input_matrix.copy() for _ in range(N)]
    ts = wip_targets(pend, state, target_vel)
    p.update_initial_state(np.asarray(state, dtype=float))
    p.update_goal_state(ts[-4:])
    p.update_target_states(ts[:-4])
    return pend, p


def random_ltv(seed=7, nx=5, nu=2, N=7):
    rng = np.random.default_rng(seed)
    A = [np.eye(nx) + 0.3 * rng.standard_normal((nx, nx)) for _ in range(N)]
    B = [rng.standard_normal((nx, nu)) for _ in range(N)]
    mks = [1 + (k % 4) for k in range(N)]
    C = [rng.standard_normal((mk, nx)) for mk in mks]
    D = [rng.standard_normal((mk, nu)) for mk in mks]
    x0 = 0.1 * rng.standard_normal(nx)
    # keep x0 strictly feasible for u=0 so the QP is feasible: e_k > C_k x_k(u=0)
    e, x = [], x0.copy()
    for k in range(N):
        e.append(C[k] @ x + 0.05 + np.abs(rng.standard_normal(mks[k])) * 0.5)
        x = A[k] @ x
    p = MPCProblem(
        transition_state_matrix=A,
        transition_input_matrix=B,
        ineq_state_matrix=C,
        ineq_input_matrix=D,
        ineq_vector=e,
        initial_state=x0,
        goal_state=rng.standard_normal(nx),
        nb_timesteps=N,
        terminal_cost_weight=2.0,
        stage_state_cost_weight=0.5,
        stage_input_cost_weight=1e-2,
    )
    p.update_target_states(rng.standard_normal(N * nx))
    return p


def synthetic_ltv_small(seed=3, nx=12, nu=4, N=8):
    """Config-5 family at a fixture-sized horizon (full size N=64 is tested by properties)."""
    rng = np.random.default_rng(seed)
    A, B = [], []
    for _ in range(N):
        Q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
        A.append(0.98 * Q)
        B.append(rng.standard_normal((nx, nu)) / np.sqrt(nx))
    D = np.vstack([np.eye(nu), -np.eye(nu), np.zeros((8, nu))])
    Cs = np.zeros((8, nx))
    for r in range(4):
        Cs[r, r] = 1.0
        Cs[4 + r, r] = -1.0
    C = np.vstack([np.zeros((8, nx)), Cs])
    e = np.concatenate([np.ones(8), 5.0 * np.ones(8)])
    p = MPCProblem(
        transition_state_matrix=A,
        transition_input_matrix=B,
        ineq_state_matrix=C,
        ineq_input_matrix=D,
        ineq_vector=e,
        initial_state=rng.standard_normal(nx),
        goal_state=np.zeros(nx),
        nb_timesteps=N,
        terminal_cost_weight=10.0,
        stage_state_cost_weight=1.0,
        stage_input_cost_weight=1e-2,
    )
    p.update_target_states(np.zeros(N * nx))
    return p


def main():
    # ---- config 1 -------------------------------------------------------
    record("triple_integrator", triple_integrator())
    # two perturbed instances of the config-2 sweep family
    record("triple_integrator_x0a", triple_integrator(x0=(0.31, -0.22, 1.7), goal=(1.2, 0.0, 0.0)))
    record("triple_integrator_x0b", triple_integrator(x0=(-0.45, 0.4, -2.3), goal=(0.6, 0.0, 0.0)))
    # ---- config 4 base + sweep members ---------------------------------
    record("humanoid_one_step", humanoid())
    record("humanoid_x0a", humanoid(x0=(0.02, 0.1, -0.2)))
    # ---- WIP: reference test (known answer U*=0), default N=12 ----------
    record("wip_n12_zero", wip(12, 0.1, np.zeros(4), 0.0)[1])
    record("wip_n12_moving", wip(12, 0.1, np.array([0.05, -0.03, 0.1, 0.08]), 0.5)[1])
    # ---- config 3: N=50 @ T=0.024 (LTI arrays and the same as N-lists) ---
    st = np.array([0.04, 0.06, -0.12, 0.15])
    record("wip_n50_lti", wip(50, 0.024, st, 0.5)[1])
    record("wip_n50_ltv_lists", wip(50, 0.024, st, 0.5, ltv_lists=True)[1])
    # a state far enough for the input box to be active
    record("wip_n50_saturated", wip(50, 0.024, np.array([0.0, 0.35, 0.0, 1.2]), 0.5)[1])
    # ---- random LTV with per-step varying m_k, C and D both set ----------
    record("random_ltv_ragged", random_ltv())
    record("random_ltv_ragged_b", random_ltv(seed=11, nx=4, nu=3, N=6))
    # ---- config-5 family, small horizon -----------------------------------
    record("synthetic_ltv_n8", synthetic_ltv_small())

    # ---- quirk fixtures --------------------------------------
    # quirk 1: constructor drops target_states
    p = triple_integrator()
    q1 = MPCProblem(
        transition_state_matrix=p.transition_state_matrix,
        transition_input_matrix=p.transition_input_matrix,
        ineq_state_matrix=p.ineq_state_matrix,
        ineq_input_matrix=None,
        ineq_vector=p.ineq_vector,
        initial_state=np.array([0.1, 0.0, 0.0]),
        goal_state=np.array([1.0, 0.0, 0.0]),
        nb_timesteps=16,
        terminal_cost_weight=1.0,
        stage_state_cost_weight=0.3,
        stage_input_cost_weight=1e-4,
        target_states=np.ones(48),
    )
    assert q1.target_states is None
    # quirk 2: stage cost set but targets undefined -> q holds only the terminal term
    record("quirk_targets_dropped_partial_q", q1, solve=False)
    # quirk 3: weights below the 1e-10 threshold enter P but not q
    q3 = triple_integrator(x0=(0.2, 0.1, 0.0))
    q3.stage_state_cost_weight = 1e-11
    q3.update_target_states(np.full(48, 0.5))
    record("quirk_weight_threshold", q3, solve=False)
    # quirk 2b: goal undefined with terminal weight set -> q == 0
    q2 = triple_integrator()
    q2.goal_state = None
    record("quirk_goal_undefined_zero_q", q2, solve=False)
    # integer inputs -> float64 outputs
    qi = MPCProblem(
        transition_state_matrix=np.array([[1, 1], [0, 1]]),
        transition_input_matrix=np.array([[0], [1]]),
        ineq_state_matrix=None,
        ineq_input_matrix=np.array([[1], [-1]]),
        ineq_vector=np.array([2, 2]),
        initial_state=np.array([3, 0]),
        goal_state=np.array([0, 0]),
        nb_timesteps=5,
        terminal_cost_weight=1,
        stage_state_cost_weight=None,
        stage_input_cost_weight=1,
    )
    qq = record("integer_inputs", qi)
    assert qq.P.dtype == np.float64 and qq.G.dtype == np.float64

    # ---- update_cost_vector / update_constraint_vector pairs ---------------
    p = humanoid()
    qp = MPCQP(p)
    h0 = qp.h.copy()
    qp.update_constraint_vector(p)
    assert np.array_equal(h0, qp.h)  # reference tests/test_update_constraint_vector.py
    p.update_initial_state(np.array([0.01, -0.05, 0.1]))
    p.update_goal_state(np.array([0.25, 0.0, 0.0]))
    qp.update_cost_vector(p)
    qp.update_constraint_vector(p)
    fresh = MPCQP(p)
    np.savez_compressed(
        os.path.join(OUT, "humanoid_update_vectors.npz"),
        new_initial_state=p.initial_state,
        new_goal_state=p.goal_state,
        q_updated=qp.q,
        h_updated=qp.h,
        q_fresh=fresh.q,
        h_fresh=fresh.h,
    )

    # ---- exception messages (the error convention of the boundary) ---------
    msgs = {}
    try:
        MPCQP(
            MPCProblem(
                transition_state_matrix=np.eye(2),
                transition_input_matrix=np.ones((2, 1)),
                ineq_state_matrix=None,
                ineq_input_matrix=np.ones((1, 1)),
                ineq_vector=np.ones(1),
                nb_timesteps=3,
                terminal_cost_weight=1.0,
                stage_state_cost_weight=None,
                stage_input_cost_weight=1.0,
            )
        )
    except ProblemDefinitionError as exn:
        msgs["initial_state_undefined"] = str(exn)
    for key, kw in (
        ("no_state_cost", dict(terminal_cost_weight=None, stage_state_cost_weight=None, stage_input_cost_weight=1.0)),
        ("nonpositive_input_weight", dict(terminal_cost_weight=1.0, stage_state_cost_weight=None, stage_input_cost_weight=0.0)),
    ):
        try:
            MPCProblem(
                transition_state_matrix=np.eye(2),
                transition_input_matrix=np.ones((2, 1)),
                ineq_state_matrix=None,
                ineq_input_matrix=np.ones((1, 1)),
                ineq_vector=np.ones(1),
                nb_timesteps=3,
                **kw,
            )
        except ProblemDefinitionError as exn:
            msgs[key] = str(exn)
    p5 = random_ltv()
    for key, fn, arg in (
        ("bad_initial_state", p5.update_initial_state, np.zeros(6)),
        ("bad_goal_state", p5.update_goal_state, np.zeros((2, 3))),
        ("bad_target_states", p5.update_target_states, np.zeros(11)),
    ):
        try:
            fn(arg)
        except StateError as exn:
            msgs[key] = str(exn)
    p6 = triple_integrator()
    p6.goal_state = None
    try:
        p6.has_terminal_cost
    except ProblemDefinitionError as exn:
        msgs["goal_undefined"] = str(exn)
    p7 = triple_integrator()
    p7.stage_state_cost_weight = 1.0
    try:
        p7.has_stage_state_cost
    except ProblemDefinitionError as exn:
        msgs["targets_undefined"] = str(exn)
    np.savez_compressed(
        os.path.join(OUT, "exception_messages.npz"), **{k: np.array(v) for k, v in msgs.items()}
    )
    for k, v in msgs.items():
        print(f"  msg[{k}] = {v!r}")

    # ---- WIP plant step (systems/wheeled_inverted_pendulum.py integrate) ----
    pend = WheeledInvertedPendulum()
    rng = np.random.default_rng(5)
    states = rng.standard_normal((16, 4)) * np.array([0.2, 0.3, 0.5, 0.8])
    accels = rng.uniform(-10, 10, 16)
    dt = 0.1 / 15
    nxt = np.array([pend.integrate(s, a, dt) for s, a in zip(states, accels)])
    Aw = pend.build_mpc_problem().transition_state_matrix
    Bw = pend.build_mpc_problem().transition_input_matrix
    np.savez_compressed(
        os.path.join(OUT, "wip_plant.npz"),
        states=states, accels=accels, dt=np.array(dt), next_states=nxt,
        A_default=Aw, B_default=Bw, omega=np.array(pend.omega),
        horizon_duration=np.array(pend.horizon_duration),
    )


if __name__ == "__main__":
    main()
