"""Pin the CPU oracle against the golden vectors captured from the reference.

CPU-only (no GPU): covers oracle/condense_np.py, oracle/mpc_oracle.c.
"""
import os

import numpy as np
import pytest

import oracle
from golden_util import GOLDEN, all_cases, kkt_residuals, load_case

BUILD_KEYS = ("P", "q", "G", "h", "Phi", "Psi", "phi_last", "psi_last", "e")


def _close(a, b, rtol=1e-13):
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    return float(np.abs(a - b).max()) <= rtol * scale if a.size else True


@pytest.mark.parametrize("name", all_cases())
def test_numpy_condense_matches_reference(name):
    p, z = load_case(name)
    cq = oracle.condense(p)
    for key in BUILD_KEYS:
        got, want = getattr(cq, key), z["out_" + key]
        assert got.shape == want.shape, key
        assert got.dtype == np.float64
        # same operations in the same order as mpc_qp.py -> bitwise equal here
        assert np.array_equal(got, want), (name, key, np.abs(got - want).max())


@pytest.mark.parametrize("name", all_cases())
def test_c_condense_matches_reference(name):
    p, z = load_case(name)
    out = oracle.condense_one(p)
    for key in ("P", "q", "G", "h", "Phi", "Psi", "phi_last", "psi_last"):
        assert out[key].shape == z["out_" + key].shape, key
        assert _close(out[key], z["out_" + key]), (name, key)


@pytest.mark.parametrize("name", all_cases(solved_only=True))
def test_gi_matches_certified_solution(name):
    p, z = load_case(name)
    P, q, G, h = z["out_P"], z["out_q"], z["out_G"], z["out_h"]
    x, lam, status, iters = oracle.gi_solve(P, q, G, h)
    assert status == 0
    scale = max(1.0, np.abs(z["U_star"]).max())
    assert np.abs(x - z["U_star"]).max() <= 1e-10 * scale  # (round 6: 1e-8 until the oracle refined its final active set)
    stat, prim, dual, comp = kkt_residuals(P, q, G, h, x, lam)
    qs = 1.0 + np.abs(q).max()
    assert stat <= 1e-9 * qs and prim <= 1e-9 and dual <= 1e-12 and comp <= 1e-8 * qs
    assert abs(0.5 * x @ P @ x + q @ x - float(z["obj_star"])) <= 1e-9 * max(1.0, abs(float(z["obj_star"])))
    assert set(np.nonzero(lam > 1e-9 * (1 + lam.max()))[0]) <= set(z["active_set"].tolist())


def test_reference_known_answer_wip_zero():
    """tests/test_wheeled_inverted_pendulum.py:23-41: x0 = goal = targets = 0
    -> the plan's first input keeps the plant at rest (U* = 0)."""
    p, z = load_case("wip_n12_zero")
    U, status, _ = oracle.solve_mpc_like_reference(p)
    assert status == 0 and np.abs(U).max() <= 6e-8
    assert np.array_equal(z["U_star"], np.zeros(12))


def test_reference_humanoid_first_rows_zero():
    """tests/test_humanoid_one_step.py:72-75: |G[0:2]| == 0 (Psi_0 = 0, D = None)."""
    p, _ = load_case("humanoid_one_step")
    assert np.linalg.norm(oracle.condense(p).G[0:2]) == 0.0
    assert np.linalg.norm(oracle.condense_one(p)["G"][0:2]) == 0.0


def test_reference_update_constraint_vector():
    """tests/test_update_constraint_vector.py:71-80 + the q/h update pair."""
    p, _ = load_case("humanoid_one_step")
    cq = oracle.condense(p)
    assert np.array_equal(cq.h, oracle.constraint_vector(cq, p))
    z = np.load(__import__("os").path.join(__import__("golden_util").GOLDEN, "humanoid_update_vectors.npz"))
    p.update_initial_state(z["new_initial_state"])
    p.update_goal_state(z["new_goal_state"])
    assert np.allclose(oracle.cost_vector(cq, p), z["q_updated"], rtol=0, atol=1e-15)
    assert np.allclose(oracle.constraint_vector(cq, p), z["h_updated"], rtol=0, atol=1e-15)
    assert np.allclose(z["q_updated"], z["q_fresh"], atol=1e-14)


@pytest.mark.parametrize("name", all_cases(solved_only=True))
def test_rollout_matches_plan_states(name):
    p, z = load_case(name)
    X = oracle.integrate(p, p.initial_state, z["plan_inputs"])
    assert np.array_equal(X, z["plan_states"])
    s = oracle.capi.stack_problem(p)
    X2 = oracle.rollout_one(s["A"], s["B"], p.initial_state, z["plan_inputs"])
    assert np.allclose(X2, z["plan_states"], rtol=1e-13, atol=1e-13 * max(1, np.abs(X).max()))


def test_batch_driver_equals_single_calls():
    p, z = load_case("triple_integrator")
    s = oracle.capi.stack_problem(p)
    rng = np.random.default_rng(0)
    B = 5
    x0 = np.stack([z["initial_state"] + 0.1 * rng.standard_normal(3) for _ in range(B)])
    goal = np.tile(z["goal_state"], (B, 1))
    U, lam, status, iters = oracle.build_solve_batch(
        s["nx"], s["nu"], s["N"], s["mk"], s["flags"], s["wt"], s["wx"], s["wu"],
        s["A"], s["B"], s["C"], s["D"], s["e"], x0, goal, None)
    assert (status == 0).all()
    for b in range(B):
        p.update_initial_state(x0[b])
        Ub, st, _ = oracle.solve_mpc_like_reference(p)
        assert np.abs(U[b] - Ub.ravel()).max() <= 1e-9 * max(1, np.abs(Ub).max())


def test_infeasible_and_not_pd_statuses():
    P = np.eye(2)
    x, lam, st, _ = oracle.gi_solve(P, np.zeros(2), np.array([[1.0, 0.0], [-1.0, 0.0]]), np.array([-1.0, -1.0]))
    assert st == 2
    x, lam, st, _ = oracle.gi_solve(np.array([[1.0, 2.0], [2.0, 1.0]]), np.zeros(2), np.zeros((0, 2)), np.zeros(0))
    assert st == 3


# ---------------------------------------------------------------- LIPM walking loop
def test_lipm_schedule_restatement_matches_reference_fixture_bitwise():
    """oracle/lipm_np.py against the schedule captured from the reference example's own PhaseStepper /
    update_goal_and_constraints / integrate (tools/gen_golden_lipm.py): 80 consecutive MPC steps."""
    import os

    from oracle import lipm_np as L

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "lipm_schedule.npz"))
    p = L.parameters()
    assert p["omega"] == float(d["omega"])
    assert np.array_equal(p["zmp_from_state"], d["zmp_from_state"]) and np.array_equal(p["dcm_from_state"], d["dcm_from_state"])
    A, B, C = L.model(p)
    assert np.array_equal(A, d["A"]) and np.array_equal(B, d["B"]) and np.array_equal(C, d["C"])
    assert np.array_equal(L.initial_state(p), d["init_state"])
    w = L.new_walker(p, index=int(d["initial_index"]))
    for s in range(d["e"].shape[0]):
        e, goal = L.goal_and_constraints(p, w)
        assert np.array_equal(e, d["e"][s]) and np.array_equal(goal, d["goal"][s]), s
        assert (w["index"], w["stride_index"], w["support"]) == (d["index"][s], d["stride_index"][s], d["support_pos"][s])
        L.advance(p, w)
    nxt = np.stack([L.integrate(d["plant_state"][i], d["plant_jerk"][i], d["plant_dt"][i]) for i in range(32)])
    assert np.array_equal(nxt, d["plant_next"])


def test_lipm_oracle_loop_walks():
    """The restated loop with the CPU oracle solver: every QP solvable, the CoM follows the footsteps
    and the ZMP stays inside the support foot during single support."""
    from oracle import lipm_np as L

    p = L.parameters()
    w = L.new_walker(p)

    def solve(problem):
        U, st, _ = oracle.solve_mpc_like_reference(problem)
        return None if st != 0 else U[:, 0]

    X, U0, S = L.closed_loop(p, w, L.initial_state(p), 40, solve=solve)
    assert (S == 0).all()
    assert np.abs(X[:, 0]).max() < 0.2  # the CoM sways between footholds at +-0.09
    assert np.ptp(X[:, 0]) > 0.05


# ---------------------------------------------------------------- stage-wise restatement
def _load_stagewise(name):
    from qpmpc_amd import MPCProblem

    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = MPCProblem(z["A"], z["B"], z["C"], None, z["e"], int(z["nb_timesteps"]), float(z["terminal_cost_weight"]),
                   float(z["stage_state_cost_weight"]), float(z["stage_input_cost_weight"]), initial_state=z["initial_state"],
                   goal_state=z["goal_state"])
    p.update_target_states(z["target_states"])
    return p, z


@pytest.mark.parametrize("name", ["stagewise_triple_n64", "stagewise_triple_n256", "stagewise_triple_n1024",
                                  "stagewise_triple_n1024_b"])
def test_stagewise_oracle_matches_reference_built_dense_minimiser(name):
    """The Riccati-based matrix-free active set (oracle/stagewise_np.py) against the minimiser of the QP that
    the REFERENCE condensed (tools/gen_golden_stagewise.py): same inputs, same U*, same active rows; its KKT
    residuals are evaluated without any condensed matrix (adjoint recursion)."""
    from oracle import stagewise_np as S

    p, z = _load_stagewise(name)
    sp = S.from_mpc_problem(p)
    U, lam, st, it = S.solve_stagewise(sp)
    assert st == 0
    Us = z["U_star"]
    assert np.abs(U - Us).max() <= 1e-7 * max(1.0, np.abs(Us).max())  # cond(P) up to 1e9 on the dense side
    assert set(np.flatnonzero(lam > 1e-9)) <= set(z["active_set"].tolist())
    kk = S.kkt_residuals_stagewise(sp, U, lam)
    assert kk["stationarity"] <= 1e-9 and kk["primal"] <= 1e-11 and kk["dual"] == 0.0 and kk["complementarity"] <= 1e-10
    # the fixture's own solution passes the same matrix-free KKT check (pins kkt_residuals_stagewise itself)
    lam_s = np.zeros(lam.shape)
    lam_s[z["active_set"]] = z["lambda_active"]
    kf = S.kkt_residuals_stagewise(sp, Us, lam_s)
    assert kf["stationarity"] <= 1e-8 and kf["primal"] <= 1e-12 and kf["complementarity"] <= 1e-10


def test_stagewise_oracle_equals_dense_oracle_on_the_reference_fixtures():
    """Every certified fixture of the dense path, through the stage-wise restatement: <= 1e-9."""
    from oracle import stagewise_np as S

    for name in all_cases(solved_only=True):
        p, z = load_case(name)
        U, lam, st, it = S.solve_stagewise(S.from_mpc_problem(p))
        assert st == 0, name
        assert np.abs(U - z["U_star"]).max() <= 1e-9 * max(1.0, np.abs(z["U_star"]).max()), name


def test_thin_qr_stagewise_restatement_equals_the_certified_fixtures_and_the_long_horizons():
    """oracle/stagewise_qr_np.py -- the method the wide stage-wise kernel runs since round 6 (whitened Riccati records, thin QR of
    the active rows, evaluations from scratch, the most violated CACHED row first) -- against every certified fixture of the dense
    path (<= 1e-9) and the reference-built long-horizon minimisers (N = 64, 256: <= 1e-7, cond(P) up to 1e9 on the dense side),
    in float64; in float32 arithmetic (what config 5's instantiation computes) against the float64 fixtures at 1e-3."""
    from oracle import stagewise_np as S
    from oracle import stagewise_qr_np as SQ

    for name in all_cases(solved_only=True):
        p, z = load_case(name)
        sp = S.from_mpc_problem(p)
        for rows in (1, 8):  # (one cached row: every iteration is followed by an evaluation; eight: the kernel's general layout)
            U, lam, st, it = SQ.solve_stagewise_qr(sp, cached_rows=rows)
            assert st == 0, (name, rows)
            assert np.abs(U - z["U_star"]).max() <= 1e-9 * max(1.0, np.abs(z["U_star"]).max()), (name, rows)
            kk = S.kkt_residuals_stagewise(sp, U, lam)
            assert kk["stationarity"] <= 1e-8 * (1.0 + np.abs(z["out_q"]).max()) and kk["primal"] <= 1e-9 and kk["dual"] == 0.0, (name, kk)
        U32, _, st32, _ = SQ.solve_stagewise_qr(sp, tol=1e-6, dtype=np.float32)
        assert st32 == 0, name
        assert np.abs(U32 - z["U_star"]).max() <= 1e-3 * max(1.0, np.abs(z["U_star"]).max()), name
    for name in ("stagewise_triple_n64", "stagewise_triple_n256"):
        p, z = _load_stagewise(name)
        U, lam, st, it = SQ.solve_stagewise_qr(S.from_mpc_problem(p))
        assert st == 0
        assert np.abs(U - z["U_star"]).max() <= 1e-7 * max(1.0, np.abs(z["U_star"]).max())


def test_thin_qr_stagewise_restatement_on_nearly_fully_active_problems():
    """Random LTV problems with 4-6 tight rows per step (most of the ~40 variables pinned; the family of tools/stress_tight.py,
    where the explicit-inverse operator of rounds 2-5 lost problems): statuses and plans equal to the dense C oracle's."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from oracle import stagewise_qr_np as SQ
    from oracle.stagewise_np import StageProblem

    rng = np.random.default_rng(3)
    nx, nu, N, mk, B = 6, 2, 20, 5, 6
    A = np.eye(nx) + 0.008 * rng.standard_normal((B, N, nx, nx))
    Bm = rng.standard_normal((B, N, nx, nu))
    Cm = rng.standard_normal((B, N, mk, nx))
    D = rng.standard_normal((B, N, mk, nu))
    x0 = 0.1 * rng.standard_normal((B, nx))
    e = np.zeros((B, N, mk))
    for b in range(B):
        x = x0[b].copy()
        for k in range(N):
            e[b, k] = Cm[b, k] @ x + 0.15 * (0.05 + 0.5 * np.abs(rng.standard_normal(mk)))
            x = A[b, k] @ x
    w = dict(A=A, B=Bm, C=Cm, D=D, e=e, N=N, wt=2.0, wx=0.5, wu=1e-2, x0=x0, goal=rng.standard_normal((B, nx)),
             targets=rng.standard_normal((B, N * nx)))
    Uo, _, sto, _ = oracle.solve_workload(w)
    solved = 0
    for b in range(B):
        sp = StageProblem(A[b], Bm[b], Cm[b], D[b], e[b], x0[b], w["goal"][b], w["targets"][b].reshape(N, nx), 2.0, 0.5, 1e-2)
        U, _, st, _ = SQ.solve_stagewise_qr(sp)
        assert (st == 0) == (sto[b] == 0), (b, st, sto[b])
        if st == 0:
            solved += 1
            assert np.abs(U - Uo[b]).max() <= 1e-7 * max(1.0, np.abs(Uo[b]).max()), b
    assert solved >= 2
