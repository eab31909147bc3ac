"""CPU tests (-m "not gpu") of the host side: the reference-API mirror, the
batched containers' layout logic, the C ABI's exported symbols and the loud
failure without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import qpmpc_amd
from golden_util import GOLDEN, load_case
from qpmpc_amd import (BackendError, BatchMPCProblem, MPCProblem, Plan, ProblemDefinitionError, Solution,
                       StateError, _capi)
from qpmpc_amd import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_public_names_match_reference():
    for name in ("MPCProblem", "MPCQP", "Plan", "solve_mpc"):  # qpmpc/__init__.py:14-19
        assert hasattr(qpmpc_amd, name)
    assert issubclass(qpmpc_amd.ProblemDefinitionError, qpmpc_amd.QPMPCException)
    assert issubclass(qpmpc_amd.StateError, qpmpc_amd.QPMPCException)
    assert issubclass(qpmpc_amd.PlanError, qpmpc_amd.QPMPCException)


def _simple(**kw):
    args = dict(
        transition_state_matrix=np.eye(2), transition_input_matrix=np.ones((2, 1)), ineq_state_matrix=None,
        ineq_input_matrix=np.ones((1, 1)), ineq_vector=np.ones(1), nb_timesteps=3, terminal_cost_weight=1.0,
        stage_state_cost_weight=None, stage_input_cost_weight=1.0)
    args.update(kw)
    return MPCProblem(**args)


def test_exception_messages_match_reference():
    msgs = {k: str(v) for k, v in np.load(os.path.join(GOLDEN, "exception_messages.npz")).items()}
    with pytest.raises(ProblemDefinitionError) as ei:
        _simple(terminal_cost_weight=None)
    assert str(ei.value) == msgs["no_state_cost"]
    with pytest.raises(ProblemDefinitionError) as ei:
        _simple(stage_input_cost_weight=0.0)
    assert str(ei.value) == msgs["nonpositive_input_weight"]
    p, _ = load_case("random_ltv_ragged")
    for key, fn, arg in (("bad_initial_state", p.update_initial_state, np.zeros(6)),
                         ("bad_goal_state", p.update_goal_state, np.zeros((2, 3))),
                         ("bad_target_states", p.update_target_states, np.zeros(11))):
        with pytest.raises(StateError) as ei:
            fn(arg)
        assert str(ei.value) == msgs[key]
    p, _ = load_case("triple_integrator")
    p.goal_state = None
    with pytest.raises(ProblemDefinitionError) as ei:
        p.has_terminal_cost
    assert str(ei.value) == msgs["goal_undefined"]
    p, _ = load_case("triple_integrator")
    p.stage_state_cost_weight = 1.0
    with pytest.raises(ProblemDefinitionError) as ei:
        p.has_stage_state_cost
    assert str(ei.value) == msgs["targets_undefined"]
    # undefined x0 is rejected before any launch (mpc_qp.py:49-50), GPU or not
    with pytest.raises(ProblemDefinitionError) as ei:
        qpmpc_amd.MPCQP(_simple())
    assert str(ei.value) == msgs["initial_state_undefined"]
    with pytest.raises(ProblemDefinitionError):
        qpmpc_amd.solve_mpc(_simple(), solver="hip_gi")


def test_solve_mpc_signature_follows_upstream():
    """`solver` is a required argument (qpmpc/solve_mpc.py:18) and the HIP solver refuses keyword
    arguments it cannot honour instead of dropping them."""
    p = _simple(initial_state=np.zeros(2))
    with pytest.raises(TypeError):
        qpmpc_amd.solve_mpc(p)
    with pytest.raises(TypeError, match="eps_abs"):
        qpmpc_amd.solve_mpc(p, solver="hip_gi", eps_abs=1e-9)


def test_batch_of_problems_with_different_weights_is_refused():
    ps = [_simple(initial_state=np.zeros(2)) for _ in range(3)]
    ps[2].stage_input_cost_weight = 0.5
    with pytest.raises(ProblemDefinitionError, match="share the three cost weights"):
        qpmpc_amd.BatchMPCProblem.from_problems(ps)


def test_problem_container_semantics():
    A = [np.eye(2) * (k + 1) for k in range(3)]
    p = _simple(transition_state_matrix=A, target_states=np.ones(6), initial_state=np.ones((2, 1)), goal_state=np.ones(2))
    assert p.target_states is None  # constructor drops it (mpc_problem.py:129)
    assert p.initial_state.shape == (2,)  # setters flatten
    assert p.get_transition_state_matrix(2) is A[2]  # list => time-varying
    assert p.get_transition_input_matrix(2) is p.transition_input_matrix  # array => shared
    assert p.get_ineq_state_matrix(1) is None
    assert (p.state_dim, p.input_dim, p.nb_timesteps) == (2, 1, 3)
    assert p.has_terminal_cost and not p.has_stage_state_cost
    p.terminal_cost_weight = 1e-11
    assert not p.has_terminal_cost  # threshold of mpc_problem.py:146
    p.update_target_states(np.arange(6.0).reshape(3, 2))
    assert p.target_states.shape == (6,)
    assert "MPCProblem(goal_state=" in repr(p)


def test_plan_semantics_without_gpu():
    p = _simple(initial_state=np.zeros(2))
    empty = Plan(p, Solution(None, found=False))
    assert empty.is_empty and empty.inputs is None and empty.first_input is None and empty.states is None
    full = Plan(p, Solution(None, x=np.arange(3.0), found=True))
    assert full.inputs.shape == (3, 1) and full.first_input[0] == 0.0 and not full.is_empty


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = _simple(initial_state=np.zeros(2))
    with pytest.raises(BackendError):
        qpmpc_amd.solve_mpc(p, solver="hip_gi")
    with pytest.raises(BackendError):
        qpmpc_amd.MPCQP(p)
    with pytest.raises(BackendError):
        p.integrate(np.zeros(2), np.zeros((3, 1)))
    bp = W.to_batch_problem(W.triple_integrator_batch(4), device=torch.device("cpu"))
    with pytest.raises(BackendError):
        qpmpc_amd.solve_mpc_batch(bp)


def test_unknown_solver_without_qpsolvers():
    p = _simple(initial_state=np.zeros(2))
    try:
        import qpsolvers  # noqa: F401
        pytest.skip("qpsolvers installed")
    except ImportError:
        pass
    with pytest.raises(BackendError, match="qpsolvers"):
        qpmpc_amd.solve_mpc(p, solver="proxqp")


def test_c_abi_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mpcqp.h")).read()
    declared = set(re.findall(r"\b(mpcqp_[a-z_]+)\s*\(", header))
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    lib = _capi.load()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.mpcqp_abi_version() == _capi.ABI_VERSION
    assert lib.mpcqp_error_string(-2).decode().startswith("no kernel for these dimensions")
    # host-only entry point: LDS budget of the configs
    b = C.c_size_t(0)
    d = _capi.Dims(3, 1, 16, 2, _capi.F64, 5, 1.0, 0.0, 1e-6)
    assert lib.mpcqp_lds_bytes(C.byref(d), C.byref(b)) == 0 and 0 < b.value <= 160 * 1024
    d = _capi.Dims(4, 1, 50, 2, _capi.F64, 15, 10.0, 1.0, 1e-3)
    assert lib.mpcqp_lds_bytes(C.byref(d), C.byref(b)) == 0 and b.value <= 160 * 1024
    # bad arguments are refused on the host, before any launch
    assert lib.mpcqp_solve_batch(0, 0, 0, None, None, None, None, 1, None, None, None, None, None, None, 0, None) == -1
    assert lib.mpcqp_build_solve_batch(None, None, 1, None, None, None, None, None, None, 0, None) == -1
    # (ABI 7) several control periods per launch: the period count is checked first, then the arguments like the one-period call
    d3 = _capi.Dims(4, 1, 50, 2, _capi.F64, 15, 10.0, 1.0, 1e-3)
    args = (C.byref(d3), None, 1024, None, None, None, None, None, None, 0, None, None, 0.024, 0.5, 1.0, 9.81, 15)
    assert lib.mpcqp_wip_periods_batch(*args, 0, None) == _capi.EINVAL
    assert lib.mpcqp_wip_periods_batch(*args, 20, None) == _capi.EINVAL  # (NULL problem)
    assert lib.mpcqp_wip_period_batch(*args, None) == _capi.EINVAL
    # (ABI 9) the pairing order's counting sort: scratch of 1024 int32 per chunk of 4096 problems; arguments checked first
    assert lib.mpcqp_order_workspace_bytes(65536) == 16 * 1024 * 4 and lib.mpcqp_order_workspace_bytes(1) == 4096
    assert lib.mpcqp_order_workspace_bytes(0) == 0
    assert lib.mpcqp_order_by_count(None, 8, None, None, 0, None) == _capi.EINVAL
    assert lib.mpcqp_order_by_count(8, 8, 8, None, 0, None) == _capi.EWORKSPACE  # (non-NULL dummies: refused before any launch)
    assert C.sizeof(_capi.SolveOpts) == 64  # (the header's struct: 2 x int32, double, ptr, 2 x int32, ptr, size_t, 2 x int32, ptr)
    # workspace queries are host-only: config 2 needs none, config 5 (n=256, m=1024, f32) does
    d = _capi.Dims(3, 1, 16, 2, _capi.F64, 5, 1.0, 0.0, 1e-6)
    assert lib.mpcqp_workspace_bytes(C.byref(d), 4096, 1, C.byref(b)) == 0 and b.value == 0
    d = _capi.Dims(12, 4, 64, 16, _capi.F32, 15, 10.0, 1.0, 1e-2)
    assert lib.mpcqp_workspace_bytes(C.byref(d), 2, 1, C.byref(b)) == 0 and b.value > 2 * 1_000_000
    assert lib.mpcqp_workspace_bytes(C.byref(d), 2, 0, C.byref(b)) == 0 and b.value == 2 * 780 * 257 * 4


def test_batch_container_layout_and_flags():
    cpu = torch.device("cpu")
    w = W.triple_integrator_batch(8)
    bp = W.to_batch_problem(w, device=cpu)
    assert (bp.batch_size, bp.nb_variables, bp.nb_constraints) == (8, 16, 32)
    cp = bp.c_problem()
    assert (cp.A.batch_stride, cp.A.step_stride) == (16 * 9, 9)
    assert (cp.e.batch_stride, cp.e.step_stride) == (32, 2)
    assert cp.D.ptr is None and cp.targets.ptr is None
    assert (cp.x0.batch_stride, cp.goal.batch_stride) == (3, 3)
    assert bp.cost_flags() == _capi.P_TERMINAL | _capi.Q_TERMINAL
    ws = W.triple_integrator_batch(8, heterogeneous=False)
    cs = W.to_batch_problem(ws, device=cpu).c_problem()
    assert (cs.A.batch_stride, cs.A.step_stride, cs.C.batch_stride, cs.C.step_stride) == (0, 0, 0, 0)
    wh = W.humanoid_batch(5)
    ch = W.to_batch_problem(wh, device=cpu).c_problem()
    assert (ch.e.batch_stride, ch.e.step_stride) == (0, 2)  # shared, time-varying
    ww = W.wip_batch(3)
    bw = W.to_batch_problem(ww, device=cpu)
    assert bw.cost_flags() == 15 and bw.c_problem().C.ptr is None
    assert (bw.c_problem().A.batch_stride, bw.c_problem().A.step_stride) == (0, 16)
    # flags: weight below threshold enters P but not q; missing targets drop the stage term of q
    bw.stage_state_cost_weight = 1e-11
    assert bw.cost_flags() == _capi.P_TERMINAL | _capi.P_STAGE | _capi.Q_TERMINAL
    bw.stage_state_cost_weight = 1.0
    bw.target_states = None
    assert bw.cost_flags() == _capi.P_TERMINAL | _capi.P_STAGE | _capi.Q_TERMINAL
    bw.goal_state = None
    assert bw.cost_flags() == _capi.P_TERMINAL | _capi.P_STAGE
    with pytest.raises(StateError):
        bw.update_initial_state(np.zeros((3, 5)))
    with pytest.raises(StateError):
        bw.update_target_states(np.zeros((3, 7)))
    with pytest.raises(ProblemDefinitionError):
        BatchMPCProblem(np.eye(2), np.ones((2, 1)), None, np.ones((1, 1)), np.ones(1), 3, None, None, 1.0,
                        np.zeros((4, 2)), device=cpu)
    with pytest.raises(ProblemDefinitionError):
        BatchMPCProblem(np.eye(2), np.ones((2, 1)), None, np.ones((1, 1)), np.ones(1), 3, 1.0, None, -1.0,
                        np.zeros((4, 2)), device=cpu)


def test_from_problems_pads_ragged_rows():
    p, z = load_case("random_ltv_ragged")
    bp = BatchMPCProblem.from_problems([p, p], device=torch.device("cpu"))
    assert bp.ineq_dim == 4 and bp.batch_size == 2
    assert len(bp.valid_rows) == z["out_G"].shape[0]
    e = bp.e.numpy()
    assert (e[0, 0, 1:] == 1e30).all() and e[0, 3, 3] != 1e30  # m_k = 1,2,3,4,1,2,3
    assert (bp.C.numpy()[0, 0, 1:] == 0).all()


def test_workload_generators_follow_the_survey():
    w = W.triple_integrator_batch(4096)
    assert W.algorithmic_bytes_per_problem(w) == 342 * 8  # 342 doubles
    assert abs(W.algorithmic_build_flops(3, 1, 16, 2, False, True) - 1.1e4) < 3e3
    assert np.abs(w["x0"][:, 2]).max() < 3.0  # k=0 row stays feasible
    z = np.load(os.path.join(GOLDEN, "triple_integrator.npz"))
    A, B, Cm, e = W.triple_integrator_matrices()
    assert np.array_equal(A, z["A"]) and np.array_equal(B, z["B"]) and np.array_equal(Cm, z["C"])
    zh = np.load(os.path.join(GOLDEN, "humanoid_one_step.npz"))
    A, B, Cm, e, goal = W.humanoid_matrices()
    assert np.array_equal(A, zh["A"]) and np.array_equal(Cm, zh["C"]) and np.array_equal(goal, zh["goal_state"])
    for k in range(16):
        assert np.array_equal(e[k], zh[f"e_{k}"])
    s = W.synthetic_ltv_batch(2, N=5)
    assert s["A"].shape == (2, 5, 12, 12) and s["e"].shape == (16,)
    assert np.allclose(np.abs(np.linalg.eigvals(s["A"][0, 0])), 0.98)


def test_wip_model_constants_match_reference():
    from qpmpc_amd.systems import WheeledInvertedPendulum

    z = np.load(os.path.join(GOLDEN, "wip_plant.npz"))
    pend = WheeledInvertedPendulum()
    A, B = pend.discretized_dynamics()
    assert np.array_equal(A, z["A_default"]) and np.array_equal(B, z["B_default"])
    assert pend.omega == float(z["omega"]) and pend.horizon_duration == float(z["horizon_duration"])
    assert pend.horizon_duration > 0.1 and pend.omega > 0.1  # test_wheeled_inverted_pendulum.py:19-21
    nxt = np.array([pend.integrate(s, a, float(z["dt"])) for s, a in zip(z["states"], z["accels"])])
    np.testing.assert_allclose(nxt, z["next_states"], rtol=0, atol=1e-15)
    got = pend.integrate_batch(torch.tensor(z["states"]), torch.tensor(z["accels"]), float(z["dt"]))
    np.testing.assert_allclose(got.numpy(), z["next_states"], rtol=1e-14, atol=1e-15)
    prob = pend.build_mpc_problem(terminal_cost_weight=10.0, stage_state_cost_weight=1.0)
    zz = np.load(os.path.join(GOLDEN, "wip_n12_zero.npz"))
    assert np.array_equal(prob.ineq_input_matrix, zz["D"]) and np.array_equal(prob.ineq_vector, zz["e"])
    ts = pend.target_states(np.array([0.05, -0.03, 0.1, 0.08]), 0.5)
    zm = np.load(os.path.join(GOLDEN, "wip_n12_moving.npz"))
    assert np.array_equal(ts[:-4], zm["target_states"]) and np.array_equal(ts[-4:], zm["goal_state"])


def test_hand_written_dpp_instructions_keep_their_wait_states():
    """The v_fmac_f64_dpp of mpcqp_pair.hip and mpcqp_quad.hip (+ its second unit, mpcqp_quadw.hip) are inline asm: the compiler cannot insert the two wait states a DPP read needs
    after a VALU write of the same register, the source does (dpp_ready). tools/check_dpp_hazards.py verifies it on the
    gfx950 assembly of every instantiation (hipcc cross-compiles without a GPU), and flags a made-up violation."""
    import os, sys

    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import check_dpp_hazards as chk

    bad, ndpp, _ = chk.check("f:\n\tv_add_f64 v[0:1], v[2:3], v[4:5]\n\tv_fmac_f64_dpp v[6:7], v[0:1], v[8:9] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
    assert ndpp == 1 and len(bad) == 1
    bad, _, _ = chk.check("f:\n\tv_add_f64 v[0:1], v[2:3], v[4:5]\n\ts_nop 1\n\tv_fmac_f64_dpp v[6:7], v[0:1], v[8:9] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
    assert not bad
    # (mpcqp_stage.hip used the same instruction in round 4; its sweeps and recursion run on v_mfma_f64_4x4x4 since round 5, whose
    # wait states the compiler inserts itself)
    # (the units are compiled to assembly side by side: mpcqp_quadw.hip -- the wide instantiations of mpcqp_quad.hip -- and
    # mpcqp_quad4.hip -- its four-rows-per-lane copy -- take two to three minutes each)
    from concurrent.futures import ThreadPoolExecutor

    units = ("mpcqp_pair.hip", "mpcqp_quad.hip", "mpcqp_quadw.hip", "mpcqp_quad4.hip", "mpcqp_quad4w.hip")
    with ThreadPoolExecutor(max_workers=5) as pool:
        asms = list(pool.map(lambda u: chk.device_asm(os.path.join(os.path.dirname(tools), "qpmpc_amd", "csrc", u)), units))
    for unit, asm in zip(units, asms):
        bad, ndpp, nasm = chk.check(asm)
        assert nasm > 1000 and not bad, (unit, bad[:3])


def test_integration_md_shows_the_build_command_of_the_build_script():
    """INTEGRATION.md section 1 is the one-line hipcc command a maintainer copies. It once missed a translation unit (a library
    that does not load). It is generated -- `python -m qpmpc_amd.build --print-command` -- and compared here."""
    import os

    from qpmpc_amd import build

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    assert build.one_line_command() in text
    for unit in build._UNITS:
        assert os.path.exists(os.path.join(root, "qpmpc_amd", "csrc", unit))
