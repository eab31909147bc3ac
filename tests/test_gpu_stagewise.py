"""GPU tests (-m gpu) of the stage-wise formulation (``mpcqp_stagewise_solve_batch``): the same
QP as the condensed path, solved without forming P or G (Riccati-based matrix-free dual active set). Pinned on
(a) fixtures whose minimiser comes from the REFERENCE-built dense QP at N = 64 / 256 / 1024
(tools/gen_golden_stagewise.py), (b) the condensed HIP path on BASELINE configs 2 and 3, (c) the NumPy
restatement oracle/stagewise_np.py including its matrix-free KKT residuals."""
import os

import numpy as np
import pytest

import oracle
from golden_util import GOLDEN

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _load(name):
    from qpmpc_amd import MPCProblem

    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = MPCProblem(z["A"], z["B"], z["C"], None, z["e"], int(z["nb_timesteps"]), float(z["terminal_cost_weight"]),
                   float(z["stage_state_cost_weight"]), float(z["stage_input_cost_weight"]), initial_state=z["initial_state"],
                   goal_state=z["goal_state"])
    p.update_target_states(z["target_states"])
    return p, z


@pytest.mark.parametrize("name", ["stagewise_triple_n64", "stagewise_triple_n256", "stagewise_triple_n1024",
                                  "stagewise_triple_n1024_b"])
def test_stagewise_kernel_matches_reference_built_minimiser(name):
    from oracle import stagewise_np as S
    from qpmpc_amd import BatchMPCProblem, solve_mpc_batch

    p, z = _load(name)
    bp = BatchMPCProblem.from_problems([p] * 3)  # three copies: every wavefront must give the same answer
    plan = solve_mpc_batch(bp, formulation="stagewise", return_multipliers=True)
    torch.cuda.synchronize()
    st = plan.status.cpu().numpy()
    assert (st == 0).all(), st
    U = plan.U.cpu().numpy()
    lam = plan.multipliers.cpu().numpy()
    assert np.array_equal(U[0], U[1]) and np.array_equal(U[0], U[2])
    Us = z["U_star"]
    assert np.abs(U[0] - Us).max() <= 1e-7 * max(1.0, np.abs(Us).max())  # cond(P) up to 9e8 on the dense side
    sp = S.from_mpc_problem(p)
    Uo, lo, sto, ito = S.solve_stagewise(sp)
    # (the kernel's Riccati products sum their four terms in a lane-dependent order: at N = 1024, cond(P) ~ 1e9, the two
    # float64 evaluations of the same method differ by a few 1e-9)
    assert np.abs(U[0] - Uo).max() <= (1e-8 if p.nb_timesteps >= 1024 else 1e-9) * max(1.0, np.abs(Uo).max())
    assert int(plan.iters[0].item()) == ito  # same method, same pivots
    kk = S.kkt_residuals_stagewise(sp, U[0], lam[0])  # the kernel's own (u, lambda), no condensed matrix involved
    assert kk["stationarity"] <= 1e-8 and kk["primal"] <= 1e-10 and kk["dual"] == 0.0 and kk["complementarity"] <= 1e-9, kk
    assert set(np.flatnonzero(lam[0] > 1e-9)) <= set(z["active_set"].tolist())


def test_stagewise_equals_condensed_path_on_configs_2_and_3():
    from qpmpc_amd import solve_mpc_batch
    from qpmpc_amd import workloads as W

    for w, nb in ((W.triple_integrator_batch(512), 512), (W.wip_batch(128), 128), (W.humanoid_batch(512), 512)):
        bp = W.to_batch_problem(w)
        dense = solve_mpc_batch(bp)
        stage = solve_mpc_batch(bp, formulation="stagewise")
        torch.cuda.synchronize()
        sd, ss = dense.status.cpu().numpy(), stage.status.cpu().numpy()
        assert np.array_equal(sd, ss), (w["name"], np.flatnonzero(sd != ss)[:10])
        ok = sd == 0
        Ud, Us = dense.U.cpu().numpy()[ok], stage.U.cpu().numpy()[ok]
        scale = np.maximum(1.0, np.abs(Ud).max(axis=1, keepdims=True))
        assert (np.abs(Ud - Us) / scale).max() <= 1e-8, w["name"]
        assert (stage.U.cpu().numpy()[~ok] == 0).all()


def test_stagewise_runs_where_the_condensed_path_is_too_large():
    """N = 1024 (n = 1024 > 256, where the dense kernels stop): the stage-wise path solves a batch of 64 different problems
    -- through the default entry point too --; every solution is KKT-certified without any condensed matrix."""
    from oracle import stagewise_np as S
    from qpmpc_amd import BackendError, BatchMPCProblem, solve_mpc_batch

    p, _ = _load("stagewise_triple_n1024")
    rng = np.random.default_rng(11)
    B = 64
    bp = BatchMPCProblem.from_problems([p])
    x0 = np.stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B), rng.uniform(-1.5, 1.5, B)], 1)
    goal = np.stack([rng.uniform(-4, 4, B), np.zeros(B), np.zeros(B)], 1)
    big = BatchMPCProblem(bp.A[0, 0], bp.B[0, 0], bp.C[0, 0], None, bp.e[0, 0], 1024, p.terminal_cost_weight,
                          p.stage_state_cost_weight, p.stage_input_cost_weight, x0, goal_state=goal,
                          target_states=np.tile(goal, (1, 1024)))
    # (round 3) the default entry point serves it as well: what does not fit on chip goes to the stage-wise kernels
    auto = solve_mpc_batch(big)
    plan = solve_mpc_batch(big, formulation="stagewise", return_multipliers=True)
    torch.cuda.synchronize()
    assert (auto.status.cpu().numpy() == 0).all()
    assert np.abs(auto.U.cpu().numpy() - plan.U.cpu().numpy()).max() <= 1e-9 * max(1.0, float(plan.U.abs().max()))
    st = plan.status.cpu().numpy()
    assert (st == 0).all(), st
    U, lam = plan.U.cpu().numpy(), plan.multipliers.cpu().numpy()
    assert (lam > 0).sum(axis=1).max() >= 20  # the boxes are active
    for b in range(0, B, 7):
        p.update_initial_state(x0[b])
        p.update_goal_state(goal[b])
        p.update_target_states(np.tile(goal[b], 1024))
        sp = S.from_mpc_problem(p)
        kk = S.kkt_residuals_stagewise(sp, U[b], lam[b])
        assert kk["stationarity"] <= 1e-7 and kk["primal"] <= 1e-9 and kk["dual"] == 0.0 and kk["complementarity"] <= 1e-8, (b, kk)


def test_stagewise_refuses_unsupported_dimensions():
    """nx = 20 is served by the general stage-wise kernel since round 4 (float64); what has no stage-wise kernel at all is a
    system wider than 32 states: MPCQP_EUNSUPPORTED from the entry point, MPCQP_ETOOLARGE from the default dispatch once the
    dense path cannot hold it either (n > 256)."""
    from qpmpc_amd import BackendError, solve_mpc_batch
    from qpmpc_amd import workloads as W

    def widened(nx, N):
        w = W.synthetic_ltv_batch(2, N=N)
        pad = nx - 12
        w["A"] = np.concatenate([np.concatenate([w["A"], np.zeros((2, N, 12, pad))], axis=3), np.zeros((2, N, pad, nx))], axis=2)
        w["B"] = np.concatenate([w["B"], np.zeros((2, N, pad, 4))], axis=2)
        w["C"] = np.concatenate([w["C"], np.zeros((16, pad))], axis=1)
        w["x0"] = np.concatenate([w["x0"], np.zeros((2, pad))], axis=1)
        w["goal"], w["targets"] = np.zeros(nx), np.zeros(N * nx)
        return w

    ok = solve_mpc_batch(W.to_batch_problem(widened(20, 8)), formulation="stagewise")
    torch.cuda.synchronize()
    assert (ok.status.cpu().numpy() == 0).all()
    with pytest.raises(BackendError, match="-6"):
        solve_mpc_batch(W.to_batch_problem(widened(40, 8)), formulation="stagewise")
    with pytest.raises(BackendError, match="-2"):
        solve_mpc_batch(W.to_batch_problem(widened(40, 80)))


# ---------------------------------------------------------------- wider systems (mpcqp_stagew.hip)
def _random_ltv(rng, B, nx, nu, N, mk):
    A = np.eye(nx) + 0.08 * rng.standard_normal((B, N, nx, nx))
    Bm = rng.standard_normal((B, N, nx, nu))
    Cm = rng.standard_normal((B, N, mk, nx))
    D = rng.standard_normal((B, N, mk, nu))
    x0 = 0.1 * rng.standard_normal((B, nx))
    e = np.zeros((B, N, mk))
    for b in range(B):
        x = x0[b].copy()
        for k in range(N):
            e[b, k] = Cm[b, k] @ x + 0.05 + 0.5 * np.abs(rng.standard_normal(mk))
            x = A[b, k] @ x
    return dict(A=A, B=Bm, C=Cm, D=D, e=e, N=N, wt=2.0, wx=0.5, wu=1e-2, x0=x0,
                goal=rng.standard_normal((B, nx)), targets=rng.standard_normal((B, N * nx)))


@pytest.mark.parametrize("nx,nu,N,mk", [(6, 2, 20, 3), (5, 3, 33, 2), (9, 4, 16, 4), (16, 4, 12, 3), (12, 1, 70, 2)])
def test_wide_stagewise_kernel_random_ltv_vs_oracles(nx, nu, N, mk):
    """nx > 4 or nu > 2: the wide stage-wise kernel, float64, against the dense C oracle and (iteration counts included) against
    the NumPy restatement of its method -- thin QR of the whitened active rows, evaluations from scratch, the most violated
    CACHED row first (oracle/stagewise_qr_np.py; eight right-hand sides per backward sweep in the general constraint layout)."""
    from oracle import stagewise_np as S
    from oracle import stagewise_qr_np as SQ
    from qpmpc_amd import solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(100 * nx + N)
    w = _random_ltv(rng, 24, nx, nu, N, mk)
    plan = solve_mpc_batch(W.to_batch_problem(w), formulation="stagewise", return_multipliers=True)
    torch.cuda.synchronize()
    U, st, lam = plan.U.cpu().numpy(), plan.status.cpu().numpy(), plan.multipliers.cpu().numpy()
    Uo, _, sto, _ = oracle.solve_workload(w)
    assert np.array_equal(st == 0, sto == 0), (st, sto)
    ok = sto == 0
    assert ok.sum() >= 18
    scale = np.maximum(1.0, np.abs(Uo[ok]).max(axis=1, keepdims=True))
    assert (np.abs(U[ok] - Uo[ok]) / scale).max() <= 1e-7
    for b in np.flatnonzero(ok)[:4]:
        sp = S.from_mpc_problem(W.problem_from_workload(w, int(b)))
        Us, ls, sts, its = SQ.solve_stagewise_qr(sp, cached_rows=8)
        assert sts == 0 and int(plan.iters[b].item()) == its
        assert np.abs(U[b] - Us).max() <= 1e-9 * max(1.0, np.abs(Us).max())
        kk = S.kkt_residuals_stagewise(sp, U[b], lam[b])
        assert kk["stationarity"] <= 1e-9 and kk["primal"] <= 1e-10 and kk["dual"] == 0.0, kk


def test_wide_stagewise_config5_dimensions_f64_and_f32():
    """BASELINE configs[4]'s problems (nx = 12, nu = 4, N = 64: n = 256, m = 1024) WITHOUT P, G or any factor of them:
    float64 against the dense oracle at 1e-7, float32 at the config's 1e-3, and against the condensed float32 HIP path."""
    from oracle.parallel import solve_workload_parallel
    from qpmpc_amd import solve_mpc_batch
    from qpmpc_amd import workloads as W
    from qpmpc_amd.distributed import shard_workload

    w = W.synthetic_ltv_batch(96)
    Uo, _, sto, _ = solve_workload_parallel(w, shard_workload)
    assert (sto == 0).all()
    scale = np.maximum(1.0, np.abs(Uo).max(axis=1, keepdims=True))
    p64 = solve_mpc_batch(W.to_batch_problem(w), formulation="stagewise")
    p32 = solve_mpc_batch(W.to_batch_problem(w, dtype=torch.float32), formulation="stagewise")
    d32 = solve_mpc_batch(W.to_batch_problem(w, dtype=torch.float32))
    torch.cuda.synchronize()
    assert (p64.status == 0).all() and (p32.status == 0).all()
    assert (np.abs(p64.U.cpu().numpy() - Uo) / scale).max() <= 1e-7
    assert (np.abs(p32.U.double().cpu().numpy() - Uo) / scale).max() <= 1e-3
    assert (np.abs(p32.U.double().cpu().numpy() - d32.U.double().cpu().numpy()) / scale).max() <= 2e-3


def test_wide_kernel_equals_narrow_kernel_where_both_apply():
    """MPCQP_OPT_STAGE_WIDE sends a small system (config 3's wheeled inverted pendulum, nx = 4, nu = 1, float64) through the
    wide kernel (MFMA sweeps, packed [C | D]): same minimiser, same iteration counts as the narrow kernel."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    bp = W.to_batch_problem(W.wip_batch(512, seed=11))
    a = solve_mpc_batch(bp, formulation="stagewise", return_multipliers=True)
    b = solve_mpc_batch(bp, formulation="stagewise", return_multipliers=True, flags=_capi.OPT_STAGE_WIDE)
    torch.cuda.synchronize()
    assert torch.equal(a.status, b.status) and (a.status == 0).all()
    assert torch.equal(a.iters, b.iters) and int(a.iters.max()) >= 2
    assert float((a.U - b.U).abs().max()) <= 1e-9 * max(1.0, float(a.U.abs().max()))
    assert float((a.multipliers - b.multipliers).abs().max()) <= 1e-7 * max(1.0, float(a.multipliers.abs().max()))


def test_large_problems_are_dispatched_to_the_stagewise_kernel_and_agree_with_the_condensed_path():
    """mpcqp_build_solve_batch hands problems that do not fit the on-chip kernels to the wide stage-wise kernel;
    MPCQP_OPT_FORCE_CONDENSED keeps the HBM-resident dense path. Both against the oracle; PreparedSolve(formulation=
    'stagewise') reuses its workspace and returns the same plan as the one-shot call."""
    from oracle.parallel import solve_workload_parallel
    from qpmpc_amd import PreparedSolve, _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W
    from qpmpc_amd.distributed import shard_workload

    w = W.synthetic_ltv_batch(64, seed=5)
    Uo, _, sto, _ = solve_workload_parallel(w, shard_workload)
    scale = np.maximum(1.0, np.abs(Uo).max(axis=1, keepdims=True))
    for dt, tol in ((torch.float64, 1e-7), (torch.float32, 1e-3)):
        bp = W.to_batch_problem(w, dtype=dt)
        auto = solve_mpc_batch(bp)
        dense = solve_mpc_batch(bp, flags=_capi.OPT_FORCE_CONDENSED)
        explicit = solve_mpc_batch(bp, formulation="stagewise", max_active=256)
        ps = PreparedSolve(bp, formulation="stagewise", max_active=256)
        ps.launch()
        ps.launch()
        torch.cuda.synchronize()
        assert torch.equal(auto.U, explicit.U) and torch.equal(auto.iters, explicit.iters)  # the same kernel ran
        assert torch.equal(ps.plan.U, explicit.U)
        for plan in (auto, dense):
            assert np.array_equal(plan.status.cpu().numpy() == 0, sto == 0)
            assert (np.abs(plan.U.double().cpu().numpy() - Uo) / scale).max() <= tol
        if dt == torch.float32:  # nothing is squared into P: the stage-wise path is the more accurate one in float32
            e_auto = (np.abs(auto.U.double().cpu().numpy() - Uo) / scale).max()
            e_dense = (np.abs(dense.U.double().cpu().numpy() - Uo) / scale).max()
            assert e_auto <= 5e-5 and e_auto <= e_dense


@pytest.mark.parametrize("nx,nu,N,mk", [(6, 2, 20, 3), (9, 4, 16, 4), (16, 4, 12, 3), (12, 1, 70, 2), (7, 3, 5, 5)])
def test_wide_stagewise_kernel_float32_random_ltv(nx, nu, N, mk):
    """float32 through every row-length instantiation (nx rounded up to 8, 12, 16; one- and two-block stacked matrices;
    horizons shorter than the sweeps' request ring; mk that does not divide 64; C, D changing along the horizon):
    against the float64 C oracle at 2e-4 relative, statuses equal."""
    from qpmpc_amd import solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(7 * nx + N)
    w = _random_ltv(rng, 32, nx, nu, N, mk)
    plan = solve_mpc_batch(W.to_batch_problem(w, dtype=torch.float32), formulation="stagewise")
    torch.cuda.synchronize()
    U, st = plan.U.double().cpu().numpy(), plan.status.cpu().numpy()
    Uo, _, sto, _ = oracle.solve_workload(w)
    ok = (sto == 0) & (st == 0)
    assert ok.sum() >= 24 and (st[sto == 0] == 0).mean() >= 0.9
    scale = np.maximum(1.0, np.abs(Uo[ok]).max(axis=1, keepdims=True))
    assert (np.abs(U[ok] - Uo[ok]) / scale).max() <= 2e-4


def test_wide_stagewise_kernel_infeasible_and_slot_overflow_statuses():
    """Contradictory rows (u_0 <= -1 and -u_0 <= -1) are reported MPCQP_INFEASIBLE with a zeroed plan, like the dense path;
    a problem that needs more active rows than max_active allows comes back unsolved (SLOTS_FULL), never a wrong plan."""
    from qpmpc_amd import solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(3)
    w = _random_ltv(rng, 8, 6, 2, 12, 3)
    w["C"][4:, 0, 0, :] = 0.0
    w["C"][4:, 0, 1, :] = 0.0
    w["D"][4:, 0, 0, :] = [1.0, 0.0]
    w["D"][4:, 0, 1, :] = [-1.0, 0.0]
    w["e"][4:, 0, 0] = -1.0
    w["e"][4:, 0, 1] = -1.0
    bp = W.to_batch_problem(w)
    plan = solve_mpc_batch(bp, formulation="stagewise")
    dense = solve_mpc_batch(bp, flags=0)
    torch.cuda.synchronize()
    st = plan.status.cpu().numpy()
    assert (st[:4] == 0).all() and (st[4:] == 2).all(), st  # MPCQP_INFEASIBLE = 2
    assert np.array_equal(st, dense.status.cpu().numpy())
    assert float(plan.U[4:].abs().max()) == 0.0
    assert float((plan.U[:4] - dense.U[:4]).abs().max()) <= 1e-8
    # slot overflow: one slot only
    tight = solve_mpc_batch(bp, formulation="stagewise", max_active=1, retry_slots=False)
    torch.cuda.synchronize()
    need = plan.iters.cpu().numpy()
    st1 = tight.status.cpu().numpy()
    for b in range(4):
        if st1[b] == 0:
            assert float((tight.U[b] - plan.U[b]).abs().max()) <= 1e-9
        else:
            assert st1[b] == 4 and need[b] > 1 and float(tight.U[b].abs().max()) == 0.0  # MPCQP_SLOTS_FULL = 4
    assert (st1[:4] != 0).any()
    # ... and the host solves those again with more slots (the default)
    again = solve_mpc_batch(bp, formulation="stagewise", max_active=1)
    torch.cuda.synchronize()
    assert np.array_equal(again.status.cpu().numpy(), st)
    assert float((again.U - plan.U).abs().max()) <= 1e-9


def test_stagewise_kernels_keep_going_when_every_slot_is_taken():
    """With nu = 1 the active set can reach n rows (the plan fully determined by constraints) and still has to SWAP rows
    to get to the minimiser: drops must be possible with full slots (a stress run, tools/stress_stagewise.py, found
    problems of this family reported MAX_ITER while the condensed path solved them). Same random sequence as that run."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(12345)

    def draw(B, nx, nu, N, mk, tight):
        A = np.eye(nx) + 0.08 * rng.standard_normal((B, N, nx, nx))
        Bm = rng.standard_normal((B, N, nx, nu))
        Cm = rng.standard_normal((B, N, mk, nx))
        D = rng.standard_normal((B, N, mk, nu))
        x0 = 0.1 * rng.standard_normal((B, nx))
        e = np.zeros((B, N, mk))
        for b in range(B):
            x = x0[b].copy()
            for k in range(N):
                e[b, k] = Cm[b, k] @ x + tight * (0.05 + 0.5 * np.abs(rng.standard_normal(mk)))
                x = A[b, k] @ x
        return dict(A=A, B=Bm, C=Cm, D=D, e=e, N=N, wt=2.0, wx=0.5, wu=1e-2, x0=x0,
                    goal=rng.standard_normal((B, nx)), targets=rng.standard_normal((B, N * nx)))

    w = None
    for _ in range(7):  # the seventh problem family of the stress run: nx = 14, nu = 1, N = 60, mk = 6
        nx, nu = int(rng.integers(2, 17)), int(rng.integers(1, 5))
        N = min(int(rng.integers(3, max(4, 256 // nu))), 60)
        mk = int(rng.integers(1, 7))
        w = draw(256, nx, nu, N, mk, float(rng.choice([0.2, 1.0, 3.0])))
    assert (nx, nu, N, mk) == (14, 1, 60, 6)
    bp = W.to_batch_problem(w)
    dense = solve_mpc_batch(bp, flags=_capi.OPT_FORCE_CONDENSED)
    wide = solve_mpc_batch(bp, formulation="stagewise", max_active=60)
    torch.cuda.synchronize()
    assert (dense.status == 0).all() and (wide.status == 0).all()
    assert int(wide.iters.max()) > 60  # more iterations than slots: rows were swapped
    scale = dense.U.abs().amax(dim=1).clamp(min=1.0)
    assert float(((dense.U - wide.U).abs().amax(dim=1) / scale).max()) <= 1e-9


@pytest.mark.parametrize("case", ["narrow_serial", "narrow_scan", "wide_f64", "wide_f32", "wide_nx16"])
def test_indefinite_hessian_is_status_not_pd_on_the_mpc_path(case):
    """A negative state weight makes the condensed Hessian indefinite (mpc_problem.py:104-107 only checks w_u > 0). The
    condensed kernels' Cholesky reports MPCQP_NOT_PD (3); the stage-wise kernels -- which mpcqp_build_solve_batch picks by
    itself for 16 < n <= 128 (small systems) and for everything that does not fit on chip -- must report the same (the
    pivots S_k of the Riccati recursion are the Schur complements of P), not decay through NaN to MAX_ITER or return a
    stationary non-minimiser as solved. Same batch with its weight restored: solved again (nothing sticks)."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    if case == "narrow_serial":
        w, dt = W.wip_batch(6), torch.float64  # nx = 4, nu = 1, N = 50: narrow kernel, serial sweeps
    elif case == "narrow_scan":
        w, dt = W.wip_batch(6, N=120), torch.float64  # chunked scans (the factor image does not fit LDS)
    elif case == "wide_nx16":
        rng = np.random.default_rng(5)
        w, dt = W.synthetic_ltv_batch(4, N=20), torch.float64  # nx = 16: the tiled (LDS) recursion of the wide kernel
        A = np.zeros((4, 20, 16, 16))
        A[:, :, :12, :12] = w["A"]
        A[:, :, 12:, 12:] = 0.5 * np.eye(4)
        w["A"] = A
        w["B"] = np.concatenate([w["B"], rng.standard_normal((4, 20, 4, 4)) / 4.0], axis=2)
        w["C"] = np.concatenate([w["C"], np.zeros((16, 4))], axis=1)
        w["x0"] = np.concatenate([w["x0"], np.zeros((4, 4))], axis=1)
        w["goal"] = np.zeros(16)
        w["targets"] = np.zeros(20 * 16)
    else:
        w, dt = W.synthetic_ltv_batch(4, N=64), torch.float64 if case == "wide_f64" else torch.float32
    good = dict(w)
    w = dict(w)
    w["wt"] = -50.0
    _, _, sto, _ = oracle.solve_workload(w)
    assert (sto == 3).all(), sto
    for ww, want in ((w, 3), (good, 0)):
        plan = solve_mpc_batch(W.to_batch_problem(ww, dtype=dt), return_multipliers=True)
        torch.cuda.synchronize()
        st = plan.status.cpu().numpy()
        assert (st == want).all(), (case, want, st)
        if want == 3:
            assert (plan.U.cpu().numpy() == 0).all() and (plan.multipliers.cpu().numpy() == 0).all()
            assert (plan.iters.cpu().numpy() == 0).all()
    if case.startswith("narrow"):  # the condensed kernels on the same batch
        plan = solve_mpc_batch(W.to_batch_problem(w, dtype=dt), flags=_capi.OPT_FORCE_CONDENSED)
        torch.cuda.synchronize()
        assert (plan.status.cpu().numpy() == 3).all()


@pytest.mark.parametrize("name", ["stagewise_triple_n1024", "stagewise_triple_n1024_b"])
def test_solve_mpc_takes_any_horizon_through_the_default_entry_point(name):
    """solve_mpc(problem, "hip_gi") -- the reference's own call, solve_mpc.py:42-44 -- on the N = 1024 fixtures (n = 1024:
    the dense kernels stop at 256): the minimiser of the REFERENCE-built dense QP to 1e-7 |U|, states rolled out."""
    from qpmpc_amd import solve_mpc

    p, f = _load(name)
    plan = solve_mpc(p, "hip_gi")
    assert not plan.is_empty
    U = np.asarray(plan.inputs).ravel()
    Us = f["U_star"].ravel()
    assert np.abs(U - Us).max() <= 1e-7 * max(1.0, np.abs(Us).max())
    assert plan.states.shape == (1025, 3)


def test_slot_overflow_is_retried_with_more_slots():
    """N = 4096 (n = 4096, m = 16384): with the default 128 slots a few problems want more rows active at once and the kernel
    reports MPCQP_SLOTS_FULL (4) for them; solve_mpc_batch solves those again with 256, 512, ... slots: 100 % solved, and the
    re-solved ones equal a solve that had enough slots from the start."""
    import os, sys

    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    from bench_stagewise import long_batch
    from qpmpc_amd import _capi, solve_mpc_batch

    bp = long_batch(192, 4096, 1.0 / 128)
    raw = solve_mpc_batch(bp, retry_slots=False)
    torch.cuda.synchronize()
    st = raw.status.cpu().numpy()
    assert set(np.unique(st)) <= {0, _capi.SLOTS_FULL}
    plan = solve_mpc_batch(bp)
    torch.cuda.synchronize()
    assert (plan.status.cpu().numpy() == 0).all()
    if (st == _capi.SLOTS_FULL).any():
        idx = np.flatnonzero(st == _capi.SLOTS_FULL)
        sub = bp.select(torch.as_tensor(idx, device=bp.device))
        wide = solve_mpc_batch(sub, formulation="stagewise", max_active=1024, retry_slots=False)
        torch.cuda.synchronize()
        assert (wide.status.cpu().numpy() == 0).all()
        a, b = plan.U.cpu().numpy()[idx], wide.U.cpu().numpy()
        assert np.abs(a - b).max() <= 1e-7 * max(1.0, np.abs(b).max())
    ok = st == 0
    assert np.array_equal(plan.U.cpu().numpy()[ok], raw.U.cpu().numpy()[ok])


def test_config5_at_the_per_gpu_size_through_the_default_dispatch_against_the_oracle():
    """BASELINE configs[4] at the size ONE GPU of the eight gets (1024 of the 8192 problems, bench.py's own slice), through
    the default entry point (-> wide stage-wise kernel): float32 at the config's 1e-3, float64 at 1e-7, against the C oracle
    on every host core (280 problems/s there: a few seconds)."""
    from oracle.parallel import solve_workload_parallel
    from qpmpc_amd import solve_mpc_batch
    from qpmpc_amd import workloads as W
    from qpmpc_amd.distributed import shard_workload

    w = W.synthetic_ltv_batch_slice(0, 1024)
    Uo, _, sto, _ = solve_workload_parallel(w, shard_workload)
    assert (sto == 0).all()
    scale = np.maximum(1.0, np.abs(Uo).max(axis=1, keepdims=True))
    p32 = solve_mpc_batch(W.to_batch_problem(w, dtype=torch.float32))
    p64 = solve_mpc_batch(W.to_batch_problem(w))
    torch.cuda.synchronize()
    assert (p32.status == 0).all() and (p64.status == 0).all()
    e32 = float((np.abs(p32.U.double().cpu().numpy() - Uo) / scale).max())
    e64 = float((np.abs(p64.U.cpu().numpy() - Uo) / scale).max())
    print("config 5, 1024 problems: float32", e32, "float64", e64, "mean iterations", float(p32.iters.float().mean()))
    assert e32 <= 1e-3 and e64 <= 1e-7


def test_small_batch_instantiation_gives_the_same_plans_as_the_default_one():
    """A float32 launch that does not fill the SIMDs (batch <= 4 x CUs) takes the small-batch instantiation of the wide kernel
    (one wavefront per SIMD, twelve right-hand sides per backward sweep, deeper request rings, vectors in registers); a larger
    launch the default one (seven). Since round 6 the cached rows steer the selection, so the iterates may differ -- the
    minimiser cannot: the first 1024 problems of config 5 solved alone and as part of a batch of 1100 give the same statuses
    and the same plans to float32 accuracy, and need about as many iterations."""
    from qpmpc_amd import solve_mpc_batch
    from qpmpc_amd import workloads as W

    w = W.synthetic_ltv_batch_slice(0, 1100)
    w1 = {k: (v[:1024] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 1100 else v) for k, v in w.items()}
    big = solve_mpc_batch(W.to_batch_problem(w, dtype=torch.float32))
    small = solve_mpc_batch(W.to_batch_problem(w1, dtype=torch.float32))
    torch.cuda.synchronize()
    assert (big.status == 0).all() and (small.status == 0).all()
    scale = big.U[:1024].abs().max(dim=1, keepdim=True).values.clamp(min=1.0)
    assert float(((big.U[:1024] - small.U).abs() / scale).max()) <= 1e-4
    assert abs(float(big.iters[:1024].float().mean()) - float(small.iters.float().mean())) <= 0.5


# ---------------------------------------------------------------- wide systems (mpcqp_stageg.hip)
@pytest.mark.parametrize("nx,nu,N,mk,dtype,tol", [
    (24, 6, 64, 4, "f64", 1e-7), (32, 8, 40, 6, "f64", 1e-7), (24, 6, 64, 4, "f32", 1e-3), (32, 8, 40, 6, "f32", 1e-3),
    (20, 6, 16, 8, "f64", 1e-7),   # n = 96 <= 256: the dense HBM-resident path keeps it
    (20, 5, 60, 3, "f64", 1e-7),   # nu = 5 alone puts it beyond the MFMA kernel
])
def test_wide_systems_any_horizon_against_the_oracle(nx, nu, N, mk, dtype, tol):
    """solve_mpc accepts any (nx, nu, N) upstream (qpmpc/solve_mpc.py:42-44, qpmpc/mpc_qp.py:39-122). Systems wider than the
    MFMA stage-wise kernel's tiles (nx > 16 or nu > 4) on horizons the dense path cannot hold (n = N nu > 256) used to come
    back MPCQP_ETOOLARGE; the general stage-wise kernel serves them (nx <= 32, nu <= 8, float64 arithmetic; float32
    launches are converted). Plans against the float64 C oracle through the default dispatch."""
    import os
    import sys

    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd.workloads import to_batch_problem

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from stress_stagewise import random_ltv

    rng = np.random.default_rng(1000 * nx + N)
    w = random_ltv(rng, 8, nx, nu, N, mk, 3.0)
    w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
    td = torch.float32 if dtype == "f32" else torch.float64
    bp = to_batch_problem(w, dtype=td)
    plan = solve_mpc_batch(bp, return_multipliers=True)
    torch.cuda.synchronize()
    U, st = plan.U.double().cpu().numpy(), plan.status.cpu().numpy()
    Uo, lamo, sto, _ = oracle.solve_workload(w)
    assert np.array_equal(st == 0, sto == 0), (st, sto)
    ok = sto == 0
    assert ok.sum() >= 2
    scale = np.maximum(1.0, np.abs(Uo[ok]).max(axis=1, keepdims=True))
    assert (np.abs(U[ok] - Uo[ok]) / scale).max() <= tol
    lam = plan.multipliers.double().cpu().numpy()
    assert (lam >= 0).all()
    if N * nu > 256 and dtype == "f64":
        # the same kernel through the stage-wise entry point
        again = solve_mpc_batch(bp, formulation="stagewise")
        torch.cuda.synchronize()
        assert np.array_equal(again.status.cpu().numpy(), st)
        assert (np.abs(again.U.cpu().numpy()[ok] - U[ok]) / scale).max() <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("nx,nu,N,mk", [(2, 1, 20, 2), (3, 2, 24, 3), (4, 2, 40, 2), (3, 1, 90, 1), (4, 1, 33, 4)])
def test_factor_options_of_the_narrow_kernel_for_every_dimension(nx, nu, N, mk):
    """MPCQP_OPT_KEEP_FACTOR / REUSE_FACTOR / PIPELINE_FACTOR of mpcqp_stage.hip for the dimensions the closed-loop tests do
    not reach (they run nx = 4, nu = 1): the factor kept by one launch, reused by the next, and rebuilt by the second
    wavefront of the pipelined instantiation (operands padded and transposed into LDS, round 4) must give the plans of the
    plain launch -- same recursion, same sweeps: bitwise. Replaces qpmpc/solve_mpc.py:42-44 / mpc_qp.py:129-163 (build once,
    update, re-solve) like the closed loops."""
    import os, sys

    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    from stress_stagewise import random_ltv
    from qpmpc_amd import PreparedSolve, _capi
    from qpmpc_amd.workloads import to_batch_problem

    rng = np.random.default_rng(1000 * nx + 100 * nu + N)
    # (nx = 3: a batch that is no multiple of four -- the pipelined instantiation packs four problems per workgroup and ONE
    # factor wavefront runs their recursions in the four lane quads of its matrix-core products; the last quads repeat a problem)
    w = random_ltv(rng, 23 if nx == 3 else 24, nx, nu, N, mk, tight=1.0)
    bp = to_batch_problem(w)
    plain = PreparedSolve(bp, formulation="stagewise")
    plain.launch()
    torch.cuda.synchronize()
    U0, st0 = plain.U.clone(), plain.status.clone()
    assert int((st0 == 0).sum()) >= 12  # (a meaningful share of the batch is solvable)
    run = PreparedSolve(bp, formulation="stagewise")
    o = run._opts
    base = o.flags & ~(_capi.OPT_KEEP_FACTOR | _capi.OPT_REUSE_FACTOR | _capi.OPT_PIPELINE_FACTOR)
    # keep (image 0) -> reuse it -> pipelined: solve with image 0, the second wavefront factors into image 1 -> solve with image 1
    for flags, slot in ((_capi.OPT_KEEP_FACTOR, 0), (_capi.OPT_REUSE_FACTOR, 0), (_capi.OPT_PIPELINE_FACTOR, 0),
                        (_capi.OPT_PIPELINE_FACTOR, 1), (_capi.OPT_REUSE_FACTOR, 0)):
        o.flags, o.factor_slot = base | flags, slot
        run.U.zero_()
        run.launch()
        torch.cuda.synchronize()
        assert torch.equal(run.status, st0), (flags, slot)
        assert torch.equal(run.U, U0), (flags, slot, float((run.U - U0).abs().max()))
