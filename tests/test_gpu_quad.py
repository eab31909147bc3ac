"""GPU tests (-m gpu): the four-problems-per-wavefront kernel (csrc/mpcqp_quad.hip) -- the cold fused build+solve of problems with
two rows per step and nx = 2 .. 4 (lean build: terminal cost only, state rows only -- BASELINE configs 1, 2, 4; general build, round 6:
input rows, stage cost), replacing qpmpc/mpc_qp.py:53-149 and the
qpsolvers call at qpmpc/solve_mpc.py:43 like the two-per-wavefront kernel it is dispatched next to. The dispatch takes it by
batch size (2049 problems and more on an MI355X; beyond 4096 its slim LDS carve, two wavefronts per SIMD); MPCQP_OPT_FOUR_PER_WAVE
forces it, MPCQP_OPT_TWO_PER_WAVE keeps the other.

Tolerances (float64): plans |u - u_ref|_inf <= 1e-7 max(1, |u_ref|_inf) against the C oracle (BASELINE.json: 1e-6), observed
~1e-12; statuses equal; iteration counts equal to the two-per-wavefront kernel's (same method, same pivots)."""
import os
import sys

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
if TOOLS not in sys.path:
    sys.path.insert(0, TOOLS)


def _lean_family(rng, batch, nx, nu, N, tight, lti):
    from stress_stagewise import random_ltv

    w = random_ltv(rng, batch, nx, nu, N, 2, tight)
    w["wx"] = w["targets"] = w["D"] = None
    if lti:  # time-invariant A, C (stride 0 along the horizon)
        w["A"] = np.ascontiguousarray(w["A"][:, :1])
        w["C"] = np.ascontiguousarray(w["C"][:, :1])
    return w


def shard(w, count):
    """the first `count` problems of a workload (the oracle is a CPU loop)"""
    out = dict(w)
    for k in ("A", "B", "C", "D", "e", "x0", "goal", "targets"):
        v = w.get(k)
        if v is not None and getattr(v, "ndim", 0) >= 1 and v.shape[0] == len(w["x0"]):
            out[k] = v[:count]
    return out


class _first:
    """the first `count` problems of a plan"""

    def __init__(self, plan, count):
        self.U, self.status = plan.U[:count], plan.status[:count]


def _check_against_oracle(w, plan, tol=1e-7):
    U, st = plan.U.cpu().numpy(), plan.status.cpu().numpy()
    Uo, lamo, sto, _ = oracle.solve_workload(w)
    assert np.array_equal(st == 0, sto == 0), np.flatnonzero((st == 0) != (sto == 0))
    ok = sto == 0
    if ok.any():
        scale = np.maximum(1.0, np.abs(Uo[ok]).max(axis=1, keepdims=True))
        assert (np.abs(U[ok] - Uo[ok]) / scale).max() <= tol
    assert not np.isnan(U).any() and (U[~ok] == 0).all()  # (no plan: zeros, never NaN)
    return ok


def test_config2_at_the_benchmarked_size_takes_the_kernel_and_matches_the_oracle():
    """4096 heterogeneous triple-integrator problems (BASELINE config 2): the default dispatch (four per wavefront at this size),
    the forced one and the two-per-wavefront kernel give the oracle's plans, and each other's iteration counts."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    w = W.triple_integrator_batch(4096)
    bp = W.to_batch_problem(w)
    auto = solve_mpc_batch(bp, return_multipliers=True)
    four = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE, return_multipliers=True)
    two = solve_mpc_batch(bp, flags=_capi.OPT_TWO_PER_WAVE, return_multipliers=True)
    torch.cuda.synchronize()
    assert torch.equal(auto.U, four.U) and torch.equal(auto.iters, four.iters)  # (the same kernel ran)
    assert torch.equal(four.status, two.status) and torch.equal(four.iters, two.iters)
    assert float((four.U - two.U).abs().max()) <= 1e-9
    assert float((four.multipliers - two.multipliers).abs().max()) <= 1e-9 * max(1.0, float(two.multipliers.abs().max()))
    ok = _check_against_oracle(w, four, tol=1e-8)
    assert ok.all()


def test_config4_share_of_one_gpu_matches_the_oracle():
    """8192 humanoid problems (BASELINE config 4's share of one of eight GPUs, examples/humanoid_one_step.py:42-80 with the sweep's
    states): default dispatch = this kernel; statuses and plans against the oracle."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    w = W.humanoid_batch(8192, seed=2)
    bp = W.to_batch_problem(w)
    auto = solve_mpc_batch(bp)
    four = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
    torch.cuda.synchronize()
    assert torch.equal(auto.U, four.U)
    ok = _check_against_oracle(w, four, tol=1e-7)
    assert ok.sum() > 6000 and set(np.unique(four.status.cpu().numpy())) <= {0, 2}  # (a state no plan exists for: status 2, zeros)


@pytest.mark.parametrize("nx,nu", [(3, 1), (4, 1), (3, 2), (4, 2)])
def test_every_horizon_and_tightness_against_the_oracle(nx, nu):
    """Random LTV families over the kernel's envelope -- every horizon N = 2 .. 16 / nu (rows 16 .. 2 N - 1 sit in the lanes'
    second register row: N = 9 .. 15 leaves that row partly empty), loose to very tight bounds (partial steps, drops), A and C per
    step or time-invariant --: statuses and plans against the oracle, iteration counts against the two-per-wavefront kernel."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(100 * nx + nu)
    drops = 0
    for N in range(2, 16 // nu + 1):
        for tight, lti in ((3.0, False), (0.2, True), (0.05, False), (0.05, True)):
            w = _lean_family(rng, 61, nx, nu, N, tight, lti)  # (an odd batch: the last wavefront holds one problem)
            bp = W.to_batch_problem(w)
            four = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
            two = solve_mpc_batch(bp, flags=_capi.OPT_TWO_PER_WAVE)
            torch.cuda.synchronize()
            ok = _check_against_oracle(w, four)
            same = (four.iters == two.iters).cpu().numpy()[ok]
            assert same.mean() >= 0.95, (N, tight, lti, same.mean())  # (ties between rows may break differently: rare)
            drops += int((four.iters.cpu().numpy()[ok] > N * nu).sum())
    assert drops > 0  # the partial-step / drop path really ran


def test_launches_of_several_rounds_take_the_slim_carve():
    """More than four problems per SIMD of the device (4097 and more on an MI355X): the kernel's instantiation with the 20-KB LDS
    carve -- packed L^-T, T' rho by row sums, the leaving slot's row fetched by ds_bpermute -- two wavefronts per SIMD. Same
    families as above at 4400 problems each (drops, partly empty second rows, inconsistent rows), against the oracle and, bit for
    bit in the statuses and iteration counts, against the one-round instantiation on the same problems in chunks."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W
    from qpmpc_amd.distributed import shard_workload

    simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    batch = 4 * simds + 304
    rng = np.random.default_rng(77)
    drops = 0
    for nx, nu, N, tight, lti in ((3, 1, 16, 0.05, False), (4, 1, 11, 0.05, True), (3, 2, 8, 0.2, False), (4, 1, 13, 0.2, True), (3, 1, 5, 3.0, False)):
        w = _lean_family(rng, batch, nx, nu, N, tight, lti)
        slim = solve_mpc_batch(W.to_batch_problem(w), return_multipliers=True)  # (default dispatch at this size)
        torch.cuda.synchronize()
        ok = _check_against_oracle(w, slim)
        drops += int((slim.iters.cpu().numpy()[ok] > N * nu).sum())
        # the same problems in two launches of one round each (the roomy carve)
        parts = [solve_mpc_batch(W.to_batch_problem(shard_workload(w, r, 2)), flags=_capi.OPT_FOUR_PER_WAVE, return_multipliers=True) for r in range(2)]
        torch.cuda.synchronize()
        st = torch.cat([p.status for p in parts])
        it = torch.cat([p.iters for p in parts])
        U = torch.cat([p.U for p in parts])
        assert torch.equal(st, slim.status) and torch.equal(it, slim.iters)
        good = st == 0
        assert float((U[good] - slim.U[good]).abs().max()) <= 1e-9 * max(1.0, float(U[good].abs().max()))
    assert drops > 0


def test_inconsistent_rows_are_never_reported_solved():
    """Problems whose rows are inconsistent with their bounds (generated consistent, then A and C replaced by their first step):
    the kernel's statuses are the oracle's, and what it reports solved is the oracle's plan (qpsolvers reports found=False for the
    others, qpmpc/plan.py:35-40)."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(99)
    unsolved = 0
    for _ in range(16):
        nx, N = int(rng.choice([3, 4])), int(rng.integers(4, 16))
        w = _lean_family(rng, 128, nx, 1, N, float(rng.choice([0.05, 0.2])), True)
        ok = _check_against_oracle(w, solve_mpc_batch(W.to_batch_problem(w), flags=_capi.OPT_FOUR_PER_WAVE), tol=1e-6)
        unsolved += int((~ok).sum())
    assert unsolved > 50


def test_small_and_ragged_batches_and_a_pairing_order():
    """Batches of 1, 2, 3, 5 and 4097 problems (idle rows of the last wavefront repeat its last problem and store nothing), and a
    launch with MpcqpSolveOpts.order (row i of the launch takes problem order[i]; outputs stay indexed by problem)."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    for batch in (1, 2, 3, 5, 4097):
        w = W.triple_integrator_batch(batch, seed=batch)
        plan = solve_mpc_batch(W.to_batch_problem(w), flags=_capi.OPT_FOUR_PER_WAVE)
        torch.cuda.synchronize()
        assert _check_against_oracle(w, plan, tol=1e-8).all()
    w = W.humanoid_batch(1001, seed=4)
    bp = W.to_batch_problem(w)
    ref = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
    order = torch.randperm(1001, device="cuda").to(torch.int32)
    got = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE, order=order)
    torch.cuda.synchronize()
    # (bit for bit: a row's arithmetic does not depend on which problems share its wavefront)
    assert torch.equal(ref.status, got.status) and torch.equal(ref.iters, got.iters) and torch.equal(ref.U, got.U)


@pytest.mark.parametrize("family", ["humanoid_4096", "humanoid_9000", "wip12", "triple"])
def test_shared_model_launches_four_per_wavefront(family):
    """mpcqp_solve_model_batch on a model factored once (mpc_qp.py:129-163 taken to its end): the dispatch takes this layout from
    2049 problems (forced below); it agrees with the two-per-wavefront launch of the same model and with the C oracle. wip12 has
    a stage cost and input rows: the model mode takes every layout whose condensed problem fits the rows."""
    from qpmpc_amd import SharedModel, _capi
    from qpmpc_amd import workloads as W

    if family.startswith("humanoid"):
        w = W.humanoid_batch(int(family.split("_")[1]))
    elif family == "triple":
        w = W.triple_integrator_batch(777, heterogeneous=False)
    else:
        w = W.wip_batch(515, N=12, sampling_period=0.1)
        w["x0"][:64, 1] += 0.25  # some loops hit the input box
        ts = np.stack([w["pendulum"].target_states(x, 0.5) for x in w["x0"]])
        w["goal"], w["targets"] = ts[:, -4:], ts[:, :-4]
    bp = W.to_batch_problem(w)
    model = SharedModel(bp)
    args = (bp.initial_state, bp.goal_state, bp.target_states)
    four = model.solve(*args, return_multipliers=True, flags=_capi.OPT_FOUR_PER_WAVE)
    two = model.solve(*args, return_multipliers=True, flags=_capi.OPT_TWO_PER_WAVE)
    torch.cuda.synchronize()
    assert torch.equal(four.status, two.status) and torch.equal(four.iters, two.iters)
    ok = two.status.cpu().numpy() == 0
    assert ok.mean() > 0.9
    Uf, Ut = four.U.cpu().numpy()[ok], two.U.cpu().numpy()[ok]
    assert (np.abs(Uf - Ut) / np.maximum(1.0, np.abs(Ut).max(axis=1, keepdims=True))).max() < 1e-9
    lf, lt = four.multipliers.cpu().numpy()[ok], two.multipliers.cpu().numpy()[ok]
    assert np.abs(lf - lt).max() <= 1e-6 * max(1.0, np.abs(lt).max())
    if family.startswith("humanoid"):  # large enough for the default rule
        dflt = model.solve(*args)
        torch.cuda.synchronize()
        assert torch.equal(dflt.U, four.U) and torch.equal(dflt.status, four.status)
    Uo, _, sto, _ = oracle.solve_workload(w, count=64)
    okc = sto == 0
    assert np.array_equal(okc, four.status.cpu().numpy()[:64] == 0)
    scale = np.maximum(1.0, np.abs(Uo[okc]).max(axis=1, keepdims=True))
    assert (np.abs(four.U.cpu().numpy()[:64][okc] - Uo[okc]) / scale).max() <= 1e-6


def test_shared_model_with_bounds_per_problem_and_an_order():
    """mpcqp_solve_model_bounds_batch (matrices shared, inequality vectors per problem: update_constraint_vector, mpc_qp.py:151-163)
    through this layout: equal to the fused build of every problem, ragged batch, and bit for bit under a pairing order."""
    from qpmpc_amd import BatchMPCProblem, SharedModel, _capi, solve_mpc_batch

    rng = np.random.default_rng(15)
    B, N, T = 2999, 16, 0.1
    A = np.array([[1.0, T, T**2 / 2.0], [0.0, 1.0, T], [0.0, 0.0, 1.0]])
    Bm = np.array([T**3 / 6.0, T**2 / 2.0, T]).reshape((3, 1))
    Cm = np.array([[1.0, 0.0, -0.0856], [-1.0, 0.0, 0.0856]])
    centre = rng.uniform(-0.2, 0.2, (B, N, 1))
    e = np.concatenate([centre + rng.uniform(0.03, 0.2, (B, N, 1)), -centre + rng.uniform(0.03, 0.2, (B, N, 1))], axis=2)
    x0 = np.concatenate([rng.uniform(-0.05, 0.05, (B, 1)), rng.uniform(-0.1, 0.1, (B, 1)), rng.uniform(-0.2, 0.2, (B, 1))], axis=1)
    goal = np.concatenate([rng.uniform(-0.3, 0.3, (B, 1)), np.zeros((B, 2))], axis=1)
    prob = BatchMPCProblem(A, Bm, Cm, None, e, N, 1.0, None, 1e-3, x0, goal_state=goal)
    fused = solve_mpc_batch(prob, flags=_capi.OPT_TWO_PER_WAVE)
    model = SharedModel(prob)
    assert model.per_problem_bounds
    run = model.prepare(prob)  # 2999 problems: the default rule takes four per wavefront
    run.launch()
    forced = model.prepare(prob, flags=_capi.OPT_FOUR_PER_WAVE)
    forced.launch()
    torch.cuda.synchronize()
    assert torch.equal(run.U, forced.U) and torch.equal(run.status, forced.status)
    st = fused.status.cpu().numpy()
    assert (run.status.cpu().numpy() == st).all()
    ok = st == 0
    assert ok.sum() > B // 2
    Uf, Um = fused.U.cpu().numpy()[ok], run.U.cpu().numpy()[ok]
    assert (np.abs(Uf - Um) / np.maximum(1.0, np.abs(Uf).max(axis=1, keepdims=True))).max() < 1e-8
    order = torch.randperm(B, device="cuda").to(torch.int32)
    perm = model.prepare(prob, flags=_capi.OPT_FOUR_PER_WAVE, order=order)
    perm.launch()
    torch.cuda.synchronize()
    assert torch.equal(perm.status, run.status) and torch.equal(perm.iters, run.iters) and torch.equal(perm.U, run.U)
    with pytest.raises(Exception, match="-6"):  # next to another override
        model.prepare(prob, flags=_capi.OPT_FOUR_PER_WAVE | _capi.OPT_ONE_PER_WAVE).launch()


def test_forcing_the_kernel_where_it_does_not_apply_is_refused():
    """MPCQP_OPT_FOUR_PER_WAVE with nine rows per step, with a warm state, next to another override or on another kernel's
    dimensions: MPCQP_EUNSUPPORTED before any launch."""
    from qpmpc_amd import BackendError, WarmState, _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W
    from stress_stagewise import random_ltv

    rng = np.random.default_rng(1)
    w = random_ltv(rng, 8, 3, 1, 6, 9, 1.0)  # nine rows per step
    with pytest.raises(BackendError, match="-6"):
        solve_mpc_batch(W.to_batch_problem(w), flags=_capi.OPT_FOUR_PER_WAVE)
    w = W.triple_integrator_batch(8)
    bp = W.to_batch_problem(w)
    with pytest.raises(BackendError, match="-6"):
        solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE | _capi.OPT_TWO_PER_WAVE)
    with pytest.raises(BackendError, match="-6"):
        solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE, warm_state=WarmState(bp))
    wide = random_ltv(rng, 8, 17, 2, 8, 2, 1.0)  # another kernel's dimensions (nx > 16)
    with pytest.raises(BackendError, match="-6"):
        solve_mpc_batch(W.to_batch_problem(wide), flags=_capi.OPT_FOUR_PER_WAVE)


def _general_family(rng, batch, nx, nu, N, tight, rows, stage, mk=2):
    """random LTV problems with mk rows per step: state rows, input rows or both; with or without a stage cost"""
    from stress_stagewise import random_ltv

    w = random_ltv(rng, batch, nx, nu, N, mk, tight)
    if rows == "c":
        w["D"] = None
    elif rows == "d":  # an input box per step: e > 0 keeps u = 0 feasible
        w["C"] = None
        w["e"] = tight * (0.05 + 0.5 * np.abs(rng.standard_normal(w["e"].shape)))
    if not stage:
        w["wx"] = w["targets"] = None
    return w


@pytest.mark.parametrize("nx,nu", [(2, 1), (2, 2), (3, 1), (4, 1), (3, 2), (4, 2), (3, 3), (4, 4), (5, 1), (5, 2), (6, 1), (6, 2), (6, 3),
                                   (7, 1), (8, 2), (9, 3), (10, 1), (12, 4), (13, 2), (16, 1), (16, 4)])
def test_general_build_every_layout_against_the_oracle(nx, nu):
    """Round 6: the kernel's general build -- input rows D_k next to / instead of the state rows C_k, a stage cost (the Gram matrix over
    every Psi_k, targets per step), nx = 2, 5, 6 (three / four operand registers per step), nx = 7 .. 16 (operands streamed per step,
    padded sizes 8 / 12 / 16) -- over every horizon that fits sixteen variables, loose to very tight bounds: statuses and
    plans against the oracle; for nx = 3, 4 iteration counts against the two-per-wavefront kernel's generic build
    (qpmpc/mpc_qp.py:53-149 both)."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(1000 * nx + nu)
    drops = 0
    for N in range(2, 16 // nu + 1):
        for tight, rows, stage in ((3.0, "cd", True), (0.2, "d", True), (0.05, "c", True), (0.05, "cd", False), (0.2, "d", False)):
            if nx == 2 and rows == "c" and not stage:
                continue  # (the lean layout: the other tests)
            w = _general_family(rng, 61, nx, nu, N, tight, rows, stage)
            bp = W.to_batch_problem(w)
            four = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
            torch.cuda.synchronize()
            ok = _check_against_oracle(w, four)
            if nx in (3, 4):
                two = solve_mpc_batch(bp, flags=_capi.OPT_TWO_PER_WAVE)
                torch.cuda.synchronize()
                same = (four.iters == two.iters).cpu().numpy()[ok]
                assert same.mean() >= 0.95, (N, tight, rows, stage, same.mean())
            drops += int((four.iters.cpu().numpy()[ok] > N * nu).sum())
    assert drops > 0 or nu > 2  # (the partial-step / drop path ran; horizons of at most five steps rarely get there)


@pytest.mark.parametrize("mk", [1, 3, 4])
def test_general_build_one_to_four_rows_per_step(mk):
    """... and with one, three or four rows per step (m = N mk <= 32 rows): against the oracle and the two-per-wavefront kernel."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(50 + mk)
    checked = 0
    for nx, nu in ((2, 1), (3, 1), (4, 1), (3, 2), (4, 2), (5, 1), (6, 2)):
        for N in range(2, min(16 // nu, 32 // mk) + 1):
            tight, rows, stage = [(3.0, "cd", True), (0.1, "d", False), (0.05, "c", True), (0.2, "cd", False)][(N + nx) % 4]
            w = _general_family(rng, 61, nx, nu, N, tight, rows, stage, mk)
            bp = W.to_batch_problem(w)
            four = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
            torch.cuda.synchronize()
            ok = _check_against_oracle(w, four)
            checked += int(ok.sum())
            if nx in (3, 4):
                two = solve_mpc_batch(bp, flags=_capi.OPT_TWO_PER_WAVE)
                torch.cuda.synchronize()
                assert ((four.iters == two.iters).cpu().numpy()[ok]).mean() >= 0.95, (nx, nu, N)
    assert checked > 1000


def test_the_reference_wip_example_at_4096_takes_the_kernel():
    """The reference's own example problem (examples/wheeled_inverted_pendulum.py:90-94: N = 12, input box, stage + terminal cost),
    4096 of them with their own states and targets: the default dispatch is the four-per-wavefront kernel (bit-equal to the forced
    launch), slim carve beyond one round; plans against the oracle and the two-per-wavefront kernel."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    for batch in (4096, 4500):
        w = W.wip_batch(batch, N=12, sampling_period=0.1, seed=5)
        w["x0"][: batch // 4, 1] += 0.4  # some loops hit the input box
        ts = np.stack([w["pendulum"].target_states(x, 0.5) for x in w["x0"]])
        w["goal"], w["targets"] = ts[:, -4:], ts[:, :-4]
        bp = W.to_batch_problem(w)
        auto = solve_mpc_batch(bp)
        four = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
        two = solve_mpc_batch(bp, flags=_capi.OPT_TWO_PER_WAVE)
        torch.cuda.synchronize()
        assert torch.equal(auto.U, four.U) and torch.equal(auto.iters, four.iters)
        assert torch.equal(four.status, two.status) and torch.equal(four.iters, two.iters)
        assert float((four.U - two.U).abs().max()) <= 1e-9 * max(1.0, float(two.U.abs().max()))
        assert int((four.iters > 0).sum()) > 0  # (the box is active somewhere)
        Uo, _, sto, _ = oracle.solve_workload(w, count=256)
        assert (sto == 0).all() and (four.status[:256] == 0).all()
        scale = np.maximum(1.0, np.abs(Uo).max(axis=1, keepdims=True))
        assert (np.abs(four.U.cpu().numpy()[:256] - Uo) / scale).max() <= 1e-7


def test_default_dispatch_of_the_general_layouts_and_of_double_integrators():
    """The general layouts (input rows, stage cost, mk != 2) take this kernel at EVERY batch size (its build beats the two-per-wavefront
    kernel's generic one from a single wavefront up, tools/ab_quad_general.py); the lean layout from more than two problems per SIMD.
    nx = 2 has no two-per-wavefront instantiation and is dispatched directly. Default launches are bit-equal to the forced ones;
    plans against the oracle."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(8)
    simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    cases = [(2, 1, 14, "cd", True, 300), (2, 2, 7, "d", False, 5), (3, 1, 12, "d", True, 1), (4, 2, 6, "cd", True, 700),
             (5, 1, 16, "c", False, 9), (6, 2, 8, "cd", True, 130), (8, 2, 8, "cd", True, 70), (12, 4, 4, "c", False, 33),
             (2, 1, 16, "c", False, 2 * simds + 1)]  # (the last one: the lean layout with nx = 2 at the size the rule takes it)
    for nx, nu, N, rows, stage, batch in cases:
        w = _general_family(rng, batch, nx, nu, N, 0.2, rows, stage)
        bp = W.to_batch_problem(w)
        auto = solve_mpc_batch(bp)
        four = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
        torch.cuda.synchronize()
        assert torch.equal(auto.U, four.U) and torch.equal(auto.iters, four.iters) and torch.equal(auto.status, four.status)
        count = min(batch, 256)
        _check_against_oracle(shard(w, count), _first(four, count))
    # the lean layout of a small batch stays where it was (another kernel): same plans
    w = _general_family(rng, 300, 2, 1, 16, 0.2, "c", False)
    bp = W.to_batch_problem(w)
    auto, four = solve_mpc_batch(bp), solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
    torch.cuda.synchronize()
    assert torch.equal(auto.status, four.status)
    assert float((auto.U - four.U).abs().max()) <= 1e-8 * max(1.0, float(four.U.abs().max()))


def test_stress_campaign_with_drops():
    """tools/stress_pair.py's lean rounds forced through this kernel: statuses equal to the C oracle's and to the one-per-wavefront
    and workgroup kernels', plans within 1e-7 relative."""
    from stress_pair import run

    worst, flagged, drops = run(24, 96, seed=20260930, verbose=False, lean_only=True, flags_lean=4096)
    assert flagged == 0, (worst, flagged)
    assert worst < 1e-7 and drops > 0


def test_stress_campaign_through_the_general_build():
    """... and tools/stress_pair.py's other rounds (stage costs, input rows, one to four rows per step, nx = 3, 4; tight enough for
    partial steps and drops) forced through the kernel's general build: the same checks."""
    from stress_pair import run

    worst, flagged, drops = run(32, 96, seed=20261001, verbose=False, flags_lean=4096, flags_other=4096)
    assert flagged == 0, (worst, flagged)
    assert worst < 1e-7 and drops > 0


def test_random_campaign_over_the_whole_envelope_of_the_general_build():
    """tools/stress_quad_general.py: nx 2..4, nu 1..4, every horizon with n <= 16, one to four rows per step, state / input rows or both,
    with and without a stage cost, per-step or time-invariant operands, loose to very tight bounds -- 80 rounds of 64 problems forced
    through the kernel against the C oracle (six seeds of 150 x 96 ran clean when the build was written: worst 2.5e-11)."""
    from stress_quad_general import run

    worst, flagged, solved, drops = run(80, 64, seed=7, verbose=False)
    assert flagged == 0 and worst <= 1e-7, (worst, flagged)
    assert solved > 4000 and drops > 0


@pytest.mark.parametrize("nx", [2, 3, 4, 5, 6, 8])
def test_more_than_32_rows_take_the_four_rows_per_lane_copy(nx):
    """csrc/mpcqp_quad4.hip (end of round 6): n <= 16 with 33 .. 64 rows (three / four rows per step on horizons of 9 .. 16 steps) --
    default dispatch = forced launch, bit for bit; statuses and plans against the oracle; iteration counts against the
    one-per-wavefront kernel that served these problems before (same method)."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(640 + nx)
    drops = solved = 0
    for N, mk in ((16, 4), (16, 3), (13, 3), (12, 4), (9, 4), (11, 3)):
        for tight, rows, stage in ((3.0, "cd", True), (0.2, "c", False), (0.05, "cd", True), (0.1, "d", False)):
            w = _general_family(rng, 45, nx, 1, N, tight, rows, stage, mk)
            bp = W.to_batch_problem(w)
            auto = solve_mpc_batch(bp)
            four = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
            one = solve_mpc_batch(bp, flags=_capi.OPT_ONE_PER_WAVE)
            torch.cuda.synchronize()
            assert torch.equal(auto.U, four.U) and torch.equal(auto.iters, four.iters) and torch.equal(auto.status, four.status)
            ok = _check_against_oracle(w, four)
            same = (four.iters == one.iters).cpu().numpy()[ok]
            assert ok.sum() == 0 or same.mean() >= 0.9, (N, mk, tight, rows, stage, same.mean())
            solved += int(ok.sum())
            drops += int((four.iters.cpu().numpy()[ok] > N).sum())
    assert solved > 500 and drops > 0


def test_more_than_32_rows_launches_beyond_three_wavefronts_per_cu_take_the_slim_carve():
    """... and its second LDS carve (36.9 KB per wavefront: four on a CU, still one per SIMD) for launches of more than three wavefronts
    per CU (3073 problems and more on an MI355X): the same problems in two launches of the roomy carve give the same statuses and
    iteration counts bit for bit and plans within 1e-9; the first 256 against the oracle."""
    from qpmpc_amd import solve_mpc_batch
    from qpmpc_amd import workloads as W
    from qpmpc_amd.distributed import shard_workload

    simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    batch = 3 * simds + 120
    rng = np.random.default_rng(64)
    for nx, N, mk, tight, rows, stage in ((3, 16, 4, 0.2, "cd", True), (6, 12, 3, 0.05, "c", False)):
        w = _general_family(rng, batch, nx, 1, N, tight, rows, stage, mk)
        slim = solve_mpc_batch(W.to_batch_problem(w))
        parts = [solve_mpc_batch(W.to_batch_problem(shard_workload(w, r, 2))) for r in range(2)]
        torch.cuda.synchronize()
        st = torch.cat([p.status for p in parts])
        it = torch.cat([p.iters for p in parts])
        U = torch.cat([p.U for p in parts])
        assert torch.equal(st, slim.status) and torch.equal(it, slim.iters)
        good = st == 0
        assert int(good.sum()) > batch // 2
        assert float((U[good] - slim.U[good]).abs().max()) <= 1e-9 * max(1.0, float(U[good].abs().max()))
        _check_against_oracle(shard(w, 256), _first(slim, 256))


def test_five_to_eight_rows_per_step_take_the_four_rows_per_lane_copy_too():
    """... and up to EIGHT rows per step (a box on two states next to a box on two inputs: mk = 8), m <= 64: horizons of up to eight
    steps, any nu -- also where m <= 32 (mk > 4 is outside mpcqp_quad.hip's build). Default = forced launch; oracle; iteration counts
    against the one-per-wavefront kernel."""
    from qpmpc_amd import _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    rng = np.random.default_rng(88)
    solved = 0
    for nx, nu, N, mk in ((4, 2, 8, 8), (3, 1, 8, 5), (4, 2, 6, 6), (2, 2, 4, 8), (6, 2, 5, 7), (8, 4, 4, 8), (5, 1, 9, 7), (7, 2, 8, 8)):
        for tight, rows, stage in ((3.0, "cd", True), (0.2, "c", False), (0.05, "cd", False), (0.1, "d", True)):
            w = _general_family(rng, 37, nx, nu, N, tight, rows, stage, mk)
            bp = W.to_batch_problem(w)
            auto = solve_mpc_batch(bp)
            four = solve_mpc_batch(bp, flags=_capi.OPT_FOUR_PER_WAVE)
            one = solve_mpc_batch(bp, flags=_capi.OPT_ONE_PER_WAVE)
            torch.cuda.synchronize()
            assert torch.equal(auto.U, four.U) and torch.equal(auto.iters, four.iters) and torch.equal(auto.status, four.status)
            ok = _check_against_oracle(w, four)
            same = (four.iters == one.iters).cpu().numpy()[ok]
            assert ok.sum() == 0 or same.mean() >= 0.9, (nx, nu, N, mk, tight, rows, stage, same.mean())
            solved += int(ok.sum())
    assert solved > 600
