"""GPU tests (-m gpu) of the solver warm start (MpcqpSolveOpts.warm_state):
the previous solve's active set and active-set operator seed the next one. The reference reaches
its backends' warm start through **kwargs (qpmpc/solve_mpc.py:20,43) inside receding-horizon loops
(examples/wheeled_inverted_pendulum.py:99-118). Whatever the stored state, results must equal the
cold solve / the CPU oracle: the kernel re-checks the KKT conditions of what it returns.
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _solve(bp, **kw):
    from qpmpc_amd import solve_mpc_batch

    plan = solve_mpc_batch(bp, return_multipliers=True, **kw)
    torch.cuda.synchronize()
    return plan.U.cpu().numpy(), plan.status.cpu().numpy(), plan.iters.cpu().numpy(), plan.multipliers.cpu().numpy()


def _scale(U):
    return np.maximum(1.0, np.abs(U).max(axis=1, keepdims=True))


@pytest.mark.parametrize("family", ["triple", "humanoid"])
def test_warm_start_from_own_solution_needs_no_iteration(family):
    from qpmpc_amd import WarmState
    from qpmpc_amd import workloads as W

    w = W.triple_integrator_batch(1023) if family == "triple" else W.humanoid_batch(1023)
    bp = W.to_batch_problem(w)
    ws = WarmState(bp)
    U0, st0, it0, lam0 = _solve(bp, warm_state=ws)  # cold, stores the state
    act = ws.active_set.cpu().numpy()
    ok = st0 == 0
    assert ok.sum() >= 900
    # the stored set is the set of rows with a positive multiplier (up to weakly active rows)
    for b in np.flatnonzero(ok)[:64]:
        stored = set(int(a) for a in act[b] if a >= 0)
        assert set(np.flatnonzero(lam0[b] > 1e-9)) <= stored
    assert (act[~ok] == -1).all()
    U1, st1, it1, lam1 = _solve(bp, warm_state=ws, warm_start=True)
    assert np.array_equal(st0, st1)
    assert (it1[ok] == 0).all(), it1[ok].max()
    assert (np.abs(U1[ok] - U0[ok]) / _scale(U0[ok])).max() <= 1e-9
    assert np.abs(lam1[ok] - lam0[ok]).max() <= 1e-6 * max(1.0, np.abs(lam0[ok]).max())
    # infeasible / failed items restart cold and report the same status
    assert np.array_equal(it1[~ok], it0[~ok])


def test_warm_start_receding_horizon_saves_iterations_same_plans():
    """A 2048-problem triple-integrator receding horizon: every period the state advances by the first
    input and the goal moves; warm-started periods must give the cold plans with fewer iterations."""
    from qpmpc_amd import PreparedSolve, WarmState
    from qpmpc_amd import workloads as W

    w = W.triple_integrator_batch(2048, heterogeneous=False)
    bp_c, bp_w = W.to_batch_problem(w), W.to_batch_problem(w)
    ws = WarmState(bp_w)
    cold, warm = PreparedSolve(bp_c), PreparedSolve(bp_w, warm_state=ws)
    A = torch.as_tensor(w["A"], device="cuda")
    Bm = torch.as_tensor(w["B"], device="cuda").reshape(3)
    it_c = it_w = 0
    for period in range(12):
        cold.launch()
        warm.launch()
        torch.cuda.synchronize()
        sc, sw = cold.status.cpu().numpy(), warm.status.cpu().numpy()
        assert np.array_equal(sc, sw), period
        ok = sc == 0
        Uc, Uw = cold.U.cpu().numpy(), warm.U.cpu().numpy()
        assert (np.abs(Uw[ok] - Uc[ok]) / _scale(Uc[ok])).max() <= 1e-8, period
        if period >= 1:
            it_c += int(cold.iters.sum().item())
            it_w += int(warm.iters.sum().item())
        # plant: x+ = A x + B u0 (both copies get the cold plan so the two sequences stay identical)
        x = bp_c.initial_state
        xn = x @ A.T + cold.U[:, :1] * Bm
        bp_c.initial_state.copy_(xn)
        bp_w.initial_state.copy_(xn)
        warm.set_warm_start(True)
    assert it_w < 0.6 * it_c, (it_w, it_c)


def test_warm_start_with_a_foreign_or_garbage_state_is_still_correct():
    """The state is not trusted: seeded with another batch's solution, with random bytes, and with a
    state stored for different matrices, the plans must still match the oracle."""
    from qpmpc_amd import WarmState
    from qpmpc_amd import workloads as W

    w = W.triple_integrator_batch(512, seed=7)
    bp = W.to_batch_problem(w)
    Uo, _, sto, _ = oracle.solve_workload(w)
    ok = sto == 0
    # (a) another batch's active sets
    other = W.to_batch_problem(W.triple_integrator_batch(512, seed=8))
    ws = WarmState(other)
    _solve(other, warm_state=ws)
    U, st, _, _ = _solve(bp, warm_state=ws, warm_start=True)
    assert np.array_equal(st == 0, ok)
    assert (np.abs(U[ok] - Uo[ok]) / _scale(Uo[ok])).max() <= 1e-8
    # (b) random bytes (NaNs, huge values, out-of-range and duplicate row ids)
    g = torch.Generator(device="cuda").manual_seed(3)
    ws.buffer.copy_(torch.randint(0, 256, ws.buffer.shape, dtype=torch.uint8, device="cuda", generator=g))
    ids = ws.active_set
    ids.copy_(torch.randint(-3, 40, ids.shape, dtype=torch.int32, device="cuda", generator=g))
    ws.buffer[:, 16 * 16 * 8:] = ids.view(torch.uint8)
    U, st, _, _ = _solve(bp, warm_state=ws, warm_start=True)
    assert np.array_equal(st == 0, ok)
    assert (np.abs(U[ok] - Uo[ok]) / _scale(Uo[ok])).max() <= 1e-8
    # (c) a state stored for different matrices (humanoid sweep), same dimensions
    hb = W.to_batch_problem(W.humanoid_batch(512))
    ws2 = WarmState(hb)
    _solve(hb, warm_state=ws2)
    U, st, _, _ = _solve(bp, warm_state=ws2, warm_start=True)
    assert np.array_equal(st == 0, ok)
    assert (np.abs(U[ok] - Uo[ok]) / _scale(Uo[ok])).max() <= 1e-8


def test_warm_start_is_refused_where_it_is_not_implemented():
    from qpmpc_amd import BackendError, WarmState, _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    small = W.to_batch_problem(W.triple_integrator_batch(4))
    ws = WarmState(small)
    with pytest.raises(BackendError, match="-6"):  # MPCQP_EUNSUPPORTED through the C ABI
        solve_mpc_batch(small, warm_state=ws, flags=_capi.OPT_FORCE_LDS)
    # the wide stage-wise kernel keeps row ids only: an operator-mode start is refused, the row-id start is served
    big = W.to_batch_problem(W.synthetic_ltv_batch(2, N=64))  # nx = 12, n = 256
    wb = WarmState(big)
    assert wb.kind == "stagew"
    with pytest.raises(BackendError, match="-6"):
        solve_mpc_batch(big, warm_state=wb, warm_start=True)


def test_warm_state_of_another_batch_is_refused_before_any_launch():
    """The state is indexed by problem: one allocated for a smaller batch (or a short raw buffer) must be refused on
    the host and, behind it, by the C ABI (MpcqpSolveOpts.warm_state_bytes -> MPCQP_EWORKSPACE), never read out of bounds."""
    import ctypes as C

    from qpmpc_amd import BackendError, PreparedSolve, WarmState, _capi, solve_mpc_batch
    from qpmpc_amd import batch as Bt
    from qpmpc_amd import workloads as W

    small = W.to_batch_problem(W.triple_integrator_batch(8))
    large = W.to_batch_problem(W.triple_integrator_batch(64))
    ws = WarmState(small)
    with pytest.raises(BackendError, match="WarmState holds 8 problems"):
        solve_mpc_batch(large, warm_state=ws)
    with pytest.raises(BackendError, match="WarmState holds 8 problems"):
        PreparedSolve(large, warm_state=ws)
    raw = torch.zeros(64 * ws.bytes_per_problem - 8, dtype=torch.uint8, device="cuda")
    with pytest.raises(BackendError, match="warm_state tensor"):
        solve_mpc_batch(large, warm_state=raw)
    # the C ABI's own check, below the host's
    lib = _capi.load()
    dims, cp = large.dims(), large.c_problem()
    opts = Bt._opts(warm_state=ws.buffer)
    U = torch.empty((64, 16), dtype=torch.float64, device="cuda")
    st = torch.full((64,), -7, dtype=torch.int32, device="cuda")
    rc = lib.mpcqp_build_solve_batch(C.byref(dims), C.byref(cp), 64, C.byref(opts), U.data_ptr(), None, st.data_ptr(), None,
                                     None, 0, Bt._stream_ptr())
    torch.cuda.synchronize()
    assert rc == -5 and (st.cpu().numpy() == -7).all()
    # the right size is accepted
    plan = solve_mpc_batch(large, warm_state=WarmState(large))
    torch.cuda.synchronize()
    assert (plan.status.cpu().numpy() == 0).all()


def _saturating_wip(batch, seed=9):
    """Config 3's problems with states that drive the input into its box for a stretch of the horizon."""
    from qpmpc_amd import workloads as W

    w = W.wip_batch(batch, seed=seed)
    w["x0"][:, 1] += 0.3
    w["x0"][:, 3] += 1.0
    pend = w["pendulum"]
    ts = np.stack([pend.target_states(x, 0.5) for x in w["x0"]])
    w["goal"], w["targets"] = ts[:, -4:], ts[:, :-4]
    return w


def test_stage_kernel_warm_start_same_plans_fewer_sweeps_less_time():
    """Warm start of the stage-wise kernel (config 3's size, n = 50: the reference's loop
    examples/wheeled_inverted_pendulum.py:99-118 would pass initvals through solve_mpc's **kwargs, solve_mpc.py:20,43): the
    active rows' vectors and W stay in the PreparedSolve's workspace, the row ids in the WarmState. Re-solving the same batch
    needs NO iteration; a perturbed batch (the non-shifting case: re-targeting / re-linearising around nearby states) gives the
    cold plans = the oracle's with fewer iterations AND in less time (the shifting-horizon case stays cold: every active row
    changes identity each period)."""
    from qpmpc_amd import PreparedSolve, WarmState
    from qpmpc_amd import workloads as W

    w = _saturating_wip(1024)
    bp = W.to_batch_problem(w)
    ws = WarmState(bp)
    assert ws.kind == "stage"
    run = PreparedSolve(bp, return_multipliers=True, warm_state=ws)
    cold = PreparedSolve(bp, return_multipliers=True)
    run.launch()
    cold.launch()
    torch.cuda.synchronize()
    U0, it0 = run.U.clone(), run.iters.clone()
    assert torch.equal(U0, cold.U) and torch.equal(it0, cold.iters) and (run.status == 0).all()
    assert int(it0.max()) >= 3 and float(it0.float().mean()) >= 1.0  # the boxes are active
    act = ws.active_set.cpu().numpy()
    lam = run.lam.cpu().numpy()
    for b in range(0, 1024, 97):
        assert set(np.flatnonzero(lam[b] > 1e-9)) <= set(int(a) for a in act[b] if a >= 0)
    run.set_warm_start(True)
    run.launch()
    torch.cuda.synchronize()
    assert (run.status == 0).all() and int(run.iters.max()) == 0
    assert float((run.U - U0).abs().max()) <= 1e-9 * max(1.0, float(U0.abs().max()))
    # perturbed states: same matrices, new x0 / goal / targets
    rng = np.random.default_rng(3)
    w2 = dict(w)
    w2["x0"] = w["x0"] + 0.01 * rng.standard_normal(w["x0"].shape)
    pend = w["pendulum"]
    ts = np.stack([pend.target_states(x, 0.5) for x in w2["x0"]])
    w2["goal"], w2["targets"] = ts[:, -4:], ts[:, :-4]
    bp2 = W.to_batch_problem(w2)
    for dst, src in ((bp, bp2),):
        dst.initial_state.copy_(src.initial_state)
        dst.goal_state.copy_(src.goal_state)
        dst.target_states.copy_(src.target_states)
    cold.launch()
    run.launch()
    torch.cuda.synchronize()
    Uo, _, sto, _ = oracle.solve_workload(w2)
    ok = sto == 0
    assert np.array_equal(run.status.cpu().numpy() == 0, ok) and np.array_equal(cold.status.cpu().numpy() == 0, ok)
    assert (np.abs(run.U.cpu().numpy() - Uo)[ok] / _scale(Uo[ok])).max() <= 1e-7
    assert float(run.iters.float().mean()) < 0.5 * float(cold.iters.float().mean())
    # time, not only iterations: the same perturbed batch again and again, cold vs warm
    def per_launch(solver, reps=30):
        solver.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            solver.launch()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    t_cold, t_warm = per_launch(cold), per_launch(run)
    print(f"stage kernel, 1024 saturating WIP problems: cold {t_cold:.1f} us ({float(cold.iters.float().mean()):.2f} iterations), "
          f"warm {t_warm:.1f} us ({float(run.iters.float().mean()):.2f})")
    assert t_warm <= t_cold


def test_stage_kernel_warm_start_with_a_foreign_or_garbage_state_is_still_correct():
    """The record of another PreparedSolve (other workspace: the tag does not match), random bytes, and a record taken for
    DIFFERENT matrices in the same workspace: the plans still equal the oracle's (cold restart, never a wrong plan)."""
    from qpmpc_amd import PreparedSolve, WarmState
    from qpmpc_amd import workloads as W

    w = _saturating_wip(256, seed=13)
    Uo, _, sto, _ = oracle.solve_workload(w)
    ok = sto == 0

    def check(run):
        torch.cuda.synchronize()
        assert np.array_equal(run.status.cpu().numpy() == 0, ok)
        assert (np.abs(run.U.cpu().numpy() - Uo)[ok] / _scale(Uo[ok])).max() <= 1e-7

    bp = W.to_batch_problem(w)
    ws_a, ws_b = WarmState(bp), WarmState(bp)
    a, b = PreparedSolve(bp, warm_state=ws_a), PreparedSolve(bp, warm_state=ws_b)
    a.launch()
    b.launch()
    check(a)
    # another solver's record: b's workspace holds b's vectors, but the record says "workspace of a"
    ws_b.buffer.copy_(ws_a.buffer)
    b.set_warm_start(True)
    b.launch()
    check(b)
    assert int(b.iters.max()) >= 3  # (it restarted cold)
    # garbage
    ws_b.buffer.copy_(torch.randint(0, 256, ws_b.buffer.shape, dtype=torch.uint8, device="cuda"))
    b.launch()
    check(b)
    # same workspace, matrices changed under the state (contract broken on purpose): a different sampling period
    w3 = _saturating_wip(256, seed=13)
    a.set_warm_start(True)
    a.launch()
    check(a)
    assert int(a.iters.max()) == 0
    bp.e.mul_(0.5)  # the bounds may change freely ...
    w_half = dict(w)
    w_half["e"] = w["e"] * 0.5
    Uh, _, sth, _ = oracle.solve_workload(w_half)
    a.launch()
    torch.cuda.synchronize()
    okh = sth == 0
    assert np.array_equal(a.status.cpu().numpy() == 0, okh)
    assert (np.abs(a.U.cpu().numpy() - Uh)[okh] / _scale(Uh[okh])).max() <= 1e-7
    bp.B.mul_(1.3)  # ... the matrices may not: the kept factor AND the stored vectors are stale
    a.set_warm_start(False)  # (a caller who changes the matrices must rebuild: KEEP_FACTOR again)
    a.launch()
    w_b = dict(w_half)
    w_b["B"] = w["B"] * 1.3
    Ub, _, stb, _ = oracle.solve_workload(w_b)
    torch.cuda.synchronize()
    okb = stb == 0
    assert np.array_equal(a.status.cpu().numpy() == 0, okb)
    assert (np.abs(a.U.cpu().numpy() - Ub)[okb] / _scale(Ub[okb])).max() <= 1e-7


def test_wide_stagewise_active_set_warm_start_same_plans_fewer_sweeps():
    """MPCQP_WARM_ACTIVE_SET in the wide stage-wise kernel (config 5's dimensions, float64 and float32): the rows active at
    the end of the previous solve ride along with the first backward sweep, so the iterations find their vectors ready. Same
    statuses and same plans as the cold solve -- the cached rows steer the SELECTION since round 6, not the minimiser; the
    iteration counts stay close -- for the same batch, for perturbed states, for garbage ids (timings printed, not asserted:
    256 problems are a latency-bound launch either way)."""
    import time

    from qpmpc_amd import PreparedSolve, WarmState, _capi
    from qpmpc_amd import workloads as W

    w = W.synthetic_ltv_batch(256)
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 1e-5)):
        bp_c, bp_w = W.to_batch_problem(w, dtype=dtype), W.to_batch_problem(w, dtype=dtype)
        ws = WarmState(bp_w)
        cold, warm = PreparedSolve(bp_c), PreparedSolve(bp_w, warm_state=ws)
        g = torch.Generator(device="cuda").manual_seed(2)
        x0 = bp_c.initial_state.clone()
        for period in range(4):
            cold.launch()
            warm.launch()
            torch.cuda.synchronize()
            assert torch.equal(cold.status, warm.status), (dtype, period)
            assert abs(float(cold.iters.float().mean()) - float(warm.iters.float().mean())) <= 1.0, (dtype, period)
            ok = cold.status == 0
            assert bool(ok.any())
            d = (warm.U[ok] - cold.U[ok]).abs().max().item()
            assert d <= tol * max(1.0, cold.U[ok].abs().max().item()), (dtype, period, d)
            if period == 0:
                act = ws.active_set
                assert int((act >= 0).sum().item()) > 0  # the record holds the active rows
            xn = x0 + 0.02 * torch.randn(x0.shape, dtype=x0.dtype, device="cuda", generator=g)
            bp_c.initial_state.copy_(xn)
            bp_w.initial_state.copy_(xn)
            warm.set_warm_start("active_set", warm_shift=0)
        # garbage ids (out of range, duplicates, negative counts) are hints only
        rec = ws.buffer.view(torch.int32)
        rec.copy_(torch.randint(-5, 2000, rec.shape, dtype=torch.int32, device="cuda", generator=g))
        cold.launch()
        warm.launch()
        torch.cuda.synchronize()
        assert torch.equal(cold.status, warm.status)
        ok = cold.status == 0
        assert (warm.U[ok] - cold.U[ok]).abs().max().item() <= tol * max(1.0, cold.U[ok].abs().max().item())

    def timed(run):
        for _ in range(3):
            run.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run.launch()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    tc, tw = timed(cold), timed(warm)
    print(f"wide stage-wise kernel, 256 config-5 problems (float32): cold {tc * 100:.3f} ms, warm rows {tw * 100:.3f} ms per launch")


def test_wide_stagewise_warm_rows_under_the_default_selection_rule_against_the_oracle():
    """The same warm start under the DEFAULT rule (lazy slacks: the rows that ride along do steer the selection, so the iterates
    may differ from the cold solve's): plans and statuses against the C oracle over four periods of perturbed states, float64."""
    from qpmpc_amd import PreparedSolve, WarmState
    from qpmpc_amd import workloads as W

    w = W.synthetic_ltv_batch(96)
    bp = W.to_batch_problem(w)
    ws = WarmState(bp)
    run = PreparedSolve(bp, warm_state=ws)
    rng = np.random.default_rng(8)
    x0 = w["x0"].copy()
    for period in range(4):
        run.launch()
        torch.cuda.synchronize()
        Uo, _, sto, _ = oracle.solve_workload(w)
        ok = sto == 0
        assert np.array_equal(run.status.cpu().numpy() == 0, ok), period
        assert (np.abs(run.U.cpu().numpy() - Uo)[ok] / _scale(Uo[ok])).max() <= 1e-7, period
        w = dict(w)
        w["x0"] = x0 + 0.02 * rng.standard_normal(x0.shape)
        bp.initial_state.copy_(torch.from_numpy(w["x0"]).to(bp.initial_state))
        run.set_warm_start("active_set", warm_shift=0)


def test_float32_problems_of_the_small_kernels_size_keep_the_float64_kernels_record():
    """A float32 launch of at most 160 variables is solved by the float64 kernels on converted copies (promote_f32), so its
    WarmState is THEIR record: the library says which (mpcqp_warm_state_kind), the binding does not re-derive the dispatch.
    n = 16: the operator record -- warm re-solve without an iteration; nx = 3, N = 40: the narrow stage-wise kernel's record."""
    from qpmpc_amd import PreparedSolve, WarmState
    from qpmpc_amd import workloads as W

    w = W.triple_integrator_batch(64)
    bp32 = W.to_batch_problem(w, dtype=torch.float32)
    ws = WarmState(bp32)
    assert ws.kind == "pair" and ws.bytes_per_problem == WarmState(W.to_batch_problem(w)).bytes_per_problem
    run = PreparedSolve(bp32, warm_state=ws)
    run.launch()
    torch.cuda.synchronize()
    U0, it0 = run.U.clone(), run.iters.clone()
    assert (run.status == 0).all() and int(it0.max()) > 3
    act = ws.active_set
    assert act.shape == (64, 16) and int((act >= 0).sum()) == int(it0.sum())  # (no drops on this family: one slot per iteration)
    run.set_warm_start(True)
    run.launch()
    torch.cuda.synchronize()
    assert (run.status == 0).all() and int(run.iters.max()) == 0
    assert float((run.U - U0).abs().max()) <= 1e-4 * max(1.0, float(U0.abs().max()))
    # the narrow stage-wise kernel's size in float32: its record, its PreparedSolve-only warm start
    wn = W.wip_batch(32, N=40)
    bpn = W.to_batch_problem(wn, dtype=torch.float32)
    wsn = WarmState(bpn)
    assert wsn.kind == "stage" and wsn.bytes_per_problem == WarmState(W.to_batch_problem(wn)).bytes_per_problem
    rn = PreparedSolve(bpn, warm_state=wsn)
    rn.launch()
    torch.cuda.synchronize()
    Un = rn.U.clone()
    rn.set_warm_start(True)
    rn.launch()
    torch.cuda.synchronize()
    assert (rn.status == 0).all() and int(rn.iters.max()) == 0
    assert float((rn.U - Un).abs().max()) <= 1e-4 * max(1.0, float(Un.abs().max()))
    assert int((wsn.active_set >= -1).all())


# ---------------------------------------------------------------- seed steps (MPCQP_OPT_SEED_VIOLATED, MPCQP_WARM_ACTIVE_SET)
@pytest.mark.parametrize("family", ["triple", "humanoid", "random"])
def test_seeded_start_gives_the_plain_iterations_minimiser(family):
    """Rows violated at the unconstrained minimiser entered by seed steps (no selection, no ratio test, negative
    multipliers repaired by drops) before the Goldfarb-Idnani iterations: same statuses, same plans as the plain
    iterations and as the oracle -- infeasible items of the humanoid sweep and inconsistent random rows included."""
    from qpmpc_amd import _capi
    from qpmpc_amd import workloads as W

    if family == "random":
        import sys, os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        from stress_stagewise import random_ltv

        rng = np.random.default_rng(77)
        w = random_ltv(rng, 384, 4, 2, 8, 2, 0.2)
    else:
        w = W.triple_integrator_batch(1023) if family == "triple" else W.humanoid_batch(2047)
    bp = W.to_batch_problem(w)
    U0, st0, it0, lam0 = _solve(bp)
    U1, st1, it1, lam1 = _solve(bp, flags=_capi.OPT_SEED_VIOLATED)
    Uo, _, sto, _ = oracle.solve_workload(w)
    assert np.array_equal(st0, st1) and np.array_equal(st1 == 0, sto == 0)
    ok = st0 == 0
    assert ok.any()
    assert (np.abs(U1[ok] - U0[ok]) / _scale(U0[ok])).max() <= 1e-8
    assert (np.abs(U1[ok] - Uo[ok]) / _scale(Uo[ok])).max() <= 1e-8
    assert np.abs(lam1[ok] - lam0[ok]).max() <= 1e-6 * max(1.0, np.abs(lam0[ok]).max())
    assert (lam1[ok] >= 0).all()


def test_active_set_warm_start_receding_horizon_same_plans():
    """MPCQP_WARM_ACTIVE_SET: last period's active ROWS (ids only), moved one step down the horizon, enter first by
    seed steps on the new period's own matrices. Plans and statuses must be those of the cold solve every period."""
    from qpmpc_amd import PreparedSolve, WarmState
    from qpmpc_amd import workloads as W

    w = W.triple_integrator_batch(1024, heterogeneous=False)
    bp_c, bp_w = W.to_batch_problem(w), W.to_batch_problem(w)
    ws = WarmState(bp_w)
    cold, warm = PreparedSolve(bp_c), PreparedSolve(bp_w, warm_state=ws)
    A = torch.as_tensor(w["A"], device="cuda")
    Bm = torch.as_tensor(w["B"], device="cuda").reshape(3)
    for period in range(10):
        cold.launch()
        warm.launch()
        torch.cuda.synchronize()
        sc, sw = cold.status.cpu().numpy(), warm.status.cpu().numpy()
        assert np.array_equal(sc, sw), period
        ok = sc == 0
        Uc, Uw = cold.U.cpu().numpy(), warm.U.cpu().numpy()
        assert (np.abs(Uw[ok] - Uc[ok]) / _scale(Uc[ok])).max() <= 1e-8, period
        xn = bp_c.initial_state @ A.T + cold.U[:, :1] * Bm
        bp_c.initial_state.copy_(xn)
        bp_w.initial_state.copy_(xn)
        warm.set_warm_start("active_set", warm_shift=2)  # mk = 2 rows per step
    # the record of such a launch holds row ids only; an operator-mode start that meets it must fall back to a cold start
    warm.set_warm_start(True)
    cold.launch()
    warm.launch()
    torch.cuda.synchronize()
    ok = cold.status.cpu().numpy() == 0
    assert np.array_equal(cold.status.cpu().numpy(), warm.status.cpu().numpy())
    assert (np.abs(warm.U.cpu().numpy()[ok] - cold.U.cpu().numpy()[ok]) / _scale(cold.U.cpu().numpy()[ok])).max() <= 1e-8


def test_active_set_warm_start_with_garbage_ids_and_lipm_loop():
    """Stored ids are not trusted: random, duplicate and out-of-range rows (and shifts past the horizon) still end at the
    oracle's plans; the LIPM walking loop warm-started from shifted active sets follows the cold loop's trajectory."""
    from qpmpc_amd import WarmState
    from qpmpc_amd import workloads as W
    from qpmpc_amd.closed_loop import LIPMWalkingLoop

    w = W.humanoid_batch(512)
    bp = W.to_batch_problem(w)
    Uo, _, sto, _ = oracle.solve_workload(w)
    ok = sto == 0
    ws = WarmState(bp)
    g = torch.Generator(device="cuda").manual_seed(11)
    for shift in (0, 2, 7, 40, -3):
        ids = torch.randint(-3, 40, ws.active_set.shape, dtype=torch.int32, device="cuda", generator=g)
        ws.buffer[:, 16 * 16 * 8:] = ids.view(torch.uint8)
        U, st, _, _ = _solve(bp, warm_state=ws, warm_start="active_set", warm_shift=shift)
        assert np.array_equal(st, sto), shift
        assert (np.abs(U[ok] - Uo[ok]) / _scale(Uo[ok])).max() <= 1e-8, shift
    rng = np.random.default_rng(3)
    strides = np.stack([-rng.uniform(0.12, 0.2, 96), rng.uniform(0.12, 0.2, 96)], axis=1)
    a = LIPMWalkingLoop(96, strides=strides, index=np.arange(96) % 8)
    b = LIPMWalkingLoop(96, strides=strides, index=np.arange(96) % 8, warm_start="active_set")
    a.step(30)
    b.step(30)
    torch.cuda.synchronize()
    assert a.stats()["failed"] == 0 and b.stats()["failed"] == 0
    assert float((a.states - b.states).abs().max()) <= 1e-9
