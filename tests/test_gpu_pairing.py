"""GPU tests (-m gpu): the pairing order of the small-problem fused kernel (MpcqpSolveOpts.order, ABI 9) and the device-side
counting sort that makes one from last period's iteration counts (mpcqp_order_by_count). The reference solves one problem per
call (qpmpc/solve_mpc.py:43), so WHICH two problems share a wavefront must never show in the results: statuses and iteration
counts are compared exactly with the natural order's, plans to rounding (1e-8); the launch time is what changes (BASELINE config 4 =
examples/humanoid_one_step.py:42-80 over 65,536 initial states)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch", [1, 5, 1023, 1025, 70001])
def test_order_by_count_is_a_permutation_sorted_longest_first(batch):
    from qpmpc_amd import pairing_order

    g = torch.Generator().manual_seed(batch)
    counts = torch.randint(-3, 1500, (batch,), generator=g, dtype=torch.int32).cuda()
    order = pairing_order(counts)
    torch.cuda.synchronize()
    o = order.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(o), np.arange(batch))
    clamped = np.clip(counts.cpu().numpy(), 0, 1023)[o]
    assert np.all(np.diff(clamped) <= 0)


@pytest.mark.parametrize("kind,batch", [("triple", 777), ("triple", 9000), ("humanoid", 5001)])
def test_results_do_not_depend_on_the_pairing_order(kind, batch):
    from qpmpc_amd import pairing_order, solve_mpc_batch
    from qpmpc_amd import workloads as W

    w = W.triple_integrator_batch(batch, seed=5) if kind == "triple" else W.humanoid_batch(batch, seed=7)
    bp = W.to_batch_problem(w)
    ref = solve_mpc_batch(bp, return_multipliers=True)
    torch.cuda.synchronize()
    assert int((ref.status == 0).sum()) > batch // 2
    g = torch.Generator().manual_seed(1)
    for order in (pairing_order(ref.iters), torch.randperm(batch, generator=g).to(torch.int32).cuda()):
        got = solve_mpc_batch(bp, return_multipliers=True, order=order)
        torch.cuda.synchronize()
        assert torch.equal(got.status, ref.status) and torch.equal(got.iters, ref.iters)
        # (a problem that moves to the other half of a wavefront sees that half's order of summation: rounding only -- 1e-15 on
        # the triple integrator, amplified to 1.3e-10 by the conditioning of the humanoid problem; the contract is 1e-6)
        ok = ref.status == 0  # (unsolved items carry no plan)
        scale = ref.U[ok].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
        assert float(((got.U[ok] - ref.U[ok]).abs() / scale).max()) <= 1e-8
        lscale = ref.multipliers[ok].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
        assert float(((got.multipliers[ok] - ref.multipliers[ok]).abs() / lscale).max()) <= 1e-8


def test_prepared_solve_takes_a_new_order_every_period():
    from qpmpc_amd import PreparedSolve, pairing_order
    from qpmpc_amd import workloads as W

    bp = W.to_batch_problem(W.triple_integrator_batch(3000, seed=9))
    prep = PreparedSolve(bp)
    prep.launch()
    torch.cuda.synchronize()
    U0, it0 = prep.U.clone(), prep.iters.clone()
    prep.set_order(pairing_order(prep.iters))
    prep.launch()
    torch.cuda.synchronize()
    assert torch.equal(prep.iters, it0) and float((prep.U - U0).abs().max()) <= 1e-11 * max(1.0, float(U0.abs().max()))
    prep.set_order(None)
    prep.launch()
    torch.cuda.synchronize()
    assert torch.equal(prep.U, U0)


def test_order_is_refused_where_no_kernel_takes_it():
    from qpmpc_amd import BackendError, WarmState, _capi, solve_mpc_batch
    from qpmpc_amd import workloads as W

    bp = W.to_batch_problem(W.triple_integrator_batch(64, seed=5))
    order = torch.arange(64, dtype=torch.int32, device="cuda")
    for kw in ({"flags": _capi.OPT_FORCE_LDS}, {"flags": _capi.OPT_ONE_PER_WAVE}, {"warm_state": WarmState(bp)},
               {"formulation": "stagewise"}):
        with pytest.raises(BackendError):
            solve_mpc_batch(bp, order=order, **kw)
    wide = W.to_batch_problem(W.wip_batch(8))  # n = 50: the stage-wise kernel's size
    with pytest.raises(BackendError):
        solve_mpc_batch(wide, order=torch.arange(8, dtype=torch.int32, device="cuda"))
    with pytest.raises(BackendError):
        solve_mpc_batch(bp, order=order.to(torch.int64))
    with pytest.raises(BackendError):  # (the kernel indexes unchecked: the host refuses an order of another length)
        solve_mpc_batch(bp, order=order[:32].contiguous())


def test_walking_loop_launches_take_an_order_set_from_outside():
    """examples/lipm_walking_controller.py:307-335 for 3000 walkers, 40 periods: the loop's prepared solve re-paired by last
    period's counts (in place, behind the launch that still reads the old order) must leave every trajectory where it was, to
    rounding -- for the per-problem build and for the model factored once (mpcqp_solve_model_bounds_batch takes the order too)."""
    from qpmpc_amd import pairing_order
    from qpmpc_amd.closed_loop import LIPMWalkingLoop

    rng = np.random.default_rng(3)
    B = 3000
    kw = dict(strides=np.stack([-rng.uniform(0.12, 0.2, B), rng.uniform(0.12, 0.2, B)], axis=1),
              foot_size=rng.uniform(0.05, 0.08, B), index=rng.integers(0, 8, B))
    for shared in (False, True):
        ref, got = LIPMWalkingLoop(B, shared_model=shared, **kw), LIPMWalkingLoop(B, shared_model=shared, **kw)
        ref.step(40)
        order = torch.empty_like(got.solver.iters)
        for _ in range(40):
            got.step(1)
            pairing_order(got.solver.iters, out=order)  # (after the launch that reads the old order, in stream order)
            got.solver.set_order(order)
        torch.cuda.synchronize()
        assert ref.stats()["failed"] == got.stats()["failed"] and ref.stats()["mean_iters"] == got.stats()["mean_iters"]
        assert float((ref.states - got.states).abs().max()) <= 1e-9
        assert torch.equal(ref.index, got.index)


def test_shared_model_solve_takes_the_order():
    """Config 4 the way the reference would run it -- MPCQP built once, update_cost_vector / update_constraint_vector per state
    (qpmpc/mpc_qp.py:129-163) --: the model-mode launch with an order gives the natural order's results."""
    from qpmpc_amd import SharedModel, pairing_order
    from qpmpc_amd import workloads as W

    bp = W.to_batch_problem(W.humanoid_batch(6001, seed=11))
    run = SharedModel(bp).prepare(bp, return_multipliers=True)
    run.launch()
    torch.cuda.synchronize()
    U0, st0, it0 = run.U.clone(), run.status.clone(), run.iters.clone()
    run.set_order(pairing_order(run.iters))
    run.launch()
    torch.cuda.synchronize()
    ok = st0 == 0
    assert torch.equal(run.status, st0) and torch.equal(run.iters, it0)
    scale = U0[ok].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
    assert float(((run.U[ok] - U0[ok]).abs() / scale).max()) <= 1e-8


@pytest.mark.parametrize("nx,nu,N,mk,with_c,with_d,wx", [
    (4, 2, 6, 3, True, True, 0.5),     # the kernel's <4, 0> instantiation: C and D, stage + terminal cost
    (3, 4, 4, 8, True, True, 0.5),     # <3, 0>: n = 16, m = 32 exactly
    (4, 1, 12, 2, True, False, None),  # <4, 2>: the register-pipelined chain
])
def test_every_instantiation_of_the_kernel_takes_the_order(nx, nu, N, mk, with_c, with_d, wx):
    """Random LTV families of the small-problem kernel's other template instantiations, odd batch, reversed and sorted orders."""
    from test_gpu_parity import _random_ltv_workload

    from qpmpc_amd import pairing_order, solve_mpc_batch
    from qpmpc_amd import workloads as W

    batch = 1001
    bp = W.to_batch_problem(_random_ltv_workload(np.random.default_rng(7 * nx + N), batch, nx, nu, N, mk, with_c, with_d, wx=wx))
    ref = solve_mpc_batch(bp, return_multipliers=True)
    torch.cuda.synchronize()
    ok = ref.status == 0
    assert int(ok.sum()) > 900
    for order in (pairing_order(ref.iters), torch.arange(batch - 1, -1, -1, dtype=torch.int32, device="cuda")):
        got = solve_mpc_batch(bp, return_multipliers=True, order=order)
        torch.cuda.synchronize()
        assert torch.equal(got.status, ref.status) and torch.equal(got.iters, ref.iters)
        scale = ref.U[ok].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
        assert float(((got.U[ok] - ref.U[ok]).abs() / scale).max()) <= 1e-8
