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
    assert np.abs(U[0] - Uo).max() <= 1e-9 * max(1.0, np.abs(Uo).max())
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
    """N = 1024 (n = 1024 > 256): mpcqp_build_solve_batch returns MPCQP_ETOOLARGE, the stage-wise path solves a
    batch of 64 different problems; every solution is KKT-certified without any condensed matrix."""
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
    with pytest.raises(BackendError, match="-2"):
        solve_mpc_batch(big)
    plan = solve_mpc_batch(big, formulation="stagewise", return_multipliers=True)
    torch.cuda.synchronize()
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
    from qpmpc_amd import BackendError, solve_mpc_batch
    from qpmpc_amd import workloads as W

    bp = W.to_batch_problem(W.synthetic_ltv_batch(2, N=8))  # nx = 12
    with pytest.raises(BackendError, match="-6"):
        solve_mpc_batch(bp, formulation="stagewise")
