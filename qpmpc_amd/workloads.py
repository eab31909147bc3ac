"""Synthetic problem batches of BASELINE.json's configs.

Pure NumPy data generators (seeded); nothing here computes a solution. Each
function returns a dict of float64 arrays laid out as ``BatchMPCProblem``
expects plus the scalar weights, so the same inputs can be handed to the HIP
path and to the CPU oracle.
"""
from __future__ import annotations

import numpy as np

from .systems import WheeledInvertedPendulum


def _pack(A, B, C, D, e, N, wt, wx, wu, x0, goal=None, targets=None, name=""):
    return dict(A=A, B=B, C=C, D=D, e=e, N=N, wt=wt, wx=wx, wu=wu, x0=x0, goal=goal, targets=targets, name=name)


def triple_integrator_matrices(N: int = 16, horizon: float = 1.0):
    """A, B, C, e of examples/triple_integrator.py:15-26."""
    T = horizon / N
    A = np.array([[1.0, T, T * T / 2.0], [0.0, 1.0, T], [0.0, 0.0, 1.0]])
    B = np.array([[T ** 3 / 6.0], [T * T / 2.0], [T]])
    C = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, -1.0]])
    e = np.array([3.0, 3.0])
    return A, B, C, e


def triple_integrator_batch(batch: int = 4096, seed: int = 20250614, heterogeneous: bool = True):
    """Config 2: batch of config-1 problems with per-problem x0 and goal.

    heterogeneous=True stacks A, B, C, e per problem and per step (a real build
    per problem, the headline mode); False shares one copy (stride-0 operands).
    """
    N = 16
    A, B, C, e = triple_integrator_matrices(N)
    rng = np.random.default_rng(seed)
    x0 = np.stack([rng.uniform(-0.5, 0.5, batch), rng.uniform(-0.5, 0.5, batch), rng.uniform(-2.5, 2.5, batch)], 1)
    goal = np.stack([rng.uniform(0.5, 1.5, batch), np.zeros(batch), np.zeros(batch)], 1)
    if heterogeneous:
        A = np.ascontiguousarray(np.broadcast_to(A, (batch, N, 3, 3)))
        B = np.ascontiguousarray(np.broadcast_to(B, (batch, N, 3, 1)))
        C = np.ascontiguousarray(np.broadcast_to(C, (batch, N, 2, 3)))
        e = np.ascontiguousarray(np.broadcast_to(e, (batch, N, 2)))
    return _pack(A, B, C, None, e, N, 1.0, None, 1e-6, x0, goal, name="triple_integrator_N16")


def humanoid_matrices():
    """Data of examples/humanoid_one_step.py:17-29,42-80 (N=16, T=0.15625)."""
    N, horizon = 16, 2.5
    T = horizon / N
    A = np.array([[1.0, T, T * T / 2.0], [0.0, 1.0, T], [0.0, 0.0, 1.0]])
    B = np.array([[T ** 3 / 6.0], [T * T / 2.0], [T]])
    zmp = np.array([1.0, 0.0, -0.8 / 9.81])
    C = np.array([zmp, -zmp])
    dsp, ssp, foot, start, end = 0.1, 0.7, 0.1, 0.0, 0.3
    n0, n1, n2 = int(round(dsp / T)), int(round(ssp / T)), int(round(dsp / T))
    e = np.zeros((N, 2))
    for i in range(N):
        if i < n0:
            e[i] = (1000.0, 1000.0)
        elif i - n0 <= n1:
            e[i] = (start + 0.5 * foot, -(start - 0.5 * foot))
        elif i - n0 - n1 < n2:
            e[i] = (1000.0, 1000.0)
        else:
            e[i] = (end + 0.5 * foot, -(end - 0.5 * foot))
    return A, B, C, e, np.array([end, 0.0, 0.0])


def humanoid_batch(batch: int = 65536, seed: int = 2):
    """Config 4: initial-state sweep of the humanoid one-step problem (LTI shared
    dynamics, per-step e_k); infeasible x0 must come back with status 2."""
    A, B, C, e, goal = humanoid_matrices()
    rng = np.random.default_rng(seed)
    x0 = np.stack([rng.uniform(-0.04, 0.04, batch), rng.uniform(-0.15, 0.15, batch), rng.uniform(-0.3, 0.3, batch)], 1)
    return _pack(A, B, C, None, e, 16, 1.0, None, 1e-3, x0, goal, name="humanoid_one_step_N16")


def wip_batch(batch: int = 1024, N: int = 50, sampling_period: float = 0.024, seed: int = 1,
              target_vel: float = 0.5, ltv: bool = True):
    """Config 3: wheeled inverted pendulum, N=50 at T=0.024 s (the default T=0.1
    gives an indefinite P at N=50). ltv=True passes A, B per step."""
    pend = WheeledInvertedPendulum(nb_timesteps=N, sampling_period=sampling_period)
    A, B = pend.discretized_dynamics()
    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal((batch, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
    ts = np.stack([pend.target_states(x, target_vel) for x in x0])
    if ltv:
        A = np.ascontiguousarray(np.broadcast_to(A, (N, 4, 4)))
        B = np.ascontiguousarray(np.broadcast_to(B, (N, 4, 1)))
    D = np.array([[1.0], [-1.0]])
    e = np.array([pend.max_ground_accel, pend.max_ground_accel])
    out = _pack(A, B, None, D, e, N, 10.0, 1.0, 1e-3, x0, ts[:, -4:], ts[:, :-4], name=f"wip_N{N}")
    out["pendulum"] = pend
    return out


def synthetic_ltv_batch(batch: int = 8192, nx: int = 12, nu: int = 4, N: int = 64, seed: int = 3):
    """Config 5: random stable LTV dynamics, input box + 4-state box (mk = 16)."""
    rng = np.random.default_rng(seed)
    A = np.empty((batch, N, nx, nx))
    for b in range(batch):
        Qs, _ = np.linalg.qr(rng.standard_normal((N, nx, nx)))
        A[b] = 0.98 * Qs
    B = rng.standard_normal((batch, N, nx, nu)) / np.sqrt(nx)
    D = np.vstack([np.eye(nu), -np.eye(nu), np.zeros((8, nu))])
    C = np.zeros((16, nx))
    for r in range(4):
        C[8 + r, r] = 1.0
        C[12 + r, r] = -1.0
    e = np.concatenate([np.ones(8), 5.0 * np.ones(8)])
    x0 = rng.standard_normal((batch, nx))
    return _pack(A, B, C, D, e, N, 10.0, 1.0, 1e-2, x0, np.zeros(nx), np.zeros(N * nx), name=f"synthetic_ltv_nx{nx}_nu{nu}_N{N}")


def synthetic_ltv_batch_slice(lo: int, hi: int, nx: int = 12, nu: int = 4, N: int = 64, seed: int = 3):
    """Problems lo..hi-1 of ONE global config-5 problem set (per-problem seeding), so that a batch
    strong-sharded over any number of ranks is the same set of problems (bench.py --config 5)."""
    batch = hi - lo
    A = np.empty((batch, N, nx, nx))
    B = np.empty((batch, N, nx, nu))
    x0 = np.empty((batch, nx))
    for i in range(batch):
        rng = np.random.default_rng([seed, lo + i])
        Qs, _ = np.linalg.qr(rng.standard_normal((N, nx, nx)))
        A[i] = 0.98 * Qs
        B[i] = rng.standard_normal((N, nx, nu)) / np.sqrt(nx)
        x0[i] = rng.standard_normal(nx)
    D = np.vstack([np.eye(nu), -np.eye(nu), np.zeros((8, nu))])
    C = np.zeros((16, nx))
    for r in range(4):
        C[8 + r, r] = 1.0
        C[12 + r, r] = -1.0
    e = np.concatenate([np.ones(8), 5.0 * np.ones(8)])
    return _pack(A, B, C, D, e, N, 10.0, 1.0, 1e-2, x0, np.zeros(nx), np.zeros(N * nx), name=f"synthetic_ltv_nx{nx}_nu{nu}_N{N}")


def problem_from_workload(w: dict, b: int):
    """Host ``MPCProblem`` of item ``b`` of a workload dict, with per-step LISTS where the workload varies
    along the horizon (the reference's LTV convention, mpc_problem.py:177-245) and arrays where it does not."""
    from .mpc_problem import MPCProblem

    N = int(w["N"])

    def field(key, block_ndim):
        a = w[key]
        if a is None:
            return None
        a = np.asarray(a)
        extra = a.ndim - block_ndim
        if extra == 2:
            a = a[b] if a.shape[0] > 1 else a[0]
            extra = 1
        if extra == 1:
            return [a[k] for k in range(N)] if a.shape[0] == N else a[0]
        return a

    def state(key):
        a = w[key]
        if a is None:
            return None
        a = np.asarray(a)
        return a[b] if a.ndim == 2 else a

    p = MPCProblem(field("A", 2), field("B", 2), field("C", 2), field("D", 2), field("e", 1), N, w["wt"], w["wx"],
                   w["wu"], initial_state=state("x0"), goal_state=state("goal"))
    if w["targets"] is not None:
        p.update_target_states(state("targets"))
    return p


def to_batch_problem(w: dict, dtype=None, device=None):
    """Upload a workload dict to the device as a ``BatchMPCProblem``."""
    from .batch import BatchMPCProblem

    return BatchMPCProblem(
        w["A"], w["B"], w["C"], w["D"], w["e"], w["N"], w["wt"], w["wx"], w["wu"], w["x0"],
        goal_state=w["goal"], target_states=w["targets"], dtype=dtype, device=device,
    )


def algorithmic_bytes_per_problem(w: dict, esz: int = 8) -> int:
    """Inputs read once + U written once, counting only operands
    that are stored per problem (shared, stride-0 operands cost nothing per item)."""
    batch = w["x0"].shape[0]
    N = w["N"]
    total = 0
    for key in ("A", "B", "C", "D", "e"):
        a = w[key]
        if a is not None and a.ndim >= 3 and a.shape[0] == batch and a.size > a[0].size:
            total += a[0].size
    for key in ("x0", "goal", "targets"):
        a = w[key]
        if a is not None and a.ndim == 2 and a.shape[0] == batch:
            total += a.shape[1]
    nu = w["B"].shape[-1]
    total += N * nu
    return total * esz


def algorithmic_build_flops(nx: int, nu: int, N: int, mk: int, stage: bool, terminal: bool) -> float:
    """F_build of SURVEY.md 8d (dense, as the reference computes)."""
    n, m = N * nu, N * mk
    f = 2 * N * nx ** 3 + 2 * N * nx * nx * n + 2 * m * nx * n + 2 * m * nx * nx + 2 * m * nx
    if stage:
        f += 2 * N * nx * n * n
    if terminal:
        f += 2 * nx * n * n
    return float(f + 2 * N * nx * nx + 2 * N * nx * n)


def stagewise_executed_flops(nx: int, nu: int, N: int, mk: int, iters: float) -> float:
    """Floating-point operations the stage-wise kernels EXECUTE per problem (useful ones: one copy of every product, not
    the redundant lanes): the Riccati recursion (P A, P B, B'PA, B'PB, K = S^-1 B'PA, A_cl = A - B K, P_k = A'P A_cl,
    symmetrised), the two sweeps of the unconstrained minimiser, the initial slacks, and per active-set iteration one more
    sweep pair plus the slack update over the m rows and the |A| slots (|A| ~ iters / 2 on average)."""
    ricc = 2.0 * (2 * nx ** 3 + 3 * nx * nx * nu + 2 * nx * nu * nu) + nu ** 3
    sweep = 2.0 * (nx * nx + 2 * nx * nu) + 2.0 * nu * nu  # one step of one sweep
    slack0 = 2.0 * mk * (nx + nu)
    per_iter = 2 * N * sweep + N * slack0 + 2.0 * N * mk * (1.0 + 0.5 * iters)
    return float(N * (ricc + 2 * sweep + slack0) + iters * per_iter)


def algorithmic_solve_flops(n: int, m: int, iters: float) -> float:
    """Active-set model of SURVEY.md 8d: n^3/3 + iters (2 m n + 4 n^2)."""
    return n ** 3 / 3.0 + iters * (2.0 * m * n + 4.0 * n * n)
