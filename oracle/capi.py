"""ORACLE (test infrastructure): ctypes access to oracle/liboracle.so.

Nothing under qpmpc_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

FLAG_P_TERMINAL, FLAG_P_STAGE, FLAG_Q_TERMINAL, FLAG_Q_STAGE = 1, 2, 4, 8
PAD_BOUND = 1e30  # padded (absent) constraint rows: 0 . u <= 1e30


class _Operand(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_int64), ("step_stride", C.c_int64)]


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "mpc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_gi_solve.restype = C.c_int
        _lib.oracle_build_solve_batch.restype = C.c_int
    return _lib


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def flags_of(problem) -> int:
    """P-inclusion follows "is not None" (mpc_qp.py:102,104); q-inclusion follows
    the >1e-10 thresholds AND defined states (mpc_problem.py:141-166 + the
    swallowed exception of mpc_qp.py:119-122)."""
    f = 0
    wt, wx = problem.terminal_cost_weight, problem.stage_state_cost_weight
    if wt is not None:
        f |= FLAG_P_TERMINAL
    if wx is not None:
        f |= FLAG_P_STAGE
    t_on = wt is not None and wt > 1e-10
    s_on = wx is not None and wx > 1e-10
    if t_on and problem.goal_state is None:
        return f  # raise happens before anything is accumulated
    if t_on:
        f |= FLAG_Q_TERMINAL
    if s_on and problem.target_states is not None:
        f |= FLAG_Q_STAGE
    return f


def stack_problem(problem) -> dict:
    """Dense per-step stacks of a duck-typed MPCProblem; ragged m_k padded with
    zero rows and PAD_BOUND bounds."""
    N, nx, nu = problem.nb_timesteps, problem.state_dim, problem.input_dim
    A = np.stack([np.asarray(problem.get_transition_state_matrix(k), dtype=float).reshape(nx, nx) for k in range(N)])
    B = np.stack([np.asarray(problem.get_transition_input_matrix(k), dtype=float).reshape(nx, nu) for k in range(N)])
    es = [np.asarray(problem.get_ineq_vector(k), dtype=float).ravel() for k in range(N)]
    mks = [len(v) for v in es]
    mk = max(mks)
    Cs = [problem.get_ineq_state_matrix(k) for k in range(N)]
    Ds = [problem.get_ineq_input_matrix(k) for k in range(N)]
    Cst = None if all(c is None for c in Cs) else np.zeros((N, mk, nx))
    Dst = None if all(d is None for d in Ds) else np.zeros((N, mk, nu))
    est = np.full((N, mk), PAD_BOUND)
    for k in range(N):
        est[k, : mks[k]] = es[k]
        if Cs[k] is not None:
            Cst[k, : mks[k]] = np.asarray(Cs[k], dtype=float).reshape(mks[k], nx)
        if Ds[k] is not None:
            Dst[k, : mks[k]] = np.asarray(Ds[k], dtype=float).reshape(mks[k], nu)
    rows = np.concatenate([k * mk + np.arange(mks[k]) for k in range(N)])
    return dict(
        nx=nx, nu=nu, N=N, mk=mk, mks=mks, rows=rows, A=A, B=B, C=Cst, D=Dst, e=est,
        x0=None if problem.initial_state is None else np.asarray(problem.initial_state, dtype=float),
        goal=None if problem.goal_state is None else np.asarray(problem.goal_state, dtype=float),
        targets=None if problem.target_states is None else np.asarray(problem.target_states, dtype=float),
        flags=flags_of(problem),
        wt=0.0 if problem.terminal_cost_weight is None else float(problem.terminal_cost_weight),
        wx=0.0 if problem.stage_state_cost_weight is None else float(problem.stage_state_cost_weight),
        wu=float(problem.stage_input_cost_weight),
    )


def condense_one(problem):
    """C restatement of the build for one problem; returns (P, q, G, h) with the
    reference's row count (padding rows removed)."""
    lib = _load()
    s = stack_problem(problem)
    nx, nu, N, mk = s["nx"], s["nu"], s["N"], s["mk"]
    n, m = N * nu, N * mk
    P, q, G, h = np.zeros((n, n)), np.zeros(n), np.zeros((m, n)), np.zeros(m)
    Psi, xf = np.zeros(((N + 1) * nx, n)), np.zeros((N + 1, nx))
    Phi = np.zeros(((N + 1) * nx, nx))
    lib.oracle_condense_one(
        C.c_int(nx), C.c_int(nu), C.c_int(N), C.c_int(mk),
        _dp(s["A"]), C.c_int64(nx * nx), _dp(s["B"]), C.c_int64(nx * nu),
        _dp(s["C"]), C.c_int64(mk * nx), _dp(s["D"]), C.c_int64(mk * nu),
        _dp(s["e"]), C.c_int64(mk), _dp(s["x0"]), _dp(s["goal"]), _dp(s["targets"]),
        C.c_int(s["flags"]), C.c_double(s["wt"]), C.c_double(s["wx"]), C.c_double(s["wu"]),
        _dp(P), _dp(q), _dp(G), _dp(h), _dp(Psi), _dp(xf), _dp(Phi),
    )
    rows = s["rows"]
    return dict(P=P, q=q, G=G[rows], h=h[rows], Psi=Psi[: N * nx], psi_last=Psi[N * nx:],
                Phi=Phi[: N * nx], phi_last=Phi[N * nx:], xf=xf)


def gi_solve(P, q, G, h, max_iter: int = 10000, tol: float = 1e-12):
    """Goldfarb-Idnani on a dense QP. Returns (x, lam, status, iters)."""
    lib = _load()
    P = np.ascontiguousarray(P, dtype=float)
    q = np.ascontiguousarray(q, dtype=float)
    G = np.ascontiguousarray(G, dtype=float).reshape(-1, P.shape[0])
    h = np.ascontiguousarray(h, dtype=float)
    n, m = P.shape[0], G.shape[0]
    x, lam, it = np.zeros(n), np.zeros(max(m, 1)), C.c_int(0)
    st = lib.oracle_gi_solve(C.c_int(n), C.c_int(m), _dp(P), _dp(q), _dp(G), _dp(h), C.c_int(max_iter),
                             C.c_double(tol), _dp(x), _dp(lam), C.byref(it))
    return x, lam[:m], int(st), int(it.value)


def rollout_one(A, B, x0, U):
    lib = _load()
    A = np.ascontiguousarray(A, dtype=float)
    B = np.ascontiguousarray(B, dtype=float)
    N, nx, nu = A.shape[0], A.shape[1], B.shape[2]
    U = np.ascontiguousarray(U, dtype=float).reshape(N, nu)
    x0 = np.ascontiguousarray(x0, dtype=float)
    X = np.zeros((N + 1, nx))
    lib.oracle_rollout_one(C.c_int(nx), C.c_int(nu), C.c_int(N), _dp(A), C.c_int64(nx * nx), _dp(B),
                           C.c_int64(nx * nu), _dp(x0), _dp(U), _dp(X))
    return X


def _operand(a: Optional[np.ndarray], per_item: int, per_step: int, batch: int, steps: int) -> _Operand:
    """Describe an array shaped [B?, N?, ...] by element strides (0 = shared)."""
    if a is None:
        return _Operand(None, 0, 0)
    size = a.size
    if size == per_step:  # shared across batch and steps
        bs, ks = 0, 0
    elif size == per_step * steps:  # [N, ...]
        bs, ks = 0, per_step
    elif size == per_step * batch and steps != batch:
        bs, ks = per_step, 0
    elif size == per_step * steps * batch:
        bs, ks = per_step * steps, per_step
    else:
        raise ValueError(f"operand of size {size} fits neither layout")
    del per_item
    return _Operand(a.ctypes.data, bs, ks)


def build_solve_batch(nx, nu, N, mk, flags, wt, wx, wu, A, B, Cm, D, e, x0, goal, targets,
                      max_iter: int = 10000, tol: float = 1e-12, layout=None):
    """Whole path over a batch, single thread, float64.

    Arrays are C-contiguous float64; ``layout`` maps operand name -> (batch_stride,
    step_stride) in elements, defaults inferred from sizes ([B,N,..], [N,..] or [..]).
    Returns (U [B,n], lam [B,m], status [B], iters [B]).
    """
    lib = _load()
    batch = x0.shape[0]
    n, m = N * nu, N * mk
    arrs = dict(A=A, B=B, C=Cm, D=D, e=e, x0=x0, goal=goal, targets=targets)
    per = dict(A=nx * nx, B=nx * nu, C=mk * nx, D=mk * nu, e=mk, x0=nx, goal=nx, targets=nx * N)
    keep, ops = [], {}
    for name, a in arrs.items():
        if a is not None:
            a = np.ascontiguousarray(a, dtype=float)
            keep.append(a)
        if layout and name in layout and a is not None:
            ops[name] = _Operand(a.ctypes.data, layout[name][0], layout[name][1])
        elif name in ("x0", "goal", "targets"):
            ops[name] = _Operand(None, 0, 0) if a is None else _Operand(
                a.ctypes.data, 0 if a.size == per[name] else per[name], 0)
        else:
            ops[name] = _operand(a, 0, per[name], batch, N)
    U = np.zeros((batch, n))
    lam = np.zeros((batch, max(m, 1)))
    status = np.zeros(batch, dtype=np.int32)
    iters = np.zeros(batch, dtype=np.int32)
    lib.oracle_build_solve_batch(
        C.c_int(nx), C.c_int(nu), C.c_int(N), C.c_int(mk), C.c_int(flags), C.c_double(wt), C.c_double(wx),
        C.c_double(wu), C.byref(ops["A"]), C.byref(ops["B"]), C.byref(ops["C"]), C.byref(ops["D"]),
        C.byref(ops["e"]), C.byref(ops["x0"]), C.byref(ops["goal"]), C.byref(ops["targets"]),
        C.c_int64(batch), C.c_int(max_iter), C.c_double(tol), _dp(U), _dp(lam), _dp(status), _dp(iters))
    return U, lam[:, :m], status, iters


def solve_mpc_like_reference(problem, max_iter: int = 10000, tol: float = 1e-12):
    """The reference's execution model for ONE problem (solve_mpc.py:42-44):
    NumPy condensing in a Python loop (oracle.condense_np, the restatement of
    mpc_qp.py) followed by a native dense active-set solve (what qpsolvers'
    quadprog backend does). Returns (U [N,nu] or None, status, iters)."""
    from .condense_np import condense

    cq = condense(problem)
    x, _lam, st, it = gi_solve(cq.P, cq.q, cq.G, cq.h, max_iter, tol)
    if st != 0:
        return None, st, it
    return x.reshape(problem.nb_timesteps, problem.input_dim), st, it


def _strides(a, block_ndim: int, block: int, N: int):
    """(batch_stride, step_stride) of an operand shaped [..], [N,..] or [B,N|1,..]."""
    extra = a.ndim - block_ndim
    if extra == 0:
        return (0, 0)
    if extra == 1:
        return (0, block if a.shape[0] > 1 else 0)
    steps = a.shape[1]
    return (block * steps if a.shape[0] > 1 else 0, block if steps > 1 else 0)


def workload_flags(w: dict) -> int:
    f = 0
    if w["wt"] is not None:
        f |= FLAG_P_TERMINAL
        if w["wt"] > 1e-10 and w["goal"] is not None:
            f |= FLAG_Q_TERMINAL
    if w["wx"] is not None:
        f |= FLAG_P_STAGE
        if w["wx"] > 1e-10 and w["targets"] is not None:
            f |= FLAG_Q_STAGE
    return f


def solve_workload(w: dict, max_iter: int = 10000, tol: float = 1e-12, count: Optional[int] = None):
    """Run the all-C oracle on a workload dict (qpmpc_amd.workloads layout: A, B,
    C, D, e, N, wt, wx, wu, x0, goal, targets). ``count`` limits it to the first
    problems of the batch. Returns (U, lam, status, iters)."""
    A = np.ascontiguousarray(w["A"], dtype=float)
    Bm = np.ascontiguousarray(w["B"], dtype=float)
    e = np.ascontiguousarray(w["e"], dtype=float)
    nx, nu, N, mk = A.shape[-1], Bm.shape[-1], int(w["N"]), e.shape[-1]
    x0 = np.ascontiguousarray(w["x0"], dtype=float)
    if count is not None:
        x0 = x0[:count]
    layout = dict(A=_strides(A, 2, nx * nx, N), B=_strides(Bm, 2, nx * nu, N), e=_strides(e, 1, mk, N))
    Cm = D = None
    if w["C"] is not None:
        Cm = np.ascontiguousarray(w["C"], dtype=float)
        layout["C"] = _strides(Cm, 2, mk * nx, N)
    if w["D"] is not None:
        D = np.ascontiguousarray(w["D"], dtype=float)
        layout["D"] = _strides(D, 2, mk * nu, N)
    goal = None if w["goal"] is None else np.ascontiguousarray(w["goal"], dtype=float)
    tgt = None if w["targets"] is None else np.ascontiguousarray(w["targets"], dtype=float)
    return build_solve_batch(nx, nu, N, mk, workload_flags(w), w["wt"] or 0.0, w["wx"] or 0.0, w["wu"],
                             A, Bm, Cm, D, e, x0, goal, tgt, max_iter=max_iter, tol=tol, layout=layout)
