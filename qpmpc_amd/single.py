"""One problem per call, without the batch containers.

``solve_mpc(problem, solver="hip_gi")`` is the reference's own calling pattern
(examples/*.py: one ``solve_mpc`` per control period, solve_mpc.py:16-44). Going through
``BatchMPCProblem`` costs a dozen small host-to-device copies and four synchronising reads
per call; here every operand of the problem is packed into ONE pinned host buffer, uploaded
with one copy into a cached device buffer, solved (``mpcqp_build_solve_batch``) and rolled
out (``mpcqp_rollout_batch``) on the same stream, and ``U``, the multipliers, the states,
``status`` and ``iters`` come back in ONE copy. LTI fields (arrays rather than per-step
lists) are uploaded once with a zero step stride, like the reference stores them.
Problems with ragged per-step row counts take the general path.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, Optional, Tuple

import numpy as np

from . import _capi

_RUNNERS: Dict[Tuple, "_Runner"] = {}


def _field(value, N: int, shape) -> Tuple[Optional[np.ndarray], int]:
    """(stacked array [steps, *shape] or None, steps) of an LTI-array-or-per-step-list field."""
    if value is None:
        return None, 0
    if isinstance(value, list):
        if len(value) != N or any(v is None for v in value):
            raise ValueError("ragged")
        return np.stack([np.asarray(v, dtype=np.float64).reshape(shape) for v in value]), N
    return np.asarray(value, dtype=np.float64).reshape((1,) + tuple(shape)), 1


class _Runner:
    """Buffers and C structs for one problem layout (dimensions + which fields are per-step)."""

    def __init__(self, nx, nu, N, mk, steps, has_goal, has_targets):
        import torch

        self.torch = torch
        self.lib = _capi.load()
        self.device = _capi.require_gpu()
        self.nx, self.nu, self.N, self.mk = nx, nu, N, mk
        n, m = N * nu, N * mk
        self.n, self.m = n, m
        sizes = [steps[0] * nx * nx, steps[1] * nx * nu, steps[2] * mk * nx, steps[3] * mk * nu, steps[4] * mk,
                 nx, nx if has_goal else 0, N * nx if has_targets else 0]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        total = int(self.offsets[-1])
        self.h_in = torch.empty((total,), dtype=torch.float64).pin_memory()
        self.h_in_np = self.h_in.numpy()
        self.d_in = torch.empty((total,), dtype=torch.float64, device=self.device)
        # outputs: U [n], lam [m], X [(N+1) nx], then status, iters as int32 (one float64 slot)
        self.out_f = n + m + (N + 1) * nx
        self.d_out = torch.empty((self.out_f + 1,), dtype=torch.float64, device=self.device)
        self.h_out = torch.empty((self.out_f + 1,), dtype=torch.float64).pin_memory()
        self.h_out_np = self.h_out.numpy()
        self.h_out_i32 = self.h_out_np[self.out_f:].view(np.int32)
        blocks = [nx * nx, nx * nu, mk * nx, mk * nu, mk]

        def problem_at(base):
            def op(i, present=True):
                if not present or sizes[i] == 0:
                    return _capi.Operand(None, 0, 0)
                step = blocks[i] if (i < 5 and steps[i] > 1) else 0
                return _capi.Operand(base + 8 * int(self.offsets[i]), 0, step)

            return _capi.Problem(op(0), op(1), op(2), op(3), op(4), op(5), op(6, has_goal), op(7, has_targets))

        self.cp = problem_at(self.d_in.data_ptr())
        # Small problems skip the two copy commands: ROCm maps pinned host memory into the device's address
        # space at the same address, so the kernels read the operands from, and write the results to, the
        # pinned buffers directly (a few KB over PCIe either way); one call = launches + one synchronisation.
        self.zero_copy = 8 * (total + self.out_f + 1) <= 32 * 1024
        self.cp_host = problem_at(self.h_in.data_ptr())
        ob = self.h_out.data_ptr() if self.zero_copy else self.d_out.data_ptr()
        self.pU, self.pLam, self.pX = ob, ob + 8 * n, ob + 8 * (n + m)
        self.pStatus, self.pIters = ob + 8 * self.out_f, ob + 8 * self.out_f + 4
        self.ws = None
        self.ws_key = None
        self.c_host = self.c_ws = None  # condense_single's staging, created on first use

    def workspace(self, dims):
        key = (dims.flags,)
        if self.ws_key != key:
            nbytes = C.c_size_t(0)
            _capi.check(self.lib.mpcqp_workspace_bytes(C.byref(dims), 1, 1, C.byref(nbytes)), "mpcqp_workspace_bytes")
            self.ws = self.torch.empty((nbytes.value,), dtype=self.torch.uint8, device=self.device) if nbytes.value else None
            self.ws_key = key
        return (None, 0) if self.ws is None else (self.ws.data_ptr(), self.ws.numel())


def _pack(problem):
    """Runner for the problem's layout with the operands written into its pinned buffer, plus the
    dims struct; ``None`` when the problem needs the general path (ragged per-step row counts)."""
    N, nx, nu = problem.nb_timesteps, problem.state_dim, problem.input_dim
    try:
        e, se = _field(problem.ineq_vector, N, (-1,))
        if e is None:
            return None
        mk = e.shape[-1]
        A, sa = _field(problem.transition_state_matrix, N, (nx, nx))
        B, sb = _field(problem.transition_input_matrix, N, (nx, nu))
        Cm, sc = _field(problem.ineq_state_matrix, N, (mk, nx))
        Dm, sd = _field(problem.ineq_input_matrix, N, (mk, nu))
    except ValueError:  # ragged rows or None inside a list
        return None
    goal, targets = problem.goal_state, problem.target_states
    # staging buffers are per layout, per device AND per host thread (two threads solving the same layout
    # must not share pinned / device buffers; after torch.cuda.set_device the buffers must follow)
    import torch as _t

    key = (nx, nu, N, mk, sa, sb, sc, sd, se, goal is not None, targets is not None,
           _t.cuda.current_device() if _t.cuda.is_available() else -1, threading.get_ident())
    r = _RUNNERS.get(key)
    if r is None:
        r = _RUNNERS[key] = _Runner(nx, nu, N, mk, (sa, sb, sc, sd, se), goal is not None, targets is not None)
    o, buf = r.offsets, r.h_in_np
    for i, arr in enumerate((A, B, Cm, Dm, e)):
        if arr is not None:
            buf[o[i]:o[i + 1]] = arr.ravel()
    buf[o[5]:o[6]] = np.asarray(problem.initial_state, dtype=np.float64).ravel()
    if goal is not None:
        buf[o[6]:o[7]] = np.asarray(goal, dtype=np.float64).ravel()
    if targets is not None:
        buf[o[7]:o[8]] = np.asarray(targets, dtype=np.float64).ravel()
    wt, wx = problem.terminal_cost_weight, problem.stage_state_cost_weight
    flags = (_capi.P_TERMINAL if wt is not None else 0) | (_capi.P_STAGE if wx is not None else 0)
    t_on, s_on = wt is not None and wt > 1e-10, wx is not None and wx > 1e-10
    if not (t_on and goal is None):  # the reference raises before accumulating anything (mpc_qp.py:119-122)
        if t_on:
            flags |= _capi.Q_TERMINAL
        if s_on and targets is not None:
            flags |= _capi.Q_STAGE
    dims = _capi.Dims(nx, nu, N, mk, _capi.F64, flags, 0.0 if wt is None else float(wt), 0.0 if wx is None else float(wx),
                      float(problem.stage_input_cost_weight))
    return r, dims


def condense_single(problem):
    """(Phi_all [(N+1) nx, nx], Psi_all [(N+1) nx, n], P, G, q, h) of one host ``MPCProblem`` as NumPy
    arrays: one upload, ``mpcqp_condense_batch`` with a batch of one, one download (``MPCQP.__init__``,
    qpmpc/mpc_qp.py:39-122); ``None`` when the problem needs the general path."""
    packed = _pack(problem)
    if packed is None:
        return None
    r, dims = packed
    torch, lib = r.torch, r.lib
    nx, N, n, m = r.nx, r.N, r.n, r.m
    sizes = [(N + 1) * nx * nx, (N + 1) * nx * n, n * n, m * n, n, m]
    total = sum(sizes)
    if r.c_host is None:
        nbytes = C.c_size_t(0)
        rc = lib.mpcqp_workspace_bytes(C.byref(dims), 1, 0, C.byref(nbytes))
        if rc != 0:
            return None  # too large for the condensing kernels: the general path reports it
        r.c_ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=r.device) if nbytes.value else None
        r.c_host = torch.empty((total,), dtype=torch.float64).pin_memory()
    # the outputs and a copy of the operands belong to the caller (MPCQP keeps Phi/Psi/C/e on the device for
    # update_cost_vector / update_constraint_vector); the staging buffers are the runner's
    c_out = torch.empty((total,), dtype=torch.float64, device=r.device)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    base = c_out.data_ptr()
    ptr = [base + 8 * int(o) for o in offs[:-1]]
    stream, raw_stream = _capi.current_stream()
    r.d_in.copy_(r.h_in, non_blocking=True)
    rc = lib.mpcqp_condense_batch(C.byref(dims), C.byref(r.cp), 1, ptr[2], ptr[4], ptr[3] if m else None,
                                  ptr[5] if m else None, ptr[0], ptr[1],
                                  None if r.c_ws is None else r.c_ws.data_ptr(), 0 if r.c_ws is None else r.c_ws.numel(),
                                  C.c_void_p(raw_stream))
    _capi.check(rc, "mpcqp_condense_batch")
    r.c_host.copy_(c_out, non_blocking=True)
    own_in = r.d_in.clone()
    stream.synchronize()
    flat = r.c_host.numpy().copy()
    shapes = [((N + 1) * nx, nx), ((N + 1) * nx, n), (n, n), (m, n), (n,), (m,)]
    arrays = tuple(flat[offs[i]:offs[i + 1]].reshape(shapes[i]) for i in range(6))
    # operands of this problem inside its own copy of the input buffer
    shift = own_in.data_ptr() - r.d_in.data_ptr()
    cp = _capi.Problem()
    for name in ("A", "B", "C", "D", "e", "x0", "goal", "targets"):
        op = getattr(r.cp, name)
        setattr(cp, name, _capi.Operand(op.ptr + shift if op.ptr else None, op.batch_stride, op.step_stride))
    keep = {"out": c_out, "in": own_in, "cp": cp, "dims": dims, "Phi": ptr[0], "Psi": ptr[1],
            "nx": nx, "N": N, "n": n, "m": m, "device": r.device}
    return arrays, keep


def solve_single(problem, max_iter=None, feas_tol=None):
    """(x [n] | None, z [m] | None, X [N+1, nx] | None, status, iters) for one host ``MPCProblem``, or
    ``None`` when the problem needs the general path (ragged per-step row counts)."""
    packed = _pack(problem)
    if packed is None:
        return None
    r, dims = packed
    torch, lib = r.torch, r.lib
    N, nx = r.N, r.nx
    opts = _capi.SolveOpts(int(max_iter or 0), 0, float(feas_tol or 0.0))  # remaining fields: NULL / 0
    stream, raw_stream = _capi.current_stream()
    sp = C.c_void_p(raw_stream)
    cp = r.cp_host if r.zero_copy else r.cp
    if not r.zero_copy:
        r.d_in.copy_(r.h_in, non_blocking=True)
    ws_ptr, ws_len = r.workspace(dims)
    rc = lib.mpcqp_build_solve_batch(C.byref(dims), C.byref(cp), 1, C.byref(opts), r.pU, r.pLam, r.pStatus, r.pIters,
                                     ws_ptr, ws_len, sp)
    _capi.check(rc, "mpcqp_build_solve_batch")
    rc = lib.mpcqp_rollout_batch(C.byref(dims), C.byref(cp.A), C.byref(cp.B), C.byref(cp.x0), r.pU, 1, r.pX, sp)
    _capi.check(rc, "mpcqp_rollout_batch")
    if not r.zero_copy:
        r.h_out.copy_(r.d_out, non_blocking=True)
    stream.synchronize()
    out = r.h_out_np
    status, iters = int(r.h_out_i32[0]), int(r.h_out_i32[1])
    if status in (_capi.SLOTS_FULL, _capi.MAX_ITER) or (status == _capi.INFEASIBLE and r.n > 16):
        # the stage-wise kernel's slots were too few, the kernel gave up on a degenerate problem, or a stage-wise kernel says
        # "infeasible" (re-checked by another formulation before it stands: batch._retry_unsolved): the batch path solves
        # again (more slots / the other formulations of the solver)
        return None
    if status != 0:
        return None, None, None, status, iters
    n, m = r.n, r.m
    return out[:n].copy(), out[n:n + m].copy(), out[n + m:r.out_f].reshape(N + 1, nx).copy(), status, iters
