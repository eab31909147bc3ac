"""Batched problems on the device: the data-parallel entry points.

A batch is ``B`` independent MPC problems with common dimensions. Operands live
in HBM as packed row-major tensors; an operand that is the same for every
problem (or every step) is stored ONCE and addressed with a zero stride, which
is how the reference's "array, not list" (LTI) case is expressed
(qpmpc/mpc_problem.py:177-245).

    A        [B|1, N|1, nx, nx]      B  [B|1, N|1, nx, nu]
    C        [B|1, N|1, mk, nx]|None D  [B|1, N|1, mk, nu]|None
    e        [B|1, N|1, mk]
    x0       [B, nx]   goal [B|1, nx]|None   targets [B|1, N*nx]|None

``solve_mpc_batch`` replaces a Python loop of ``solve_mpc`` calls
(qpmpc/solve_mpc.py:42-44) by one fused kernel launch; one problem per
workgroup, no host round trip.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _capi
from .exceptions import BackendError, ProblemDefinitionError, StateError

PAD_BOUND = 1e30  # bound of padded inequality rows (0 . u <= 1e30 is never active)
HIP_SOLVERS = ("hip", "hip_gi", "mpcqp_hip")


def _torch():
    import torch

    return torch


def _dtype_code(dtype) -> int:
    torch = _torch()
    if dtype == torch.float64:
        return _capi.F64
    if dtype == torch.float32:
        return _capi.F32
    raise ProblemDefinitionError(f"dtype must be torch.float64 or torch.float32, not {dtype}")


def _as_tensor(x, dtype, device):
    torch = _torch()
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype)
    return torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=dtype, device=device)


def _canon(x, tail: Sequence[int], name: str):
    """Bring an operand to [Bx, Nx, *tail] with Bx, Nx possibly 1; contiguous."""
    if x is None:
        return None
    t = len(tail)
    if x.dim() == t:
        x = x.reshape(1, 1, *x.shape)
    elif x.dim() == t + 1:
        x = x.reshape(1, *x.shape)
    elif x.dim() != t + 2:
        raise ProblemDefinitionError(f"{name}: expected {t}, {t + 1} or {t + 2} dimensions, got {x.dim()}")
    if tuple(x.shape[2:]) != tuple(tail):
        raise ProblemDefinitionError(f"{name}: trailing shape {tuple(x.shape[2:])} != {tuple(tail)}")
    return x.contiguous()


class BatchMPCProblem:
    """``B`` linear time-variant MPC problems resident on one GPU.

    Same cost/constraint model and the same validation as
    :class:`qpmpc_amd.MPCProblem` (reference mpc_problem.py:88-139): weights are
    shared by the batch; states may differ per problem.
    """

    def __init__(
        self,
        transition_state_matrix,
        transition_input_matrix,
        ineq_state_matrix,
        ineq_input_matrix,
        ineq_vector,
        nb_timesteps: int,
        terminal_cost_weight: Optional[float],
        stage_state_cost_weight: Optional[float],
        stage_input_cost_weight: float,
        initial_state,
        goal_state=None,
        target_states=None,
        dtype=None,
        device=None,
    ) -> None:
        torch = _torch()
        if stage_input_cost_weight <= 0.0:
            raise ProblemDefinitionError("stage non-negative control weight needed for regularization")
        if terminal_cost_weight is None and stage_state_cost_weight is None:
            raise ProblemDefinitionError("either terminal or stage state cost should be set")
        self.dtype = dtype if dtype is not None else torch.float64
        self.device = device if device is not None else _capi.require_gpu()
        _dtype_code(self.dtype)
        N = int(nb_timesteps)
        A = _as_tensor(transition_state_matrix, self.dtype, self.device)
        Bm = _as_tensor(transition_input_matrix, self.dtype, self.device)
        nx, nu = int(A.shape[-1]), int(Bm.shape[-1])
        self.state_dim, self.input_dim, self.nb_timesteps = nx, nu, N
        self.A = _canon(A, (nx, nx), "transition_state_matrix")
        self.B = _canon(Bm, (nx, nu), "transition_input_matrix")
        e = _as_tensor(ineq_vector, self.dtype, self.device)
        mk = int(e.shape[-1])
        self.ineq_dim = mk
        self.e = _canon(e, (mk,), "ineq_vector")
        self.C = _canon(_as_tensor(ineq_state_matrix, self.dtype, self.device), (mk, nx), "ineq_state_matrix")
        self.D = _canon(_as_tensor(ineq_input_matrix, self.dtype, self.device), (mk, nu), "ineq_input_matrix")
        for name, op in (("A", self.A), ("B", self.B), ("C", self.C), ("D", self.D), ("e", self.e)):
            if op is not None and op.shape[1] not in (1, N):
                raise ProblemDefinitionError(f"{name}: step dimension {op.shape[1]} is neither 1 nor N={N}")
        self.terminal_cost_weight = terminal_cost_weight
        self.stage_state_cost_weight = stage_state_cost_weight
        self.stage_input_cost_weight = float(stage_input_cost_weight)
        self.initial_state = None
        self.goal_state = None
        self.target_states = None
        if initial_state is None:
            raise ProblemDefinitionError("initial state is undefined")
        self.update_initial_state(initial_state)
        if goal_state is not None:
            self.update_goal_state(goal_state)
        if target_states is not None:
            self.update_target_states(target_states)
        B_ = self.batch_size
        for name, op in (("A", self.A), ("B", self.B), ("C", self.C), ("D", self.D), ("e", self.e)):
            if op is not None and op.shape[0] not in (1, B_):
                raise ProblemDefinitionError(f"{name}: batch dimension {op.shape[0]} is neither 1 nor B={B_}")
        # row mask for problems whose per-step row count was padded (from_problems)
        self.valid_rows: Optional[np.ndarray] = None

    # ------------------------------------------------------------------ state
    @property
    def batch_size(self) -> int:
        return int(self.initial_state.shape[0])

    @property
    def nb_variables(self) -> int:
        return self.nb_timesteps * self.input_dim

    @property
    def nb_constraints(self) -> int:
        return self.nb_timesteps * self.ineq_dim

    def _state(self, x, width: int, what: str, per_batch_required: bool):
        x = _as_tensor(x, self.dtype, self.device)
        if x.dim() == 1 or (not per_batch_required and x.numel() == width):
            x = x.reshape(1, -1)
        x = x.reshape(x.shape[0], -1)
        if x.shape[1] != width:
            raise StateError(f"{what} of shape {tuple(x.shape)} does not match dimension ({width})")
        return x.contiguous()

    def update_initial_state(self, initial_state) -> None:
        """x0 per problem, [B, nx] (a 1-D vector makes a batch of one)."""
        x = _as_tensor(initial_state, self.dtype, self.device)
        if x.dim() == 1:
            x = x.reshape(1, -1)
        x = x.reshape(x.shape[0], -1)
        if x.shape[1] != self.state_dim:
            raise StateError(
                f"Initial state of shape {tuple(x.shape)} does not match state dimension ({self.state_dim})"
            )
        if self.initial_state is not None and x.shape[0] != self.batch_size:
            raise StateError(f"batch size changed from {self.batch_size} to {x.shape[0]}")
        self.initial_state = x.contiguous()

    def update_goal_state(self, goal_state) -> None:
        """Goal per problem [B, nx] or shared [nx]."""
        g = self._state(goal_state, self.state_dim, "goal state", False)
        if g.shape[0] not in (1, self.batch_size):
            raise StateError(f"goal state batch {g.shape[0]} is neither 1 nor {self.batch_size}")
        self.goal_state = g

    def update_target_states(self, target_states) -> None:
        """Stage targets per problem [B, N*nx] (or [B, N, nx]) or shared [N*nx]."""
        width = self.state_dim * self.nb_timesteps
        t = _as_tensor(target_states, self.dtype, self.device)
        if t.numel() == width:
            t = t.reshape(1, width)
        else:
            t = t.reshape(t.shape[0], -1)
        if t.shape[1] != width or t.shape[0] not in (1, self.batch_size):
            raise StateError(
                f"Reference state trajectory of shape {tuple(t.shape)} does not match "
                f"nb_timesteps * state dimension = {self.nb_timesteps} * {self.state_dim} = {width}"
            )
        self.target_states = t.contiguous()

    def select(self, index) -> "BatchMPCProblem":
        """The sub-batch of the problems ``index`` (a 1-D int64 device tensor): per-problem operands are gathered,
        shared ones (batch dimension 1) stay shared."""
        sub = BatchMPCProblem.__new__(BatchMPCProblem)
        sub.__dict__.update(self.__dict__)
        for name in ("A", "B", "C", "D", "e", "initial_state", "goal_state", "target_states"):
            t = getattr(self, name)
            if t is not None and t.shape[0] != 1:
                setattr(sub, name, t.index_select(0, index).contiguous())
        if self.initial_state.shape[0] == 1:  # (a batch of one keeps its only problem)
            sub.initial_state = self.initial_state
        return sub

    # ------------------------------------------------------------ C ABI views
    def cost_flags(self) -> int:
        """MPCQP_P_* / MPCQP_Q_* bits (see include/mpcqp.h for why they differ)."""
        f = 0
        wt, wx = self.terminal_cost_weight, self.stage_state_cost_weight
        if wt is not None:
            f |= _capi.P_TERMINAL
        if wx is not None:
            f |= _capi.P_STAGE
        t_on = wt is not None and wt > 1e-10
        s_on = wx is not None and wx > 1e-10
        if t_on and self.goal_state is None:
            return f  # the reference raises before accumulating anything (mpc_qp.py:119-122)
        if t_on:
            f |= _capi.Q_TERMINAL
        if s_on and self.target_states is not None:
            f |= _capi.Q_STAGE
        return f

    def dims(self) -> _capi.Dims:
        return _capi.Dims(
            self.state_dim, self.input_dim, self.nb_timesteps, self.ineq_dim, _dtype_code(self.dtype),
            self.cost_flags(),
            0.0 if self.terminal_cost_weight is None else float(self.terminal_cost_weight),
            0.0 if self.stage_state_cost_weight is None else float(self.stage_state_cost_weight),
            self.stage_input_cost_weight,
        )

    @staticmethod
    def _operand(t) -> _capi.Operand:
        if t is None:
            return _capi.Operand(None, 0, 0)
        block = int(np.prod(t.shape[2:])) if t.dim() > 2 else 0
        steps = t.shape[1] if t.dim() > 2 else 1
        bs = 0 if t.shape[0] == 1 else int(t.stride(0))
        ks = 0 if (t.dim() <= 2 or steps == 1) else block
        return _capi.Operand(t.data_ptr(), bs, ks)

    def c_problem(self) -> _capi.Problem:
        return _capi.Problem(
            self._operand(self.A), self._operand(self.B), self._operand(self.C), self._operand(self.D),
            self._operand(self.e), self._operand(self.initial_state), self._operand(self.goal_state),
            self._operand(self.target_states),
        )

    # ------------------------------------------------------------- factories
    @classmethod
    def from_problems(cls, problems: List, dtype=None, device=None) -> "BatchMPCProblem":
        """Stack host ``MPCProblem`` objects of equal dimensions into one batch.

        Per-step row counts may vary (lists of different-height C_k/D_k/e_k);
        they are padded to the maximum with zero rows and ``PAD_BOUND`` bounds and
        ``valid_rows`` remembers which rows are real.
        """
        p0 = problems[0]
        N, nx, nu = p0.nb_timesteps, p0.state_dim, p0.input_dim
        if p0.initial_state is None:
            raise ProblemDefinitionError("initial state is undefined")
        mks = [len(np.asarray(p0.get_ineq_vector(k)).ravel()) for k in range(N)]
        mk = max(mks)
        Bn = len(problems)
        A = np.zeros((Bn, N, nx, nx))
        Bm = np.zeros((Bn, N, nx, nu))
        e = np.full((Bn, N, mk), PAD_BOUND)
        anyC = any(p.get_ineq_state_matrix(k) is not None for p in problems for k in range(N))
        anyD = any(p.get_ineq_input_matrix(k) is not None for p in problems for k in range(N))
        Cm = np.zeros((Bn, N, mk, nx)) if anyC else None
        Dm = np.zeros((Bn, N, mk, nu)) if anyD else None
        weights = (p0.terminal_cost_weight, p0.stage_state_cost_weight, p0.stage_input_cost_weight)
        for b, p in enumerate(problems):
            if (p.nb_timesteps, p.state_dim, p.input_dim) != (N, nx, nu):
                raise ProblemDefinitionError("problems of a batch must share (N, nx, nu)")
            if (p.terminal_cost_weight, p.stage_state_cost_weight, p.stage_input_cost_weight) != weights:
                raise ProblemDefinitionError(
                    "problems of a batch must share the three cost weights "
                    f"(problem 0: {weights}, problem {b}: "
                    f"{(p.terminal_cost_weight, p.stage_state_cost_weight, p.stage_input_cost_weight)}); "
                    "group the problems by weights and solve one batch per group")
            if p.initial_state is None:
                raise ProblemDefinitionError("initial state is undefined")
            for k in range(N):
                A[b, k] = np.asarray(p.get_transition_state_matrix(k), dtype=float).reshape(nx, nx)
                Bm[b, k] = np.asarray(p.get_transition_input_matrix(k), dtype=float).reshape(nx, nu)
                ek = np.asarray(p.get_ineq_vector(k), dtype=float).ravel()
                if len(ek) != mks[k]:
                    raise ProblemDefinitionError("problems of a batch must share per-step row counts")
                e[b, k, : mks[k]] = ek
                Ck, Dk = p.get_ineq_state_matrix(k), p.get_ineq_input_matrix(k)
                if Ck is not None:
                    Cm[b, k, : mks[k]] = np.asarray(Ck, dtype=float).reshape(mks[k], nx)
                if Dk is not None:
                    Dm[b, k, : mks[k]] = np.asarray(Dk, dtype=float).reshape(mks[k], nu)
        x0 = np.stack([np.asarray(p.initial_state, dtype=float) for p in problems])
        goals = [p.goal_state for p in problems]
        tgts = [p.target_states for p in problems]
        bp = cls(
            A, Bm, Cm, Dm, e, N, p0.terminal_cost_weight, p0.stage_state_cost_weight,
            p0.stage_input_cost_weight, x0,
            goal_state=None if any(g is None for g in goals) else np.stack(goals),
            target_states=None if any(t is None for t in tgts) else np.stack(tgts),
            dtype=dtype, device=device,
        )
        bp.valid_rows = np.concatenate([k * mk + np.arange(mks[k]) for k in range(N)])
        return bp


def _stream_ptr():
    """hipStream_t of torch's current stream; raises BackendError without a GPU
    (every launch goes through here, so nothing can silently run elsewhere)."""
    return C.c_void_p(_capi.current_stream()[1])


def _require_on_gpu(*tensors) -> None:
    for t in tensors:
        if t is not None and t.device.type != "cuda":
            raise BackendError(f"operand on {t.device}: the HIP path needs device-resident tensors (no CPU fallback)")


def _workspace(problem: "BatchMPCProblem", for_solve: bool):
    """Caller-owned scratch for problems that do not fit the on-chip path (None when
    they do): a uint8 tensor of exactly ``mpcqp_workspace_bytes`` bytes."""
    torch = _torch()
    lib = _capi.load()
    dims = problem.dims()
    nbytes = C.c_size_t(0)
    rc = lib.mpcqp_workspace_bytes(C.byref(dims), problem.batch_size, 1 if for_solve else 0, C.byref(nbytes))
    _capi.check(rc, "mpcqp_workspace_bytes")
    if nbytes.value == 0:
        return None
    return torch.empty((nbytes.value,), dtype=torch.uint8, device=problem.device)


def _ws_args(ws):
    return (None, 0) if ws is None else (ws.data_ptr(), ws.numel())


def _warm_mode(warm_start) -> int:
    """False / True / "operator" / "active_set" (or the MPCQP_WARM_* integers) -> MpcqpSolveOpts.warm_start."""
    if isinstance(warm_start, str):
        try:
            return {"operator": _capi.WARM_OPERATOR, "active_set": _capi.WARM_ACTIVE_SET}[warm_start]
        except KeyError:
            raise BackendError(f"warm_start must be False, True, 'operator' or 'active_set', not {warm_start!r}") from None
    return int(warm_start) if warm_start else 0


def _opts(max_iter=None, feas_tol=None, flags: int = 0, warm_state=None, warm_start=False, probe=None, warm_shift: int = 0,
          order=None):
    """``MpcqpSolveOpts``. ``warm_state``: a :class:`WarmState` (or a uint8 device tensor) that every solve
    updates and, with ``warm_start``, starts from -- ``True`` / ``"operator"``: the stored active set and operator
    (matrices unchanged); ``"active_set"``: the stored rows only, moved down by ``warm_shift`` rows (receding horizon:
    ``warm_shift = mk`` per step the horizon advanced); ``flags``: the explicit dispatch overrides
    ``_capi.OPT_*`` (tests); ``probe``: an int64 device tensor for the developer stamps; ``order``: an int32 device tensor,
    a permutation of the batch (``MpcqpSolveOpts.order``: which problems share a wavefront in the small-problem kernel --
    :func:`pairing_order` of last period's iteration counts)."""
    o = _capi.SolveOpts()
    o.max_iter, o.flags, o.feas_tol = int(max_iter or 0), int(flags), float(feas_tol or 0.0)
    if warm_state is not None:
        buf = warm_state.buffer if isinstance(warm_state, WarmState) else warm_state
        o.warm_state, o.warm_start, o.warm_shift = buf.data_ptr(), _warm_mode(warm_start), int(warm_shift)
        o.warm_state_bytes = int(buf.numel() * buf.element_size())  # the C side refuses a buffer smaller than the launch needs
    if probe is not None:
        o.probe = probe.data_ptr()
    if order is not None:
        torch = _torch()
        if order.dtype != torch.int32 or not order.is_cuda or not order.is_contiguous():
            raise BackendError("order must be a contiguous int32 tensor on the GPU (a permutation of the batch)")
        o.order = order.data_ptr()
    return o


def pairing_order(counts, out=None):
    """``mpcqp_order_by_count``: the batch sorted by ``counts`` (int32 device tensor -- last period's ``plan.iters``), longest
    first, as an int32 permutation for ``order=``. A device-side counting sort on torch's current stream, no synchronisation.
    The small-problem kernel runs two problems per wavefront for max(trips) of the two: pairing equals with equals shortens
    launches of several rounds by up to 12 % (BASELINE config 4's 65,536 problems); the plans do not depend on it."""
    torch = _torch()
    lib = _capi.load()
    _require_on_gpu(counts)
    if counts.dtype != torch.int32 or not counts.is_contiguous():
        raise BackendError("counts must be a contiguous int32 tensor")
    n = counts.numel()
    order = out if out is not None else torch.empty((n,), dtype=torch.int32, device=counts.device)
    ws = torch.empty((max(1, lib.mpcqp_order_workspace_bytes(n)),), dtype=torch.uint8, device=counts.device)
    _capi.check(lib.mpcqp_order_by_count(counts.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr()),
                "mpcqp_order_by_count")
    return order


class WarmState:
    """Per-problem active set and active-set operator kept in HBM between the periods of
    receding-horizon loops (``MpcqpSolveOpts.warm_state``). Valid while the problems' matrices and
    weights stay the same; states, goals, targets and bounds may change. A stale state only costs a
    cold restart (the kernel re-checks the KKT conditions of what it returns)."""

    def __init__(self, problem: "BatchMPCProblem"):
        torch = _torch()
        lib = _capi.load()
        dims = problem.dims()
        nbytes = C.c_size_t(0)
        _capi.check(lib.mpcqp_warm_state_bytes(C.byref(dims), C.byref(nbytes)), "mpcqp_warm_state_bytes")
        if nbytes.value == 0:
            raise BackendError(
                "warm start is available for problems with n = N*nu <= 16 variables and m <= 32 rows (the active-set "
                "operator persists; float32 launches of that size are solved in float64), for small systems (nx <= 4, nu <= 2) with "
                "16 < n <= 128 (the active rows' vectors "
                "persist in the solver's workspace) and for what the wide stage-wise kernel takes (nx <= 16, nu <= 4: the active "
                f"rows' ids); got n={problem.nb_variables}, m={problem.nb_constraints}, {problem.dtype}")
        self.bytes_per_problem = int(nbytes.value)
        self.batch_size = problem.batch_size
        self.device = problem.device
        # "pair": the operator T (16 x 16 doubles) first, then the slots' constraint ids (int32); "stage": int32 record
        # [nq, slots, workspace tag (2), row steps ..., row indices ...] -- the vectors themselves stay in the workspace of
        # the PreparedSolve that is re-launched (MPCQP_OPT_REUSE_FACTOR contract), so this kind only acts there
        # "stagew" (round 4): the wide stage-wise kernel -- int32 count, then the active rows' ids; started from with
        # warm_start="active_set" only (the rows' vectors ride along with the first sweeps)
        # (asked of the library: the dispatch is its business -- a float32 problem of at most 160 variables is solved by the
        # float64 kernels and keeps their record)
        kind = C.c_int32(0)
        _capi.check(lib.mpcqp_warm_state_kind(C.byref(dims), C.byref(kind)), "mpcqp_warm_state_kind")
        self.kind = {1: "pair", 2: "stage", 3: "stagew"}[kind.value]
        self._mk = problem.ineq_dim
        self._ids_at = 16 * 16 * 8 if self.kind == "pair" else 0
        self.buffer = torch.zeros((problem.batch_size, self.bytes_per_problem), dtype=torch.uint8, device=problem.device)
        if self.kind == "pair":
            self.buffer[:, self._ids_at:] = 0xFF  # constraint ids -1: empty set

    def check(self, problem: "BatchMPCProblem") -> None:
        """Raise ``BackendError`` unless this state was allocated for ``problem``'s batch, dimensions and device
        (the kernel indexes it by problem: a smaller buffer would be read and written out of bounds)."""
        lib = _capi.load()
        dims, nbytes = problem.dims(), C.c_size_t(0)
        _capi.check(lib.mpcqp_warm_state_bytes(C.byref(dims), C.byref(nbytes)), "mpcqp_warm_state_bytes")
        if nbytes.value != self.bytes_per_problem or problem.batch_size != self.batch_size:
            raise BackendError(
                f"WarmState holds {self.batch_size} problems of {self.bytes_per_problem} bytes; this launch needs "
                f"{problem.batch_size} of {nbytes.value}: allocate WarmState(problem) for the batch it is used with")
        if problem.device != self.device:
            raise BackendError(f"WarmState lives on {self.device}, the problem on {problem.device}")

    @property
    def active_set(self):
        """int32 [B, slots]: constraint row (k * mk + r) held by each slot after the last solve, -1 = empty."""
        torch = _torch()
        if self.kind == "pair":
            return self.buffer[:, self._ids_at:].contiguous().view(torch.int32)
        if self.kind == "stagew":
            rec = self.buffer.view(torch.int32)  # [B, 1 + slots (+ padding)]
            rows, nq = rec[:, 1:], rec[:, 0:1]
            live = torch.arange(rows.shape[1], device=rec.device)[None, :] < nq
            return torch.where(live, rows, torch.full_like(rows, -1))
        rec = self.buffer.view(torch.int32)  # [B, 4 + 2 slots (+ padding)]
        slots = int(rec[0, 1].item()) if int(rec[0, 1].item()) > 0 else (rec.shape[1] - 4) // 2
        nq = rec[:, 0:1]
        k, r = rec[:, 4:4 + slots], rec[:, 4 + slots:4 + 2 * slots]
        rows = k * self._mk + r
        live = torch.arange(slots, device=rec.device)[None, :] < nq
        return torch.where(live, rows, torch.full_like(rows, -1))


def _check_order(problem, opt_kw) -> None:
    """The kernel indexes by ``order[i]`` unchecked (MpcqpSolveOpts.order): at least the length is checked here."""
    order = opt_kw.get("order")
    if order is not None and order.numel() != problem.batch_size:
        raise BackendError(f"order holds {order.numel()} indices, the batch {problem.batch_size} problems")


def _check_warm(problem: "BatchMPCProblem", opt_kw) -> None:
    """Host-side check of ``warm_state=`` against the launch (batch, dimensions, device); raw uint8 tensors are
    checked for size and device (the C ABI checks the size again: ``MpcqpSolveOpts.warm_state_bytes``)."""
    ws = opt_kw.get("warm_state")
    if ws is None:
        return
    if isinstance(ws, WarmState):
        ws.check(problem)
        return
    lib = _capi.load()
    dims, nbytes = problem.dims(), C.c_size_t(0)
    _capi.check(lib.mpcqp_warm_state_bytes(C.byref(dims), C.byref(nbytes)), "mpcqp_warm_state_bytes")
    need = nbytes.value * problem.batch_size
    if ws.device != problem.device or ws.numel() * ws.element_size() < need or nbytes.value == 0:
        raise BackendError(f"warm_state tensor: {ws.numel() * ws.element_size()} bytes on {ws.device}; this launch needs "
                           f"{need} bytes on {problem.device} (mpcqp_warm_state_bytes per problem: {nbytes.value})")


class BatchPlan:
    """Solutions of a batch (the batched counterpart of ``Plan``, plan.py:18-109).

    ``inputs`` [B, N, nu]; rows of problems that were not solved are zero and
    flagged in ``found`` (status: 0 solved, 1 iteration limit, 2 infeasible,
    3 P not positive definite).
    """

    def __init__(self, problem: BatchMPCProblem, U, status, iters, multipliers=None):
        self.problem = problem
        self.U = U
        self.status = status
        self.iters = iters
        self.multipliers = multipliers
        self._states = None

    @property
    def inputs(self):
        p = self.problem
        return self.U.view(p.batch_size, p.nb_timesteps, p.input_dim)

    @property
    def found(self):
        return self.status == 0

    @property
    def first_input(self):
        return self.inputs[:, 0, :]

    @property
    def states(self):
        """[B, N+1, nx], rolled out on the device and memoised (plan.py:81-109)."""
        if self._states is None:
            self._states = rollout_batch(self.problem, self.U)
        return self._states


def solve_mpc_batch(problem: BatchMPCProblem, solver: str = "hip_gi", return_multipliers: bool = False,
                    max_iter: Optional[int] = None, feas_tol: Optional[float] = None, formulation: str = "condensed",
                    max_active: Optional[int] = None, retry_slots: bool = True, retry_unsolved: bool = False,
                    **opt_kw) -> BatchPlan:
    """Build and solve every problem of the batch in ONE fused launch
    (``mpcqp_build_solve_batch``; replaces solve_mpc.py:42-44 per problem). Any problem size is served: what does not
    fit one CU's LDS goes to the stage-wise kernels (systems with nx <= 16, nu <= 4, any horizon) or, for wider systems,
    to the dense HBM-resident path (n = N*nu <= 256) and the general stage-wise kernel (nx <= 32, nu <= 8, any horizon).

    ``formulation="stagewise"`` asks for the uncondensed solver explicitly (``mpcqp_stagewise_solve_batch``:
    Riccati-based dual active set, O(N) memory and O(N) work per iteration); ``max_active`` bounds the active rows
    it can hold (default min(n, m, 128)). A problem that needs more comes back ``MPCQP_SLOTS_FULL`` from the kernel and
    -- ``retry_slots`` -- is solved again with twice the slots until it fits (this reads the statuses: a
    synchronisation, only for problems with more than 128 variables and rows).

    ``retry_unsolved``: problems that come back ``MPCQP_MAX_ITER`` are solved once more through the OTHER formulations of the
    same solver on the GPU (the LDS workgroup kernel -- Householder-based, the sturdiest of them --, the condensed kernels where a
    wide system went to the general stage-wise kernel, then the stage-wise one):
    a handful of degenerate problems in 10^4 (hundreds of iterations, rows nearly conflicting) end the mid-size dense kernel's
    verification rounds unsolved while the others -- and the reference's backends -- solve them. It reads the statuses (a
    synchronisation), hence off by default here and on in ``solve_mpc``.

    ``opt_kw``: ``warm_state`` (a :class:`WarmState`, updated by every solve) with ``warm_start=True``
    to begin from it, ``flags`` (``_capi.OPT_*`` dispatch overrides for cross-checks), ``probe``."""
    if solver not in HIP_SOLVERS:
        raise BackendError(f"solver '{solver}' is not a batched backend; available: {HIP_SOLVERS}")
    torch = _torch()
    lib = _capi.load()
    _require_on_gpu(problem.initial_state)
    Bn, n, m = problem.batch_size, problem.nb_variables, problem.nb_constraints
    U = torch.empty((Bn, n), dtype=problem.dtype, device=problem.device)
    lam = torch.empty((Bn, m), dtype=problem.dtype, device=problem.device) if return_multipliers else None
    status = torch.empty((Bn,), dtype=torch.int32, device=problem.device)
    iters = torch.empty((Bn,), dtype=torch.int32, device=problem.device)
    _check_warm(problem, opt_kw)
    _check_order(problem, opt_kw)
    ws_ = opt_kw.get("warm_state")
    if isinstance(ws_, WarmState) and ws_.kind == "stage" and opt_kw.get("warm_start"):
        # the stage-wise kernel's warm start continues from the vectors its previous launch left in the SAME workspace;
        # this function allocates a fresh one per call, so the request could only ever be a silent cold start
        raise BackendError("a stage-kind WarmState warm-starts through PreparedSolve (one workspace kept across launches), "
                           "not through solve_mpc_batch(warm_start=True)")
    dims, cp, opts = problem.dims(), problem.c_problem(), _opts(max_iter, feas_tol, **opt_kw)
    if formulation == "stagewise":
        nbytes = C.c_size_t(0)
        # (MPCQP_OPT_STAGE_GENERAL: the general kernel's workspace is asked for with a negative slot count, include/mpcqp.h)
        general = bool((opt_kw.get("flags") or 0) & _capi.OPT_STAGE_GENERAL)
        ask = (-max(int(max_active or 0), 1)) if general else int(max_active or 0)
        _capi.check(lib.mpcqp_stagewise_workspace_bytes(C.byref(dims), Bn, ask, C.byref(nbytes)),
                    "mpcqp_stagewise_workspace_bytes")
        ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=problem.device)
        rc = lib.mpcqp_stagewise_solve_batch(
            C.byref(dims), C.byref(cp), Bn, C.byref(opts), int(max_active or 0), U.data_ptr(),
            None if lam is None else lam.data_ptr(), status.data_ptr(), iters.data_ptr(), ws.data_ptr(), ws.numel(),
            _stream_ptr())
        _capi.check(rc, "mpcqp_stagewise_solve_batch")
        plan = BatchPlan(problem, U, status, iters, lam)
        plan._workspace = (ws, opt_kw)
        if retry_slots:
            _retry_slots_full(plan, int(max_active or (256 if general else 128)), max_iter, feas_tol, opt_kw)
        return plan
    if formulation != "condensed":
        raise ProblemDefinitionError(f"formulation must be 'condensed' or 'stagewise', not {formulation!r}")
    ws = _workspace(problem, True)
    rc = lib.mpcqp_build_solve_batch(
        C.byref(dims), C.byref(cp), Bn, C.byref(opts), U.data_ptr(),
        None if lam is None else lam.data_ptr(), status.data_ptr(), iters.data_ptr(), *_ws_args(ws), _stream_ptr())
    _capi.check(rc, "mpcqp_build_solve_batch")
    plan = BatchPlan(problem, U, status, iters, lam)
    plan._workspace = (ws, opt_kw)  # keep the scratch (and the opts' tensors) alive until the stream has consumed them
    if retry_slots and min(n, m) > 256:
        # (the automatic dispatch sends what can have more than 128 active rows to the wide stage-wise kernel, which holds
        # min(n, m, 256) of them -- mpcqp_capi.hip)
        _retry_slots_full(plan, 256, max_iter, feas_tol, opt_kw)
    if retry_unsolved and not (opt_kw.get("flags") or 0):
        _retry_unsolved(plan, max_iter, feas_tol, opt_kw)
    return plan


def _retry_unsolved(plan: "BatchPlan", max_iter, feas_tol, opt_kw) -> None:
    """``MPCQP_MAX_ITER`` items of a default-dispatch solve, once more through the other formulations (see solve_mpc_batch)."""
    problem = plan.problem
    kw = {k: v for k, v in opt_kw.items() if k not in ("warm_state", "warm_start", "probe", "flags", "order")}
    # (FORCE_CONDENSED: wide systems take the general stage-wise kernel by default, the condensed kernels are their other formulation)
    # ... and an INFEASIBLE verdict of a stage-wise kernel is confirmed by another formulation before it stands. Since round 6 the wide
    # stage-wise kernel keeps a thin QR factorisation of the whitened active rows and needs none of this (tools/stress_tight.py at
    # STRESS_TIGHT 0.5 .. 0.05 without any re-solve: tests/test_gpu_stress.py); the NARROW one (nx <= 4, nu <= 2, 16 < n <= 128) still
    # keeps the explicit inverse of the active rows' Gram matrix, and when every variable is pinned its error can turn the sign of a
    # step in the multipliers: one wrong verdict in 384 rounds of that family at STRESS_TIGHT = 0.05. Problems of the small-problem
    # kernels' size (n <= 16) never go there.
    recheck = problem.nb_variables > 16
    # (second, behind the workgroup kernel, which is quick where it applies: the general stage-wise kernel -- thin QR of the whitened
    # active rows, the formulation that stays accurate when nearly every variable is pinned; float64, nx <= 32, nu <= 8. It is the one
    # that settles the tight families of tools/stress_tight.py, so it goes before the dense path and the other stage-wise kernels.)
    for attempt in ({"flags": _capi.OPT_FORCE_LDS}, {"formulation": "stagewise", "flags": _capi.OPT_STAGE_GENERAL},
                    {"flags": _capi.OPT_FORCE_CONDENSED}, {"formulation": "stagewise"}):
        left = plan.status == _capi.MAX_ITER
        if recheck:
            left = left | (plan.status == _capi.INFEASIBLE)
        if not bool(left.any().item()):
            return
        index = left.nonzero().flatten()
        try:
            again = solve_mpc_batch(problem.select(index), return_multipliers=plan.multipliers is not None, max_iter=max_iter,
                                    feas_tol=feas_tol, retry_unsolved=False, **attempt, **kw)
        except BackendError:  # (this formulation does not serve these dimensions)
            continue
        better = again.status == _capi.SOLVED
        if not bool(better.any().item()):
            continue
        sub = better.nonzero().flatten()
        dst = index.index_select(0, sub)
        plan.U.index_copy_(0, dst, again.U.index_select(0, sub))
        plan.status.index_copy_(0, dst, again.status.index_select(0, sub))
        plan.iters.index_copy_(0, dst, again.iters.index_select(0, sub))
        if plan.multipliers is not None:
            plan.multipliers.index_copy_(0, dst, again.multipliers.index_select(0, sub))
        plan._states = None


def _retry_slots_full(plan: "BatchPlan", held: int, max_iter, feas_tol, opt_kw) -> None:
    """Problems that came back ``MPCQP_SLOTS_FULL`` -- more rows active at once than the launch's ``held`` slots -- are
    solved again through the stage-wise entry point with twice the slots, until they fit (at most min(n, m) rows can
    be active) -- the reference's backends have no such cap (solve_mpc.py:42-44). Reading the statuses synchronises;
    this is only reached for problems with more than 128 variables and rows, where the cap exists."""
    torch = _torch()
    problem = plan.problem
    cap = min(problem.nb_variables, problem.nb_constraints)
    kw = {k: v for k, v in opt_kw.items() if k not in ("warm_state", "warm_start", "probe")}
    while held < cap:
        full = plan.status == _capi.SLOTS_FULL
        if not bool(full.any().item()):
            return
        index = full.nonzero().flatten()
        held = min(2 * held, cap)
        try:
            again = solve_mpc_batch(problem.select(index), return_multipliers=plan.multipliers is not None, max_iter=max_iter,
                                    feas_tol=feas_tol, formulation="stagewise", max_active=held, retry_slots=False, **kw)
        except BackendError:  # (more slots than the kernel of these dimensions holds -- the general one stops at 1024: the items keep
            return            # their MPCQP_SLOTS_FULL status, an empty plan, instead of an exception out of a batch call)
        plan.U.index_copy_(0, index, again.U)
        plan.status.index_copy_(0, index, again.status)
        plan.iters.index_copy_(0, index, again.iters)
        if plan.multipliers is not None:
            plan.multipliers.index_copy_(0, index, again.multipliers)
        plan._states = None


class PreparedSolve:
    """A fused build+solve bound to fixed device buffers: ``launch()`` costs one
    C call (no allocation, no struct marshalling), so a receding-horizon loop or a
    benchmark can enqueue steps back-to-back or capture them in a HIP graph.

    Update problem data IN PLACE (``problem.initial_state.copy_(x)``) between
    launches; if a tensor of the problem is replaced, call ``rebind()``.
    """

    def __init__(self, problem: BatchMPCProblem, return_multipliers: bool = False,
                 max_iter: Optional[int] = None, feas_tol: Optional[float] = None, formulation: str = "condensed",
                 max_active: Optional[int] = None, **opt_kw):
        torch = _torch()
        self._lib = _capi.load()
        _require_on_gpu(problem.initial_state)
        if formulation not in ("condensed", "stagewise"):
            raise ProblemDefinitionError(f"formulation must be 'condensed' or 'stagewise', not {formulation!r}")
        self.problem = problem
        self._stagewise, self._max_active = formulation == "stagewise", int(max_active or 0)
        self._opt_kw = opt_kw  # tensors referenced by the opts struct stay alive with the object
        Bn, n, m = problem.batch_size, problem.nb_variables, problem.nb_constraints
        self.U = torch.empty((Bn, n), dtype=problem.dtype, device=problem.device)
        self.lam = torch.empty((Bn, m), dtype=problem.dtype, device=problem.device) if return_multipliers else None
        self.status = torch.empty((Bn,), dtype=torch.int32, device=problem.device)
        self.iters = torch.empty((Bn,), dtype=torch.int32, device=problem.device)
        _check_warm(problem, opt_kw)
        _check_order(problem, opt_kw)
        self._opts = _opts(max_iter, feas_tol, **opt_kw)
        ws_ = opt_kw.get("warm_state")
        # the stage-wise kernel's warm start continues from the vectors its previous launch left in THIS object's
        # workspace: the first launch keeps the Riccati factor, set_warm_start(True) switches to re-using it
        self._stage_warm = isinstance(ws_, WarmState) and ws_.kind == "stage"
        # (warm_start=True at construction: the first launch has nothing to continue from -- it factors, keeps the factor
        # and starts cold; launch() switches to the re-used factor + warm start right after it)
        self._stage_warm_pending = self._stage_warm and bool(self._opts.warm_start)
        if self._stage_warm:
            self._opts.flags |= _capi.OPT_KEEP_FACTOR
        if self._stagewise:
            dims, nbytes = problem.dims(), C.c_size_t(0)
            _capi.check(self._lib.mpcqp_stagewise_workspace_bytes(C.byref(dims), Bn, self._max_active, C.byref(nbytes)),
                        "mpcqp_stagewise_workspace_bytes")
            self._ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=problem.device)
        else:
            self._ws = _workspace(problem, True)
        self.rebind()

    def rebind(self) -> None:
        self._dims, self._cp = self.problem.dims(), self.problem.c_problem()
        head = (C.byref(self._dims), C.byref(self._cp), self.problem.batch_size, C.byref(self._opts))
        if self._stagewise:
            head = head + (self._max_active,)
        self._args = head + (
            self.U.data_ptr(), None if self.lam is None else self.lam.data_ptr(),
            self.status.data_ptr(), self.iters.data_ptr(), *_ws_args(self._ws),
        )
        self._entry = self._lib.mpcqp_stagewise_solve_batch if self._stagewise else self._lib.mpcqp_build_solve_batch

    def set_warm_start(self, on, warm_shift: Optional[int] = None) -> None:
        """Begin the next launches from the ``warm_state`` given at construction (or from the empty set): ``True`` /
        ``"operator"`` -- the stored active set and operator --, ``"active_set"`` -- the stored rows only, moved down by
        ``warm_shift`` rows (receding horizon: ``mk`` per step the horizon advanced; small problems only)."""
        if "warm_state" not in self._opt_kw:
            raise BackendError("PreparedSolve was built without warm_state=WarmState(problem)")
        self._opts.warm_start = _warm_mode(on)
        if warm_shift is not None:
            self._opts.warm_shift = int(warm_shift)
        if self._stage_warm:  # (contract of both: A, B, C, D and the weights are those of the launch before)
            keep, reuse = _capi.OPT_KEEP_FACTOR, _capi.OPT_REUSE_FACTOR
            self._opts.flags = (self._opts.flags & ~(keep | reuse)) | (reuse if on else keep)

    def set_order(self, order) -> None:
        """Pairing order of the next launches (:func:`pairing_order`; ``None``: natural). The tensor is kept alive here."""
        _check_order(self.problem, {"order": order})
        probe = _opts(order=order)  # (dtype / device / layout checks)
        self._opt_kw["order"] = order
        self._opts.order = probe.order

    def launch(self, stream=None) -> None:
        """Enqueue one fused build+solve of the whole batch on ``stream``
        (default: torch's current stream). Asynchronous."""
        sp = _stream_ptr() if stream is None else C.c_void_p(stream.cuda_stream)
        rc = self._entry(*self._args, sp)
        if rc != 0:
            _capi.check(rc, "mpcqp_stagewise_solve_batch" if self._stagewise else "mpcqp_build_solve_batch")
        if self._stage_warm_pending:
            self._stage_warm_pending = False
            self.set_warm_start(True)

    @property
    def plan(self) -> BatchPlan:
        return BatchPlan(self.problem, self.U, self.status, self.iters, self.lam)


class SharedModel:
    """Everything of a batch that does NOT depend on x0 / goal / targets, factored once.

    For problems that share A, B, C, D, e and the weights (an initial-state sweep such as
    BASELINE config 4, or the successive steps of time-invariant receding-horizon loops)
    this is the reference's own fast path -- build ``MPCQP`` once, then only
    ``update_cost_vector`` / ``update_constraint_vector`` (mpc_qp.py:129-163) and re-solve --
    taken to its end: L = chol(P), M = G L^-T, L^-T and the linear maps from the states to
    h and L^-1 q are kept in HBM (``mpcqp_factor_model``); every solve then starts at the
    active-set loop (``mpcqp_solve_model_batch``).
    """

    def __init__(self, template: BatchMPCProblem):
        torch = _torch()
        lib = _capi.load()
        _require_on_gpu(template.initial_state)
        for name in ("A", "B", "C", "D"):
            op = getattr(template, name)
            if op is not None and op.shape[0] != 1:
                raise ProblemDefinitionError(f"SharedModel: operand {name} differs across the batch")
        # e may differ per problem (bounds that move while the matrices stay, e.g. the ZMP bounds of the LIPM walking
        # controller): the model is then factored with the first problem's e and every solve passes its own
        # (mpcqp_solve_model_bounds_batch)
        self.per_problem_bounds = template.e is not None and template.e.shape[0] != 1
        self.template = template
        nx, N = template.state_dim, template.nb_timesteps
        n, m = template.nb_variables, template.nb_constraints
        nb = 1 + 2 * nx + N * nx
        dev, dt = template.device, template.dtype
        x0 = torch.zeros((nb, nx), dtype=dt, device=dev)
        goal = torch.zeros((nb, nx), dtype=dt, device=dev)
        tgt = torch.zeros((nb, N * nx), dtype=dt, device=dev)
        x0[1: 1 + nx] = torch.eye(nx, dtype=dt, device=dev)
        goal[1 + nx: 1 + 2 * nx] = torch.eye(nx, dtype=dt, device=dev)
        tgt[1 + 2 * nx:] = torch.eye(N * nx, dtype=dt, device=dev)
        pseudo = BatchMPCProblem(
            template.A, template.B, template.C, template.D, template.e[:1] if self.per_problem_bounds else template.e, N,
            template.terminal_cost_weight,
            template.stage_state_cost_weight, template.stage_input_cost_weight, x0, goal_state=goal,
            target_states=tgt, dtype=dt, device=dev)
        self.dims = pseudo.dims()  # cost flags from the weights (goal and targets are defined here)
        qp = BatchMPCQP(pseudo, keep_propagators=False)
        nbytes = C.c_size_t(0)
        _capi.check(lib.mpcqp_model_bytes(C.byref(self.dims), C.byref(nbytes)), "mpcqp_model_bytes")
        self.model = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)
        rc = lib.mpcqp_factor_model(C.byref(self.dims), qp.P[0].data_ptr(), qp.G[0].data_ptr() if m else None,
                                    qp.q.data_ptr(), qp.h.data_ptr() if m else None, self.model.data_ptr(),
                                    nbytes.value, _stream_ptr())
        _capi.check(rc, "mpcqp_factor_model")
        self._keep = qp  # inputs of the asynchronous factorisation

    def problem_for(self, x0, goal=None, targets=None, e=None) -> BatchMPCProblem:
        """A batch sharing this model's operands with the given states (and, optionally, its own bounds e)."""
        t = self.template
        return BatchMPCProblem(t.A, t.B, t.C, t.D, t.e if e is None else e, t.nb_timesteps, t.terminal_cost_weight,
                               t.stage_state_cost_weight, t.stage_input_cost_weight, x0, goal_state=goal,
                               target_states=targets, dtype=t.dtype, device=t.device)

    def prepare(self, problem: BatchMPCProblem, return_multipliers: bool = False,
                max_iter: Optional[int] = None, feas_tol: Optional[float] = None, **opt_kw) -> "PreparedModelSolve":
        return PreparedModelSolve(self, problem, return_multipliers, max_iter, feas_tol, **opt_kw)

    def solve(self, x0, goal=None, targets=None, return_multipliers: bool = False,
              max_iter: Optional[int] = None, feas_tol: Optional[float] = None, **opt_kw) -> BatchPlan:
        run = self.prepare(self.problem_for(x0, goal, targets), return_multipliers, max_iter, feas_tol, **opt_kw)
        run.launch()
        return run.plan


class PreparedModelSolve:
    """``PreparedSolve`` for a ``SharedModel``: one C call per launch; update the problem's
    states IN PLACE between launches."""

    def __init__(self, model: SharedModel, problem: BatchMPCProblem, return_multipliers: bool = False,
                 max_iter: Optional[int] = None, feas_tol: Optional[float] = None, **opt_kw):
        torch = _torch()
        self._lib = _capi.load()
        self.model, self.problem = model, problem
        self._opt_kw = opt_kw
        flags = model.dims.flags
        if (flags & _capi.Q_TERMINAL) and problem.goal_state is None:
            raise ProblemDefinitionError("MPC problem has terminal cost but the goal state is undefined")
        if (flags & _capi.Q_STAGE) and problem.target_states is None:
            raise ProblemDefinitionError("MPC problem has a stage state cost but the reference trajectory is undefined")
        Bn, n, m = problem.batch_size, problem.nb_variables, problem.nb_constraints
        self.U = torch.empty((Bn, n), dtype=problem.dtype, device=problem.device)
        self.lam = torch.empty((Bn, m), dtype=problem.dtype, device=problem.device) if return_multipliers else None
        self.status = torch.empty((Bn,), dtype=torch.int32, device=problem.device)
        self.iters = torch.empty((Bn,), dtype=torch.int32, device=problem.device)
        self._opts = _opts(max_iter, feas_tol, **opt_kw)
        self.rebind()

    def rebind(self) -> None:
        p = self.problem
        self._own_e = p.e is not None and p.e.shape[0] != 1  # bounds per problem
        self._ops = (p._operand(p.initial_state), p._operand(p.goal_state), p._operand(p.target_states),
                     p._operand(p.e) if self._own_e else None)
        tail = (C.byref(self._ops[0]), C.byref(self._ops[1]), C.byref(self._ops[2]), p.batch_size, C.byref(self._opts),
                self.U.data_ptr(), None if self.lam is None else self.lam.data_ptr(), self.status.data_ptr(),
                self.iters.data_ptr())
        head = (C.byref(self.model.dims), self.model.model.data_ptr())
        self._args = head + ((C.byref(self._ops[3]),) if self._own_e else ()) + tail

    def set_order(self, order) -> None:
        """Pairing order of the next launches (:func:`pairing_order`; ``None``: natural), as ``PreparedSolve.set_order``."""
        _check_order(self.problem, {"order": order})
        probe = _opts(order=order)
        self._opt_kw["order"] = order
        self._opts.order = probe.order

    def launch(self, stream=None) -> None:
        sp = _stream_ptr() if stream is None else C.c_void_p(stream.cuda_stream)
        if self._own_e:
            rc = self._lib.mpcqp_solve_model_bounds_batch(*self._args, sp)
        else:
            rc = self._lib.mpcqp_solve_model_batch(*self._args, sp)
        if rc != 0:
            _capi.check(rc, "mpcqp_solve_model_batch")

    @property
    def plan(self) -> BatchPlan:
        return BatchPlan(self.problem, self.U, self.status, self.iters, self.lam)


class BatchMPCQP:
    """Condensed QPs of a batch, kept in HBM (batched ``MPCQP``, mpc_qp.py:21-163)."""

    def __init__(self, problem: BatchMPCProblem, keep_propagators: bool = True):
        torch = _torch()
        lib = _capi.load()
        _require_on_gpu(problem.initial_state)
        Bn, n, m = problem.batch_size, problem.nb_variables, problem.nb_constraints
        nx, N = problem.state_dim, problem.nb_timesteps
        mk = lambda *shape: torch.empty(shape, dtype=problem.dtype, device=problem.device)  # noqa: E731
        self.P, self.q, self.G, self.h = mk(Bn, n, n), mk(Bn, n), mk(Bn, m, n), mk(Bn, m)
        self.Phi_all = mk(Bn, (N + 1) * nx, nx) if keep_propagators else None
        self.Psi_all = mk(Bn, (N + 1) * nx, n) if keep_propagators else None
        dims, cp = problem.dims(), problem.c_problem()
        self._ws = _workspace(problem, False)
        rc = lib.mpcqp_condense_batch(
            C.byref(dims), C.byref(cp), Bn, self.P.data_ptr(), self.q.data_ptr(), self.G.data_ptr(),
            self.h.data_ptr(), None if self.Phi_all is None else self.Phi_all.data_ptr(),
            None if self.Psi_all is None else self.Psi_all.data_ptr(), *_ws_args(self._ws), _stream_ptr())
        _capi.check(rc, "mpcqp_condense_batch")
        self.nb_timesteps, self.state_dim = N, nx

    def _update(self, problem: BatchMPCProblem, do_q: bool, do_h: bool) -> None:
        if self.Phi_all is None:
            raise ProblemDefinitionError("BatchMPCQP was built with keep_propagators=False")
        lib = _capi.load()
        dims, cp = problem.dims(), problem.c_problem()
        rc = lib.mpcqp_update_vectors_batch(
            C.byref(dims), C.byref(cp), self.Phi_all.data_ptr(), int(self.Phi_all.stride(0)),
            self.Psi_all.data_ptr(), int(self.Psi_all.stride(0)), problem.batch_size,
            self.q.data_ptr() if do_q else None, self.h.data_ptr() if do_h else None, _stream_ptr())
        _capi.check(rc, "mpcqp_update_vectors_batch")

    def update_cost_vector(self, problem: BatchMPCProblem) -> None:
        """q for new x0 / goal / targets (mpc_qp.py:129-149)."""
        self._update(problem, True, False)

    def update_constraint_vector(self, problem: BatchMPCProblem) -> None:
        """h = e - C Phi x0 for a new x0 (mpc_qp.py:151-163)."""
        self._update(problem, False, True)

    def solve(self, return_multipliers: bool = False, max_iter=None, feas_tol=None, **opt_kw):
        return solve_qp_batch(self.P, self.q, self.G, self.h, return_multipliers, max_iter, feas_tol, **opt_kw)


def solve_qp_batch(P, q, G, h, return_multipliers: bool = False, max_iter=None, feas_tol=None, **opt_kw):
    """Batched dense QP solve, ``min 1/2 x'Px + q'x s.t. Gx <= h`` per item
    (replaces ``qpsolvers.solve_problem`` at solve_mpc.py:43).
    Returns (x [B,n], lam [B,m] | None, status [B], iters [B])."""
    torch = _torch()
    lib = _capi.load()
    _require_on_gpu(P, q, G, h)
    P, q = P.contiguous(), q.contiguous()
    Bn, n = q.shape
    m = 0 if G is None else int(G.shape[1])
    G = None if G is None else G.contiguous()
    h = None if h is None else h.contiguous()
    x = torch.empty((Bn, n), dtype=P.dtype, device=P.device)
    lam = torch.empty((Bn, m), dtype=P.dtype, device=P.device) if return_multipliers else None
    status = torch.empty((Bn,), dtype=torch.int32, device=P.device)
    iters = torch.empty((Bn,), dtype=torch.int32, device=P.device)
    opts = _opts(max_iter, feas_tol, **opt_kw)
    nbytes = C.c_size_t(0)
    _capi.check(lib.mpcqp_solve_workspace_bytes(n, m, _dtype_code(P.dtype), Bn, C.byref(nbytes)),
                "mpcqp_solve_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=P.device) if nbytes.value else None
    rc = lib.mpcqp_solve_batch(
        n, m, _dtype_code(P.dtype), P.data_ptr(), q.data_ptr(), None if m == 0 else G.data_ptr(),
        None if m == 0 else h.data_ptr(), Bn, C.byref(opts), x.data_ptr(),
        None if lam is None else lam.data_ptr(), status.data_ptr(), iters.data_ptr(), *_ws_args(ws), _stream_ptr())
    _capi.check(rc, "mpcqp_solve_batch")
    return x, lam, status, iters


def rollout_batch(problem: BatchMPCProblem, U, initial_state=None):
    """X [B, N+1, nx] from U [B, N*nu] (``mpcqp_rollout_batch``; mpc_problem.py:316-335)."""
    torch = _torch()
    lib = _capi.load()
    x0 = problem.initial_state if initial_state is None else initial_state
    Bn = int(x0.shape[0])
    U = U.reshape(Bn, -1).contiguous()
    X = torch.empty((Bn, problem.nb_timesteps + 1, problem.state_dim), dtype=problem.dtype, device=problem.device)
    dims = problem.dims()
    opA, opB, opx = problem._operand(problem.A), problem._operand(problem.B), problem._operand(x0)
    rc = lib.mpcqp_rollout_batch(C.byref(dims), C.byref(opA), C.byref(opB), C.byref(opx), U.data_ptr(), Bn,
                                 X.data_ptr(), _stream_ptr())
    _capi.check(rc, "mpcqp_rollout_batch")
    return X


def rollout_single(problem, initial_state, inputs) -> np.ndarray:
    """``MPCProblem.integrate`` for one host problem, computed by the rollout kernel."""
    torch = _torch()
    device = _capi.require_gpu()
    N, nx, nu = problem.nb_timesteps, problem.state_dim, problem.input_dim
    A = np.stack([np.asarray(problem.get_transition_state_matrix(k), dtype=float).reshape(nx, nx) for k in range(N)])
    Bm = np.stack([np.asarray(problem.get_transition_input_matrix(k), dtype=float).reshape(nx, nu) for k in range(N)])
    bp = BatchMPCProblem.__new__(BatchMPCProblem)
    bp.dtype, bp.device = torch.float64, device
    bp.state_dim, bp.input_dim, bp.nb_timesteps, bp.ineq_dim = nx, nu, N, 0
    bp.A = _canon(_as_tensor(A, bp.dtype, device), (nx, nx), "A")
    bp.B = _canon(_as_tensor(Bm, bp.dtype, device), (nx, nu), "B")
    bp.C = bp.D = bp.e = bp.goal_state = bp.target_states = None
    bp.terminal_cost_weight, bp.stage_state_cost_weight, bp.stage_input_cost_weight = 1.0, None, 1.0
    bp.initial_state = _as_tensor(np.asarray(initial_state, dtype=float).reshape(1, nx), bp.dtype, device)
    U = _as_tensor(np.asarray(inputs, dtype=float).reshape(1, N * nu), bp.dtype, device)
    return rollout_batch(bp, U).cpu().numpy()[0]
