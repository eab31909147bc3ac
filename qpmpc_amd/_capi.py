"""ctypes binding of libmpcqp_hip.so (the C ABI of include/mpcqp.h).

The library is looked up in-tree (qpmpc_amd/lib/) only. If it is missing or a
call fails this module raises ``BackendError``: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

from .exceptions import BackendError

F64, F32 = 0, 1
P_TERMINAL, P_STAGE, Q_TERMINAL, Q_STAGE = 1, 2, 4, 8
SOLVED, MAX_ITER, INFEASIBLE, NOT_PD, SLOTS_FULL = 0, 1, 2, 3, 4
EINVAL, EWORKSPACE, EUNSUPPORTED = -1, -5, -6
ABI_VERSION = 11
OPT_FORCE_LDS, OPT_FORCE_GWS, OPT_FORCE_DENSE_G, OPT_ONE_PER_WAVE, OPT_FORCE_CONDENSED, OPT_STAGE_WIDE = 1, 2, 4, 8, 16, 32
OPT_KEEP_FACTOR, OPT_REUSE_FACTOR, OPT_PIPELINE_FACTOR, OPT_SEED_VIOLATED, OPT_EXACT_SELECTION = 64, 128, 256, 512, 1024
OPT_TWO_PER_WAVE, OPT_FOUR_PER_WAVE = 2048, 4096
OPT_STAGE_GENERAL = 8192
WARM_OPERATOR, WARM_ACTIVE_SET = 1, 2

# MPCQP_LIB (dev only) points at another build of the same sources for A/B timing.
LIB_PATH = os.environ.get("MPCQP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmpcqp_hip.so")

EXPORTS = (
    "mpcqp_abi_version",
    "mpcqp_error_string",
    "mpcqp_lds_bytes",
    "mpcqp_workspace_bytes",
    "mpcqp_warm_state_bytes",
    "mpcqp_warm_state_kind",
    "mpcqp_solve_workspace_bytes",
    "mpcqp_condense_batch",
    "mpcqp_condense_phase_batch",
    "mpcqp_update_vectors_batch",
    "mpcqp_solve_batch",
    "mpcqp_build_solve_batch",
    "mpcqp_stagewise_workspace_bytes",
    "mpcqp_stagewise_solve_batch",
    "mpcqp_rollout_batch",
    "mpcqp_model_bytes",
    "mpcqp_factor_model",
    "mpcqp_solve_model_batch",
    "mpcqp_solve_model_bounds_batch",
    "mpcqp_accumulate_stats",
    "mpcqp_order_workspace_bytes",
    "mpcqp_order_by_count",
    "mpcqp_wip_advance_batch",
    "mpcqp_wip_advance_stats_batch",
    "mpcqp_wip_period_batch",
    "mpcqp_wip_periods_batch",
    "mpcqp_lipm_advance_batch",
    "mpcqp_lipm_advance_stats_batch",
)


class Dims(C.Structure):
    _fields_ = [
        ("nx", C.c_int32), ("nu", C.c_int32), ("N", C.c_int32), ("mk", C.c_int32),
        ("dtype", C.c_int32), ("flags", C.c_int32),
        ("w_terminal", C.c_double), ("w_stage", C.c_double), ("w_input", C.c_double),
    ]


class Operand(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_int64), ("step_stride", C.c_int64)]


class Problem(C.Structure):
    _fields_ = [(name, Operand) for name in ("A", "B", "C", "D", "e", "x0", "goal", "targets")]


class SolveOpts(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int32), ("flags", C.c_int32), ("feas_tol", C.c_double),
        ("warm_state", C.c_void_p), ("warm_start", C.c_int32), ("factor_slot", C.c_int32),
        ("probe", C.c_void_p), ("warm_state_bytes", C.c_size_t), ("warm_shift", C.c_int32), ("reserved_", C.c_int32),
        ("order", C.c_void_p),
    ]


_lib = None


def load():
    """dlopen the in-tree library and declare its prototypes (once)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # torch must be imported first: its wheel bundles the HIP runtime this process
    # uses, and stream handles passed to the library are only valid inside that
    # runtime instance (loading the system libamdhip64 first creates a second one).
    import torch  # noqa: F401

    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exn:  # pragma: no cover - depends on the host's ROCm install
        raise BackendError(f"cannot load {LIB_PATH}: {exn}") from exn
    vp, i64, i32p = C.c_void_p, C.c_int64, C.POINTER(C.c_int32)
    lib.mpcqp_abi_version.restype = C.c_int
    lib.mpcqp_abi_version.argtypes = []
    lib.mpcqp_error_string.restype = C.c_char_p
    lib.mpcqp_error_string.argtypes = [C.c_int]
    lib.mpcqp_lds_bytes.restype = C.c_int
    lib.mpcqp_lds_bytes.argtypes = [C.POINTER(Dims), C.POINTER(C.c_size_t)]
    lib.mpcqp_workspace_bytes.restype = C.c_int
    lib.mpcqp_workspace_bytes.argtypes = [C.POINTER(Dims), i64, C.c_int32, C.POINTER(C.c_size_t)]
    lib.mpcqp_warm_state_bytes.restype = C.c_int
    lib.mpcqp_warm_state_bytes.argtypes = [C.POINTER(Dims), C.POINTER(C.c_size_t)]
    lib.mpcqp_warm_state_kind.restype = C.c_int
    lib.mpcqp_warm_state_kind.argtypes = [C.POINTER(Dims), C.POINTER(C.c_int32)]
    lib.mpcqp_solve_workspace_bytes.restype = C.c_int
    lib.mpcqp_solve_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, i64, C.POINTER(C.c_size_t)]
    lib.mpcqp_condense_phase_batch.restype = C.c_int
    lib.mpcqp_condense_phase_batch.argtypes = [C.POINTER(Dims), C.POINTER(Problem), i64, C.c_int32, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.mpcqp_condense_batch.restype = C.c_int
    lib.mpcqp_condense_batch.argtypes = [C.POINTER(Dims), C.POINTER(Problem), i64, vp, vp, vp, vp, vp, vp, vp,
                                         C.c_size_t, vp]
    lib.mpcqp_update_vectors_batch.restype = C.c_int
    lib.mpcqp_update_vectors_batch.argtypes = [C.POINTER(Dims), C.POINTER(Problem), vp, i64, vp, i64, i64, vp, vp, vp]
    lib.mpcqp_solve_batch.restype = C.c_int
    lib.mpcqp_solve_batch.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, i64,
                                      C.POINTER(SolveOpts), vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.mpcqp_build_solve_batch.restype = C.c_int
    lib.mpcqp_build_solve_batch.argtypes = [C.POINTER(Dims), C.POINTER(Problem), i64, C.POINTER(SolveOpts),
                                            vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.mpcqp_stagewise_workspace_bytes.restype = C.c_int
    lib.mpcqp_stagewise_workspace_bytes.argtypes = [C.POINTER(Dims), i64, C.c_int32, C.POINTER(C.c_size_t)]
    lib.mpcqp_stagewise_solve_batch.restype = C.c_int
    lib.mpcqp_stagewise_solve_batch.argtypes = [C.POINTER(Dims), C.POINTER(Problem), i64, C.POINTER(SolveOpts), C.c_int32,
                                                vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.mpcqp_model_bytes.restype = C.c_int
    lib.mpcqp_model_bytes.argtypes = [C.POINTER(Dims), C.POINTER(C.c_size_t)]
    lib.mpcqp_factor_model.restype = C.c_int
    lib.mpcqp_factor_model.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.mpcqp_solve_model_batch.restype = C.c_int
    lib.mpcqp_solve_model_batch.argtypes = [C.POINTER(Dims), vp, C.POINTER(Operand), C.POINTER(Operand),
                                            C.POINTER(Operand), i64, C.POINTER(SolveOpts), vp, vp, vp, vp, vp]
    lib.mpcqp_solve_model_bounds_batch.restype = C.c_int
    lib.mpcqp_solve_model_bounds_batch.argtypes = [C.POINTER(Dims), vp, C.POINTER(Operand), C.POINTER(Operand),
                                                   C.POINTER(Operand), C.POINTER(Operand), i64, C.POINTER(SolveOpts),
                                                   vp, vp, vp, vp, vp]
    lib.mpcqp_accumulate_stats.restype = C.c_int
    lib.mpcqp_accumulate_stats.argtypes = [vp, vp, i64, vp, vp]
    lib.mpcqp_order_workspace_bytes.restype = C.c_size_t
    lib.mpcqp_order_workspace_bytes.argtypes = [i64]
    lib.mpcqp_order_by_count.restype = C.c_int
    lib.mpcqp_order_by_count.argtypes = [vp, i64, vp, vp, C.c_size_t, vp]
    lib.mpcqp_wip_advance_batch.restype = C.c_int
    lib.mpcqp_wip_advance_batch.argtypes = [C.c_int32, vp, vp, i64, vp, C.c_int32, C.c_double, C.c_double, C.c_double,
                                            C.c_double, C.c_int32, vp, vp, vp, i64, vp]
    lib.mpcqp_wip_period_batch.restype = C.c_int
    lib.mpcqp_wip_period_batch.argtypes = [C.POINTER(Dims), C.POINTER(Problem), i64, C.POINTER(SolveOpts), vp, vp, vp, vp, vp,
                                           C.c_size_t, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, vp]
    lib.mpcqp_wip_periods_batch.restype = C.c_int
    lib.mpcqp_wip_periods_batch.argtypes = lib.mpcqp_wip_period_batch.argtypes[:-1] + [C.c_int32, vp]
    lib.mpcqp_wip_advance_stats_batch.restype = C.c_int
    lib.mpcqp_wip_advance_stats_batch.argtypes = [C.c_int32, vp, vp, i64, vp, vp, vp, C.c_int32, C.c_double, C.c_double,
                                                  C.c_double, C.c_double, C.c_int32, vp, vp, vp, i64, vp]
    lib.mpcqp_lipm_advance_batch.restype = C.c_int
    lib.mpcqp_lipm_advance_batch.argtypes = [C.c_int32, vp, vp, i64, vp, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.mpcqp_lipm_advance_stats_batch.restype = C.c_int
    lib.mpcqp_lipm_advance_stats_batch.argtypes = [C.c_int32, vp, vp, i64, vp, vp, vp, C.c_int32, C.c_double, C.c_int32,
                                                   C.c_int32, C.c_int32, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.mpcqp_rollout_batch.restype = C.c_int
    lib.mpcqp_rollout_batch.argtypes = [C.POINTER(Dims), C.POINTER(Operand), C.POINTER(Operand),
                                        C.POINTER(Operand), vp, i64, vp, vp]
    del i32p
    if lib.mpcqp_abi_version() != ABI_VERSION:
        raise BackendError(f"ABI mismatch: library {lib.mpcqp_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    """Turn a non-zero return code of the C ABI into ``BackendError``."""
    if code != 0:
        msg = load().mpcqp_error_string(code).decode()
        if code == -2:  # MPCQP_ETOOLARGE: name the envelope instead of leaving the caller guessing
            msg += (". Served: everything that fits 160 KiB of LDS; beyond that any horizon for systems with nx <= 32, "
                    "nu <= 8 (stage-wise kernels)")
        raise BackendError(f"{what} failed with code {code}: {msg}")


def require_gpu():
    """Return the torch CUDA(ROCm) device or raise: the product path needs a GPU."""
    import torch

    if not torch.cuda.is_available():
        raise BackendError(
            "no ROCm GPU visible to torch: qpmpc_amd computes only on MI355X-class "
            "devices and has no CPU fallback"
        )
    return torch.device("cuda", torch.cuda.current_device())


_gpu_ok = False
_streams = {}


def current_stream():
    """(torch stream object, raw hipStream_t as int) of torch's CURRENT stream, without the ~8 us that
    ``torch.cuda.current_stream()`` spends on device-index handling per call: the raw handle comes from torch's C
    layer and the stream object is cached per (device, handle). Raises ``BackendError`` without a GPU (every launch
    goes through here, so nothing can silently run elsewhere)."""
    global _gpu_ok
    import torch

    if not _gpu_ok:
        require_gpu()
        _gpu_ok = True
    try:
        dev = torch._C._cuda_getDevice()
        raw = torch._C._cuda_getCurrentRawStream(dev)
    except AttributeError:  # (another torch build: the public, slower way)
        s = torch.cuda.current_stream()
        return s, s.cuda_stream
    s = _streams.get((dev, raw))
    if s is None:
        s = torch.cuda.current_stream()
        if s.cuda_stream != raw:  # (should not happen: fall back to what torch says)
            return s, s.cuda_stream
        _streams[(dev, raw)] = s
    return s, raw
