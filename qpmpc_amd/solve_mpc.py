"""``solve_mpc``: one MPC problem in, one ``Plan`` out.

Drop-in for the reference's entry point (qpmpc/solve_mpc.py:16-44). With a HIP
solver name the whole of ``MPCQP(problem)`` + ``solve_problem`` runs in one fused
kernel (``mpcqp_build_solve_batch``, batch of one). Any other name follows the
reference literally -- condense (on the GPU) then hand ``mpc_qp.problem`` to
``qpsolvers.solve_problem`` -- and needs that package installed.
"""
from __future__ import annotations

import logging

import numpy as np

from .batch import HIP_SOLVERS, BatchMPCProblem, solve_mpc_batch
from .exceptions import BackendError, ProblemDefinitionError
from .mpc_problem import MPCProblem
from .mpc_qp import MPCQP
from .plan import Plan, Solution

available_solvers = list(HIP_SOLVERS)


_HIP_KWARGS = ("max_iter", "feas_tol", "warm_state", "warm_start", "initvals", "verbose")


def solve_mpc(problem: MPCProblem, solver: str, sparse: bool = False, **kwargs) -> Plan:
    """Solve a linear time-variant MPC problem.

    Args:
        problem: problem to solve (its initial state must be set).
        solver: required, like upstream (qpmpc/solve_mpc.py:18). ``"hip_gi"`` (aliases ``"hip"``,
            ``"mpcqp_hip"``) selects the fused MI355X path; any qpsolvers backend name works if
            qpsolvers is installed (the condensing still runs on the GPU).
        sparse: only meaningful for qpsolvers backends (CSC wrappers, as upstream).
        kwargs: forwarded verbatim to ``qpsolvers.solve_problem`` for qpsolvers backends. The HIP
            solver honours ``max_iter``, ``feas_tol``, ``warm_state=WarmState(...)`` with
            ``warm_start=True`` (solver warm start from the previous call's active set, see
            ``qpmpc_amd.WarmState``); it accepts and ignores ``initvals`` (a primal guess is of no use
            to a dual active-set method -- qpsolvers' quadprog wrapper ignores it the same way, with a
            warning) and ``verbose``. Any other keyword raises ``TypeError``: nothing is dropped silently.

    Supported sizes of the HIP path (the dispatch table is DESIGN.md 3.0): problems of at most 16 variables and 32 rows take the
    fused on-chip kernels (four or two problems per wavefront); beyond that the stage-wise kernels, which never form the
    condensed QP, serve every horizon -- the narrow one (``nx <= 4``, ``nu <= 2``, up to 128 variables), the wide one
    (``nx <= 16``, ``nu <= 4``; thin-QR active-set operator, float32 and float64) and the general one (``nx <= 32``,
    ``nu <= 8``, float64, one workgroup per problem: slower). ``nu > 8`` is served by the dense HBM-resident path up to
    ``n = N*nu <= 256``; only ``nx > 32`` or ``nu > 8`` together with ``n > 256`` has no kernel (``MPCQP_ETOOLARGE`` ->
    ``BackendError`` naming the envelope). A problem that wants more rows active at once than a stage-wise kernel's slots
    hold is solved again with more slots. float32 problems with at most 160 variables are solved in float64 on converted
    operands (mpcqp_capi.hip).

    Returns:
        A ``Plan``; empty (``is_empty``) when no solution was found.
    """
    if problem.initial_state is None:
        raise ProblemDefinitionError("initial state is undefined")
    if solver in HIP_SOLVERS:
        unknown = sorted(set(kwargs) - set(_HIP_KWARGS))
        if unknown:
            raise TypeError(f"solve_mpc(solver='{solver}') got keyword arguments it cannot honour: {unknown}; "
                            f"supported: {list(_HIP_KWARGS)}")
        if kwargs.get("initvals") is not None:
            logging.warning("hip_gi: warm-start values ignored (dual active-set method); use warm_state=")
        warm_state, warm_start = kwargs.get("warm_state"), bool(kwargs.get("warm_start", False))
        from .single import solve_single

        fast = None if warm_state is not None else solve_single(problem, kwargs.get("max_iter"), kwargs.get("feas_tol"))
        if fast is not None:  # one upload, one fused launch + roll-out, one download
            x, z, X, status, iters = fast
            plan = Plan(problem, Solution(None, x=x, z=z, found=(status == 0), extras={"status": status, "iters": iters}))
            if X is not None:  # already rolled out on the device; used if the initial state is still the same
                plan._precomputed_rollout = (np.asarray(problem.initial_state, dtype=float).ravel().copy(), X)
            return plan
        bp = BatchMPCProblem.from_problems([problem])
        opt_kw = {} if warm_state is None else {"warm_state": warm_state, "warm_start": warm_start}
        bplan = solve_mpc_batch(bp, solver=solver, return_multipliers=True, retry_unsolved=warm_state is None,
                                max_iter=kwargs.get("max_iter"), feas_tol=kwargs.get("feas_tol"), **opt_kw)
        status = int(bplan.status[0].item())
        x = bplan.U[0].cpu().numpy()
        z = bplan.multipliers[0].cpu().numpy()[bp.valid_rows]
        qpsol = Solution(None, x=x if status == 0 else None, z=z if status == 0 else None,
                         found=(status == 0), extras={"status": status, "iters": int(bplan.iters[0].item())})
        return Plan(problem, qpsol)
    try:
        from qpsolvers import solve_problem
    except ImportError as exn:
        raise BackendError(
            f"solver '{solver}' needs the qpsolvers package, which is not installed; "
            f"HIP solvers available: {available_solvers}"
        ) from exn
    mpc_qp = MPCQP(problem, sparse=sparse)
    qpsol = solve_problem(mpc_qp.problem, solver=solver, **kwargs)
    return Plan(problem, qpsol)


__all__ = ["solve_mpc", "available_solvers"]
