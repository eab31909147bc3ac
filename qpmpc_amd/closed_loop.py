"""Receding-horizon closed loops on the device (BASELINE config 3).

Batched counterpart of the loop in the reference's
examples/wheeled_inverted_pendulum.py:99-118: at every MPC step each loop
(1) rebuilds its reference ramp from its current state (:65-83, :101-108),
(2) builds and solves its MPC problem -- one fused launch for the whole batch,
replacing the per-step ``solve_mpc`` call (:109) -- and (3) applies the first
input to the nonlinear plant for NB_SUBSTEPS Taylor sub-steps (:110-111,
qpmpc/systems/wheeled_inverted_pendulum.py:127-160). States, references and
inputs stay in HBM; the host only enqueues work.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

import ctypes as C

from . import _capi
from .batch import BatchMPCProblem, PreparedSolve, SharedModel, WarmState, _dtype_code, _stream_ptr
from .systems import WheeledInvertedPendulum

NB_SUBSTEPS = 15  # examples/wheeled_inverted_pendulum.py:31


def wip_problem(pendulum: WheeledInvertedPendulum, x0, ltv: bool = True, dtype=None, device=None) -> BatchMPCProblem:
    """Batch of WIP problems (weights of the reference example, :90-94) for states x0 [B,4]."""
    A, B = pendulum.discretized_dynamics()
    N = pendulum.nb_timesteps
    if ltv:  # "LTV = pass A, B as N-lists" 
        A = np.ascontiguousarray(np.broadcast_to(A, (N, 4, 4)))
        B = np.ascontiguousarray(np.broadcast_to(B, (N, 4, 1)))
    D = np.array([[1.0], [-1.0]])
    e = np.array([pendulum.max_ground_accel, pendulum.max_ground_accel])
    x0 = np.asarray(x0, dtype=float)
    ts = np.stack([pendulum.target_states(x, 0.0) for x in x0])
    return BatchMPCProblem(A, B, None, D, e, N, 10.0, 1.0, 1e-3, x0, goal_state=ts[:, -4:],
                           target_states=ts[:, :-4], dtype=dtype, device=device)


class WIPClosedLoop:
    """``B`` independent wheeled-inverted-pendulum control loops advancing in lock step."""

    def __init__(self, x0, nb_timesteps: int = 50, sampling_period: float = 0.024, target_vel: float = 0.5,
                 ltv: bool = True, max_iter: Optional[int] = None, shared_model: bool = False, fused_period: bool = True,
                 reuse_factor: bool = False, pipeline_factor: bool = False, periods_per_launch: int = 1):
        """``fused_period``: run a whole period (solve + plant + next problem + bookkeeping) in ONE launch,
        ``mpcqp_wip_period_batch``, where the solver kernel supports it (else, and with ``shared_model``, two
        launches per period). ``reuse_factor``: the dynamics and weights never change along the loop, so the
        stage-wise kernel keeps its Riccati factor in the workspace after the first period and starts from it
        afterwards (``MPCQP_OPT_KEEP_FACTOR`` / ``MPCQP_OPT_REUSE_FACTOR``: build once, re-solve -- the reference's
        ``update_cost_vector`` / ``update_constraint_vector`` usage; ignored by kernels that rebuild anyway).
        ``pipeline_factor``: the factor is REBUILT every period, like the reference's ``solve_mpc`` does, but by a second
        wavefront that works on the next period's factor while the first one solves this period with the factor the
        previous launch left (``MPCQP_OPT_PIPELINE_FACTOR``; the operands of the next period are known here: they do
        not change). Same trajectories, bit for bit; falls back to the plain rebuilding period where the kernel does
        not offer it. ``periods_per_launch``: with the fused period, ``step(n)`` runs up to that many consecutive periods
        per launch (``mpcqp_wip_periods_batch``: the wavefront that solved period t carries on with period t + 1; the
        loops never interact, so nothing has to meet at a period's end) -- same trajectories bit for bit, without the
        dispatch gap between the periods. The first period of an episode is always a launch of its own."""
        import torch

        self.pendulum = WheeledInvertedPendulum(nb_timesteps=nb_timesteps, sampling_period=sampling_period)
        self.problem = wip_problem(self.pendulum, x0, ltv=ltv)
        self.target_vel = float(target_vel)
        # shared_model=True: the dynamics, constraints and weights never change along the
        # loop, so P, its factor and M are computed ONCE (SharedModel) instead of at every
        # step; False reproduces the reference, which rebuilds everything per step
        # (solve_mpc.py:42).
        if shared_model:
            self.solver = SharedModel(self.problem).prepare(self.problem, max_iter=max_iter)
        else:
            self.solver = PreparedSolve(self.problem, max_iter=max_iter)
        self.states = self.problem.initial_state.clone()
        N, T = nb_timesteps, sampling_period
        dev, dt = self.states.device, self.states.dtype
        # ramp[k] = k*T*target_vel for k = 0..N (reference position offsets)
        self._ramp = torch.arange(N + 1, device=dev, dtype=dt) * (T * self.target_vel)
        self.mpc_steps = 0
        self._stats = torch.zeros((2,), dtype=torch.int64, device=dev)  # [failed, sum of iterations], mpcqp_accumulate_stats
        # per-loop [failed, iterations] counters of the fused period (no atomics in the solver kernel's epilogue);
        # the public counters ``failed`` / ``iters_total`` (properties below) add both
        self._loopstats = torch.zeros((self.problem.batch_size, 2), dtype=torch.int64, device=dev)
        self._fused = bool(fused_period) and not shared_model  # cleared by the first launch if the kernel cannot do it
        self._period_args = None
        self._ppl = max(1, int(periods_per_launch))
        self._reuse = bool(reuse_factor) and not shared_model
        self._pipe = bool(pipeline_factor) and not shared_model and not self._reuse
        if self._reuse or self._pipe:
            self.solver._opts.flags |= _capi.OPT_KEEP_FACTOR

    def _write_references(self) -> None:
        """target_states / goal_state / initial_state of every loop from its current state, in place (so the bound
        pointers stay valid): one launch of the plant kernel with zero sub-steps (``mpcqp_wip_advance_batch``, nsub = 0)."""
        p, pend = self.problem, self.pendulum
        rc = _capi.load().mpcqp_wip_advance_batch(
            _dtype_code(p.dtype), self.states.data_ptr(), self.solver.U.data_ptr(), p.nb_variables, None, pend.nb_timesteps,
            pend.sampling_period, self.target_vel, pend.length, pend.GRAVITY, 0, p.initial_state.data_ptr(),
            p.goal_state.data_ptr(), p.target_states.data_ptr(), p.batch_size, _stream_ptr())
        _capi.check(rc, "mpcqp_wip_advance_batch")

    def _write_references_torch(self) -> None:
        """The same with torch operations (kept as a cross-check of the kernel: tests/)."""
        p, N = self.problem, self.pendulum.nb_timesteps
        pos = self.states[:, 0:1] + self._ramp[None, :]  # [B, N+1]
        tgt = p.target_states.view(-1, N, 4)
        tgt.zero_()
        tgt[:, :, 0] = pos[:, :N]
        tgt[:, :, 2] = self.target_vel
        p.goal_state.zero_()
        p.goal_state[:, 0] = pos[:, N]
        p.goal_state[:, 2] = self.target_vel
        p.initial_state.copy_(self.states)

    def reset(self, x0) -> None:
        """Put every loop back to state ``x0`` [B, 4] (a new episode; counters restart)."""
        import torch

        self.states.copy_(torch.as_tensor(np.asarray(x0, dtype=float), dtype=self.states.dtype, device=self.states.device))
        self.mpc_steps = 0
        self._stats.zero_()
        self._loopstats.zero_()
        # (the kept factor stays valid across episodes: it depends on the dynamics and the weights only)

    @property
    def failed(self):
        """0-d int64 device tensor: periods of any loop whose plan was not found, whichever launch counted them
        (the two-launch period's global counter + the fused period's per-loop counters)."""
        return self._stats[0] + self._loopstats[:, 0].sum()

    @property
    def iters_total(self):
        """0-d int64 device tensor: active-set iterations of all solves so far."""
        return self._stats[1] + self._loopstats[:, 1].sum()

    def step(self, nb_mpc_steps: int = 1):
        """Advance every loop by ``nb_mpc_steps`` MPC periods: per period one solver launch
        and one fused plant + reference + bookkeeping launch (``mpcqp_wip_advance_stats_batch``). Asynchronous."""
        lib = _capi.load()
        p, pend = self.problem, self.pendulum
        if self.mpc_steps == 0:
            self._write_references()
        left = int(nb_mpc_steps)
        while left > 0:
            left -= 1
            if self._reuse and self.mpc_steps == 1:  # the first period of the episode left the factor in the workspace
                self.solver._opts.flags = (self.solver._opts.flags & ~_capi.OPT_KEEP_FACTOR) | _capi.OPT_REUSE_FACTOR
            if self._pipe:
                o = self.solver._opts
                if self.mpc_steps == 0:  # the first period factors for itself and keeps the factor in image 0
                    o.flags = (o.flags & ~_capi.OPT_PIPELINE_FACTOR) | _capi.OPT_KEEP_FACTOR
                    o.factor_slot = 0
                else:  # period t solves with image (t - 1) % 2 while the factor for period t + 1 goes into image t % 2
                    o.flags = (o.flags & ~_capi.OPT_KEEP_FACTOR) | _capi.OPT_PIPELINE_FACTOR
                    o.factor_slot = (self.mpc_steps - 1) % 2
            if self._fused:
                if self._period_args is None:  # converted once: the call is launch-rate-bound otherwise
                    import ctypes as C

                    self._period_args = tuple(self.solver._args) + (
                        C.c_void_p(self.states.data_ptr()), C.c_void_p(self._loopstats.data_ptr()),
                        C.c_double(pend.sampling_period), C.c_double(self.target_vel), C.c_double(pend.length),
                        C.c_double(pend.GRAVITY), C.c_int32(NB_SUBSTEPS))
                # periods of this launch: the episode's first period stands alone (it keeps its factor for the others)
                k = 1 if self.mpc_steps == 0 else min(self._ppl, left + 1)
                rc = lib.mpcqp_wip_periods_batch(*self._period_args, k, _stream_ptr())
                if rc == _capi.EUNSUPPORTED and self._pipe and self.mpc_steps > 0:
                    # (this horizon has no factor image to pipeline through: plain rebuilding periods)
                    self._pipe = False
                    self.solver._opts.flags &= ~(_capi.OPT_PIPELINE_FACTOR | _capi.OPT_KEEP_FACTOR)
                    rc = lib.mpcqp_wip_periods_batch(*self._period_args, k, _stream_ptr())
                if rc == _capi.EUNSUPPORTED and k > 1:
                    # (this horizon has no multi-period instantiation; the fused single-period launch may still exist)
                    self._ppl = k = 1
                    rc = lib.mpcqp_wip_periods_batch(*self._period_args, k, _stream_ptr())
                if rc == _capi.EUNSUPPORTED:
                    self._fused = False  # (another kernel serves this size: two launches per period)
                else:
                    _capi.check(rc, "mpcqp_wip_periods_batch")
                    self.mpc_steps += k
                    left -= k - 1
                    continue
            if self._pipe:
                rc = self.solver._entry(*self.solver._args, _stream_ptr())
                if rc == _capi.EUNSUPPORTED:  # (a kernel without factor images serves this size: plain rebuilds)
                    self._pipe = False
                    self.solver._opts.flags &= ~(_capi.OPT_PIPELINE_FACTOR | _capi.OPT_KEEP_FACTOR)
                    rc = self.solver._entry(*self.solver._args, _stream_ptr())
                _capi.check(rc, "mpcqp_build_solve_batch")
            else:
                self.solver.launch()
            rc = lib.mpcqp_wip_advance_stats_batch(
                _dtype_code(p.dtype), self.states.data_ptr(), self.solver.U.data_ptr(), p.nb_variables,
                self.solver.status.data_ptr(), self.solver.iters.data_ptr(), self._stats.data_ptr(), pend.nb_timesteps,
                pend.sampling_period, self.target_vel, pend.length, pend.GRAVITY, NB_SUBSTEPS, p.initial_state.data_ptr(),
                p.goal_state.data_ptr(), p.target_states.data_ptr(), p.batch_size, _stream_ptr())
            _capi.check(rc, "mpcqp_wip_advance_stats_batch")
            self.mpc_steps += 1
        return self.states

    def stats(self) -> Dict[str, float]:
        B = self.problem.batch_size
        solves = max(self.mpc_steps * B, 1)
        return {
            "loops": B,
            "mpc_steps": self.mpc_steps,
            "builds_and_solves": self.mpc_steps * B,
            "failed": int(self.failed.item()),
            "mean_iters": float(self.iters_total.item()) / solves,
        }


# ---------------------------------------------------------------------------
# LIPM walking controller
# ---------------------------------------------------------------------------
MAX_ZMP_DIST = 100.0  # examples/lipm_walking_controller.py:29


class LIPMWalkingLoop:
    """``B`` walkers running the model-predictive part of the LIPM walking controller in lock step.

    Batched counterpart of examples/lipm_walking_controller.py:304-333. Per MPC period every walker
    (1) rebuilds the ZMP bounds ``e_k`` of its receding horizon from its footstep phase and its goal
    state (``PhaseStepper.get_nb_steps`` + ``update_goal_and_constraints``, :134-213) -- a per-step
    *list* of inequality vectors in the reference, a ``[B, N, 2]`` operand here --, (2) solves its MPC
    problem (triple integrator, nx=3, nu=1, N=16, ZMP rows ``C``; one fused launch for the batch,
    replacing ``solve_mpc`` at :309), (3) integrates the first jerk exactly for ``substeps``
    sub-steps (:216-236, :311-312) and (4) advances its phase (:329-332). Walkers differ by initial
    phase, support foot, stride lengths and foot size; the model (A, B, C) is shared (stride 0).
    Everything stays on the device; the host only enqueues work.
    """

    def __init__(self, batch: int, strides=None, foot_size=None, index=None, stride_index=None, support=None,
                 state=None, com_height: float = 0.84, dsp_duration: float = 0.1, ssp_duration: float = 0.7,
                 gravity: float = 9.81, init_support_foot_pos: float = 0.09, nb_timesteps: int = 16,
                 sampling_period: float = 0.1, substeps: int = 15, max_iter: Optional[int] = None,
                 warm_start=False, shared_model: bool = False):
        import torch

        _capi.require_gpu()
        dev, f64 = torch.device("cuda", torch.cuda.current_device()), torch.float64
        T, N = float(sampling_period), int(nb_timesteps)
        self.nb_timesteps, self.sampling_period, self.substeps = N, T, int(substeps)
        self.nb_dsp, self.nb_ssp = int(round(dsp_duration / T)), int(round(ssp_duration / T))
        if 2 * (self.nb_dsp + self.nb_ssp) < N:  # PhaseStepper.__init__, :107-110
            from .exceptions import ProblemDefinitionError

            raise ProblemDefinitionError("there are more than two steps in the receding horizon")
        self.omega = float(np.sqrt(gravity / com_height))

        def dev_t(v, default, dtype, shape):
            a = np.broadcast_to(np.asarray(default if v is None else v), shape)
            return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)

        self.strides = dev_t(strides, (-0.18, 0.18), f64, (batch, 2))
        self.foot_size = dev_t(foot_size, 0.065, f64, (batch,))
        self.index = dev_t(index, 5, torch.int64, (batch,))  # initial index of the reference, :117
        self.stride_index = dev_t(stride_index, 0, torch.int64, (batch,))
        self.support = dev_t(support, init_support_foot_pos, f64, (batch,))
        if state is None:  # ZMP at the centre of the first foothold, DCM halfway (:300-302)
            s0 = self.support
            state = torch.stack([torch.zeros_like(s0), 0.5 * self.omega * s0, -self.omega**2 * s0], dim=1)
        self.states = dev_t(state.cpu().numpy() if hasattr(state, "cpu") else state, None, f64, (batch, 3))
        # build_mpc_problem (:59-101): shared LTI model, per-walker e / x0 / goal
        A = np.array([[1.0, T, T**2 / 2.0], [0.0, 1.0, T], [0.0, 0.0, 1.0]])
        Bm = np.array([T**3 / 6.0, T**2 / 2.0, T]).reshape((3, 1))
        zmp = np.array([1.0, 0.0, -1.0 / self.omega**2])
        self.zmp_from_state = torch.tensor(zmp, dtype=f64, device=dev)
        Cm = np.array([+zmp, -zmp])
        e0 = np.full((batch, N, 2), MAX_ZMP_DIST)
        self.problem = BatchMPCProblem(A, Bm, Cm, None, e0, N, 1.0, None, 1e-3, np.zeros((batch, 3)),
                                       goal_state=np.zeros((batch, 3)))
        # warm_start: every period begins from the previous period's active set and operator (the model
        # A, B, C is time-invariant here, only the bounds e, x0 and the goal move; MpcqpSolveOpts.warm_state)
        # warm_start="active_set": only last period's active ROWS are kept and enter first, moved one step down the horizon
        # (MPCQP_WARM_ACTIVE_SET with warm_shift = mk = 2: row (k, i) of the last period is row (k - 1, i) of this one)
        self.warm_state = WarmState(self.problem) if warm_start else None
        self._warm_mode = warm_start
        self.model = None
        if shared_model:
            # The matrices never change here, only e / x0 / goal do: factor them ONCE (the reference's own usage of
            # MPCQP, mpc_qp.py:129-163: build, then update_cost_vector / update_constraint_vector per period) and give
            # every period's solve its bounds (mpcqp_solve_model_bounds_batch). Same plans as the rebuilding loop.
            from .batch import SharedModel

            if warm_start:
                raise ValueError("LIPMWalkingLoop: warm_start and shared_model are exclusive")
            self.model = SharedModel(self.problem)
            self.solver = self.model.prepare(self.problem, max_iter=max_iter)
        else:
            self.solver = (PreparedSolve(self.problem, max_iter=max_iter, warm_state=self.warm_state) if warm_start
                           else PreparedSolve(self.problem, max_iter=max_iter))
        self._k = torch.arange(N, device=dev)
        self._problem_written = False  # e / goal / x0 of the CURRENT phase are in the problem buffers
        self.mpc_steps = 0
        self._stats = torch.zeros((2,), dtype=torch.int64, device=dev)  # [failed, sum of iterations], mpcqp_accumulate_stats
        self.failed, self.iters_total = self._stats[0], self._stats[1]

    # -- PhaseStepper.get_nb_steps (:134-165), vectorised over the batch ---------------------------
    def _segment_counts(self):
        import torch

        nb_dsp, nb_ssp, N = self.nb_dsp, self.nb_ssp, self.nb_timesteps
        zero = torch.zeros_like(self.index)
        offset = self.index
        init_dsp = torch.clamp(nb_dsp - offset, min=0)
        offset = torch.clamp(offset - nb_dsp, min=0)
        init_ssp = torch.clamp(nb_ssp - offset, min=0)
        remaining = N - init_dsp - init_ssp
        next_dsp = torch.clamp(remaining, max=nb_dsp)
        remaining = torch.maximum(zero, remaining - nb_dsp)
        next_ssp = torch.clamp(remaining, max=nb_ssp)
        remaining = torch.maximum(zero, remaining - nb_ssp)
        last_dsp = torch.clamp(remaining, max=nb_dsp)
        remaining = torch.maximum(zero, remaining - nb_dsp)
        last_ssp = torch.clamp(remaining, max=nb_ssp)
        return torch.stack([init_dsp, init_ssp, next_dsp, next_ssp, last_dsp, last_ssp], dim=1)

    def _foot_positions(self):
        nxt = self.support + self.strides.gather(1, self.stride_index[:, None])[:, 0]  # get_next_foot_pos
        last = nxt + self.strides.gather(1, ((self.stride_index + 1) % 2)[:, None])[:, 0]  # get_last_foot_pos
        return nxt, last

    def _write_goal_and_constraints(self) -> None:
        """update_goal_and_constraints (:179-213) for every walker, in place."""
        import torch

        counts = self._segment_counts()  # [B, 6]
        nxt, last = self._foot_positions()
        half = 0.5 * self.foot_size
        free = torch.full_like(self.support, MAX_ZMP_DIST)
        upper = torch.stack([free, self.support + half, free, nxt + half, free, last + half], dim=1)
        lower = torch.stack([free, -(self.support - half), free, -(nxt - half), free, -(last - half)], dim=1)
        ends = torch.cumsum(counts, dim=1)  # step k lies in segment #(ends <= k)
        seg = (self._k[None, :, None] >= ends[:, None, :]).sum(dim=2).clamp(max=5)  # [B, N]
        e = self.problem.e.view(-1, self.nb_timesteps, 2)
        e[:, :, 0] = upper.gather(1, seg)
        e[:, :, 1] = lower.gather(1, seg)
        goal = self.problem.goal_state
        goal.zero_()
        goal[:, 0] = torch.where(counts[:, 4] > 0, last, nxt)
        self.problem.initial_state.copy_(self.states)

    def _integrate(self, jerk):
        """Constant-jerk plant (:216-236), ``substeps`` exact sub-steps like the example (:311-312)."""
        import torch

        dt = self.sampling_period / self.substeps
        p, v, a = self.states[:, 0], self.states[:, 1], self.states[:, 2]
        for _ in range(self.substeps):
            p, v, a = (p + dt * (v + dt * (a / 2 + dt * jerk / 6)), v + dt * (a + dt * (jerk / 2)), a + dt * jerk)
        self.states = torch.stack([p, v, a], dim=1)

    def _advance_phase(self) -> None:
        """phase.advance(); on wrap-around the swing foot becomes the support foot (:329-332)."""
        import torch

        nxt, _ = self._foot_positions()
        index = self.index + 1
        index = torch.where(index >= self.nb_dsp + self.nb_ssp, torch.zeros_like(index), index)
        wrapped = index == 0
        self.support = torch.where(wrapped, nxt, self.support)
        self.stride_index = torch.where(wrapped, (self.stride_index + 1) % 2, self.stride_index)
        self.index = index

    def _advance_fused(self, first: bool) -> None:
        """One launch of ``mpcqp_lipm_advance_stats_batch``: plant + phase + next problem + bookkeeping (for the
        very first period only the problem of the current phase)."""
        p = self.problem
        rc = _capi.load().mpcqp_lipm_advance_stats_batch(
            _dtype_code(p.dtype), self.states.data_ptr(), None if first else self.solver.U.data_ptr(), p.nb_variables,
            None if first else self.solver.status.data_ptr(), None if first else self.solver.iters.data_ptr(),
            None if first else self._stats.data_ptr(), self.nb_timesteps, self.sampling_period, self.substeps,
            self.nb_dsp, self.nb_ssp, MAX_ZMP_DIST, self.index.data_ptr(), self.stride_index.data_ptr(),
            self.support.data_ptr(), self.strides.data_ptr(), self.foot_size.data_ptr(), p.initial_state.data_ptr(),
            p.goal_state.data_ptr(), p.e.data_ptr(), p.batch_size, _stream_ptr())
        _capi.check(rc, "mpcqp_lipm_advance_stats_batch")

    def step(self, nb_mpc_steps: int = 1, fused: bool = True):
        """Advance every walker by ``nb_mpc_steps`` MPC periods: per period one solver launch and one
        fused plant + phase + next-problem launch. ``fused=False`` does the same bookkeeping with torch
        ops (dozens of small launches; kept as a cross-check of the kernel). Asynchronous."""
        import torch

        for _ in range(nb_mpc_steps):
            if fused:
                if not self._problem_written:
                    self._advance_fused(first=True)
                self.solver.launch()
                if self.warm_state is not None and self.mpc_steps == 0:
                    self.solver.set_warm_start(self._warm_mode, warm_shift=2)  # from the second period on
                self._advance_fused(first=False)
                self._problem_written = True
            else:
                self._problem_written = False
                self._write_goal_and_constraints()
                self.solver.launch()
                if self.warm_state is not None and self.mpc_steps == 0:
                    self.solver.set_warm_start(self._warm_mode, warm_shift=2)
                ok = self.solver.status == 0
                jerk = torch.where(ok, self.solver.U[:, 0], torch.zeros_like(self.solver.U[:, 0]))
                self._integrate(jerk)
                self._advance_phase()
                _capi.load().mpcqp_accumulate_stats(self.solver.status.data_ptr(), self.solver.iters.data_ptr(),
                                                    self.problem.batch_size, self._stats.data_ptr(), _stream_ptr())
            self.mpc_steps += 1
        return self.states

    def zmp(self):
        """Current ZMP of every walker (``zmp_from_state``, :53-55)."""
        return self.states @ self.zmp_from_state

    def stats(self) -> Dict[str, float]:
        B = self.problem.batch_size
        solves = max(self.mpc_steps * B, 1)
        return {"loops": B, "mpc_steps": self.mpc_steps, "builds_and_solves": self.mpc_steps * B,
                "failed": int(self.failed.item()), "mean_iters": float(self.iters_total.item()) / solves}
