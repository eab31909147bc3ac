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
from .batch import BatchMPCProblem, PreparedSolve, SharedModel, _dtype_code, _stream_ptr
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
                 ltv: bool = True, max_iter: Optional[int] = None, shared_model: bool = False):
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
        self.failed = torch.zeros((), dtype=torch.int64, device=dev)
        self.iters_total = torch.zeros((), dtype=torch.int64, device=dev)

    def _write_references(self) -> None:
        """target_states / goal_state of every loop, in place (so the bound pointers stay valid)."""
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

    def step(self, nb_mpc_steps: int = 1):
        """Advance every loop by ``nb_mpc_steps`` MPC periods: per period one solver launch
        and one fused plant + reference launch (``mpcqp_wip_advance_batch``). Asynchronous."""
        lib = _capi.load()
        p, pend = self.problem, self.pendulum
        if self.mpc_steps == 0:
            self._write_references()
        for _ in range(nb_mpc_steps):
            self.solver.launch()
            rc = lib.mpcqp_wip_advance_batch(
                _dtype_code(p.dtype), self.states.data_ptr(), self.solver.U.data_ptr(), p.nb_variables,
                self.solver.status.data_ptr(), pend.nb_timesteps, pend.sampling_period, self.target_vel,
                pend.length, pend.GRAVITY, NB_SUBSTEPS, p.initial_state.data_ptr(), p.goal_state.data_ptr(),
                p.target_states.data_ptr(), p.batch_size, _stream_ptr())
            _capi.check(rc, "mpcqp_wip_advance_batch")
            self.failed += (self.solver.status != 0).sum()
            self.iters_total += self.solver.iters.sum()
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
