"""Wheeled inverted pendulum: problem generator and plant for BASELINE config 3.

Same model as the reference's ``qpmpc.systems.WheeledInvertedPendulum``
(qpmpc/systems/wheeled_inverted_pendulum.py): state ``[r, theta, r', theta']``,
input = ground acceleration. The MPC model is the exact zero-order-hold
discretisation (:82-103); the plant step is a second-order Taylor expansion of
the nonlinear dynamics (:127-160). ``integrate_batch`` is the torch version of
the plant step for batched closed loops on the device.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from ..mpc_problem import MPCProblem


class WheeledInvertedPendulum:
    GRAVITY: float = 9.81  # m/s^2
    INPUT_DIM: int = 1
    STATE_DIM: int = 4

    def __init__(self, length: float = 0.6, max_ground_accel: float = 10.0,
                 nb_timesteps: int = 12, sampling_period: float = 0.1):
        self.length = length
        self.max_ground_accel = max_ground_accel
        self.nb_timesteps = nb_timesteps
        self.sampling_period = sampling_period

    @property
    def omega(self) -> float:
        """Natural frequency sqrt(g / l) of the pendulum."""
        return np.sqrt(self.GRAVITY / self.length)

    @property
    def horizon_duration(self) -> float:
        return self.sampling_period * self.nb_timesteps

    def discretized_dynamics(self):
        """(A, B) of the exact discretisation over one sampling period."""
        T, w, g = self.sampling_period, self.omega, self.GRAVITY
        ch, sh = np.cosh(T * w), np.sinh(T * w)
        A = np.array([
            [1.0, 0.0, T, 0.0],
            [0.0, ch, 0.0, sh / w],
            [0.0, 0.0, 1.0, 0.0],
            [0.0, w * sh, 0.0, ch],
        ])
        B = np.array([[T**2 / 2.0], [-ch / g + 1.0 / g], [T], [-w * sh / g]])
        return A, B

    def build_mpc_problem(self, stage_input_cost_weight: float = 1e-3,
                          stage_state_cost_weight: Optional[float] = None,
                          terminal_cost_weight: Optional[float] = 1.0) -> MPCProblem:
        """LTI problem with the input box |u| <= max_ground_accel (D = [1; -1])."""
        A, B = self.discretized_dynamics()
        return MPCProblem(
            transition_state_matrix=A,
            transition_input_matrix=B,
            ineq_state_matrix=None,
            ineq_input_matrix=np.array([[1.0], [-1.0]]),
            ineq_vector=np.array([self.max_ground_accel, self.max_ground_accel]),
            nb_timesteps=self.nb_timesteps,
            terminal_cost_weight=terminal_cost_weight,
            stage_state_cost_weight=stage_state_cost_weight,
            stage_input_cost_weight=stage_input_cost_weight,
        )

    def integrate(self, state: np.ndarray, ground_accel, dt: float) -> np.ndarray:
        """One plant step of duration ``dt`` under a constant ground acceleration."""
        r, th, rd, thd = (float(v) for v in np.asarray(state).ravel())
        a = float(np.asarray(ground_accel).ravel()[0])
        thdd = self.omega ** 2 * (math.sin(th) - (a / self.GRAVITY) * math.cos(th))
        return np.array([
            r + dt * (rd + dt * (a / 2)),
            th + dt * (thd + dt * (thdd / 2)),
            rd + dt * a,
            thd + dt * thdd,
        ])

    def integrate_batch(self, states, ground_accel, dt: float):
        """Torch plant step for [B, 4] states and [B] accelerations (device-resident)."""
        import torch

        r, th, rd, thd = states.unbind(-1)
        a = ground_accel.reshape(-1)
        thdd = self.omega ** 2 * (torch.sin(th) - (a / self.GRAVITY) * torch.cos(th))
        return torch.stack(
            [r + dt * (rd + dt * (a / 2)), th + dt * (thd + dt * (thdd / 2)), rd + dt * a, thd + dt * thdd],
            dim=-1,
        )

    def target_states(self, state: np.ndarray, target_vel: float) -> np.ndarray:
        """(N+1)*nx reference ramp of examples/wheeled_inverted_pendulum.py:65-83."""
        nx, T = self.STATE_DIM, self.sampling_period
        out = np.zeros((self.nb_timesteps + 1, nx))
        out[:, 0] = state[0] + np.arange(self.nb_timesteps + 1) * T * target_vel
        out[:, 2] = target_vel
        return out.ravel()
