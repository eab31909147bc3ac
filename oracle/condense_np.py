"""ORACLE (test infrastructure, never shipped on the product path).

NumPy float64 restatement of the reference's *build half*:

    qpmpc/mpc_qp.py:39-163      MPCQP.__init__, update_cost_vector,
                                update_constraint_vector
    qpmpc/mpc_problem.py:316-335 MPCProblem.integrate

Pinned: every array this module produces is compared in tests/test_oracle.py
with tests/golden/*.npz, which tools/gen_golden.py captured from the real
reference (v3.1.0) imported in the build container.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The ``problem`` argument is duck-typed: any object exposing the
reference's ``get_*(k)`` accessors and attributes works (the product's host
container does).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np


class CondensedQP:
    """Plain record of what MPCQP keeps (mpc_qp.py:28-37)."""

    P: np.ndarray
    q: np.ndarray
    G: np.ndarray
    h: np.ndarray
    Phi: np.ndarray
    Psi: np.ndarray
    phi_last: np.ndarray
    psi_last: np.ndarray
    e: np.ndarray
    C_blocks: List[Optional[np.ndarray]]


def _terminal_cost_active(problem) -> bool:
    # mpc_problem.py:141-153 (weight set AND above 1e-10; goal must then exist)
    w = problem.terminal_cost_weight
    on = w is not None and w > 1e-10
    if on and problem.goal_state is None:
        raise ValueError("goal state undefined")
    return on


def _stage_cost_active(problem) -> bool:
    # mpc_problem.py:155-166
    w = problem.stage_state_cost_weight
    on = w is not None and w > 1e-10
    if on and problem.target_states is None:
        raise ValueError("target states undefined")
    return on


def cost_vector(cq: CondensedQP, problem) -> np.ndarray:
    """q of mpc_qp.py:129-149, including the partial fill of SURVEY quirk 2."""
    x0 = problem.initial_state
    q = np.zeros(cq.Psi.shape[1])
    try:
        if _terminal_cost_active(problem):
            c = cq.phi_last @ x0 - problem.goal_state  # :141
            q += problem.terminal_cost_weight * (c @ cq.psi_last)  # :142-144
        if _stage_cost_active(problem):
            c = cq.Phi @ x0 - problem.target_states  # :146
            q += problem.stage_state_cost_weight * (c @ cq.Psi)  # :147-149
    except ValueError:
        # MPCQP.__init__ swallows the ProblemDefinitionError (mpc_qp.py:119-122):
        # whatever was accumulated before the raise stays in q.
        pass
    return q


def constraint_vector(cq: CondensedQP, problem) -> np.ndarray:
    """h = e - C Phi x0 (mpc_qp.py:151-163), blockwise instead of block_diag.

    When every C_k is None the reference's block_diag produces an object array
    and the update raises; the intended semantics, h = e, is
    what is restated here.
    """
    x0 = problem.initial_state
    nx = cq.phi_last.shape[0]
    parts = []
    off = 0
    for k, Ck in enumerate(cq.C_blocks):
        mk = len(problem.get_ineq_vector(k))
        ek = cq.e[off: off + mk]
        if Ck is None:
            parts.append(ek)
        else:
            parts.append(ek - Ck @ (cq.Phi[k * nx: (k + 1) * nx] @ x0))
        off += mk
    return np.hstack(parts).astype(float)


def condense(problem) -> CondensedQP:
    """The loop of mpc_qp.py:53-114."""
    nu, nx, N = problem.input_dim, problem.state_dim, problem.nb_timesteps
    n = nu * N
    x0 = problem.initial_state
    if x0 is None:
        raise ValueError("initial state is undefined")  # :49-50
    phi = np.eye(nx)  # :53
    psi = np.zeros((nx, n))  # :54
    G_rows, h_rows, phis, psis, es, Cs = [], [], [], [], [], []
    for k in range(N):
        phis.append(phi)  # :60
        psis.append(psi)  # :61
        A_k = np.asarray(problem.get_transition_state_matrix(k))
        B_k = np.asarray(problem.get_transition_input_matrix(k))
        C_k = problem.get_ineq_state_matrix(k)
        D_k = problem.get_ineq_input_matrix(k)
        e_k = np.asarray(problem.get_ineq_vector(k))
        G_k = np.zeros((e_k.shape[0], n))  # :67
        h_k = e_k if C_k is None else e_k - (np.asarray(C_k) @ phi) @ x0  # :68-72
        cols = slice(k * nu, (k + 1) * nu)
        if D_k is not None:
            G_k[:, cols] = D_k  # :76
        if C_k is not None:
            G_k = G_k + np.asarray(C_k) @ psi  # :78
        G_rows.append(G_k)
        h_rows.append(h_k)
        phi = A_k @ phi  # :88
        psi = A_k @ psi  # :89
        psi[:, cols] = B_k  # :90
        es.append(e_k)
        Cs.append(None if C_k is None else np.asarray(C_k, dtype=float))
    cq = CondensedQP()
    cq.G = np.vstack(G_rows).astype(float)  # :93
    cq.h = np.hstack(h_rows).astype(float)  # :94
    cq.Phi = np.vstack(phis).astype(float)  # :95
    cq.Psi = np.vstack(psis).astype(float)  # :96
    cq.e = np.hstack(es).astype(float)  # :98
    cq.C_blocks = Cs
    P = problem.stage_input_cost_weight * np.eye(n)  # :99-101
    if problem.terminal_cost_weight is not None:
        P = P + problem.terminal_cost_weight * (psi.T @ psi)  # :102-103
    if problem.stage_state_cost_weight is not None:
        P = P + problem.stage_state_cost_weight * (cq.Psi.T @ cq.Psi)  # :104-105
    cq.P = P
    cq.phi_last = phi  # :113
    cq.psi_last = psi  # :114
    cq.q = cost_vector(cq, problem)  # :119-122
    return cq


def integrate(problem, x0: np.ndarray, U: np.ndarray) -> np.ndarray:
    """X[0]=x0, X[k+1] = A_k X[k] + B_k U[k]  (mpc_problem.py:316-335)."""
    N, nx = problem.nb_timesteps, problem.state_dim
    U = np.asarray(U, dtype=float).reshape(N, problem.input_dim)
    X = np.zeros((N + 1, nx))
    X[0] = x0
    for k in range(N):
        A_k = np.asarray(problem.get_transition_state_matrix(k))
        B_k = np.asarray(problem.get_transition_input_matrix(k))
        X[k + 1] = A_k @ X[k] + B_k @ U[k]
    return X
