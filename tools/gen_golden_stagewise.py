#!/usr/bin/env python3
"""Golden fixtures for the stage-wise solver: tests/golden/stagewise_*.npz.

Runs only in the build container (imports the REAL reference from /root/reference through the data-only
qpsolvers stand-in of tools/gen_golden.py). For long horizons the reference still BUILDS the dense
condensed QP (MPCQP, qpmpc/mpc_qp.py:39-122: P is 8 MB at N = 1024); its minimiser is then obtained
independently of this repository's solvers -- an exact active-set fixing loop in NumPy/SciPy on the
reference-built (P, q, G, h): Cholesky of P, Schur complement of the active rows, drop the most negative
multiplier / add the most violated row until the KKT conditions hold -- and certified by its KKT residuals,
which are stored. P >= w_u I > 0 makes the minimiser unique (mpc_problem.py:104-107).

Only data is written: the problem's operands and states, U_star, the active rows and multipliers, the
residuals. The condensed matrices themselves are not stored (they are what the stage-wise path avoids).
"""
from __future__ import annotations

import os
import sys

import numpy as np
from scipy.linalg import cho_factor, cho_solve

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as GG  # noqa: E402  (installs the qpsolvers stand-in and imports the reference)
from gen_golden import MPCQP, MPCProblem  # noqa: E402

OUT = GG.OUT


def exact_minimiser(P, q, G, h, max_rounds=5000):
    n, m = P.shape[0], G.shape[0]
    cf = cho_factor(P)
    Kq = cho_solve(cf, q)
    scale = 1.0 + np.abs(h)
    act: list = []
    Y = np.zeros((n, 0))  # P^-1 G_a' by column
    for _ in range(max_rounds):
        if act:
            Ga = G[act]
            S = Ga @ Y
            lam = np.linalg.solve(S, -(h[act] + Ga @ Kq))
            x = -Kq - Y @ lam
        else:
            lam, x = np.zeros(0), -Kq
        if len(lam) and lam.min() < -1e-12 * (1.0 + np.abs(lam).max()):
            j = int(np.argmin(lam))
            act.pop(j)
            Y = np.delete(Y, j, axis=1)
            continue
        viol = (G @ x - h) / scale
        viol[act] = -np.inf
        worst = int(np.argmax(viol))
        if viol[worst] > 1e-12:
            act.append(worst)
            Y = np.hstack([Y, cho_solve(cf, G[worst])[:, None]])
            continue
        break
    else:
        raise RuntimeError("active-set fixing loop did not settle")
    lam_full = np.zeros(m)
    lam_full[act] = lam
    kkt = {
        "stationarity": float(np.abs(P @ x + q + G.T @ lam_full).max()),
        "primal": float(np.maximum(G @ x - h, 0.0).max()),
        "dual": float(np.maximum(-lam_full, 0.0).max()),
        "complementarity": float(np.abs(lam_full * (G @ x - h)).max()),
    }
    return x, lam_full, np.array(sorted(act), dtype=np.int64), kkt


def triple_long(N, dt, goal, vmax, wu=1e-4, wx=1.0, wt=10.0, amax=3.0, x0=(0.0, 0.0, 0.0)):
    """Triple integrator of examples/triple_integrator.py:15-26 over a long horizon: acceleration box, optional
    velocity box, stage + terminal tracking of `goal`."""
    A = np.array([[1.0, dt, dt * dt / 2.0], [0.0, 1.0, dt], [0.0, 0.0, 1.0]])
    B = np.array([[dt ** 3 / 6.0], [dt * dt / 2.0], [dt]])
    rows, e = [[0.0, 0.0, 1.0], [0.0, 0.0, -1.0]], [amax, amax]
    if vmax is not None:
        rows += [[0.0, 1.0, 0.0], [0.0, -1.0, 0.0]]
        e += [vmax, vmax]
    p = MPCProblem(A, B, np.array(rows), None, np.array(e), N, wt, wx, wu, initial_state=np.array(x0),
                   goal_state=np.array(goal))
    p.update_target_states(np.tile(np.array(goal), N))
    return p


def record(name, p):
    qp = MPCQP(p)  # the reference's own dense build
    x, lam, act, kkt = exact_minimiser(qp.P, qp.q, qp.G, qp.h)
    sc = 1.0 + np.abs(qp.q).max()
    assert kkt["stationarity"] < 1e-8 * sc and kkt["primal"] < 1e-9 and kkt["dual"] < 1e-9, (name, kkt)
    N = p.nb_timesteps
    rec = dict(A=np.asarray(p.transition_state_matrix), B=np.asarray(p.transition_input_matrix),
               C=np.asarray(p.ineq_state_matrix), e=np.asarray(p.ineq_vector), nb_timesteps=np.array(N),
               terminal_cost_weight=np.array(p.terminal_cost_weight), stage_state_cost_weight=np.array(p.stage_state_cost_weight),
               stage_input_cost_weight=np.array(p.stage_input_cost_weight), initial_state=np.asarray(p.initial_state),
               goal_state=np.asarray(p.goal_state), target_states=np.asarray(p.target_states),
               U_star=x, active_set=act, lambda_active=lam[act], obj_star=np.array(0.5 * x @ qp.P @ x + qp.q @ x),
               cond_P=np.array(np.linalg.cond(qp.P)))
    for k, v in kkt.items():
        rec["kkt_" + k] = np.array(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: N={N} n={qp.P.shape[0]} m={qp.G.shape[0]} active={len(act)} |U|max={np.abs(x).max():.3f} "
          f"cond(P)={rec['cond_P']:.2e} kkt={kkt}")


def main():
    record("stagewise_triple_n64", triple_long(64, 1 / 16, (4.0, 0.0, 0.0), None))
    record("stagewise_triple_n256", triple_long(256, 1 / 32, (4.0, 0.0, 0.0), 1.5))
    record("stagewise_triple_n1024", triple_long(1024, 1 / 64, (4.0, 0.0, 0.0), 1.5))
    record("stagewise_triple_n1024_b", triple_long(1024, 1 / 64, (-2.5, 0.0, 0.0), 1.0, x0=(0.5, 0.8, -1.0)))


if __name__ == "__main__":
    main()
