#!/usr/bin/env python3
"""Dev probe: one round of tools/stress_tight.py general / tools/stress_general.py, re-generated from its seed: stop reasons of the
general stage-wise kernel, and for the solved problems the KKT quality of the kernel's plan and of the oracle's.
usage: probe_general_case.py tight|general SEED ROUND"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import oracle
from oracle import condense_np
from qpmpc_amd import solve_mpc_batch, workloads as W
from stress_stagewise import random_ltv
kind, seed, rnd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for it in range(rnd + 1):
    if kind == "tight":
        nx, nu = int(rng.integers(17, 33)), int(rng.integers(1, 9))
        N = int(rng.integers(20, 41)); mk = int(rng.integers(4, 7))
        w = random_ltv(rng, 8, nx, nu, N, mk, 0.5)
        w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
    else:
        import stress_general as SG  # noqa
        raise SystemExit("use tools/stress_general.py's generator")
buf = torch.zeros(8 * 16, dtype=torch.int64, device="cuda")
plan = solve_mpc_batch(W.to_batch_problem(w), probe=buf, return_multipliers=True); torch.cuda.synchronize()
t = buf.view(8, 16).cpu()
print("dims", nx, nu, N, mk, "n", N * nu, "m", N * mk)
print("stop reasons (1 iteration limit, 2 verification passes, 3 verification failed, 4 re-entered four times, 5 no step)", t[:, 8].tolist())
print("slots / worst |rho|/thr / zero multipliers at the pass limit (or 100+pass: rebuild failed)", t[:, 9:12].tolist())
Uo, lamo, sto, ito = oracle.solve_workload(w)
st = plan.status.cpu().numpy()
print("gpu status", st.tolist(), "iters", plan.iters.cpu().tolist())
print("ora status", sto.tolist(), "iters", ito.tolist())
U, lam = plan.U.cpu().numpy(), plan.multipliers.cpu().numpy()
for b in range(8):
    if st[b] != 0 or sto[b] != 0:
        continue
    P, q, G, h = condense_np.condense_problem(W.problem_from_workload(w, b))
    def kkt(u, l):
        return np.abs(P @ u + q + G.T @ l).max(), max(0.0, (G @ u - h).max()), float(np.abs(l * (G @ u - h)).max()), 0.5 * u @ P @ u + q @ u
    ku, ko = kkt(U[b], lam[b]), kkt(Uo[b], lamo[b])
    print(f"problem {b}: rel err {np.abs(U[b]-Uo[b]).max()/max(1,np.abs(Uo[b]).max()):.2e} | gpu: stationarity {ku[0]:.1e} infeas {ku[1]:.1e} compl {ku[2]:.1e} obj {ku[3]:.12e}"
          f" | oracle: {ko[0]:.1e} {ko[1]:.1e} {ko[2]:.1e} obj {ko[3]:.12e} | active gpu {int((lam[b]>0).sum())} oracle {int((lamo[b]>0).sum())} max lam {lam[b].max():.2e}")
