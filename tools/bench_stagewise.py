#!/usr/bin/env python3
"""Dev/bench: the stage-wise formulation over horizons, next to the condensed path where it exists."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import BatchMPCProblem, solve_mpc_batch
from qpmpc_amd import workloads as W

def long_batch(B, N, dt, seed=11):
    A = np.array([[1.0, dt, dt * dt / 2.0], [0.0, 1.0, dt], [0.0, 0.0, 1.0]])
    Bm = np.array([[dt ** 3 / 6.0], [dt * dt / 2.0], [dt]])
    C = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0], [0.0, -1.0, 0.0]])
    e = np.array([3.0, 3.0, 1.5, 1.5])
    rng = np.random.default_rng(seed)
    x0 = np.stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B), rng.uniform(-1.5, 1.5, B)], 1)
    goal = np.stack([rng.uniform(-4, 4, B), np.zeros(B), np.zeros(B)], 1)
    return BatchMPCProblem(A, Bm, C, None, e, N, 10.0, 1.0, 1e-4, x0, goal_state=goal, target_states=np.tile(goal, (1, N)))

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out

if __name__ == "__main__":
    for B, N, dt in ((4096, 16, 1 / 16), (4096, 64, 1 / 16), (4096, 256, 1 / 32), (4096, 1024, 1 / 64), (512, 4096, 1 / 128)):
        bp = long_batch(B, N, dt)
        ts, plan = timeit(lambda: solve_mpc_batch(bp, formulation="stagewise"))
        st = plan.status.cpu().numpy(); it = plan.iters.float().mean().item()
        line = f"N={N:5d} batch {B}: stage-wise {ts*1e3:9.2f} ms = {B/ts/1e3:9.1f} k problems/s  (solved {np.mean(st==0):.3f}, mean iters {it:.1f}, max {plan.iters.max().item()})"
        if N * 1 <= 256:
            td, pd = timeit(lambda: solve_mpc_batch(bp))
            err = float(((pd.U - plan.U).abs().max(dim=1).values / pd.U.abs().max(dim=1).values.clamp(min=1.0)).max())
            line += f" | condensed {td*1e3:8.2f} ms = {B/td/1e3:9.1f} k/s, max rel diff {err:.1e}"
        print(line, flush=True)
    w = W.triple_integrator_batch(4096); bp = W.to_batch_problem(w)
    ts, plan = timeit(lambda: solve_mpc_batch(bp, formulation="stagewise")); td, pd = timeit(lambda: solve_mpc_batch(bp))
    print(f"config 2 (N=16, 4096): stage-wise {ts*1e3:.3f} ms vs condensed {td*1e3:.3f} ms")
