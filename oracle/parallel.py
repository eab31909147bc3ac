"""ORACLE (test infrastructure): the all-C oracle on every host core.

Used only by bench.py's ``cpu_baseline`` leg (SURVEY.md 8d asks for a 1-thread and an
all-cores figure next to the GPU number). Workers are spawned processes: each one loads
liboracle.so through ctypes and solves its contiguous shard of the workload over and over
for a fixed wall-clock budget; the rate is problems solved by all workers / the budget.
Nothing under qpmpc_amd/ imports this module.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import time
from typing import Dict


def _worker(args):
    w, seconds = args
    import oracle  # noqa: F401  (spawned process: imports are its own)
    from oracle.capi import solve_workload

    batch = int(w["x0"].shape[0])
    solve_workload(w)  # warm-up: page in, build tables
    done = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        solve_workload(w)
        done += batch
    return done, time.perf_counter() - t0


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def all_cores_rate(w: Dict, seconds: float = 4.0, workers: int = 0) -> Dict:
    """Problems per second of ``workers`` processes (default: every CPU the process may run on), each
    solving the whole sample workload ``w`` repeatedly for ``seconds`` (one C call per pass, so the
    per-call Python overhead stays negligible)."""
    if workers <= 0:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
    jobs = [(w, seconds) for _ in range(workers)]
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        res = pool.map(_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    rate = sum(d / t for d, t in res)
    return {"value": rate, "cores": workers, "cpu_model": cpu_model(), "seconds_per_worker": seconds,
            "wall_s_including_spawn": wall}


def _solve_shard(w):
    from oracle.capi import solve_workload

    return solve_workload(w)


def solve_workload_parallel(w: Dict, shard, workers: int = 0):
    """``oracle.solve_workload`` over contiguous shards in spawned processes (tests that check the GPU
    against the oracle at sizes one core would need minutes for). Returns (U, lam, status, iters)."""
    import numpy as np

    if workers <= 0:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
    batch = int(w["x0"].shape[0])
    workers = max(1, min(workers, batch, 64))
    parts = [shard(w, r, workers) for r in range(workers)]
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_solve_shard, parts, chunksize=1)
    return tuple(np.concatenate([r[i] for r in res], axis=0) for i in range(4))
