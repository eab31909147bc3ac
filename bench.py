#!/usr/bin/env python3
"""Benchmark of the hot path: MPC QP builds+solves per second (batched).

A "step" is one fused build+solve of one batch of BASELINE.json's configs[1]:
4096 triple-integrator problems (nx=3, nu=1, N=16 -> n=16, m=32), float64,
heterogeneous operands (A_k, B_k, C_k, e_k stacked per problem and per step, a
real condense per problem), inputs resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run, one rank per GPU; the batch shards
by problem (each rank owns 4096 problems of its own, no data-path collective:
weak scaling). Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6  # MI355X vector/matrix fp64 peak (AMD spec; not in the guide)


def cpu_baseline(w, seconds: float = 12.0):
    """Oracle timed on the host, rank 0 only. Two figures:
    * "port": the reference's execution model -- one Python call per problem,
      NumPy condensing (oracle.condense_np, the restatement of mpc_qp.py) and a
      native dense active-set solve (what qpsolvers/quadprog does), 1 thread;
    * "port_c": the all-C oracle (condense + Goldfarb-Idnani), 1 thread.
    """
    import numpy as np

    import oracle
    from qpmpc_amd import MPCProblem

    batch = w["x0"].shape[0]
    t0 = time.perf_counter()
    done = 0
    b = 0
    while time.perf_counter() - t0 < seconds:
        p = MPCProblem(
            [w["A"][b, k] for k in range(w["N"])], [w["B"][b, k] for k in range(w["N"])],
            [w["C"][b, k] for k in range(w["N"])], None, [w["e"][b, k] for k in range(w["N"])],
            w["N"], w["wt"], w["wx"], w["wu"], initial_state=w["x0"][b], goal_state=w["goal"][b])
        U, st, _ = oracle.solve_mpc_like_reference(p)
        done += 1
        b = (b + 1) % batch
    t_ref = time.perf_counter() - t0
    t1 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t1 < max(2.0, seconds / 4):
        oracle.solve_workload(w)
        reps += 1
    t_c = time.perf_counter() - t1
    return {
        "value": done / t_ref,
        "unit": "problems/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{done} problems of the same batch, one Python call each (NumPy condense + C active-set), {t_ref:.1f} s",
        "port_c_value": reps * batch / t_c,
        "port_c_sample": f"all-C oracle, {reps} passes over the {batch}-problem batch, {t_c:.1f} s, 1 thread",
        "host_cpus": os.cpu_count(),
    }


def other_workloads():
    """Rates of the other BASELINE.json configurations on this GPU (rank 0, one GPU only; never `value`):
    a few launches each, HIP events on the launch stream. Parity for these is in tests/ (-m gpu)."""
    import numpy as np
    import torch

    from qpmpc_amd import PreparedSolve, SharedModel
    from qpmpc_amd import workloads as W
    from qpmpc_amd.closed_loop import LIPMWalkingLoop, WIPClosedLoop

    def rate(run, batch, reps):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return batch * reps / (e0.elapsed_time(e1) * 1e-3)

    out = {}
    w = W.wip_batch(1024)
    out["config3_wip_n50_fused_batch1024"] = rate(PreparedSolve(W.to_batch_problem(w)).launch, 1024, 20)
    rng = np.random.default_rng(1)
    loop = WIPClosedLoop(rng.standard_normal((1024, 4)) * np.array([0.05, 0.05, 0.1, 0.1]))
    out["config3_closed_loop_rebuild_every_step"] = rate(loop.step, 1024, 50)
    w = W.humanoid_batch(65536)
    bp = W.to_batch_problem(w)
    out["config4_humanoid_sweep_65536_fused"] = rate(PreparedSolve(bp).launch, 65536, 10)
    out["config4_humanoid_sweep_65536_shared_model"] = rate(SharedModel(bp).prepare(bp).launch, 65536, 10)
    w = W.synthetic_ltv_batch(1024)
    out["config5_synthetic_ltv_n256_m1024_f32_batch1024"] = rate(
        PreparedSolve(W.to_batch_problem(w, dtype=torch.float32)).launch, 1024, 5)
    walkers = LIPMWalkingLoop(4096, index=rng.integers(0, 8, 4096))
    out["lipm_walking_loops_4096"] = rate(walkers.step, 4096, 100)
    return {k: float(v) for k, v in out.items()} | {"unit": "problems/s (builds+solves/s for the loops)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed steps (one step = 58 us: short runs end before the GPU reaches its sustained clocks)")
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=4096, help="problems per GPU (configs[1]: 4096)")
    ap.add_argument("--shared-lti", action="store_true", help="stride-0 operands (not the headline mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--spinup", type=float, default=0.25, help="seconds of untimed launches before the warm-up (clock ramp)")
    ap.add_argument("--no-extras", action="store_true", help="skip the short runs of the other configurations")
    ap.add_argument("--no-overlap", action="store_true", help="skip the extra two-streams-in-flight measurement")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from qpmpc_amd import PreparedSolve
    from qpmpc_amd import workloads as W

    # each rank owns its own shard of the sweep (independent problems, no exchange)
    w = W.triple_integrator_batch(args.batch, seed=20250614 + rank, heterogeneous=not args.shared_lti)
    bp = W.to_batch_problem(w)
    run = PreparedSolve(bp)
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    # Untimed device spin-up before the W warm-up steps: one step is ~55 us, so a short (W, K) would be over
    # before the GPU has left its idle clocks (57 us/step measured that way against 54 us sustained).
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup:
        for _ in range(100):
            run.launch()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        run.launch()
    torch.cuda.synchronize()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # same stream the kernels are enqueued on (torch's current stream)
    for _ in range(args.steps):
        run.launch()
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch duration, HIP events

    # Extra (not `value`): two independent batches in flight on two streams, the way a
    # server or a set of unrelated control loops would submit work. It measures how much
    # of a single-stream step is ramp/tail (the step ends with its slowest wavefront).
    overlap = None
    if not args.no_overlap:
        w2 = W.triple_integrator_batch(args.batch, seed=30250614 + rank, heterogeneous=not args.shared_lti)
        runs = [run, PreparedSolve(W.to_batch_problem(w2))]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        for k in range(2 * args.warmup):
            runs[k % 2].launch(stream=streams[k % 2])
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for k in range(args.steps):
            runs[k % 2].launch(stream=streams[k % 2])
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        barrier()
        t2 = torch.tensor([el2], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        overlap = {"streams": 2, "value": args.batch * world * args.steps / float(t2.item()), "unit": "problems/s",
                   "ms_per_step": float(t2.item()) / args.steps * 1e3,
                   "note": "independent batches in flight on 2 HIP streams; not the headline value"}

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    solved = (run.status == 0).sum().to(torch.float64).reshape(1)
    it_sum = run.iters.sum().to(torch.float64).reshape(1)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(solved, op=dist.ReduceOp.SUM)
        dist.all_reduce(it_sum, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_problems = args.batch * world * args.steps

    if rank == 0:
        import oracle

        U = run.U.cpu().numpy()
        Uo, _, sto, _ = oracle.solve_workload(w)
        ok = sto == 0
        err = np.abs(U[ok] - Uo[ok])
        # KKT residuals of the kernel's own (u, lambda) on a sample, against the oracle-built QP
        from qpmpc_amd import MPCProblem

        chk = PreparedSolve(bp, return_multipliers=True)
        chk.launch()
        torch.cuda.synchronize()
        lam = chk.lam.cpu().numpy()
        kkt = {"stationarity": 0.0, "primal": 0.0, "complementarity": 0.0, "dual": 0.0}
        for b in range(0, args.batch, max(1, args.batch // 128)):
            if not ok[b]:
                continue
            p = MPCProblem(
                [w["A"][b, k] for k in range(w["N"])] if not args.shared_lti else w["A"],
                [w["B"][b, k] for k in range(w["N"])] if not args.shared_lti else w["B"],
                [w["C"][b, k] for k in range(w["N"])] if not args.shared_lti else w["C"], None,
                [w["e"][b, k] for k in range(w["N"])] if not args.shared_lti else w["e"],
                w["N"], w["wt"], w["wx"], w["wu"], initial_state=w["x0"][b], goal_state=w["goal"][b])
            cq = oracle.condense(p)
            ub, lb = U[b], lam[b]
            slack = cq.h - cq.G @ ub
            kkt["stationarity"] = max(kkt["stationarity"], float(np.abs(cq.P @ ub + cq.q + cq.G.T @ lb).max()))
            kkt["primal"] = max(kkt["primal"], float(np.maximum(-slack, 0.0).max()))
            kkt["dual"] = max(kkt["dual"], float(np.maximum(-lb, 0.0).max()))
            kkt["complementarity"] = max(kkt["complementarity"], float(np.abs(lb * slack).max()))
        nx, nu, N, mk = 3, 1, 16, 2
        n, m = N * nu, N * mk
        bytes_per_problem = W.algorithmic_bytes_per_problem(w)
        mean_iters = float(it_sum.item()) / (args.batch * world)
        flops_per_problem = W.algorithmic_build_flops(nx, nu, N, mk, False, True) + W.algorithmic_solve_flops(n, m, mean_iters)
        kernel_s = kernel_ms * 1e-3
        achieved_gbs = bytes_per_problem * args.batch / kernel_s / 1e9
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        traffic = None
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "MPC QP builds+solves/sec (batched)",
            "value": total_problems / elapsed,
            "unit": "problems/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "spinup_s": args.spinup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"batch={args.batch} triple-integrator N=16 (nx=3 nu=1, n=16 m=32), "
                            + ("shared LTI operands (stride 0)" if args.shared_lti else "heterogeneous per-problem LTV operands")
                            + ", fp64, fused condense + dual active-set solve, one launch per step",
                "batch_per_gpu": args.batch,
                "parallelism": f"batch-sharded x{world}, no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "mpcqp fused build+solve",
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_problem": bytes_per_problem,
                "algorithmic_flops_per_problem": flops_per_problem,
                "achieved_tflops_f64": flops_per_problem * args.batch / kernel_s / 1e12,
                "note": "latency-bound: 4096 problems x 2.7 KB is 11 MB per launch; the serial "
                        "active-set chain per problem, not HBM or FP64 throughput, sets the time",
            },
            "accuracy": {
                "max_abs_err_vs_oracle": float(err.max()),
                "p99_abs_err_vs_oracle": float(np.quantile(err.max(axis=1), 0.99)),
                "max_kkt_residuals_sample": kkt,
                "max_rel_err_vs_oracle": float((err / np.maximum(1.0, np.abs(Uo[ok]).max(axis=1, keepdims=True))).max()),
                "solved_frac": float(solved.item()) / (args.batch * world),
                "mean_iters": mean_iters,
            },
        }
        if overlap is not None:
            out["overlap_2_streams"] = overlap
        if not args.no_extras and world == 1:
            try:
                out["other_workloads"] = other_workloads()
            except Exception as exc:  # never at the expense of the headline line
                out["other_workloads"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
