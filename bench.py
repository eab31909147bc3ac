#!/usr/bin/env python3
"""Benchmark of the hot path: MPC QP builds+solves per second (batched).

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

A "step" is one pass of the hot path over one batch of synthetic input, inputs resident in
HBM when the timed region starts. ``--config`` picks the BASELINE.json configuration that is
timed (configs[1..4]; configs[0] is the reference's single CPU problem):

  2 (default, the headline)  batch 4096 triple integrator N=16 per GPU, float64, heterogeneous
                             per-problem LTV operands, one fused build+solve launch; weak scaling.
  3  wheeled inverted pendulum N=50 (T = 0.024 s), 1024 receding-horizon closed loops per GPU,
     float64, LTV lists: one MPC period (fused build+solve rebuilt every step like the reference,
     + the nonlinear plant for 15 sub-steps) = 1024 builds+solves; weak scaling.
  4  humanoid one-step N=16, 65,536-state sweep, float64, STRONG-sharded over the N GPUs
     (8192 per GPU at N=8); the all_gather of U/status is timed separately (``all_gather_ms``).
  5  synthetic LTV nx=12 nu=4 N=64 (n=256, m=1024), float32, batch 8192 STRONG-sharded
     (1024 per GPU at N=8); uncondensed stage-wise active-set kernels (no Gram is formed);
     ``roofline.bound`` = hbm on the problem's inputs + outputs.

N > 1 runs one rank per GPU over RCCL: either launched by ``torch.distributed.run`` (the driver's way) or,
when started as a plain ``python bench.py --gpus N``, by re-executing itself through it. Problems never
interact, so the data path has no collective. Rank 0 prints ONE JSON line.

Without a GPU (build container) the same command runs the RANK LOGIC ONLY over gloo with a runner that
computes nothing: the record then carries ``"value": null`` and ``"dry_run"`` -- it is not a measurement.
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

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6   # MI355X vector/matrix fp64 peak (AMD spec; not in the guide)
FP32_PEAK_TFLOPS = 157.3  # MI355X f32 vector = f32-input MFMA peak (MI355X_MICROARCH.md)
PERIODS_PER_LAUNCH = 50   # config 3: consecutive control periods of the closed loops per launch

CONFIGS = {
    2: dict(batch=4096, scaling="weak", dtype="f64",
            workload="batch={b} triple-integrator N=16 (nx=3 nu=1, n=16 m=32), heterogeneous per-problem LTV operands, "
                     "fp64, fused condense + dual active-set solve, one launch per step"),
    3: dict(batch=1024, scaling="weak", dtype="f64",
            workload="{b} wheeled-inverted-pendulum receding-horizon loops, N=50 T=0.024 s (nx=4 nu=1, n=50 m=100), LTV "
                     "lists, fp64; one step = one MPC period: fused build+solve rebuilt every step (the factor by a second "
                     "wavefront one period ahead) + plant (15 sub-steps); up to %d consecutive periods per launch" % PERIODS_PER_LAUNCH),
    4: dict(batch=65536, scaling="strong", dtype="f64",
            workload="humanoid one-step (LIPM) N=16 (n=16 m=32), {b}-state sweep strong-sharded over the GPUs, fp64, "
                     "fused build+solve; U/status all_gather timed separately"),
    5: dict(batch=8192, scaling="strong", dtype="f32",
            workload="synthetic LTV nx=12 nu=4 N=64 (n=256 m=1024) with input and state boxes, batch {b} strong-sharded "
                     "over the GPUs, fp32, uncondensed stage-wise active-set kernel (Riccati factor + MFMA sweeps), one problem per wavefront"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 2000 for the 40-us steps of configs 2/4, fewer for 3/5)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="override the configuration's batch (per GPU if weak, total if strong)")
    ap.add_argument("--shared-lti", action="store_true", help="config 2 with stride-0 operands (not the headline mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--spinup", type=float, default=0.25, help="seconds of untimed launches before the warm-up (clock ramp)")
    ap.add_argument("--no-extras", action="store_true", help="skip the short runs of the other configurations")
    ap.add_argument("--no-overlap", action="store_true", help="skip the extra two-streams-in-flight measurement")
    args = ap.parse_args(argv)
    defaults = {2: (2000, 200), 3: (100, 20), 4: (200, 20), 5: (20, 3)}
    if args.steps is None:
        args.steps = defaults[args.config][0]
    if args.warmup is None:
        args.warmup = defaults[args.config][1]
    return args


# ---------------------------------------------------------------------------------- workloads
def local_workload(config: int, rank: int, world: int, batch=None, shared_lti: bool = False):
    """(this rank's workload dict, problems of the whole job per step, problems of this rank per step)."""
    from qpmpc_amd import workloads as W
    from qpmpc_amd.distributed import shard_range, shard_workload

    spec = CONFIGS[config]
    b = int(batch or spec["batch"])
    if spec["scaling"] == "weak":  # every rank owns b problems of its own (independent seeds)
        if config == 2:
            w = W.triple_integrator_batch(b, seed=20250614 + rank, heterogeneous=not shared_lti)
        else:
            w = W.wip_batch(b, seed=1 + rank)
        return w, b * world, b
    # strong: ONE global problem set, whatever the number of ranks; this rank takes its contiguous slice
    if config == 4:
        w = shard_workload(W.humanoid_batch(b, seed=2), rank, world)
    else:
        lo, hi = shard_range(b, rank, world)
        w = W.synthetic_ltv_batch_slice(lo, hi, seed=3)
    return w, b, int(w["x0"].shape[0])


class _Runner:
    """One step of the configured hot path on this rank's GPU."""

    def __init__(self, config: int, w, device="cuda"):
        import numpy as np
        import torch

        from qpmpc_amd import PreparedSolve
        from qpmpc_amd import workloads as W

        self.config = config
        if config == 3:
            from qpmpc_amd.closed_loop import WIPClosedLoop

            # the factor is rebuilt every period like the reference's solve_mpc does (solve_mpc.py:42), by a second wavefront
            # working one period ahead (MPCQP_OPT_PIPELINE_FACTOR); other_workloads has the unpipelined rate
            # ... and up to PERIODS_PER_LAUNCH consecutive periods run in one launch (mpcqp_wip_periods_batch: the wavefront
            # that solved period t carries on with t + 1; same trajectories bit for bit, no dispatch gap between periods)
            self.loop = WIPClosedLoop(np.asarray(w["x0"]), pipeline_factor=True, periods_per_launch=PERIODS_PER_LAUNCH)
            self.solver = self.loop.solver
            self.launch = lambda stream=None: self.loop.step()
            self.launch_steps = lambda n: self.loop.step(n)  # K steps = K periods, in ceil(K / PERIODS_PER_LAUNCH) launches
            # the timed region is ONE EPISODE from the random initial states ("x0 ~ N(0, diag(.05,.05,
            # .1,.1)^2), ... repeat >= 100 MPC steps"): warm-up and spin-up periods must not leave the loops in
            # their constraint-free steady state
            self.reset = lambda: self.loop.reset(np.asarray(w["x0"]))
        else:
            self.bp = W.to_batch_problem(w, dtype=torch.float32 if config == 5 else None)
            self.solver = PreparedSolve(self.bp)
            self.launch = self.solver.launch

    @property
    def U(self):
        return self.solver.U

    @property
    def status(self):
        return self.solver.status

    @property
    def iters(self):
        return self.solver.iters


class _DryRunner:
    """No-GPU stand-in for ``_Runner`` (``main`` picks it when torch sees no device): computes NOTHING -- zero
    plans, status 0 -- so that argument handling, sharding, reductions, the gather and the JSON record of an
    N-rank launch can be exercised over gloo. ``main`` nulls ``value`` of such a run."""

    def __init__(self, config, w, device="cpu"):
        import torch

        b, n = int(w["x0"].shape[0]), int(w["N"]) * int(w["B"].shape[-1])
        self.U = torch.zeros((b, n), dtype=torch.float64)
        self.status = torch.zeros((b,), dtype=torch.int32)
        self.iters = torch.zeros((b,), dtype=torch.int32)

    def launch(self, stream=None):
        pass


class _Clock:
    """Timing of K launches: HIP events on the launch stream (torch's current stream) + a host clock
    around a full synchronisation. On a CPU device (the gloo test of this file's rank logic) both are the
    host clock."""

    def __init__(self, device: str):
        self.cuda = device == "cuda"

    def sync(self):
        if self.cuda:
            import torch

            torch.cuda.synchronize()

    def time(self, launch, steps: int, launch_steps=None):
        import torch

        self.sync()
        if self.cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if self.cuda:
            e0.record()
        if launch_steps is not None:
            launch_steps(steps)
        else:
            for _ in range(steps):
                launch()
        if self.cuda:
            e1.record()
        self.sync()
        wall = time.perf_counter() - t0
        return wall, (e0.elapsed_time(e1) * 1e-3 if self.cuda else wall)


# ---------------------------------------------------------------------------------- the measurement
def run_bench(args, rank: int, world: int, dist=None, make_runner=_Runner, device: str = "cuda"):
    """Everything between "process group is up" and "rank 0 has the JSON record". ``make_runner`` and
    ``device`` exist so that tests can drive this rank logic on CPU (gloo) with the solve stubbed."""
    import torch

    spec = CONFIGS[args.config]
    clock = _Clock(device)
    w, total_per_step, local_per_step = local_workload(args.config, rank, world, args.batch, args.shared_lti)
    run = make_runner(args.config, w, device)
    clock.sync()

    def barrier():
        if dist is not None:
            dist.barrier()

    def allreduce(x, op):
        if dist is not None:
            dist.all_reduce(x, op=getattr(dist.ReduceOp, op))
        return x

    dev = torch.device("cuda", torch.cuda.current_device()) if device == "cuda" else torch.device("cpu")
    # Untimed device spin-up before the W warm-up steps: a step of configs 2/4 is ~40 us, so a short (W, K)
    # would be over before the GPU has left its idle clocks.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup:
        for _ in range(20):
            run.launch()
        clock.sync()
    # the W warm-up steps go through the same region code as the timed ones (events recorded around them, full
    # synchronisation behind them): the timed region then meets no first-use cost of the event and synchronisation paths
    if args.warmup > 0 and getattr(run, "launch_steps", None) is None:
        clock.time(run.launch, args.warmup)
    else:
        for _ in range(args.warmup):
            run.launch()
    clock.sync()
    barrier()
    # A real barrier (RCCL) keeps the host busy for a few hundred microseconds during which the GPU idles and its clocks
    # drop: the first steps after it ran 27.6-29.9 us instead of 26.5 (one rank under torch.distributed.run, 20 steps).
    # Up to 20 ms of UNTIMED steps after the barrier (as the spin-up before the warm-up); the timed region still starts
    # with its own synchronisation.
    rewarm = 0
    t_spin = time.perf_counter()
    while dist is not None and time.perf_counter() - t_spin < min(args.spinup, 0.02):
        for _ in range(8):
            run.launch()
        rewarm += 8
        clock.sync()
    if hasattr(run, "reset"):
        run.reset()
    elapsed, kernel_s = clock.time(run.launch, args.steps, getattr(run, "launch_steps", None))
    barrier()
    kernel_ms = kernel_s / args.steps * 1e3  # average duration of one step's launches, HIP events

    # whole-job reductions: MAX of the elapsed time, SUMs of the counts
    t = allreduce(torch.tensor([elapsed], dtype=torch.float64, device=dev), "MAX")
    if hasattr(run, "loop"):  # closed loop: counts over the whole timed episode
        st = run.loop.stats()
        c = [st["builds_and_solves"] - st["failed"], st["mean_iters"] * st["builds_and_solves"], st["builds_and_solves"]]
        counts = allreduce(torch.tensor(c, dtype=torch.float64, device=dev), "SUM")
    else:
        counts = allreduce(torch.stack([(run.status == 0).sum(), run.iters.sum(),
                                        torch.tensor(run.status.numel(), device=run.status.device)]).to(torch.float64).to(dev), "SUM")
    elapsed = float(t.item())

    # strong-scaling configurations hand the whole batch to every rank afterwards: all_gather, timed apart
    gather_ms = None
    if spec["scaling"] == "strong":
        from qpmpc_amd.distributed import gather_batch

        for _ in range(2):
            gather_batch(run.U, total_per_step)
        clock.sync()
        barrier()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            U_all = gather_batch(run.U, total_per_step)
            st_all = gather_batch(run.status, total_per_step)
        clock.sync()
        tg = allreduce(torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev), "MAX")
        gather_ms = float(tg.item()) * 1e3
        assert U_all.shape[0] == total_per_step and st_all.shape[0] == total_per_step

    # Extra (config 2, not `value`): two independent batches in flight on two streams
    overlap = None
    if args.config == 2 and not args.no_overlap and device == "cuda":
        overlap = _overlap_two_streams(args, run, rank, world, barrier, allreduce, dev)

    if rank != 0:
        return None
    problems = float(counts[2].item())
    out = {
        "metric": "MPC QP builds+solves/sec (batched)",
        "value": total_per_step * args.steps / elapsed,
        "unit": "problems/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "spinup_s": args.spinup,
        "rewarm_after_barrier": rewarm,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": spec["scaling"],
        "vs_baseline": None,
        "dtype": spec["dtype"],
        "data": "synthetic",
        "config": {
            "workload": (spec["workload"].format(b=args.batch or spec["batch"])
                         + (" [shared LTI operands, stride 0]" if args.shared_lti else "")),
            "baseline_config_index": args.config - 1,
            "problems_per_step": total_per_step,
            "problems_per_gpu_per_step": local_per_step,
            "parallelism": f"batch-sharded x{world}, no data-path collective",
        },
        "solved_frac": float(counts[0].item()) / max(problems, 1.0),
        "mean_iters": float(counts[1].item()) / max(problems, 1.0),
    }
    if gather_ms is not None:
        out["all_gather_ms"] = gather_ms
        out["all_gather_note"] = "U [B, n] + status [B] to every rank (padded all_gather, qpmpc_amd.distributed.gather_batch); outside the timed region"
    if overlap is not None:
        out["overlap_2_streams"] = overlap
    if device == "cuda":
        out["roofline"] = _roofline(args, w, local_per_step, kernel_ms, out["mean_iters"])
        if args.config == 5:
            # north_star's "fraction of the dense-GEMM roofline": the benched path forms no Gram, so the MFMA Gram kernel of
            # the dense path (qpmpc/mpc_qp.py:99-105) is timed by itself, in this run, on this workload's problems
            out["roofline"]["gram_mfma"] = _gram_mfma_block(w)
            out["sweeps_per_problem"] = _sweeps_per_problem(w)
        out["accuracy"] = _accuracy(args, w, run)
        if args.config == 4 and world == 1:
            out["paired_by_last_counts"] = _paired_block(run, local_per_step)
        if not args.no_extras and world == 1 and args.config == 2:
            try:
                out["other_workloads"] = other_workloads()
            except Exception as exc:  # never at the expense of the headline line
                out["other_workloads"] = {"error": repr(exc)}
            try:
                out["predicted_strong_scaling"] = predicted_strong_scaling()
            except Exception as exc:
                out["predicted_strong_scaling"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, w, args.cpu_seconds)
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["all_cores_value"]
    return out


def predicted_strong_scaling():
    """No 8-GPU node was available to this build: what one GPU measures at the per-GPU SHARE of the strong-split configurations
    (config 4: 65,536 / 8 = 8192 problems; config 5: 8192 / 8 = 1024), against the full batch on one GPU. The ranks share nothing
    on the data path (qpmpc_amd/distributed.py), so rate(share) x 8 is what eight GPUs deliver before the all_gather, and
    rate(share) / rate(full) is the predicted strong-scaling efficiency. A PREDICTION from one GPU, not a measured curve."""
    import torch

    from qpmpc_amd import PreparedSolve
    from qpmpc_amd import workloads as W

    def rate(launch, problems, steps):
        for _ in range(max(3, steps // 4)):
            launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            launch()
        torch.cuda.synchronize()
        return problems * steps / (time.perf_counter() - t0)

    out = {"note": "single-GPU rates at the per-GPU share of an 8-way strong split vs the full batch; efficiency = share rate / "
                   "full rate; x8 = the eight-GPU aggregate this predicts (no data-path collective)"}
    for name, make, full, steps in (("config4_humanoid", lambda b: W.to_batch_problem(W.humanoid_batch(b)), 65536, 40),
                                    ("config5_synthetic_ltv_f32", lambda b: W.to_batch_problem(W.synthetic_ltv_batch(b), dtype=torch.float32), 8192, 5)):
        share = full // 8
        r_full = rate(PreparedSolve(make(full)).launch, full, steps)
        r_share = rate(PreparedSolve(make(share)).launch, share, steps * 4)
        out[name] = {"full_batch": full, "share_per_gpu": share, "rate_full_1gpu": r_full, "rate_share_1gpu": r_share,
                     "predicted_efficiency_8gpu": r_share / r_full, "predicted_rate_8gpu": 8 * r_share}
    return out


def _paired_block(run, problems, reps: int = 50):
    """Not the headline: the same launch with MpcqpSolveOpts.order = the batch sorted by the PREVIOUS launch's iteration counts
    (mpcqp_order_by_count: what a receding-horizon loop has at hand -- here the previous launch solved the same problems, so the
    counts are exact and this is the ceiling of the technique), the device-side sort timed with it."""
    import torch

    from qpmpc_amd import pairing_order

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    solver = run.solver
    order = torch.empty_like(solver.iters)
    natural = timed(solver.launch)
    pairing_order(solver.iters, out=order)
    solver.set_order(order)
    paired = timed(solver.launch)

    def period():
        solver.launch()
        pairing_order(solver.iters, out=order)

    with_sort = timed(period)
    solver.set_order(None)
    return {"ms_natural": natural, "ms_paired": paired, "ms_paired_plus_sort": with_sort,
            "value_paired_plus_sort": problems / (with_sort * 1e-3), "unit": "problems/s",
            "note": "order = mpcqp_order_by_count(last launch's iters), re-sorted every launch; statuses and iteration counts are those "
                    "of the natural order, plans equal to rounding (tests/test_gpu_pairing.py); not the headline value"}


def _overlap_two_streams(args, run, rank, world, barrier, allreduce, dev):
    """Not the headline: two independent batches in flight on two HIP streams, the way a server or a set of
    unrelated control loops would submit work (how much of a single-stream step is ramp and tail)."""
    import torch

    from qpmpc_amd import PreparedSolve
    from qpmpc_amd import workloads as W

    b = args.batch or CONFIGS[2]["batch"]
    w2 = W.triple_integrator_batch(b, seed=30250614 + rank, heterogeneous=not args.shared_lti)
    runs = [run.solver, PreparedSolve(W.to_batch_problem(w2))]
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
    t2 = allreduce(torch.tensor([el2], dtype=torch.float64, device=dev), "MAX")
    return {"streams": 2, "value": b * world * args.steps / float(t2.item()), "unit": "problems/s",
            "ms_per_step": float(t2.item()) / args.steps * 1e3,
            "note": "independent batches in flight on 2 HIP streams; not the headline value"}


def _sweeps_per_problem(w, dtype=None):
    """Config 5: how many serial sweeps over the horizon a problem makes in the wide stage-wise kernel (mpcqp_stagew.hip): counted BY
    THE KERNEL (developer stamps, one extra launch outside the timed region) -- every forward sweep reads 64 KB of factor records per
    problem, every backward sweep up to 48 KB: they are the launch's HBM traffic."""
    import torch

    from qpmpc_amd import solve_mpc_batch
    from qpmpc_amd import workloads as W

    bp = W.to_batch_problem(w, dtype=torch.float32)
    buf = torch.zeros(bp.batch_size * 16, dtype=torch.int64, device="cuda")
    plan = solve_mpc_batch(bp, probe=buf)
    torch.cuda.synchronize()
    code = buf.view(bp.batch_size, 16)[:, 15]
    bwd = float((code & 255).double().mean().item())
    ev = float(((code >> 8) & 255).double().mean().item())
    pol = float((code >> 16).double().mean().item())
    return {"riccati_recursion": 1.0, "forward_unconstrained_minimiser": 1.0, "backward_sweeps_of_cached_rows": bwd,
            "forward_evaluations_of_the_point": ev, "polish_steps": pol, "iterations": float(plan.iters.float().mean().item()),
            "note": "per problem, means over the batch; a forward sweep reads 64 KB of records, a backward sweep at most 48 KB"}


def _dims_of(w):
    import numpy as np

    nx, nu = int(np.asarray(w["A"]).shape[-1]), int(np.asarray(w["B"]).shape[-1])
    N, mk = int(w["N"]), int(np.asarray(w["e"]).shape[-1])
    return nx, nu, N, mk


def _traffic_from_profiles(config: int):
    """(HBM bytes per launch, source) from the rocprofv3 PMC passes stored under profiles/ for this round's kernels
    (separate --pmc runs of this same command: tools/collect_profiles.sh; FETCH_SIZE doubled as the microarchitecture
    guide prescribes for gfx950, WRITE_SIZE as reported). Newest round first; (None, None) when there is none."""
    names = [f"r0{r}_pmc_traffic_config{config}.json" for r in (6, 5, 4, 3, 2)]
    if config == 2:
        names = ["r06_pmc_traffic.json", "r05_pmc_traffic.json"] + names + ["r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"]
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            if d.get("hbm_bytes_per_launch") is not None:
                return float(d["hbm_bytes_per_launch"]), ("profiles/" + name + ": rocprofv3 --pmc FETCH_SIZE (x2, gfx950 "
                                                          "correction) + WRITE_SIZE per launch of the dominant kernel, collected "
                                                          "in separate counter passes of this command (not by this run)")
    return None, None


def _small_kernel_name(problems: int) -> str:
    """Which small-problem fused kernel a cold launch of `problems` lean problems gets (csrc/mpcqp_quad.hip, quad_pays: four per
    wavefront from more than two problems per SIMD of the device up; one round: the roomy LDS carve, several rounds: the slim one, two
    wavefronts per SIMD) -- for the report only; the library decides."""
    import torch

    simds = 4 * torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    if problems > 2 * simds:  # (quad_pays, csrc/mpcqp_quad.hip: more than two problems per SIMD)
        carve = "35.6 KB of LDS, one wavefront per SIMD" if (problems + 3) // 4 <= simds else "20 KB of LDS, two wavefronts per SIMD"
        return f"mpcqp_quad_kernel<3> (fused build+solve, FOUR problems per wavefront, one per 16-lane DPP row; {carve})"
    return "mpcqp_pair_kernel<3, 2> (fused build+solve, two problems per wavefront)"


def _roofline(args, w, local_per_step, kernel_ms, mean_iters):
    from qpmpc_amd import workloads as W

    nx, nu, N, mk = _dims_of(w)
    n, m = N * nu, N * mk
    kernel_s = kernel_ms * 1e-3
    esz = 4 if args.config == 5 else 8
    bytes_pp = W.algorithmic_bytes_per_problem(w, esz)
    stage, term = w["wx"] is not None, w["wt"] is not None
    flops_pp = W.algorithmic_build_flops(nx, nu, N, mk, stage, term) + W.algorithmic_solve_flops(n, m, mean_iters)
    gbs = bytes_pp * local_per_step / kernel_s / 1e9
    tfs = flops_pp * local_per_step / kernel_s / 1e12
    common = {"kernel_ms": kernel_ms, "algorithmic_bytes_per_problem": bytes_pp, "algorithmic_flops_per_problem": flops_pp,
              "units_per_launch": local_per_step}
    common["traffic"], common["traffic_source"] = _traffic_from_profiles(args.config)
    if args.batch:  # the stored counters belong to the configured batch
        common["traffic"], common["traffic_source"] = None, None
    if common["traffic"] is not None:
        # what the memory system actually moves per launch (counters of the stored passes) over THIS run's kernel time:
        # not `achieved` (that is algorithmic bytes by contract), but it tells whether the launch is paced by HBM
        common["traffic_rate_gbs"] = common["traffic"] / kernel_s / 1e9
        common["traffic_rate_frac_of_peak"] = common["traffic_rate_gbs"] / HBM_PEAK_GBS
        common["traffic_over_algorithmic"] = common["traffic"] / (bytes_pp * local_per_step)
    if args.config in (2, 4):
        return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "kernel": _small_kernel_name(local_per_step),
                "achieved_tflops_f64": tfs, **common,
                "note": "nominally HBM-bound (4 flop/B) but a launch moves only ~11 MB: the serial active-set chain of "
                        "each wavefront (instruction issue + dependent latency) sets the time, not HBM or FP64 throughput"}
    if args.config == 3:
        # what the stage-wise kernel EXECUTES (it forms neither P nor G), not the reference's dense flops
        ex = W.stagewise_executed_flops(nx, nu, N, mk, mean_iters)
        etf = ex * local_per_step / kernel_s / 1e12
        common["algorithmic_flops_per_problem"] = ex
        return {"bound": "mfma", "achieved": etf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": etf / FP64_PEAK_TFLOPS,
                "kernel": "mpcqp_stage_kernel<4, 1, serial, pipelined, 4> (stage-wise Riccati active set on v_mfma_f64_4x4x4; "
                          "workgroups of four loops: four solving wavefronts and ONE factor wavefront that rebuilds the four "
                          "factors of the next period, one per lane quad of its matrix-core products, into the loops' second LDS "
                          "image) with the plant step, next references and bookkeeping as its epilogue; up to periods_per_launch "
                          "consecutive periods per launch (the episode's first period is a launch of its own), kernel_ms is per "
                          "period",
                "periods_per_launch": PERIODS_PER_LAUNCH, "pipeline_factor": True,
                "achieved_gbs": gbs, "dense_equivalent_tflops": tfs, **common,
                "note": "achieved = float64 operations the kernel executes per period (Riccati recursion, sweeps, slack "
                        "passes: ~3e4 per loop) over the period; dense_equivalent_tflops prices the reference's dense "
                        "condense + solve flops, which this path does not execute. With 1024 loops there are five wavefronts "
                        "per CU and the period is the latency of one problem's serial chain (dependent float64 "
                        "operations, LDS round trips), nowhere near a throughput roof. Peak = AMD's fp64 vector=matrix figure"}
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "kernel": "mpcqp_stagew_kernel<float, 12> (one launch per step: Riccati factor, LQR sweeps and the dual "
                      "active set of one problem per wavefront; the condensed QP is never formed)",
            "dense_equivalent_tflops": tfs, **common,
            "note": "achieved = the problem's inputs + outputs (SURVEY 8d) over the launch; the kernel's own HBM traffic "
                    "is ~15x that (per-step factor records written once and re-read by every sweep pair: traffic, "
                    "traffic_over_algorithmic) and at batch 8192 it moves at ~0.6 of the HBM peak (traffic_rate_gbs): the "
                    "launch is paced by its own record traffic, at small batches by one problem's serial sweeps. "
                    "dense_equivalent_tflops prices the reference's dense condense + solve flops (which this path "
                    "does not execute) over the same time -- above the 157 TFLOP/s fp32 MFMA peak, i.e. out of reach "
                    "of any dense implementation"}


def _gram_mfma_block(w, batch=2048, reps=10):
    """In-run HIP-event time of mpcqp_gram_mfma_f32_kernel (phase 2 of mpcqp_condense_phase_batch) on the first `batch`
    problems of the config-5 workload: executed flops (the causal lower triangle's MFMAs) and the dense-equivalent flops of
    the reference's product (mpc_qp.py:99-105) against the 157.3 TFLOP/s dense float32 matrix-core peak."""
    import ctypes as C

    import numpy as np
    import torch

    from qpmpc_amd import _capi
    from qpmpc_amd import workloads as W
    from qpmpc_amd.batch import _stream_ptr

    lib = _capi.load()
    nx, nu, N, mk = _dims_of(w)
    n, m = N * nu, N * mk
    B = min(batch, int(w["x0"].shape[0]))
    sub = {k: (v[:B] if isinstance(v, np.ndarray) and v.ndim >= 2 and v.shape[0] == w["x0"].shape[0] else v) for k, v in w.items()}
    bp = W.to_batch_problem(sub, dtype=torch.float32)
    dims, cp = bp.dims(), bp.c_problem()
    dev = bp.device
    P = torch.empty((B, n, n), dtype=torch.float32, device=dev)
    q = torch.empty((B, n), dtype=torch.float32, device=dev)
    G = torch.empty((B, m, n), dtype=torch.float32, device=dev)
    h = torch.empty((B, m), dtype=torch.float32, device=dev)
    Psi = torch.empty((B, (N + 1) * nx * n), dtype=torch.float32, device=dev)
    ws = torch.empty((B * (N + 1) * nx * 4,), dtype=torch.uint8, device=dev)
    sp = _stream_ptr()
    args = (C.byref(dims), C.byref(cp), B)
    tail = (P.data_ptr(), q.data_ptr(), G.data_ptr(), h.data_ptr(), Psi.data_ptr(), ws.data_ptr(), ws.numel(), sp)
    _capi.check(lib.mpcqp_condense_phase_batch(*args, 1, *tail), "mpcqp_condense_phase_batch(1)")
    for _ in range(3):
        _capi.check(lib.mpcqp_condense_phase_batch(*args, 2, *tail), "mpcqp_condense_phase_batch(2)")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.mpcqp_condense_phase_batch(*args, 2, *tail)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # tiles of 32 x 32 of the lower triangle, active from the step whose columns reach them: the kernel's own count
    mfmas = 0
    nt = n // 32
    for k in range(1, N + 1):                      # block row k of Psi: nx rows, non-zero in columns < k nu
        rows = nx
        act = min(nt, -(-(k * nu) // 32))          # tile rows / columns that see non-zero columns
        mfmas += (act * (act + 1) // 2) * (rows / 2.0)   # one 32x32x2 instruction per two rows and tile
    executed = mfmas * 32 * 32 * 2 * 2
    dense = 2.0 * (N + 1) * nx * n * n
    peak = 157.3
    return {"kernel": "mpcqp_gram_mfma_f32_kernel<8> (v_mfma_f32_32x32x2_f32; P = w_u I + Psi' W Psi of the dense path, "
                      "mpcqp_condense_phase_batch phase 2)",
            "batch": B, "kernel_ms": ms, "executed_flops_per_problem": executed, "dense_equivalent_flops_per_problem": dense,
            "achieved_tflops_executed": executed * B / (ms * 1e-3) / 1e12, "peak_tflops_f32_mfma_dense": peak,
            "frac": executed * B / (ms * 1e-3) / 1e12 / peak,
            "dense_equivalent_tflops": dense * B / (ms * 1e-3) / 1e12,
            "source_profile": "profiles/r04_config5_dense_path.txt",
            "note": "event time over `reps` launches of the Gram product alone on torch's current stream (the stream the "
                    "library launches on); executed = the causal, symmetric part the kernel computes"}


def _accuracy(args, w, run):
    """max |u - u_oracle| on (a sample of) this rank's problems, CPU oracle as the checker."""
    import numpy as np

    import oracle

    lim = {2: None, 3: 64, 4: 4096, 5: 64}[args.config]
    U = run.U.double().cpu().numpy()
    st = run.status.cpu().numpy()
    if args.config == 3:  # the loop has moved on: compare the LAST solved problems through the oracle
        bp = run.loop.problem
        ws = dict(w)
        ws["x0"] = bp.initial_state.cpu().numpy()
        ws["goal"] = bp.goal_state.cpu().numpy()
        ws["targets"] = bp.target_states.cpu().numpy()
        # the problem buffers already hold the NEXT period's data; re-solve them on the device for the comparison
        run.solver.launch()
        import torch

        torch.cuda.synchronize()
        U, st, w = run.U.cpu().numpy(), run.status.cpu().numpy(), ws
    if args.config == 5:  # (seconds per problem on one core: the sample's problems go to the host cores, one shard each)
        from oracle.parallel import solve_workload_parallel
        from qpmpc_amd.distributed import shard_workload

        batch = int(w["x0"].shape[0])
        ws = shard_workload(w, 0, max(1, batch // lim)) if batch > lim else w
        Uo, _, sto, _ = solve_workload_parallel(ws, shard_workload)
    else:
        Uo, _, sto, _ = oracle.solve_workload(w, count=lim)
    k = Uo.shape[0]
    ok = (sto == 0) & (st[:k] == 0)
    err = np.abs(U[:k][ok] - Uo[ok])
    scale = np.maximum(1.0, np.abs(Uo[ok]).max(axis=1, keepdims=True))
    return {"checked_problems": int(k), "status_agree": bool(np.array_equal(st[:k] == 0, sto == 0)),
            "max_abs_err_vs_oracle": float(err.max()) if err.size else 0.0,
            "max_rel_err_vs_oracle": float((err / scale).max()) if err.size else 0.0,
            "tolerance": "1e-6 relative (float64) / 1e-3 (float32), tests/"}


def cpu_baseline(config: int, w, seconds: float = 10.0):
    """The oracle timed on the host cores (rank 0, N=1 only), on a bounded sample of the same workload:
    * "port" (`value`): the reference's execution model -- one Python call per problem, NumPy condensing
      (oracle.condense_np, the restatement of mpc_qp.py) + a native dense active-set solve (what
      qpsolvers/quadprog does), 1 thread;
    * "port_c": the all-C oracle (condense + Goldfarb-Idnani), 1 thread;
    * "all_cores": the all-C oracle in one spawned process per host CPU, each on its own shard."""
    import numpy as np

    import oracle
    from oracle.parallel import all_cores_rate
    from qpmpc_amd import workloads as W
    from qpmpc_amd.distributed import shard_workload

    batch = int(w["x0"].shape[0])
    t0 = time.perf_counter()
    done, b = 0, 0
    budget = seconds * 0.4
    while time.perf_counter() - t0 < budget:
        p = W.problem_from_workload(w, b)
        oracle.solve_mpc_like_reference(p)
        done += 1
        b = (b + 1) % batch
    t_ref = time.perf_counter() - t0
    sample = min(batch, {2: 1024, 3: 64, 4: 1024, 5: 2}[config])
    ws = shard_workload(w, 0, max(1, batch // sample)) if sample < batch else w
    t1 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t1 < seconds * 0.2 or reps == 0:
        oracle.solve_workload(ws)
        reps += 1
    t_c = time.perf_counter() - t1
    nb = int(ws["x0"].shape[0])
    cores = all_cores_rate(ws, seconds=max(2.0, seconds * 0.3))
    return {
        "value": done / t_ref, "unit": "problems/s", "cores": 1, "kind": "port",
        "sample": f"{done} problems of the same batch, one Python call each (NumPy condense + C active-set), {t_ref:.1f} s",
        "port_c_value": reps * nb / t_c,
        "port_c_sample": f"all-C oracle, {reps} passes over {nb} problems of the batch, {t_c:.1f} s, 1 thread",
        "all_cores_value": cores["value"], "all_cores": cores["cores"], "cpu_model": cores["cpu_model"],
        "all_cores_sample": f"all-C oracle, {cores['cores']} spawned processes (one per host CPU), each passing over the same "
                            f"{nb} problems for {cores['seconds_per_worker']:.1f} s",
        "host_cpus": os.cpu_count(),
    }


def other_workloads():
    """Rates of the other BASELINE.json configurations on this GPU (rank 0, one GPU only; never `value`):
    a few launches each, HIP events on the launch stream. Each of them can be the timed `value` with
    --config 3|4|5; parity for all of them is in tests/ (-m gpu)."""
    import numpy as np
    import torch

    from qpmpc_amd import PreparedSolve, SharedModel
    from qpmpc_amd import workloads as W
    from qpmpc_amd.closed_loop import LIPMWalkingLoop, WIPClosedLoop

    def rate(run, batch, reps):
        # (spin-up: the clocks of a GPU that has just idled through a host-side set-up ramp for tens of milliseconds -- a short
        # timed region right behind it once read 13 M/s for a 175 M/s launch)
        t0 = time.perf_counter()
        n = 0
        while n < 3 or time.perf_counter() - t0 < 0.1:
            run()
            n += 1
            if n % 8 == 0:
                torch.cuda.synchronize()
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
    loop_p = WIPClosedLoop(rng.standard_normal((1024, 4)) * np.array([0.05, 0.05, 0.1, 0.1]), pipeline_factor=True)
    loop_p.step(2)  # the first period factors for itself
    out["config3_closed_loop_rebuild_pipelined"] = rate(loop_p.step, 1024, 50)
    x0r = rng.standard_normal((1024, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
    loop_r = WIPClosedLoop(x0r, reuse_factor=True)
    loop_r.step(2)  # the first period keeps the factor
    out["config3_closed_loop_factor_reused_resolves"] = rate(loop_r.step, 1024, 50)
    for key, kw in (("config3_closed_loop_rebuild_pipelined_20_periods_per_launch", {"pipeline_factor": True}),
                    ("config3_closed_loop_factor_reused_20_periods_per_launch", {"reuse_factor": True})):
        lp = WIPClosedLoop(rng.standard_normal((1024, 4)) * np.array([0.05, 0.05, 0.1, 0.1]), periods_per_launch=20, **kw)
        lp.step(2)
        out[key] = rate(lambda: lp.step(20), 1024 * 20, 4)
    # config 2 in shared-LTI mode (reported separately and labelled): stride-0 operands, then the model
    # factored once (P, Cholesky, G L^-T hoisted out of the batch)
    bp2 = W.to_batch_problem(W.triple_integrator_batch(4096, heterogeneous=False))
    # the headline launch over 2000 launches (the driver's contract times 20: ~1 us per step of launch pipeline is inside that region)
    bph = W.to_batch_problem(W.triple_integrator_batch(4096))
    out["config2_headline_batch4096_2000_launches"] = rate(PreparedSolve(bph).launch, 4096, 2000)
    out["config2_shared_lti_operands_fused_batch4096"] = rate(PreparedSolve(bp2).launch, 4096, 200)
    out["config2_shared_lti_model_factored_once_batch4096"] = rate(SharedModel(bp2).prepare(bp2).launch, 4096, 200)
    w = W.humanoid_batch(65536)
    bp = W.to_batch_problem(w)
    out["config4_humanoid_sweep_65536_fused"] = rate(PreparedSolve(bp).launch, 65536, 10)
    shared = SharedModel(bp).prepare(bp)
    out["config4_humanoid_sweep_65536_shared_model"] = rate(shared.launch, 65536, 10)
    # ... and with the problems paired by the previous launch's iteration counts, the sort re-run every launch
    from qpmpc_amd import pairing_order

    order = torch.empty_like(shared.iters)
    shared.set_order(pairing_order(shared.iters, out=order))

    def shared_period():
        shared.launch()
        pairing_order(shared.iters, out=order)

    out["config4_humanoid_sweep_65536_shared_model_paired_by_last_counts"] = rate(shared_period, 65536, 10)

    shared.set_order(None)
    w = W.synthetic_ltv_batch(1024)
    out["config5_synthetic_ltv_n256_m1024_f32_batch1024"] = rate(
        PreparedSolve(W.to_batch_problem(w, dtype=torch.float32)).launch, 1024, 5)
    # the dense HBM-resident path (propagate + MFMA Gram + one-QP-per-workgroup solver): what systems with nx > 16 or
    # nu > 4 get -- a slow fallback, an order of magnitude behind the stage-wise kernels
    from qpmpc_amd import _capi

    out["dense_fallback_path_n256_m1024_f32_batch1024"] = rate(
        PreparedSolve(W.to_batch_problem(w, dtype=torch.float32), flags=_capi.OPT_FORCE_CONDENSED).launch, 1024, 3)
    # a mid-size problem family of a wider system (random LTV, nx = 8, nu = 2, N = 20: n = 40, m = 80, float64): the automatic
    # dispatch (wide stage-wise kernel since round 3) next to the condensed on-chip kernels it used to take
    import os as _os

    _tools = _os.path.join(ROOT, "tools")
    if _tools not in sys.path:
        sys.path.insert(0, _tools)
    from stress_stagewise import random_ltv

    wm = random_ltv(np.random.default_rng(5), 4096, 8, 2, 20, 4, 1.0)
    wm["A"] = np.eye(8) + 0.3 * (wm["A"] - np.eye(8))
    bpm = W.to_batch_problem(wm)
    out["midsize_ltv_nx8_nu2_n40_m80_f64_batch4096"] = rate(PreparedSolve(bpm).launch, 4096, 5)
    out["midsize_ltv_nx8_nu2_n40_m80_f64_batch4096_condensed_kernels"] = rate(
        PreparedSolve(bpm, flags=_capi.OPT_FORCE_CONDENSED).launch, 4096, 3)
    # the reference's own example problem (examples/wheeled_inverted_pendulum.py:90-94: N = 12, input box, stage + terminal cost), 4096
    # states: the four-per-wavefront kernel's general build since round 6 (the two-per-wavefront kernel's generic build before)
    wr = W.wip_batch(4096, N=12, sampling_period=0.1, seed=5)
    wr["x0"][:1024, 1] += 0.4
    tsr = np.stack([wr["pendulum"].target_states(x, 0.5) for x in wr["x0"]])
    wr["goal"], wr["targets"] = tsr[:, -4:], tsr[:, :-4]
    bpr = W.to_batch_problem(wr)
    out["reference_wip_example_n12_batch4096"] = rate(PreparedSolve(bpr).launch, 4096, 200)
    out["reference_wip_example_n12_batch4096_two_per_wavefront"] = rate(PreparedSolve(bpr, flags=_capi.OPT_TWO_PER_WAVE).launch, 4096, 200)
    # small problems of wider systems (n = 16): the four-per-wavefront kernel's general build since round 6 (three / four operand
    # registers per step for nx = 5, 6, streamed operands from nx = 7); they ran on the one-per-wavefront kernel before
    for key, (nxs, nus, Ns) in (("small_nx6_nu2_n16_m16_f64_batch4096", (6, 2, 8)), ("small_nx12_nu4_n16_m16_f64_batch4096", (12, 4, 4)),
                                ("small_nx7_nu1_n16_m32_f64_batch4096", (7, 1, 16))):
        ws = random_ltv(np.random.default_rng(3), 4096, nxs, nus, Ns, 16 // Ns if Ns < 8 else 2, 0.5)
        out[key] = rate(PreparedSolve(W.to_batch_problem(ws)).launch, 4096, 50)
    # ... and with more than 32 rows (four rows per step at N = 16: m = 64): the four-rows-per-lane copy of that kernel (mpcqp_quad4.hip)
    w64 = random_ltv(np.random.default_rng(3), 4096, 3, 1, 16, 4, 0.5)
    out["small_nx3_nu1_n16_m64_f64_batch4096"] = rate(PreparedSolve(W.to_batch_problem(w64)).launch, 4096, 50)
    walkers = LIPMWalkingLoop(4096, index=rng.integers(0, 8, 4096))
    out["lipm_walking_loops_4096"] = rate(walkers.step, 4096, 100)
    walkers_m = LIPMWalkingLoop(4096, index=rng.integers(0, 8, 4096), shared_model=True)
    out["lipm_walking_loops_4096_model_factored_once"] = rate(walkers_m.step, 4096, 100)
    return {k: float(v) for k, v in out.items()} | {"unit": "problems/s (builds+solves/s for the loops)"}


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _respawn(argv, n: int) -> int:
    """``python bench.py --gpus N`` started outside torch.distributed.run: launch the N ranks ourselves
    (same command line the driver uses) and hand its exit code back; rank 0 of the child prints the line."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)]
    cmd += list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main(argv=None) -> int:
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _respawn(argv, args.gpus)
    import torch

    have_gpu = torch.cuda.is_available()
    if have_gpu:
        torch.cuda.set_device(local_rank)
    elif rank == 0:
        print("bench.py: no GPU visible -- RANK-LOGIC DRY RUN over gloo, nothing is computed, value = null",
              file=sys.stderr)
    dist = None
    # (a single rank launched through torch.distributed.run also goes through RCCL: the only way to execute the
    # collectives' code path on a one-GPU box)
    if world > 1 or os.environ.get("TORCHELASTIC_RUN_ID"):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if have_gpu:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    if have_gpu:
        out = run_bench(args, rank, world, dist)
    else:
        args.spinup = 0.0
        out = run_bench(args, rank, world, dist, make_runner=_DryRunner, device="cpu")
        if out is not None:
            out["dry_run"] = "no GPU visible: rank logic over gloo with a runner that computes nothing; NOT a measurement"
            out["value"] = out["ms_per_step"] = None
            out["solved_frac"] = out["mean_iters"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
