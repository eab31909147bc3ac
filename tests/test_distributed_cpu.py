"""World-size-2 gloo test of the sharding + gather/reduce path (CPU). The local
solve is done by the CPU oracle here (the product's solve needs a GPU); what is
under test is qpmpc_amd.distributed: contiguous shards, padded all_gather in
rank order, and the statistics all_reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from qpmpc_amd import workloads as W
from qpmpc_amd.distributed import gather_batch, reduce_stats, shard_range, shard_workload


def test_shard_range_is_a_partition():
    for total in (0, 1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_shard_workload_keeps_shared_operands():
    w = W.humanoid_batch(10)
    s = shard_workload(w, 1, 3)
    assert s["x0"].shape[0] == 3 and s["A"] is not None and s["A"].shape == (3, 3)
    assert np.array_equal(s["x0"], w["x0"][4:7])
    wh = W.triple_integrator_batch(9)
    s = shard_workload(wh, 1, 2)
    assert s["A"].shape[0] == 4 and s["goal"].shape[0] == 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = W.humanoid_batch(total, seed=4)
        ws = shard_workload(w, rank, world)
        U, _, st, it = oracle.solve_workload(ws)
        Ut, stt, itt = torch.tensor(U), torch.tensor(st, dtype=torch.int32), torch.tensor(it, dtype=torch.int32)
        U_all = gather_batch(Ut, total)
        st_all = gather_batch(stt, total)
        stats = reduce_stats(stt, itt)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), U=U_all.numpy(), st=st_all.numpy(),
                 **{k: np.array(v) for k, v in stats.items()})
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_matches_single_process(tmp_path):
    total, world = 37, 2  # odd on purpose: shards of 19 and 18 rows
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    w = W.humanoid_batch(total, seed=4)
    U, _, st, it = oracle.solve_workload(w)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["U"], U) and np.array_equal(z["st"], st)
        assert z["problems"] == total and z["solved"] == (st == 0).sum() and z["infeasible"] == (st == 2).sum()
        assert abs(z["mean_iters"] - it.mean()) < 1e-12 and z["max_iters"] == it.max()


# ---------------------------------------------------------------- bench.py's own rank logic under gloo
class _OracleRunner:
    """Stand-in for bench._Runner on CPU: the solve is done by the oracle (test infrastructure) into CPU
    tensors. What is under test is bench.run_bench: argument handling, per-rank workloads (weak: own
    seeds; strong: slices of one global set), reductions, the timed all_gather and the JSON record."""

    def __init__(self, config, w, device):
        self.w = w
        self.launches = 0
        self.launch()

    def launch(self, stream=None):
        U, _, st, it = oracle.solve_workload(self.w)
        self.U, self.status, self.iters = torch.tensor(U), torch.tensor(st, dtype=torch.int32), torch.tensor(it, dtype=torch.int32)
        self.launches += 1


def _bench_worker(rank, world, port, config, batch, out_dir):
    import json
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        args = bench.parse_args(["--gpus", str(world), "--config", str(config), "--steps", "2", "--warmup", "1",
                                 "--batch", str(batch), "--spinup", "0", "--no-cpu-baseline"])
        out = bench.run_bench(args, rank, world, dist if world > 1 else None, make_runner=_OracleRunner, device="cpu")
        assert (out is not None) == (rank == 0)
        if rank == 0:
            with open(os.path.join(out_dir, f"bench_c{config}_w{world}.json"), "w") as f:
                json.dump(out, f)
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("config,batch", [(2, 24), (4, 37)])
def test_bench_rank_logic_two_ranks_gloo(tmp_path, config, batch):
    import json

    for world in (1, 2):
        mp.spawn(_bench_worker, args=(world, _free_port(), config, batch, str(tmp_path)), nprocs=world, join=True)
    one = json.load(open(tmp_path / f"bench_c{config}_w1.json"))
    two = json.load(open(tmp_path / f"bench_c{config}_w2.json"))
    for rec, world in ((one, 1), (two, 2)):
        assert rec["n_gpus"] == world and rec["steps"] == 2 and rec["warmup"] == 1 and rec["unit"] == "problems/s"
        assert rec["metric"].startswith("MPC QP builds+solves/sec") and rec["vs_baseline"] is None
        assert rec["value"] > 0 and abs(rec["value"] - rec["config"]["problems_per_step"] * 2 / (rec["ms_per_step"] * 2e-3)) < 1e-6 * rec["value"]
    if config == 2:  # weak: every rank owns `batch` problems of its own
        assert one["scaling"] == two["scaling"] == "weak"
        assert one["config"]["problems_per_step"] == batch and two["config"]["problems_per_step"] == 2 * batch
        assert two["config"]["problems_per_gpu_per_step"] == batch
        assert "all_gather_ms" not in two
    else:  # strong: ONE global set split over the ranks, gathered afterwards
        assert one["scaling"] == two["scaling"] == "strong"
        assert one["config"]["problems_per_step"] == two["config"]["problems_per_step"] == batch
        assert two["config"]["problems_per_gpu_per_step"] == 19  # rank 0 of shards (19, 18)
        assert two["all_gather_ms"] >= 0.0
        # the same global problem set whatever the number of ranks: identical whole-job statistics
        assert abs(one["solved_frac"] - two["solved_frac"]) < 1e-12 and abs(one["mean_iters"] - two["mean_iters"]) < 1e-12


@pytest.mark.parametrize("config,batch", [(2, 16), (5, 6)])
def test_bench_main_spawns_its_own_ranks(config, batch):
    """``python bench.py --gpus 2`` started WITHOUT torch.distributed.run (how a user, or a driver that only knows
    the N=1 command, starts it) must launch its two ranks itself and print ONE JSON line from rank 0. On this
    GPU-less box that is the rank-logic dry run over gloo: ``value`` is null and the record says so."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", str(config), "--steps", "2",
           "--warmup", "1", "--batch", str(batch)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1
    if torch.cuda.is_available():
        assert rec["value"] > 0
    else:
        assert rec["value"] is None and "dry_run" in rec
    if config == 2:
        assert rec["scaling"] == "weak" and rec["config"]["problems_per_step"] == 2 * batch
    else:
        assert rec["scaling"] == "strong" and rec["config"]["problems_per_step"] == batch and "all_gather_ms" in rec


def test_bench_main_in_process_goes_through_respawn(monkeypatch):
    """bench.main(["--gpus", "2", ...]) with no WORLD_SIZE in the environment re-executes through
    torch.distributed.run (the command line is the driver's) instead of exiting with a usage error."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import subprocess

    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd
        return 0

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(subprocess, "call", fake_call)
    assert bench.main(["--gpus", "2", "--config", "2", "--steps", "2", "--warmup", "1"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--config", "2", "--steps", "2", "--warmup", "1"]


def test_bench_config5_slices_are_one_global_problem_set():
    a = W.synthetic_ltv_batch_slice(0, 4, N=8)
    b = W.synthetic_ltv_batch_slice(2, 4, N=8)
    for key in ("A", "B", "x0"):
        assert np.array_equal(a[key][2:], b[key])
