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
