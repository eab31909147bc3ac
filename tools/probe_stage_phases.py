#!/usr/bin/env python3
"""Dev probe: per-phase shader-clock stamps of the stage-wise kernel. usage: probe_stage_phases.py [wip|long] [batch] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import solve_mpc_batch, workloads as W
kind = sys.argv[1] if len(sys.argv) > 1 else "wip"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if kind == "wip":
    bp = W.to_batch_problem(W.wip_batch(batch))
elif kind == "sat":  # config 3's problems with states that saturate the input box (tools/probe_saturating.py)
    w = W.wip_batch(batch, seed=9)
    w["x0"][:, 1] += 0.3
    w["x0"][:, 3] += 1.0
    ts = np.stack([w["pendulum"].target_states(x, 0.5) for x in w["x0"]])
    w["goal"], w["targets"] = ts[:, -4:], ts[:, :-4]
    bp = W.to_batch_problem(w)
elif kind in ("c5", "c5f64"):
    bp = W.to_batch_problem(W.synthetic_ltv_batch_slice(0, batch), dtype=torch.float32 if kind == "c5" else torch.float64)
else:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_stagewise import long_batch  # noqa
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    bp = long_batch(batch, N, 1.0 / max(16, N // 16))
buf = torch.zeros(batch * 16, dtype=torch.int64, device="cuda")
for _ in range(2): plan = solve_mpc_batch(bp, formulation="stagewise", probe=buf)
torch.cuda.synchronize()
t = buf.view(batch, 16).cpu().double()
names = ["riccati", "chunk matrices", "u0 backward", "u0 forward", "slacks", "active set", "verification"]
print(kind, "batch", batch, "N", bp.nb_timesteps, "mean iters", plan.iters.float().mean().item())
for i, nme in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"  {nme:16s} mean {d.mean().item():10.0f} cyc   max {d.max().item():10.0f}")
print(f"  total            mean {(t[:,7]-t[:,0]).mean().item():10.0f} cyc   max {(t[:,7]-t[:,0]).max().item():10.0f}")
if kind == "wip" or kind == "sat":
    for i, nme in zip((9, 10, 12, 13), ["sweeps", "c, r = W c", "ratio test", "slack update"]):
        print(f"    in loop: {nme:14s} mean {t[:, i].mean().item():10.0f} cyc   per iteration {t[:, i].sum().item() / max(1.0, plan.iters.float().sum().item()):8.0f}")
if kind in ("c5", "c5f64"):
    for i, nme in zip(range(8, 15), ["selection", "backward", "forward", "gmul", "step calc", "slack update", "W update"]):
        print(f"    in loop: {nme:14s} mean {t[:, i].mean().item():10.0f} cyc")
if kind in ("c5", "c5f64"):
    code = buf.view(batch, 16)[:, 15].cpu()
    nv0, tv0 = (code & 0xffff).double(), (code >> 16).double() / 1000.0
    it = plan.iters.cpu().double()
    work = (t[:, 7] - t[:, 5])
    def corr(a, b): return float(torch.corrcoef(torch.stack([a, b]))[0, 1])
    print(f"    predictors of the active-set work: corr(nv0, iters) {corr(nv0, it):.3f}  corr(tv0, iters) {corr(tv0, it):.3f}  corr(nv0, cycles) {corr(nv0, work):.3f}  corr(tv0, cycles) {corr(tv0, work):.3f}")
    # what longest-first would give: list scheduling on 3072 slots by descending key vs natural order
    import heapq
    def makespan(order, slots=3072):
        h = [0.0] * slots; heapq.heapify(h)
        for i in order: heapq.heappush(h, heapq.heappop(h) + float(work[i]))
        return max(h)
    nat = makespan(range(batch)); ideal = float(work.sum()) / 3072
    print(f"    active-set makespan on 3072 slots (cycles): natural order {nat:.0f}, by nv0 desc {makespan(torch.argsort(nv0, descending=True).tolist()):.0f}, by tv0 desc {makespan(torch.argsort(tv0, descending=True).tolist()):.0f}, oracle LPT {makespan(torch.argsort(work, descending=True).tolist()):.0f}, sum/slots {ideal:.0f}")
print(f"  makespan {(t[:,7].max()-t[:,0].min()).item():.0f} cyc; start spread {(t[:,0].max()-t[:,0].min()).item():.0f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record(); plan = solve_mpc_batch(bp, formulation="stagewise"); e1.record(); torch.cuda.synchronize()
    print(f"  call: {e0.elapsed_time(e1):.2f} ms")
