#!/usr/bin/env python3
"""Dev probe: per-phase shader-clock timestamps of the large-problem solver (config 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "config5"
if kind == "wip":  # config 3 through the mid-size kind of the same kernel (f64)
    w = W.wip_batch(batch); bp = W.to_batch_problem(w)
else:
    w = W.synthetic_ltv_batch(batch); bp = W.to_batch_problem(w, dtype=torch.float32)
buf = torch.zeros(batch * 32, dtype=torch.int64, device="cuda")
from qpmpc_amd import _capi
run = PreparedSolve(bp, probe=buf, flags=_capi.OPT_FORCE_CONDENSED)  # the condensed kernels (config 3 would otherwise take the stage-wise one)
for _ in range(2): run.launch()
torch.cuda.synchronize()
full = buf.view(batch, 32).cpu().double()
t = full[:, :8]
laps = full[:, 8:16]
it = run.iters.cpu().double()
names = ["load P", "factor+invert", "init slacks", "active-set", "solution"]
t = t[:, [0, 1, 3, 4, 5, 6]]
d = (t[:, 1:6] - t[:, 0:5])
print("batch", batch, "mean iters", it.mean().item(), "max", it.max().item())
for i, nme in enumerate(names):
    print(f"  {nme:12s} mean {d[:, i].mean().item():10.0f} cyc   max {d[:, i].max().item():10.0f}")
print(f"  total        mean {(t[:,5]-t[:,0]).mean().item():10.0f} cyc   max {(t[:,5]-t[:,0]).max().item():10.0f}")
print(f"  per-iteration (active-set / iters): {(d[:,3].sum()/it.sum()).item():.0f} cyc")
print(f"  kernel span: {(t[:,5].max()-t[:,0].min()).item():.0f} cyc  (readcyclecounter ticks at 100 MHz if s_memrealtime, else shader clock)")
lap_names = ["select", "row fetch", "M_p = L^-1 g", "r = N* M_p", "z, d2, ratio", "z_x = L^-T z", "G z_x, slacks", "N* update"]
print("  inside the active-set loop (cycles per iteration):")
for i, nme in enumerate(lap_names):
    print(f"    {nme:16s} {(laps[:, i].sum() / it.sum()).item():9.0f}")
fl = full[:, 16:21]
print("  inside factor+invert (cycles per problem):")
for i, nme in enumerate(["diag blocks", "panels", "trailing", "W_I L[I,K]", "X rows"]):
    print(f"    {nme:16s} {fl[:, i].mean().item():9.0f}")
if kind == "wip":
    print("  front end, busy cycles per wavefront (sum over the steps):", [int(full[:, 24 + w].mean().item()) for w in range(4)])
