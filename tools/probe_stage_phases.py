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
names = ["riccati", "chunk matrices", "u0 backward", "u0 forward + rows", "-", "active set", "evaluation"]
print(kind, "batch", batch, "N", bp.nb_timesteps, "mean iters", plan.iters.float().mean().item())
for i, nme in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"  {nme:16s} mean {d.mean().item():10.0f} cyc   max {d.max().item():10.0f}")
print(f"  total            mean {(t[:,7]-t[:,0]).mean().item():10.0f} cyc   max {(t[:,7]-t[:,0]).max().item():10.0f}")
if kind == "wip" or kind == "sat":
    for i, nme in zip((9, 10, 12, 13), ["sweeps", "c, r = W c", "ratio test", "slack update"]):
        print(f"    in loop: {nme:14s} mean {t[:, i].mean().item():10.0f} cyc   per iteration {t[:, i].sum().item() / max(1.0, plan.iters.float().sum().item()):8.0f}")
if kind in ("c5", "c5f64"):  # (round 6: the loop's parts of the thin-QR kernel)
    its = max(1.0, plan.iters.float().sum().item())
    for i, nme in zip(range(8, 15), ["cache lookup", "candidates+bwd", "orthogonalise", "R solve+ratio", "step: v, cached rows", "append / drop", "evaluations"]):
        print(f"    in loop: {nme:16s} mean {t[:, i].mean().item():10.0f} cyc   per iteration {t[:, i].sum().item() / its:8.0f}")
if kind in ("c5", "c5f64"):
    code = buf.view(batch, 16)[:, 15].cpu()
    print(f"    per problem: backward sweeps {(code & 255).double().mean().item():.2f}, evaluations {((code >> 8) & 255).double().mean().item():.2f}, polish steps {(code >> 16).double().mean().item():.3f}")
print(f"  makespan {(t[:,7].max()-t[:,0].min()).item():.0f} cyc; start spread {(t[:,0].max()-t[:,0].min()).item():.0f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record(); plan = solve_mpc_batch(bp, formulation="stagewise"); e1.record(); torch.cuda.synchronize()
    print(f"  call: {e0.elapsed_time(e1):.2f} ms")
