#!/usr/bin/env python3
"""Dev probe: per-phase shader-clock timestamps of the small-problem kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import PreparedSolve, workloads as W
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = W.triple_integrator_batch(batch); bp = W.to_batch_problem(w)
one = len(sys.argv) > 2 and sys.argv[2] == "w64"
SL = 8 if one else 16
buf = torch.zeros(batch * SL, dtype=torch.int64, device="cuda")
from qpmpc_amd import _capi
flags = _capi.OPT_ONE_PER_WAVE if (len(sys.argv) > 2 and sys.argv[2] == "w64") else (_capi.OPT_SEED_VIOLATED if (len(sys.argv) > 2 and sys.argv[2] == "seed") else 0)
run = PreparedSolve(bp, probe=buf, flags=flags)
for _ in range(3): run.launch()
torch.cuda.synchronize()
t = buf.view(batch, SL).cpu().double()
it = run.iters.cpu().double()
names = ["build", "cholesky", "fwd-subst", "init", "active-set", "refine+x"]
d = (t[:, 1:7] - t[:, 0:6])
print("batch", batch, "mean iters", it.mean().item())
for i, nme in enumerate(names):
    print(f"  {nme:12s} mean {d[:, i].mean().item():10.0f} cyc   max {d[:, i].max().item():10.0f}")
print(f"  total        mean {(t[:,6]-t[:,0]).mean().item():10.0f} cyc   max {(t[:,6]-t[:,0]).max().item():10.0f}")
print(f"  per-iteration (active-set / iters): {(d[:,4].sum()/it.sum()).item():.0f} cyc")
print(f"  kernel span: {(t[:,6].max()-t[:,0].min()).item():.0f} cyc")
if not one:
    sub = t[:, [0, 8, 9, 10, 11, 1]]
    for nme, a in zip(["stage operands", "x0/e loads, wsync", "chain", "gram", "q row, h"], range(5)):
        print(f"    build/{nme:18s} mean {(sub[:, a+1]-sub[:, a]).mean().item():8.0f} cyc")
print("  iterations: max", int(it.max().item()), " 99%", torch.quantile(it, 0.99).item(), " 90%", torch.quantile(it, 0.9).item())
pm = torch.maximum(it[0::2], it[1::2]) if not one else it
print("  per-wavefront max(iterations of the pair): mean", pm.mean().item(), "max", pm.max().item())
if not one:
    rt = t[:, 12:14]  # s_memrealtime (100 MHz) at the first and the last stamp of a wavefront
    st, en = rt[:, 0], rt[:, 1]
    if float((en - st).mean()) < 0:  # (the two-per-wavefront kernel stamps them the other way round)
        st, en = en, st
    q = torch.tensor([0.5, 0.9, 0.99, 1.0], dtype=torch.float64)
    print(f"  real time (100 MHz): first start -> last end {(en.max()-st.min()).item()/100:.2f} us ; wavefront mean {(en-st).mean().item()/100:.2f} us max {(en-st).max().item()/100:.2f} us")
    print("    starts after the first (50/90/99/100 %, us):", [round(v / 100, 2) for v in torch.quantile(st - st.min(), q).tolist()],
          "; ends before the last:", [round(v / 100, 2) for v in torch.quantile(en.max() - en, q).tolist()])
