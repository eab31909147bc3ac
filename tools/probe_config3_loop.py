#!/usr/bin/env python3
"""Dev probe: shader-clock stamps of one period of the config-3 closed loop (1024 loops, N = 50) in its three modes
(rebuilding / factor pipelined by a second wavefront / factor reused). usage: probe_config3_loop.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd.closed_loop import WIPClosedLoop
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # periods per launch
rng = np.random.default_rng(1)
x0 = rng.standard_normal((B, 4)) * np.array([0.05, 0.05, 0.1, 0.1])
names = ["factor (or its load)", "chunk matrices", "u0 backward", "u0 forward", "slacks", "active set", "verification", "outputs + plant epilogue"]
for name, kw in (("rebuild", {}), ("pipeline_factor", {"pipeline_factor": True}), ("reuse_factor", {"reuse_factor": True})):
    loop = WIPClosedLoop(x0.copy(), periods_per_launch=K, **kw)
    loop.step(20)
    buf = torch.zeros(B * 16, dtype=torch.int64, device="cuda")
    loop.solver._opts.probe = buf.data_ptr()
    loop._period_args = None
    loop.step(max(2, 2 * K))
    torch.cuda.synchronize()
    t = buf.view(B, 16).cpu().double()
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter(); loop.step(10 * K); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / (10 * K) * 1e6
    print(f"  wall clock with the stamps on: {wall:.2f} us per period; first stamp -> last stamp of a period: {(t[:, 8] - t[:, 0]).mean().item():.0f} ticks"
          f" = {(t[:, 8] - t[:, 0]).mean().item() / wall / 1e3:.2f} ticks per ns if that were all")
    if kw.get("pipeline_factor"):
        sol, fac = (t[:, 8] - t[:, 11]), (t[:, 10] - t[:, 9])
        per = torch.maximum(sol, fac)
        qs = torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64)
        print("  per workgroup: solving", [int(v) for v in torch.quantile(sol, qs)], "factor", [int(v) for v in torch.quantile(fac, qs)],
              "max of the two", [int(v) for v in torch.quantile(per, qs)], "(10/50/90/99/100 %)")
        hs, hf = buf.view(B, 16)[:, 14].cpu(), buf.view(B, 16)[:, 15].cpu()
        simd_s, simd_f = (hs >> 4) & 3, (hf >> 4) & 3
        cu = lambda h: ((h >> 8) & 15) | (((h >> 12) & 1) << 4) | (((h >> 13) & 7) << 5)
        print("    solver SIMD == factor SIMD in", int((simd_s == simd_f).sum()), "workgroups; mean solving there", 
              int(sol[simd_s == simd_f].mean()) if (simd_s == simd_f).any() else -1, "elsewhere", int(sol[simd_s != simd_f].mean()) if (simd_s != simd_f).any() else -1)
        for sd in range(4):
            print(f"    solver on SIMD {sd}: {int((simd_s == sd).sum())} workgroups, mean solving {int(sol[simd_s == sd].mean()) if (simd_s == sd).any() else -1};"
                  f" factor on SIMD {sd}: {int((simd_f == sd).sum())}")
        slow = sol > torch.quantile(sol, torch.tensor(0.8, dtype=torch.float64))
        joint = torch.zeros(4, 4, dtype=torch.int64); jslow = torch.zeros(4, 4, dtype=torch.int64)
        for a, b2, sl_ in zip(simd_s.tolist(), simd_f.tolist(), slow.tolist()):
            joint[a, b2] += 1; jslow[a, b2] += int(sl_)
        print("    (solver SIMD, factor SIMD) counts", joint.tolist(), "slow among them", jslow.tolist())
        # what shares a SIMD with each solving wavefront: per (XCC, CU, SIMD) the kinds of wavefronts
        xs, xf = (hs >> 32) & 15, (hf >> 32) & 15
        ks, kf = (xs * 512 + cu(hs)) * 4 + simd_s, (xf * 512 + cu(hf)) * 4 + simd_f
        ns, nf = torch.bincount(ks, minlength=8 * 512 * 4), torch.bincount(kf, minlength=8 * 512 * 4)
        for a in range(0, 3):
            for b2 in range(0, 3):
                sel = (ns[ks] == a + 1) & (nf[ks] == b2)
                if sel.any(): print(f"    solving wavefronts on a SIMD with {a + 1} solving and {b2} factor wavefronts: {int(sel.sum())}, mean solving time {int(sol[sel].mean())}, slow {int(slow[sel].sum())}")
        print("    slow workgroups:", int(slow.sum()), "; their solver SIMDs", torch.bincount(simd_s[slow], minlength=4).tolist(), "factor SIMDs", torch.bincount(simd_f[slow], minlength=4).tolist())
        cuid = (cu(hs) | ((hs >> 32) & 15) << 8)  # (XCC, SE, SH, CU)
        ids, inv = torch.unique(cuid, return_inverse=True)
        cnt = torch.bincount(inv); mean_cu = torch.bincount(inv, weights=sol) / cnt
        within = float(((sol - mean_cu[inv]) ** 2).mean().sqrt()); between = float(mean_cu.std())
        print(f"    {len(ids)} CUs, workgroups per CU min {int(cnt.min())} max {int(cnt.max())}; solving time: std between CUs {between:.0f}, within a CU {within:.0f};"
              f" slowest CUs' means {sorted(mean_cu.tolist())[-3:]}, fastest {sorted(mean_cu.tolist())[:3]}")
        slowfrac = torch.bincount(inv, weights=slow.double()) / cnt
        print("    fraction of slow workgroups per CU: histogram", torch.histc(slowfrac, bins=5, min=0, max=1).tolist())
        xcc = (hs >> 32) & 15
        print("    mean solving time per XCD", [int(sol[xcc == i].mean()) for i in range(8) if (xcc == i).any()])
        # waves per (CU, SIMD) as seen by this launch (XCC unknown: counts are summed over the 8 XCDs)
        key_s, key_f = cu(hs) * 4 + simd_s, cu(hf) * 4 + simd_f
        occ = torch.bincount(torch.cat([key_s, key_f]), minlength=1)
        print("    waves per (SE, SH, CU, SIMD) key summed over XCDs: min", int(occ[occ > 0].min()), "max", int(occ.max()))
        nsolv = torch.bincount(key_s, minlength=int(occ.numel()))
        print("    solving waves sharing a key with n solving waves (summed over XCDs): slow mean", float(nsolv[key_s[slow]].double().mean()), "fast mean", float(nsolv[key_s[~slow]].double().mean()))
        # by position in the CU (workgroups are dispatched round-robin over XCDs, then CUs): look for a systematic pattern
        for m in (8, 32, 256):
            g = torch.stack([per[i::m].mean() for i in range(m)])
            print(f"    mean period by workgroup index mod {m}: min {int(g.min())} max {int(g.max())}")
    if K > 1:  # several periods per launch: slot 12 = end of the hand-over after the period BEFORE the last one
        print(f"  {'period entry -> first stamp':26s} mean {(t[:, 0] - t[:, 11]).mean().item():9.0f} cyc")
        print(f"  {'store drain + barrier':26s} mean {(t[:, 12] - t[:, 13]).mean().item():9.0f} cyc   (after the period before the last)")
        print(f"  {'hand-over -> next entry':26s} mean {(t[:, 11] - t[:, 12]).mean().item():9.0f} cyc")
    print(name)
    for i, nme in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"  {nme:26s} mean {d.mean().item():9.0f} cyc   max {d.max().item():9.0f}")
    tot = t[:, 8] - t[:, 0]
    print(f"  {'solving wavefront, total':26s} mean {tot.mean().item():9.0f} cyc   max {tot.max().item():9.0f}")
    if kw.get("pipeline_factor"):
        f = t[:, 10] - t[:, 9]
        print(f"  {'factor wavefront':26s} mean {f.mean().item():9.0f} cyc   max {f.max().item():9.0f}; ends {(t[:, 10] - t[:, 8]).mean().item():.0f} cyc after the solving one (mean)")
    # (the shader clocks of different CUs are not synchronised: stamps of different wavefronts cannot be compared;
    # every stamp costs the wavefront ~0.9 k cycles, so the phases above are inflated by that much each)
