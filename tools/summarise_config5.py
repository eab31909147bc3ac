#!/usr/bin/env python3
"""gpurun_out/c5/ (tools/collect_config5.sh) -> profiles/r02_config5_kernel_stats.txt"""
import collections, csv

rows = list(csv.DictReader(open("gpurun_out/c5/kernel_stats_8192.csv")))
out = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python tools/bench_config5.py 8192 3",
       "# config 5: synthetic LTV nx=12 nu=4 N=64 (n=256, m=1024), float32, batch 8192 on ONE MI355X",
       "# (bench_config5.py first times the condense-only API -- propagate<false> writes G -- then the fused build+solve)",
       "# Name, Calls, AverageNs, Percentage"]
for r in rows[:4]:
    nm = r["Name"].split("(")[0].replace("void mpcqp::", "")
    out.append(f"{nm:55s} {r['Calls']:>3s} {float(r['AverageNs']):14.1f} {float(r['Percentage']):6.2f}")


def pmc(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("void mpcqp::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items() if "mpcqp" in k}


f, w = pmc("gpurun_out/c5/pmc_FETCH_SIZE.csv"), pmc("gpurun_out/c5/pmc_WRITE_SIZE.csv")
out += ["", "# HBM traffic per launch at batch 1024 (separate --pmc passes; FETCH_SIZE/WRITE_SIZE in KB; FETCH doubled for",
        "# gfx950 as MI355X_MICROARCH.md prescribes)"]
for k in f:
    out.append(f"#   {k:50s} fetch {2*f[k]/1024:9.1f} MB (raw {f[k]/1024:8.1f})   write {w.get(k,0)/1024:9.1f} MB"
               f"   -> per problem {2*f[k]/1024/1024*1e3:7.1f} / {w.get(k,0)/1024/1024*1e3:7.1f} KB")
# ---- matrix-core counters (separate pass, batch 1024): MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)
def pmc_multi(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("void mpcqp::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in acc.items() if "mpcqp" in k}


mf = pmc_multi("gpurun_out/c5/pmc_SQ_INSTS_VALU_MFMA_MOPS_F32+SQ.csv")
ga = pmc_multi("gpurun_out/c5/pmc_GRBM_GUI_ACTIVE+SQ_WAVES.csv")
out += ["", "# matrix cores per launch at batch 1024 (separate --pmc passes): v_mfma_f32_32x32x2_f32 = 4096 flop, 64 busy cycles each;",
        "# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); MFMA TFLOP/s = executed MFMA flops / kernel time"]
for k in mf:
    c, g = mf[k], ga.get(k, {})
    active = g.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    util = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (active * 1024.0) if active else 0.0
    out.append(f"#   {k:50s} SQ_INSTS_MFMA {c.get('SQ_INSTS_MFMA', 0):12.0f}  MOPS_F32 {c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0):12.0f}  "
               f"MFMA_BUSY {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):13.0f}  VALU insts {c.get('SQ_INSTS_VALU', 0):13.0f}  "
               f"GUI_ACTIVE/XCD {active:11.0f}  MfmaUtil {util:5.1f} %  ({c.get('SQ_INSTS_MFMA', 0) / 1024:.0f} MFMA per problem)")
out += ["", "# tools/bench_config5.py 8192 3:"]
out += ["#   " + b for b in open("gpurun_out/c5/bench_8192.txt").read().strip().splitlines()[-2:]]
out.append("# tools/probe_big_phases.py 1024 (shader clocks per problem inside mpcqp_bigsolve_kernel):")
out += ["#   " + l for l in open("gpurun_out/c5/phases_1024.txt").read().splitlines() if l.strip() and "amdgpu.ids" not in l]
open("profiles/r02_config5_kernel_stats.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
