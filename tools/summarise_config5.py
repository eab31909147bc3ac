#!/usr/bin/env python3
"""gpurun_out/c5/ (tools/collect_config5.sh) -> profiles/r01_config5_kernel_stats.txt"""
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
out += ["", "# tools/bench_config5.py 8192 3:"]
out += ["#   " + b for b in open("gpurun_out/c5/bench_8192.txt").read().strip().splitlines()[-2:]]
out.append("# tools/probe_big_phases.py 1024 (shader clocks per problem inside mpcqp_bigsolve_kernel):")
out += ["#   " + l for l in open("gpurun_out/c5/phases_1024.txt").read().splitlines() if l.strip() and "amdgpu.ids" not in l]
open("profiles/r01_config5_kernel_stats.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
