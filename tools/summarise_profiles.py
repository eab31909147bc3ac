#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (rocprofv3 CSVs) into the committed summaries under
profiles/: kernel stats, per-kernel counter means, and the HBM traffic JSON that
bench.py reads for roofline.traffic."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
# optional third argument: suffix of the traffic JSON ("config5" -> profiles/<rnd>_pmc_traffic_config5.json), and a
# fourth: the command line that was profiled (header of the summary)
suffix = ("_" + sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] else ""
cmdline = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 2000 --warmup 200 --no-cpu-baseline"
src = os.path.join("gpurun_out", f"prof_{tag}")
dst = "profiles"
os.makedirs(dst, exist_ok=True)
lines = []
# ---- kernel stats (rocprofv3 --kernel-trace --stats)
for f in glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    lines.append(f"# rocprofv3 --kernel-trace --stats -- {cmdline}")
    lines.append("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for r in rows[:8]:
        lines.append(",".join([r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
# ---- counters
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines.append("")
lines.append("# rocprofv3 --kernel-trace --pmc <one counter set per pass> -- " + (cmdline if suffix else "python bench.py --steps 20 --warmup 5 --no-cpu-baseline"))
lines.append("# mean value per dispatch of the mpcqp kernel")
traffic = {}
for k, cs in agg.items():
    if "mpcqp" not in k:
        continue
    lines.append(f"kernel: {k[:100]}")
    for c, v in sorted(cs.items()):
        lines.append(f"  {c:26s} {sum(v)/len(v):18.1f}   (dispatches: {len(v)})")
        traffic[c] = sum(v) / len(v)
open(os.path.join(dst, f"{rnd}_{tag}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
if "FETCH_SIZE" in traffic:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB. MI355X_MICROARCH.md (HBM): on gfx950
    # FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> doubled here;
    # WRITE_SIZE is taken as reported (uncalibrated).
    fetch = traffic["FETCH_SIZE"] * 1024.0 * 2.0
    write = traffic.get("WRITE_SIZE", 0.0) * 1024.0
    json.dump({"hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
               "raw_FETCH_SIZE_KiB": traffic["FETCH_SIZE"], "raw_WRITE_SIZE_KiB": traffic.get("WRITE_SIZE"),
               "source": f"profiles/{rnd}_{tag}_rocprof_summary.txt"},
              open(os.path.join(dst, f"{rnd}_pmc_traffic{suffix}.json"), "w"), indent=1)
print("\n".join(lines))
