#!/bin/bash
# Dev tool (GPU box): one rocprofv3 --pmc pass per counter group for a command, mean per dispatch of the kernels matching $KERNEL.
# usage: KERNEL=gram tools/pmc.sh "<counters group 1>" "<counters group 2>" ... -- <command...>
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
for g in "${groups[@]}"; do
  rm -rf /tmp/pmcx; rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmcx -o p -- "$@" > /tmp/pmcx.log 2>&1
  python3 - <<PY
import csv, glob, collections, os
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ.get("KERNEL", "mpcqp") in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"  {c:30s} {sum(v)/len(v):18.1f}  (n={len(v)})")
PY
done
