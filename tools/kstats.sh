#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel stats of a command, top kernels with short names. usage: tools/kstats.sh <command...>
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kst && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o t -- "$@" > /tmp/kst.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kst/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:10.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  {r['Percentage']}%")
PY
