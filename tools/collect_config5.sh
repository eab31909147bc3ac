#!/bin/bash
# Run ON THE GPU BOX (through gpurun): evidence for the large-problem path (config 5),
# written to gpurun_out/c5/ (scratch); tools/summarise_config5.py turns it into
# profiles/r01_config5_kernel_stats.txt. Counter passes are separate runs.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c5
mkdir -p $OUT
python $R/tools/bench_config5.py 8192 3 > $OUT/bench_8192.txt 2>&1
python $R/tools/probe_big_phases.py 1024 > $OUT/phases_1024.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -o t -- python $R/tools/bench_config5.py 8192 3 > /dev/null 2>&1
cp $(find /tmp/p5 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_8192.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p5_$c -o p -- python $R/tools/bench_config5.py 1024 1 > /dev/null 2>&1
  cp $(find /tmp/p5_$c -name "*counter_collection.csv" | head -1) $OUT/pmc_$c.csv
done
ls -la $OUT
