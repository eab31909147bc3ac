#!/bin/bash
# Run ON THE GPU BOX (through gpurun): evidence for the large-problem path (config 5),
# written to gpurun_out/c5/ (scratch); tools/summarise_config5.py turns it into
# profiles/r02_config5_kernel_stats.txt. Counter passes are separate runs (never combined with traces
# other than --kernel-trace).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c5
mkdir -p $OUT
python $R/tools/bench_config5.py 8192 3 > $OUT/bench_8192.txt 2>&1
python $R/tools/probe_big_phases.py 1024 > $OUT/phases_1024.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -o t -- python $R/tools/bench_config5.py 8192 3 > /dev/null 2>&1
cp $(find /tmp/p5 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_8192.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  tag=$(echo $c | tr ' ' '+' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p5_$tag -o p -- python $R/tools/bench_config5.py 1024 1 > $OUT/pmc_$tag.log 2>&1
  cp $(find /tmp/p5_$tag -name "*counter_collection.csv" | head -1) $OUT/pmc_$tag.csv
done
ls -la $OUT
