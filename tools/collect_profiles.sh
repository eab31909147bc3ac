#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for bench.py, written to
# gpurun_out/ (scratch); summarise with tools/summarise_profiles.py into profiles/.
# Counter passes are separate from each other and never combined with sys/hip traces.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-overlap --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SMEM"; do
  tag=$(echo $c | tr ' ' '+' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$tag -o p -- python $R/bench.py --steps 20 --warmup 5 --spinup 0 --no-cpu-baseline --no-overlap --no-extras > $OUT/pmc_$tag.log 2>&1
done
# keep only the small CSVs
find $OUT -name "*.db" -delete
du -sh $OUT
