#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for the wide stage-wise kernel at config 5's dimensions.
# Counter passes are separate from each other and never combined with sys/hip traces.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_stagew_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/run_config5_stagewise.py ${3:-8192} 10 ${2:-f32}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  tag=$(echo $c | tr ' ' '+' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$tag -o p -- $CMD > $OUT/pmc_$tag.log 2>&1
done
find $OUT -name "*.db" -delete
du -sh $OUT
