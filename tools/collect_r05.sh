#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round's evidence in one call -- GPU tests, one bench line per
# configuration, rocprofv3 kernel + PMC passes for the headline kernel (config 2) and the wide stage-wise
# kernel (config 5). Everything lands in gpurun_out/r05_$1/ (scratch); tools/summarise_profiles.py and
# tools/pmc_summary.py turn it into profiles/r05_*.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}
OUT=$R/gpurun_out/r05_$TAG
mkdir -p $OUT
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc $?" >> $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
fi
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in 2 3 4 5; do
  timeout 300 python bench.py --config $c --no-extras > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err
done
timeout 300 python bench.py --config 5 --batch 1024 --no-extras --no-cpu-baseline > $OUT/bench_config5_b1024.json 2> $OUT/bench_config5_b1024.err
cat $OUT/bench_config*.json
if [ "${SKIP_PROF:-0}" != "1" ]; then
  timeout 900 bash $R/tools/collect_profiles.sh quad_$TAG
  timeout 900 bash $R/tools/collect_stagew.sh $TAG
  timeout 600 bash $R/tools/collect_stagew.sh ${TAG}_b1024 f32 1024
  # config 3: the narrow stage-wise kernel with the whole control period (kernel trace + instruction counters)
  P3=$R/gpurun_out/prof_stage_$TAG; mkdir -p $P3
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $P3/trace -o t -- python $R/bench.py --config 3 --spinup 0 --no-extras --no-cpu-baseline > $P3/trace.log 2>&1
    for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SMEM"; do
      tag=$(echo $c | tr ' ' '+' | cut -c1-40)
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P3/pmc_$tag -o p -- python $R/bench.py --config 3 --steps 100 --no-extras --no-cpu-baseline > $P3/pmc_$tag.log 2>&1
    done
    find $P3 -name "*.db" -delete )
fi
du -sh $R/gpurun_out
