#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round's evidence in one call -- GPU tests, one bench line per
# configuration, rocprofv3 kernel + PMC passes for the headline kernel (config 2) and the wide stage-wise
# kernel (config 5). Everything lands in gpurun_out/r04_$1/ (scratch); tools/summarise_profiles.py and
# tools/pmc_summary.py turn it into profiles/r04_*.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}
OUT=$R/gpurun_out/r04_$TAG
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
  timeout 900 bash $R/tools/collect_profiles.sh pair_$TAG
  timeout 900 bash $R/tools/collect_stagew.sh $TAG
  timeout 600 bash $R/tools/collect_stagew.sh ${TAG}_b1024 f32 1024
fi
du -sh $R/gpurun_out
