#!/bin/bash
# Run ON THE GPU BOX (through gpurun): round 6's evidence in one call -- GPU tests, one bench line per configuration (and the
# single-rank RCCL launches of configs 2, 4, 5), rocprofv3 kernel + PMC passes for the headline kernel (config 2), the wide
# stage-wise kernel (config 5 at 8192 and at the 1024-problem share), HBM-traffic passes for configs 3 and 4. Everything lands in
# gpurun_out/r06_$1/ (scratch); tools/summarise_profiles.py turns it into profiles/r06_*.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}
OUT=$R/gpurun_out/r06_$TAG
mkdir -p $OUT
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc $?" >> $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
fi
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in 2 3 4 5; do
  timeout 300 python bench.py --config $c --no-extras > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err
done
timeout 300 python bench.py --config 5 --batch 1024 --no-extras --no-cpu-baseline > $OUT/bench_config5_b1024.json 2> $OUT/bench_config5_b1024.err
timeout 300 python bench.py --steps 2000 --warmup 200 --no-extras --no-cpu-baseline > $OUT/bench_config2_2000steps.json 2> /dev/null
# one rank over RCCL (backend nccl): the init / barrier / all_gather / all_reduce path of bench.py on real hardware
for c in 2 4 5; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2950$c \
    bench.py --gpus 1 --config $c --no-extras --no-cpu-baseline > $OUT/bench_rccl_single_rank_config$c.json 2> $OUT/bench_rccl_single_rank_config$c.err
done
cat $OUT/bench_config*.json $OUT/bench_rccl_*.json | cut -c1-400
if [ "${SKIP_PROF:-0}" != "1" ]; then
  timeout 900 bash $R/tools/collect_profiles.sh quad_$TAG
  timeout 900 bash $R/tools/collect_stagew.sh $TAG
  timeout 600 bash $R/tools/collect_stagew.sh ${TAG}_b1024 f32 1024
  ( cd /tmp && export TMPDIR=/tmp
    for c in 3 4; do
      P=$R/gpurun_out/prof_c${c}_$TAG; mkdir -p $P
      rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o t -- python $R/bench.py --config $c --spinup 0 --no-extras --no-cpu-baseline > $P/trace.log 2>&1
      for g in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
        tag=$(echo $g | tr ' ' '+' | cut -c1-40)
        rocprofv3 --kernel-trace --pmc $g --output-format csv -d $P/pmc_$tag -o p -- python $R/bench.py --config $c --steps 20 --spinup 0 --no-extras --no-cpu-baseline > $P/pmc_$tag.log 2>&1
      done
      find $P -name "*.db" -delete
    done )
fi
du -sh $R/gpurun_out
