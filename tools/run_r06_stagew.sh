#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the wide stage-wise kernel after its active-set operator became a thin QR factorisation
# (round 6): its tests, the tight stress families WITHOUT the host side's re-solve, config 5's bench lines.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stagewise.py -x -q -m gpu 2>&1 | tail -15
for t in 0.5 0.3 0.15 0.05; do
  echo "stress_tight wide STRESS_TIGHT=$t seeds 1-16: $(STRESS_TIGHT=$t STRESS_SEEDS=1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16 timeout 900 python tools/stress_tight.py wide 8 8 2>&1 | grep -E 'CHECK|worst' | tail -8 | tr '\n' ' ')"
done
for s in 6 7; do echo "stress_stagewise wide 20x128 seed $s: $(STRESS_SEED=$s timeout 600 python tools/stress_stagewise.py 20 128 2>&1 | grep -E 'CHECK|worst' | tr '\n' ' ')"; done
for s in 56 57; do echo "stress_f32 60x128 seed $s: $(STRESS_SEED=$s timeout 600 python tools/stress_f32.py 60 128 2>&1 | grep -E 'CHECK|worst' | tr '\n' ' ')"; done
python bench.py --config 5 > gpurun_out/r06_bench_config5.json 2> gpurun_out/r06_bench_config5.err; tail -c 600 gpurun_out/r06_bench_config5.err
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench_config5.json",):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("mean_iters"), d.get("solved_frac"), d.get("accuracy"))
    except Exception as e:
        print(f, "unreadable", e)
PY
python bench.py --config 5 --batch 1024 > gpurun_out/r06_bench_config5_b1024.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_config5_b1024.json').read().strip().splitlines()[-1]); print('b1024', d['value'], d['ms_per_step'], d.get('mean_iters'), d.get('solved_frac'))"
