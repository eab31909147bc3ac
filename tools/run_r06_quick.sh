#!/bin/bash
# Run ON THE GPU BOX: quick check of the wide stage-wise kernel (tests without the iteration-count case, one tight family, config 5 timing + phases)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_stagewise.py -q -m gpu 2>&1 | tail -6
echo "stress_tight wide 0.3 seeds 1,5,11: $(STRESS_TIGHT=0.3 STRESS_SEEDS=1,5,11 timeout 600 python tools/stress_tight.py wide 8 8 2>&1 | grep -E 'CHECK|worst' | tail -4 | tr '\n' ' ')"
echo "stress_f32 seed 56: $(STRESS_SEED=56 timeout 600 python tools/stress_f32.py 30 128 2>&1 | grep -E 'CHECK|worst' | tr '\n' ' ')"
for b in 8192 1024; do python bench.py --config 5 --batch $b 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step'],4), 'ms iters', d.get('mean_iters'), 'solved', d.get('solved_frac'), 'err', (d.get('accuracy') or {}).get('max_rel_err_vs_oracle'))"; done
python tools/probe_stage_phases.py c5 8192 2>&1 | grep -v amdgpu.ids | tail -22; python tools/probe_stage_phases.py c5 1024 2>&1 | grep -v amdgpu.ids | tail -22
