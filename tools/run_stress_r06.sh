#!/bin/bash
# Run ON THE GPU BOX (through gpurun): round 6's stress campaign on the final build; prints one line per (tool, seed).
# New this round: the wide stage-wise kernel's thin-QR operator (default dispatch of everything beyond 20 variables with nx <= 16,
# nu <= 4), the second opinion behind the narrow kernel -- the tight families run WITHOUT any host-side re-solve.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in 70 71 72 73; do echo "stress_dense 120x64 seed $s: $(STRESS_SEED=$s timeout 900 python tools/stress_dense.py 120 64 2>&1 | tail -1)"; done
for s in 21 22; do echo "stress_pair 40x256 seed $s: $(STRESS_SEED=$s timeout 600 python tools/stress_pair.py 40 256 2>&1 | tail -1)"; done
for s in 6 7 8 9; do echo "stress_stagewise wide 20x128 seed $s: $(STRESS_SEED=$s timeout 600 python tools/stress_stagewise.py 20 128 2>&1 | tail -1)"; echo "stress_stagewise narrow 20x128 seed $s: $(STRESS_SEED=$s timeout 600 python tools/stress_stagewise.py 20 128 narrow 2>&1 | tail -1)"; done
for s in 56 57 58 59; do echo "stress_f32 60x128 seed $s: $(STRESS_SEED=$s timeout 600 python tools/stress_f32.py 60 128 2>&1 | grep -E 'CHECK|worst' | tr '\n' ' ')"; done
for k in wide widef narrow general; do for t in 0.5 0.3 0.15 0.05; do
  echo "stress_tight $k STRESS_TIGHT=$t seeds 1-16 (no re-solve): $(STRESS_TIGHT=$t STRESS_SEEDS=1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16 timeout 1200 python tools/stress_tight.py $k 8 8 2>&1 | grep -E 'CHECK|^worst' | tail -3 | cut -c1-300 | tr '\n' ' ')"
done; done
for s in 1 2 3 4 5 6; do echo "stress_quad_general 150x96 seed $s (nx 2..16, every layout, forced four per wavefront): $(STRESS_QUIET=1 STRESS_SEED=$s timeout 600 python tools/stress_quad_general.py 150 96 2>&1 | grep -E 'CHECK|^worst' | tail -3 | tr '\n' ' ')"; done
for s in 1 2 3 4; do echo "stress_quad_general 150x96 seed $s with up to 64 rows (mpcqp_quad4.hip where m > 32): $(STRESS_ROWS64=1 STRESS_QUIET=1 STRESS_SEED=$s timeout 600 python tools/stress_quad_general.py 150 96 2>&1 | grep -E 'CHECK|^worst' | tail -3 | tr '\n' ' ')"; done
for t in 0.15 0.05; do
  echo "stress_tight narrow STRESS_TIGHT=$t seeds 1-16 through mpcqp_stagewise_solve_batch: $(STRESS_FORMULATION=stagewise STRESS_TIGHT=$t STRESS_SEEDS=1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16 timeout 1200 python tools/stress_tight.py narrow 8 8 2>&1 | grep -E 'CHECK|^worst' | tail -3 | cut -c1-300 | tr '\n' ' ')"
done
