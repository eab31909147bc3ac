#!/bin/bash
# Run ON THE GPU BOX (through gpurun): round 5's stress campaign on the final build; prints one line per (tool, seed).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in 70 71 72 73; do echo "stress_dense 120x64 seed $s: $(STRESS_SEED=$s python tools/stress_dense.py 120 64 2>&1 | tail -1)"; done
for s in 21 22 23 24; do echo "stress_pair 40x256 seed $s: $(STRESS_SEED=$s python tools/stress_pair.py 40 256 2>&1 | tail -1)"; done
# the lean family forced through the four-per-wavefront kernel (MPCQP_OPT_FOUR_PER_WAVE)
for s in 31 32 33 34 35 36; do echo "stress_pair LEAN four-per-wavefront 40x256 seed $s: $(STRESS_LEAN=1 STRESS_FOUR=1 STRESS_SEED=$s python tools/stress_pair.py 40 256 2>&1 | tail -1)"; done
for s in 6 7; do echo "stress_stagewise wide 20x128 seed $s: $(STRESS_SEED=$s python tools/stress_stagewise.py 20 128 2>&1 | tail -1)"; echo "stress_stagewise narrow 20x128 seed $s: $(STRESS_SEED=$s python tools/stress_stagewise.py 20 128 narrow 2>&1 | tail -1)"; done
for s in 56 57 58 59; do echo "stress_f32 60x128 seed $s: $(STRESS_SEED=$s python tools/stress_f32.py 60 128 2>&1 | grep -E 'CHECK|worst' | tr '\n' ' ')"; done
# the general stage-wise kernel (round 5: thin QR operator): the review's list
for s in 1 2 3 4 5 6 7 8 9 10 11 12; do echo "stress_general 12x8 seed $s: $(STRESS_SEED=$s python tools/stress_general.py 12 8 2>&1 | tail -1)"; done
for s in 1 2 3 4 5 6 7 8 9; do echo "stress_tight general 8x8 seed $s: $(STRESS_SEED=$s python tools/stress_tight.py general 8 8 2>&1 | tail -1)"; done
for s in 1 2 3 4 5 6 7 8 9; do echo "stress_tight wide 8x8 seed $s: $(STRESS_SEED=$s python tools/stress_tight.py wide 8 8 2>&1 | tail -1)"; echo "stress_tight narrow 8x8 seed $s: $(STRESS_SEED=$s python tools/stress_tight.py narrow 8 8 2>&1 | tail -1)"; done
