#!/bin/bash
# Run ON THE GPU BOX: the narrow stage-wise kernel (mpcqp_stage.hip) after a change of its sweeps / recursion: stress seeds + long horizons
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in 6 7 8 9 10 11 12 13 14 15 16 17; do echo "stress_stagewise narrow 30x128 seed $s: $(STRESS_SEED=$s python tools/stress_stagewise.py 30 128 narrow 2>&1 | tail -1)"; done
