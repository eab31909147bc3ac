for s in 40 41 42 43 44 45 46 47 48 49 50 51 52 53 54 55; do echo seed $s; STRESS_SEED=$s python tools/stress_f32.py 60 128 2>&1 | grep -E "CHECK|worst"; done
python -m pytest tests/test_gpu_stagewise.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
