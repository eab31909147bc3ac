import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from stress_pair import run
for seed in (1, 3, 4):
    worst, bad, drops = run(100, 192, seed=seed, verbose=True)
    print("seed", seed, "worst", worst, "flagged", bad, "drops", drops, flush=True)
