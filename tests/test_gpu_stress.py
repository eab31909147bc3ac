"""GPU tests (-m gpu): the developer stress campaigns (tools/stress_*.py) at fixed seeds, so that the round-end GPU tier
covers the DISPATCH SPACE -- random shapes over every kernel's envelope, a quarter of the small problems with rows made
inconsistent -- and not only the hand-picked shapes of the other test files. Each campaign compares the HIP path through
the C ABI with the float64 C oracle (and, where the tool does, with the other formulations of the same solver):
statuses must agree and plans must match to 1e-7 relative (float64; the reference contract is 1e-6) / 1e-3
(float32). The reference has no such test -- its solver is third party (qpmpc/solve_mpc.py:43) --;
the families are the ones the round-2/3 campaigns (profiles/r03_stress_summary.txt) ran by hand.
"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _campaign(tool, args, seed, extra_env=None):
    env = dict(os.environ, STRESS_SEED=str(seed), **(extra_env or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *map(str, args)], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    last = [line for line in out.stdout.splitlines() if line.startswith("worst rel diff")]
    assert last, out.stdout[-2000:]
    m = re.match(r"worst rel diff ([0-9.eE+-]+) rounds flagged (\d+)", last[-1])
    assert m, last[-1]
    flagged = [line for line in out.stdout.splitlines() if "<-- CHECK" in line]
    return float(m.group(1)), int(m.group(2)), flagged


@pytest.mark.parametrize("tool,args,seed,bound", [
    ("stress_pair.py", (24, 128), 3, 1e-7),             # n <= 16, m <= 32: pair / one-per-wavefront / workgroup kernels vs the oracle
    ("stress_dense.py", (24, 48), 41, 1e-7),            # default dispatch, float64, n up to ~190
    ("stress_stagewise.py", (20, 64), 2, 1e-6),         # wide stage-wise kernel vs the condensed path
    ("stress_stagewise.py", (20, 64, "narrow"), 3, 1e-6),  # narrow stage-wise kernel (nx <= 4, nu <= 2)
    ("stress_f32.py", (24, 64), 41, 1e-3),              # float32 through the default dispatch vs the float64 oracle
    ("stress_f32.py", (24, 64), 46, 1e-3),
    ("stress_general.py", (12, 8), 2, 1e-7),            # wide systems (17 <= nx <= 32, nu <= 8): the general stage-wise kernel
    # nearly fully active problems (4-6 tight rows per step, most variables pinned): the three stage-wise kernels
    ("stress_tight.py", ("narrow", 8, 8), 1, 1e-7),
    ("stress_tight.py", ("wide", 8, 8), 1, 1e-7),
    ("stress_tight.py", ("general", 8, 8), 1, 1e-7),
])
def test_stress_campaign(tool, args, seed, bound):
    worst, nflag, flagged = _campaign(tool, args, seed)
    assert nflag == 0, "\n".join(flagged)
    assert worst <= bound, (tool, seed, worst)


# Round 4's known failures of the general stage-wise kernel (17 <= nx <= 32 or nu > 4: the only kernel of those dimensions) on nearly
# fully active problems -- one plan 3.3e-6 off (stress_general seed 7), one or two problems per campaign ending MPCQP_MAX_ITER where
# the oracle solves them (stress_tight general seeds 2, 3, 8), profiles/r04_stress_summary.txt -- were strict xfails here until the
# kernel's active-set operator became a thin QR factorisation of the whitened active rows with the projected direction sent through
# the forward sweep (round 5, csrc/mpcqp_stageg.hip): all four are ordinary cases now, next to more seeds of the same families.
@pytest.mark.parametrize("tool,args,seed,bound", [
    ("stress_general.py", (12, 8), 7, 1e-7),
    ("stress_general.py", (12, 8), 11, 1e-7),
    ("stress_tight.py", ("general", 8, 8), 2, 1e-7),
    ("stress_tight.py", ("general", 8, 8), 3, 1e-7),
    ("stress_tight.py", ("general", 8, 8), 8, 1e-7),
    ("stress_tight.py", ("general", 8, 8), 5, 1e-7),
])
def test_stress_campaign_nearly_full_active_sets_of_the_general_kernel(tool, args, seed, bound):
    worst, nflag, flagged = _campaign(tool, args, seed)
    assert nflag == 0, "\n".join(flagged)
    assert worst <= bound, (tool, seed, worst)


# The review's full list (round 4, "next round" item 2): stress_general seeds 1-12 and stress_tight general | wide | narrow seeds 1-9 --
# nearly fully active problems through all three stage-wise kernels -- every seed unflagged, plans within 1e-7 of the oracle's.
# One process per family (STRESS_SEEDS: the tool loops over the seeds).
@pytest.mark.parametrize("tool,args,seeds", [
    ("stress_general.py", (12, 8), range(1, 13)),
    ("stress_tight.py", ("general", 8, 8), range(1, 10)),
    ("stress_tight.py", ("wide", 8, 8), range(1, 10)),
    ("stress_tight.py", ("narrow", 8, 8), range(1, 10)),
])
def test_stress_campaign_every_seed_of_the_review_list(tool, args, seeds):
    worst, nflag, flagged = _campaign(tool, args, 0, {"STRESS_SEEDS": ",".join(str(sd) for sd in seeds)})
    assert nflag == 0, "\n".join(flagged)
    assert worst <= 1e-7, (tool, args, worst)


# Beyond the review's list: TIGHTER rows (STRESS_TIGHT 0.3 / 0.15 / 0.05 instead of 0.5: 60-78 of ~80 variables pinned, every variable at
# 0.05). Until round 5 the wide stage-wise kernel ALONE reported MPCQP_INFEASIBLE / MPCQP_MAX_ITER for a handful of these problems per
# thousand that the oracle -- and an exact backend of the reference (qpmpc/solve_mpc.py:43) -- solves: its active-set operator was the
# explicit inverse of the active rows' Gram matrix, and only solve_mpc's host-side re-solve delivered the right answer (a strict xfail
# stood here). Round 6: the kernel keeps a thin QR factorisation of the whitened active rows (csrc/mpcqp_stagew.hip,
# oracle/stagewise_qr_np.py) -- every family below runs through the DEFAULT dispatch WITHOUT any re-solve (retry_unsolved=False: what
# solve_mpc_batch, PreparedSolve, the closed loops and the C ABI deliver), sixteen seeds each, statuses equal to the oracle's, plans
# within 1e-7.
# ("widef": the same family with constraint matrices FIXED along the horizon, 4 / 8 / 12 / 16 rows per step -- the layout in which the
# wide kernel's forward sweep forms the rows itself, config 5's.)
@pytest.mark.parametrize("kind,tight", [("wide", "0.5"), ("wide", "0.3"), ("wide", "0.15"), ("wide", "0.05"),
                                        ("widef", "0.5"), ("widef", "0.3"), ("widef", "0.15"), ("widef", "0.05"),
                                        ("narrow", "0.5"), ("narrow", "0.3"), ("narrow", "0.15"), ("narrow", "0.05")])
def test_stress_campaign_nearly_fully_active_without_any_re_solve(kind, tight):
    worst, nflag, flagged = _campaign("stress_tight.py", (kind, 8, 8), 0,
                                      {"STRESS_SEEDS": ",".join(str(sd) for sd in range(1, 17)), "STRESS_TIGHT": tight, "STRESS_RETRY": "0"})
    assert nflag == 0, "\n".join(flagged)
    assert worst <= 1e-7, (kind, tight, worst)


@pytest.mark.parametrize("tight", ["0.15", "0.05"])
def test_stagewise_entry_point_takes_the_second_opinion_too(tight):
    """mpcqp_stagewise_solve_batch (formulation="stagewise") sends nx <= 4, nu <= 2 to the narrow kernel at every size; what that kernel
    leaves MPCQP_MAX_ITER / MPCQP_INFEASIBLE is solved by the wide kernel behind it, as in mpcqp_build_solve_batch (include/mpcqp.h)."""
    worst, nflag, flagged = _campaign("stress_tight.py", ("narrow", 8, 8), 0,
                                      {"STRESS_SEEDS": ",".join(str(sd) for sd in range(1, 17)), "STRESS_TIGHT": tight, "STRESS_RETRY": "0",
                                       "STRESS_FORMULATION": "stagewise"})
    assert nflag == 0, "\n".join(flagged)
    assert worst <= 1e-7, (tight, worst)


def test_solve_mpc_delivers_what_an_exact_backend_would_on_known_hard_problems():
    """The problems of stress_tight's wide family at STRESS_TIGHT=0.3 (seeds 1 and 5; 60-78 of ~80 variables pinned) on which the wide
    stage-wise kernel of rounds 2-5 gave up or said `infeasible` although the oracle solves them: the batched entry point on its own
    (no re-solve) and the reference's own entry point, solve_mpc (qpmpc/solve_mpc.py:16-44), both return the oracle's plan."""
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oracle
    from qpmpc_amd import solve_mpc, solve_mpc_batch
    from qpmpc_amd import workloads as W
    from stress_stagewise import random_ltv

    checked = 0
    for seed in (1, 5):
        rng = np.random.default_rng(seed)
        for it in range(8):  # (the generator of tools/stress_tight.py, wide family)
            nx, nu = int(rng.integers(5, 17)), int(rng.integers(1, 5))
            N, mk = int(rng.integers(20, 41)), int(rng.integers(4, 7))
            w = random_ltv(rng, 8, nx, nu, N, mk, 0.3)
            w["A"] = np.eye(nx) + 0.1 * (w["A"] - np.eye(nx))
            raw = solve_mpc_batch(W.to_batch_problem(w))
            torch.cuda.synchronize()
            st = raw.status.cpu().numpy()
            Uo, _, sto, _ = oracle.solve_workload(w)
            assert np.array_equal(st == 0, sto == 0), (seed, it, st.tolist(), sto.tolist())
            if it in (0, 5):  # (one problem per campaign through the single-problem entry point as well)
                b = int(np.flatnonzero(sto == 0)[0]) if (sto == 0).any() else None
                if b is not None:
                    plan = solve_mpc(W.problem_from_workload(w, b), solver="hip_gi")
                    assert not plan.is_empty, (seed, it, b)
                    U = np.asarray(plan.inputs).reshape(-1)
                    assert np.abs(U - Uo[b]).max() <= 1e-7 * max(1.0, np.abs(Uo[b]).max()), (seed, it, b)
                    checked += 1
    assert checked >= 2


def test_stress_campaign_inconsistent_rows():
    """Rows made inconsistent with their bounds (infeasible and borderline problems): statuses must follow the oracle's.
    (On a few solvable-but-degenerate problems of this family ALL formulations, oracle included, only agree to 1e-3 in u --
    DESIGN section 5 --, so the plans are not compared here beyond the tool's own report.)"""
    env = dict(os.environ, STRESS_SEED="5", STRESS_INCONSISTENT="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_dense.py"), "16", "48"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [line for line in out.stdout.splitlines() if line.startswith("nx=")]
    assert len(rows) == 16
    for line in rows:
        agreement = float(re.search(r"agreement ([0-9.]+)", line).group(1))
        assert agreement == 1.0, line
