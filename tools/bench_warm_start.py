#!/usr/bin/env python3
"""Dev/bench: solver warm start on two receding-horizon loops, cold vs warm:
(a) 4096 triple-integrator loops (config-2 family, x+ = A x + B u0, goals fixed), (b) LIPM walking loops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpmpc_amd import PreparedSolve, WarmState, workloads as W
from qpmpc_amd.closed_loop import LIPMWalkingLoop

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
periods = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def triple(warm):
    w = W.triple_integrator_batch(B, heterogeneous=True)
    bp = W.to_batch_problem(w)
    ws = WarmState(bp) if warm else None
    run = PreparedSolve(bp, warm_state=ws) if warm else PreparedSolve(bp)
    A = torch.as_tensor(w["A"][0, 0], device="cuda"); Bm = torch.as_tensor(w["B"][0, 0], device="cuda").reshape(3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    its, t_solve, Us = 0, 0.0, []
    for k in range(periods):
        ev[0].record(); run.launch(); ev[1].record(); torch.cuda.synchronize()
        if k >= 1:
            its += int(run.iters.sum().item()); t_solve += ev[0].elapsed_time(ev[1])
        Us.append(run.U[:, 0].clone())
        bp.initial_state.copy_(bp.initial_state @ A.T + run.U[:, :1] * Bm)
        if warm: run.set_warm_start(warm, warm_shift=2)  # (mk = 2 rows per step)
    return its / (B * (periods - 1)), t_solve / (periods - 1) * 1e3, torch.stack(Us), int((run.status != 0).sum().item())

ic, tc, Uc, fc = triple(False)
iw, tw, Uw, fw = triple(True)
ia, ta, Ua, fa = triple("active_set")
print(f"triple integrator receding horizon, {B} loops x {periods} periods: cold {ic:.2f} iters {tc:.1f} us/period | "
      f"warm(operator) {iw:.2f} iters {tw:.1f} us/period | warm(active set, shifted) {ia:.2f} iters {ta:.1f} us/period | "
      f"max |u0 - u0_cold| {float((Uc - Uw).abs().max()):.2e} / {float((Uc - Ua).abs().max()):.2e} | failed {fc}/{fw}/{fa}")


def resolve(warm, noise):
    """The same batch solved again and again with slightly different states (an outer loop that re-linearises
    or re-targets: the horizon does not shift, so the active rows keep their identity)."""
    w = W.triple_integrator_batch(B, heterogeneous=True)
    bp = W.to_batch_problem(w)
    ws = WarmState(bp) if warm else None
    run = PreparedSolve(bp, warm_state=ws) if warm else PreparedSolve(bp)
    x0 = bp.initial_state.clone()
    g = torch.Generator(device="cuda").manual_seed(5)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    its, t_solve, Us = 0, 0.0, []
    for k in range(periods):
        ev[0].record(); run.launch(); ev[1].record(); torch.cuda.synchronize()
        if k >= 1:
            its += int(run.iters.sum().item()); t_solve += ev[0].elapsed_time(ev[1])
        Us.append(run.U.clone())
        bp.initial_state.copy_(x0 + noise * torch.randn(x0.shape, dtype=x0.dtype, device="cuda", generator=g))
        if warm: run.set_warm_start(warm, warm_shift=0)
    return its / (B * (periods - 1)), t_solve / (periods - 1) * 1e3, torch.stack(Us)

for noise in (0.0, 1e-3, 3e-2):
    ic, tc, Uc = resolve(False, noise)
    iw, tw, Uw = resolve(True, noise)
    ia, ta, Ua = resolve("active_set", noise)
    print(f"config-2 batch re-solved with x0 noise {noise:g}: cold {ic:.2f} iters {tc:.1f} us | warm(operator) {iw:.2f} iters {tw:.1f} us | "
          f"warm(active set) {ia:.2f} iters {ta:.1f} us | max |U - U_cold| {float((Uc - Uw).abs().max()):.2e} / {float((Uc - Ua).abs().max()):.2e}")

rng = np.random.default_rng(1)
strides = np.stack([-rng.uniform(0.12, 0.2, B), rng.uniform(0.12, 0.2, B)], axis=1)
res = {}
for warm in (False, True, "active_set"):
    loop = LIPMWalkingLoop(B, strides=strides, foot_size=rng.uniform(0.05, 0.08, B) * 0 + 0.065, index=np.arange(B) % 8, warm_start=warm)
    loop.step(5); torch.cuda.synchronize()
    it0 = float(loop.iters_total.item())
    t0 = time.perf_counter(); loop.step(periods); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[warm] = (dt / periods * 1e6, (float(loop.iters_total.item()) - it0) / (B * periods), loop.states.clone(), loop.stats()["failed"])
print(f"LIPM walking, {B} walkers x {periods} periods: cold {res[False][1]:.2f} iters {res[False][0]:.1f} us/period | "
      f"warm(operator) {res[True][1]:.2f} iters {res[True][0]:.1f} us/period | warm(active set, shifted) {res['active_set'][1]:.2f} iters {res['active_set'][0]:.1f} us/period | "
      f"max state diff {float((res[False][2]-res[True][2]).abs().max()):.2e} / {float((res[False][2]-res['active_set'][2]).abs().max()):.2e} | failed {res[False][3]}/{res[True][3]}/{res['active_set'][3]}")
