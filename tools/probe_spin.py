#!/usr/bin/env python3
"""Dev probe: the 20-step timed region of config 2 with the HIP runtime's default wait policy against hipDeviceScheduleSpin."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
import torch
if mode != "default":
    lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    rc = lib.hipSetDeviceFlags(ctypes.c_uint({"spin": 1, "yield": 2, "block": 4}[mode]))
    print("hipSetDeviceFlags", mode, "rc", rc)
from qpmpc_amd import PreparedSolve, workloads as W
run = PreparedSolve(W.to_batch_problem(W.triple_integrator_batch(4096)))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.25:
    for _ in range(20): run.launch()
torch.cuda.synchronize()
def region(K):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0.record()
    for _ in range(K): run.launch()
    e1.record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6, e0.elapsed_time(e1) / K * 1e3
r = [region(20) for _ in range(9)]
print(mode, "K=20: wall per step min %.2f median %.2f us; events %.2f us" % (min(a for a, _ in r), sorted(a for a, _ in r)[4], min(b for _, b in r)))
