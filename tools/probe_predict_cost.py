#!/usr/bin/env python3
"""Dev probe: GPU and host cost of the pieces of PreparedModelSolve.predict_order (config 4 through the shared model)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpmpc_amd import SharedModel, pairing_order, workloads as W
from qpmpc_amd.batch import _stream_ptr
bp = W.to_batch_problem(W.humanoid_batch(65536))
run = SharedModel(bp).prepare(bp)
run.launch(); run.predict_order(); torch.cuda.synchronize()
counts, order = run._pred
ops = run._ops
def predict():
    run._lib.mpcqp_model_predict_counts(C.byref(run.model.dims), run.model.model.data_ptr(), None, C.byref(ops[0]), C.byref(ops[1]),
                                        C.byref(ops[2]), 65536, C.byref(run._opts), counts.data_ptr(), _stream_ptr())
def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(reps): fn()
    b.record(); host = (time.perf_counter() - t0) / reps; torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3, host * 1e6
for name, fn in (("predict kernel", predict), ("pairing_order", lambda: pairing_order(counts, out=order)), ("predict_order()", run.predict_order),
                 ("launch (ordered)", run.launch)):
    g, h = timed(fn)
    print(f"{name}: {g:.1f} us between events, host {h:.1f} us per call")
