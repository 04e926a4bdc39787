#!/usr/bin/env python3
"""Where does madsim_hip_run_batch(count = 262 144) spend its wall time?  summary-only call (no per-seed copies) / caller's array
pre-touched / fresh (the Python mirror's np.zeros).  Run on a GPU box: python tools/experiment/exp_r5_runbatch.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (one ROCm runtime)
from madsim_amd import runtime, workload, _abi as A
runtime.init(0)
L = runtime.lib()
w, lim, _ = workload.bench_case("pingpong")
cfg = A.Config.default()
n = 262144
def call(out, seed0):
    s = A.Summary()
    t = time.perf_counter()
    rc = L.madsim_hip_run_batch(w.ref(), C.byref(cfg), seed0, n, C.byref(lim), out.ctypes.data_as(C.c_void_p) if out is not None else None, C.byref(s))
    dt = time.perf_counter() - t
    assert rc == 0, L.madsim_hip_last_error()
    return dt, s
buf = np.zeros(n, dtype=A.RESULT_DTYPE); buf[:] = buf       # touched
for mode in ("null", "touched", "fresh", "null", "touched", "fresh"):
    ts = []
    for r in range(9):
        out = None if mode == "null" else buf if mode == "touched" else np.empty(n, dtype=A.RESULT_DTYPE)
        dt, s = call(out, (1 << 40) + r * n)
        ts.append(dt)
    ts.sort()
    print(f"{mode:8s} median {ts[4]*1e3:.3f} ms  min {ts[0]*1e3:.3f} ms  -> {n/ts[4]/1e6:.1f} M seeds/s  kernel_ms {s.kernel_ms:.3f}")
for cnt in (65536, 131072, 262144, 524288, 1048576):
    b2 = np.zeros(cnt, dtype=A.RESULT_DTYPE)
    ts = []
    for r in range(5):
        s = A.Summary(); t = time.perf_counter()
        L.madsim_hip_run_batch(w.ref(), C.byref(cfg), (1 << 41) + r * cnt, cnt, C.byref(lim), b2.ctypes.data_as(C.c_void_p), C.byref(s)); ts.append(time.perf_counter() - t)
    ts.sort(); print(f"count {cnt:8d}: {ts[2]*1e3:.3f} ms  {cnt/ts[2]/1e6:.1f} M seeds/s")
