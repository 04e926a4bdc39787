"""CPU oracle loader — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (madsim_amd/) never does: it fails loudly when its
HIP library is missing instead of falling back to anything here.

Parity status: "parity unpinned" (see oracle/madsim_oracle.c header): the Rust
reference cannot be built in this image and ships no golden vectors for the path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from madsim_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmadsim_oracle.so")


class OracleStats(C.Structure):
    _fields_ = [("max_heap", C.c_uint32), ("max_ready", C.c_uint32), ("max_tasks", C.c_uint32),
                ("max_msgs", C.c_uint32), ("max_regs", C.c_uint32), ("max_conns", C.c_uint32), ("max_cq", C.c_uint32)]


def build(force=False):
    src = os.path.join(_HERE, "madsim_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        # MADSIM_ORACLE_LIB: another build of this same oracle (tools/sanitize_check.sh loads an ASan / UBSan one)
        L = C.CDLL(os.environ.get("MADSIM_ORACLE_LIB", _LIB))
        L.madsim_oracle_run_batch.restype = C.c_int
        L.madsim_oracle_run_batch.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64,
                                              C.POINTER(A.Limits), C.c_void_p, C.POINTER(A.Summary),
                                              C.POINTER(OracleStats)]
        L.madsim_oracle_run_batch_pure.restype = C.c_int
        L.madsim_oracle_run_batch_pure.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64,
                                                   C.POINTER(A.Limits), C.c_void_p, C.c_void_p]
        L.madsim_oracle_trace_seed.restype = C.c_int64
        L.madsim_oracle_trace_seed.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64,
                                               C.POINTER(A.Limits), C.c_void_p, C.c_uint64, C.POINTER(A.Result)]
        L.madsim_oracle_observe_seed.restype = C.c_int64
        L.madsim_oracle_observe_seed.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64,
                                                 C.POINTER(A.Limits), C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(A.Result)]
        L.oracle_seed_from_u64.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
        L.oracle_xoshiro_next.restype = C.c_uint64
        L.oracle_xoshiro_next.argtypes = [C.POINTER(C.c_uint64)]
        L.madsim_oracle_gen_range.restype = C.c_uint64
        L.madsim_oracle_gen_range.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        L.oracle_uniform_duration_params.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_int),
                                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                                     C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def run_batch(workload, seed0, count, config=None, limits=None, want_stats=False):
    """Run `count` seeds on the CPU oracle. Returns (results ndarray[RESULT_DTYPE], Summary[, stats])."""
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    out = np.zeros(count, dtype=A.RESULT_DTYPE)
    summ = A.Summary()
    st = OracleStats()
    rc = lib().madsim_oracle_run_batch(workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                       out.ctypes.data_as(C.c_void_p), C.byref(summ), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}")
    return (out, summ, st) if want_stats else (out, summ)


ME_NAMES = {1: "live tasks > 254", 2: "registrations per socket > 255", 4: "registration word aliases a dead one", 8: "queued channel payloads > 15",
            16: "connections waiting for accept1 > 8", 32: "servers per IPVS service > 6", 64: "formatted panic value > panic_dyn_max",
            128: "port-0 entry bound beside its live Endpoint", 256: "op through a port-0 entry that lost its socket",
            512: "connection ends per Endpoint guard > 127", 1024: "ephemeral port beyond the table's candidates", 2048: "live connections > 127",
            4096: "queued messages per mailbox > 255"}


def run_batch_pure(workload, seed0, count, config=None, limits=None):
    """The restatement with the workload model's ceilings OFF (madsim_oracle.c model_event): (results, events) — events[i] is the mask
    of model events seed i met (ME_NAMES); 0 = the seed stayed inside the model and its result is the reference's."""
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    out = np.zeros(count, dtype=A.RESULT_DTYPE)
    ev = np.zeros(count, dtype=np.uint32)
    rc = lib().madsim_oracle_run_batch_pure(workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                            out.ctypes.data_as(C.c_void_p), ev.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}")
    return out, ev


def expected_of_pure(pure, events):
    """What the device runner must report, DERIVED from the pure run: the pure result where the seed stayed inside the workload model,
    the verdict MADSIM_UNSUPPORTED (every other field 0) where it left it."""
    want = pure.copy()
    out = events != 0
    want[out] = np.zeros(1, dtype=A.RESULT_DTYPE)[0]
    want["verdict"][out] = A.UNSUPPORTED
    return want


def trace_seed(workload, seed, config=None, limits=None, cap=1 << 20):
    """Determinism log (rand.rs:64-88) of one seed: (bytes, Result)."""
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    buf = (C.c_uint8 * cap)()
    res = A.Result()
    n = lib().madsim_oracle_trace_seed(workload.ref(), C.byref(cfg), seed, C.byref(lim), buf, cap, C.byref(res))
    if n < 0:
        raise RuntimeError(f"oracle error {n}")
    return bytes(buf[:min(n, cap)]), res


def observe_seed(workload, seed, config=None, limits=None, cap=1 << 16):
    """(list of observed values in execution order — what obs_hash folds —, Result) of one seed."""
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    buf = (C.c_uint64 * cap)()
    res = A.Result()
    n = lib().madsim_oracle_observe_seed(workload.ref(), C.byref(cfg), seed, C.byref(lim), buf, cap, C.byref(res))
    if n < 0:
        raise RuntimeError(f"oracle error {n}")
    return [int(v) for v in buf[:min(n, cap)]], res
