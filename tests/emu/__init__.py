"""Host emulation of the device code (tests/emu/emu_shim.h): a debugging aid for GPU-less boxes.

Test-only.  Never imported by madsim_amd/.  Agreement of the emulation with the oracle checks the
kernel's *logic* on CPU; the GPU build is checked by the `-m gpu` tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from madsim_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = os.path.join(_HERE, "libmadsim_emu.so")
_SRCS = [os.path.join(_HERE, "emu_driver.cpp"), os.path.join(_HERE, "emu_shim.h"),
         os.path.join(_ROOT, "madsim_amd", "csrc", "sim_kernel.hip"),
         os.path.join(_ROOT, "madsim_amd", "csrc", "sim_kernel.h"),
         *sorted(__import__("glob").glob(os.path.join(_ROOT, "madsim_amd", "csrc", "kernel", "*.h"))),
         os.path.join(_ROOT, "madsim_amd", "csrc", "geometry.h"),
         os.path.join(_ROOT, "include", "madsim_hip.h")]


def _stale():
    return not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in _SRCS)


def build():
    """(pytest-xdist workers may arrive together: one builds under a file lock into a temporary name, the others wait)"""
    if _stale():
        import fcntl
        with open(_LIB + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if _stale():
                tmp = f"{_LIB}.{os.getpid()}.tmp"
                subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DMADSIM_EMU", "-x", "c++",
                                       "-I" + _HERE, "-o", tmp, _SRCS[0]])
                os.replace(tmp, _LIB)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.environ.get("MADSIM_EMU_LIB") or build())      # MADSIM_EMU_LIB: a sanitizer build (tools/sanitize_check.sh)
        L.madsim_emu_run_batch.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64,
                                           C.POINTER(A.Limits), C.c_void_p, C.c_int, C.c_void_p, C.c_uint64,
                                           C.POINTER(C.c_uint64)]
        L.madsim_emu_last_error.restype = C.c_char_p
        L.madsim_emu_geometry.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Limits), C.POINTER(A.Geometry)]
        _lib = L
    return _lib


def run_batch(workload, seed0, count, config=None, limits=None, num_cus=2):
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    out = np.zeros(count, dtype=A.RESULT_DTYPE)
    rc = lib().madsim_emu_run_batch(workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                    out.ctypes.data_as(C.c_void_p), num_cus, None, 0, None)
    if rc:
        raise RuntimeError(f"emu error {rc}: {lib().madsim_emu_last_error().decode()}")
    return out


def geometry_params(workload, limits=None):
    """Selected KParams of the geometry a workload gets: the layout words tools/gstate_access_model.py reads, then the op classes
    (MADSIM_FEAT_* mask), whether the state lives in global memory, and the de-duplication table (buckets, byte offset)."""
    names = ["gs_stride", "gs_planes", "max_tasks", "task_units", "n_socks", "sock_words", "mbox_regs", "mbox_msgs", "off_socks",
             "off_handles", "off_nodes", "off_clog", "off_pause", "off_greg", "off_conn", "gs_plane_words", "n_progs",
             "features", "gstate_mode", "dedup_n", "dedup_off", "narrow", "pool_n", "heap_lds", "heap_spill", "lds_per_seed"]
    kp = (C.c_uint32 * 32)()
    L = lib()
    L.madsim_emu_geometry_params.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Limits), C.POINTER(C.c_uint32)]
    rc = L.madsim_emu_geometry_params(workload.ref(), C.byref(limits or A.Limits()), kp)
    if rc:
        raise RuntimeError(L.madsim_emu_last_error().decode())
    return dict(zip(names, kp))


def trace_seed(workload, seed, config=None, limits=None, cap=1 << 20):
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    out = np.zeros(1, dtype=A.RESULT_DTYPE)
    buf = (C.c_uint8 * cap)()
    n = C.c_uint64(0)
    rc = lib().madsim_emu_run_batch(workload.ref(), C.byref(cfg), seed, 1, C.byref(lim),
                                    out.ctypes.data_as(C.c_void_p), 1, buf, cap, C.byref(n))
    if rc:
        raise RuntimeError(f"emu error {rc}: {lib().madsim_emu_last_error().decode()}")
    return bytes(buf[:min(n.value, cap)]), out[0]


def geometry(workload, limits=None):
    g = A.Geometry()
    rc = lib().madsim_emu_geometry(workload.ref(), C.byref(limits or A.Limits()), C.byref(g))
    if rc:
        raise RuntimeError(lib().madsim_emu_last_error().decode())
    return g
