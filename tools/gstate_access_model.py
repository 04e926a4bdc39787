#!/usr/bin/env python3
"""Which parts of the per-lane global state block (Variant::G) a workload touches, per executor step — on CPU.

Builds the host emulation of the kernel (tests/emu) with access counters in gs_load/gs_store, runs a batch and
attributes every access to a region of the block: task units (by unit index), socket header / owner / registrations /
queued messages / accept queue, handles, node, clog, pause, flags, connections.  The regions that dominate are the
candidates for staying in LDS (or for wider loads)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from madsim_amd import workload as W, _abi as A

EMU = os.path.join(ROOT, "tests", "emu")
LIB = os.path.join(EMU, "libmadsim_emu_gstat.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DMADSIM_EMU", "-DMADSIM_EMU_GSTAT", "-x", "c++",
                       "-I" + EMU, "-o", LIB, os.path.join(EMU, "emu_driver.cpp")])
L = C.CDLL(LIB)
L.madsim_emu_run_batch.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64, C.POINTER(A.Limits),
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
L.madsim_emu_gstat.restype = C.c_uint64
L.madsim_emu_gstat.argtypes = [C.c_int, C.c_uint32]
L.madsim_emu_geometry_params.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Limits), C.POINTER(C.c_uint32)]
which = sys.argv[1] if len(sys.argv) > 1 else "raft"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w, lim, _ = W.bench_case(which)
lim.state_mem = A.STATE_GLOBAL
kp = (C.c_uint32 * 32)()
assert L.madsim_emu_geometry_params(w.ref(), C.byref(lim), kp) == 0
names = ["gs_stride", "gs_planes", "max_tasks", "task_units", "n_socks", "sock_words", "mbox_regs", "mbox_msgs", "off_socks", "off_handles",
         "off_nodes", "off_clog", "off_pause", "off_greg", "off_conn", "gs_plane_words", "n_progs"]
P = dict(zip(names, kp))
L.madsim_emu_gstat_reset(P["gs_stride"])
cfg = A.Config.default()
out = np.zeros(count, dtype=A.RESULT_DTYPE)
assert L.madsim_emu_run_batch(w.ref(), C.byref(cfg), 0, count, C.byref(lim), out.ctypes.data_as(C.c_void_p), 1, None, 0, None) == 0
steps = int(out["steps"].sum())


def region(word):
    b = word * 4
    if b < P["gs_planes"]:
        return f"task unit {(b // 16) % P['task_units']}"
    pw = (b - P["gs_planes"]) // 4
    if pw >= P["off_conn"]: return "connections"
    if pw >= P["off_greg"]: return "flags"
    if pw >= P["off_pause"]: return "pause list"
    if pw >= P["off_clog"]: return "clog masks"
    if pw >= P["off_nodes"]: return "node region"
    if pw >= P["off_handles"]: return "join handles"
    f = (pw - P["off_socks"]) % P["sock_words"]
    if f == 0: return "socket header"
    if f == 1: return "socket owner"
    if f < 2 + P["mbox_regs"]: return "socket registrations"
    if f < 2 + P["mbox_regs"] + 2 * P["mbox_msgs"]: return "socket queued messages"
    return "socket accept queue"


tot = {}
for kind, kname in enumerate(["load32", "store32", "load128", "store128"]):
    for word in range(P["gs_stride"] // 4):
        n = L.madsim_emu_gstat(kind, word)
        if n:
            r = region(word)
            tot.setdefault(r, [0, 0, 0, 0])[kind] += n
print(f"{which}: {count} seeds, {steps} executor steps, state block {P['gs_stride']} B/seed "
      f"(tasks {P['gs_planes']} B, planes {P['gs_plane_words'] * 4} B)")
print(f"{'region':28s} {'load32':>9s} {'store32':>9s} {'load128':>9s} {'store128':>9s}   per executor step")
for r, v in sorted(tot.items(), key=lambda kv: -sum(kv[1])):
    print(f"{r:28s} " + " ".join(f"{x / steps:9.3f}" for x in v))
print(f"{'total':28s} " + " ".join(f"{sum(v[k] for v in tot.values()) / steps:9.3f}" for k in range(4)))
