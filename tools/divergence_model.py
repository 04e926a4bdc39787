#!/usr/bin/env python3
"""Divergence model of sim_kernel on CPU (no GPU needed).

Builds the host emulation of the device code (tests/emu) with region markers (REG(id) in kernel/k_*.h),
runs a batch, and for each marked region reports how often a 64-lane wave executes it per main-loop
iteration (max over lanes: lanes re-converge at the loop top) and what fraction of the lanes are active
in it.  Regions with many trips and low utilisation are where the wave's VALU time goes.
"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from madsim_amd import workload, _abi as A

EMU = os.path.join(ROOT, "tests", "emu")
LIB = os.path.join(EMU, "libmadsim_emu_regions.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DMADSIM_EMU", "-DMADSIM_EMU_REGIONS", "-x", "c++",
                       "-I" + EMU, "-o", LIB, os.path.join(EMU, "emu_driver.cpp")])
L = C.CDLL(LIB)
L.madsim_emu_run_batch.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64, C.POINTER(A.Limits),
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
NAMES = {0: "iteration top", 1: "gen_index attempt", 26: "poll (task popped)", 2: "poll round (insn fetch)",
         3: "[A] sleep deadline check", 4: "[A] recv inbox check", 5: "[A] sleep not elapsed: re-add timer",
         6: "[A] send/reply: try_send", 7: "gen_bool draw (loss)", 8: "latency draw attempt", 12: "[A] completed: post-chain",
         13: "[B] light op", 9: "[C] entry", 14: "[C] recv begin (mailbox)", 15: "[C] rand_delay", 16: "gen_range attempt (rand_delay &c)",
         17: "[C] sleep begin (+timer)", 10: "timer_add", 11: "sift_up trip", 18: "advance 50..100 ns draw attempt",
         19: "fire/idle loop trip", 20: "timer_pop", 21: "sift_down trip", 22: "fire: wake", 23: "fire: deliver",
         24: "deliver: registration scan trip", 25: "result write + seed init"}
count = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cus = int(sys.argv[2]) if len(sys.argv) > 2 else 1
which = sys.argv[3] if len(sys.argv) > 3 else "pingpong"
if which == "pingpong":
    w = workload.pingpong(4, 64)
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots, lim.mbox_regs, lim.mbox_msgs = 4, 0, 1, A.LIMIT_NONE
else:
    w, lim = {"kv": (workload.kv_rpc, workload.kv_rpc_limits), "raft": (workload.raft_election, workload.raft_election_limits),
              "topo": (workload.streaming_topology, workload.streaming_topology_limits)}[which]
    w, lim = w(), lim()
cfg = A.Config.default()
out = np.zeros(count, dtype=A.RESULT_DTYPE)
rc = L.madsim_emu_run_batch(w.ref(), C.byref(cfg), 0, count, C.byref(lim), out.ctypes.data_as(C.c_void_p), cus, None, 0, None)
assert rc == 0
trips = (C.c_double * 32)(); visits = (C.c_double * 32)(); iters = C.c_double()
L.madsim_emu_region_stats(trips, visits, C.byref(iters))
L.madsim_emu_geometry.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Limits), C.POINTER(A.Geometry)]
geo = A.Geometry(); L.madsim_emu_geometry(w.ref(), C.byref(lim), C.byref(geo)); LW = geo.lanes_per_wave
it = iters.value
print(f"{which}: {LW} seed lanes per wave")
print(f"{count} seeds, {it:.0f} wave-iterations, executor steps {int(out['steps'].sum())}, lane-steps per wave-iteration {out['steps'].sum() / it:.1f}")
print(f"{'region':44s} {'wave trips/iter':>16s} {'lane visits/iter':>17s} {'utilisation':>12s}")
for i in sorted(NAMES, key=lambda k: list(NAMES).index(k)):
    if trips[i] == 0: continue
    print(f"{NAMES[i]:44s} {trips[i] / it:16.3f} {visits[i] / it / LW:17.3f} {visits[i] / (LW * trips[i]):12.3f}")
