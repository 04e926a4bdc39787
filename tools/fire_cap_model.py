#!/usr/bin/env python3
"""Would capping the timer fires of a pass pay in the global-state builds?  (CPU only; a model.)

A wave stays in the fire loop until its slowest lane has popped every due timer: the topology pops 3.9 times per pass at wave level for 1.09 per
lane, each pop a sift through the spilled heap levels.  Cap the pops of a pass at K: a lane with more due timers carries them into the next pass
(it must not poll in between: Timer::expire fires everything due before run_all_ready goes on), i.e. it loses that pass's poll.
Replays the host-compiled kernel's per-lane pop / sift-down counts per pass (MADSIM_EMU_DUMP with region ids 20, 21) under caps.

    python tools/fire_cap_model.py topo|raft [seeds=1024] [fire share of a pass's time, default from the round-4 phase profile]

MEASURED (round 5, tools/experiment/r5_fire_cap.patch, gpurun_out/r5v; bit-exact in the emulation at caps 1 and 2, every GPU line oracle-verified):
the model says topology x0.91 at a cap of 2, election loop x0.88 at 1; the GPU says topology 4.72 / 4.51 / 4.71 G steps/s at caps 3 / 2 / 1 against 4.73
without, election loop 8.93 / 8.83 / 8.69 against 8.95, KV 12.3 / 12.3 / 12.1 against 12.7 (its build spills three more registers).  Wave trips are the
wrong currency for these latency-bound kernels: a lane's pops cost dependent round trips whichever pass they run in.  Dropped."""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from madsim_amd import workload, _abi as A
EMU = os.path.join(ROOT, "tests", "emu")
LIB = os.path.join(EMU, "libmadsim_emu_regions.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DMADSIM_EMU", "-DMADSIM_EMU_REGIONS", "-x", "c++", "-I" + EMU, "-o", LIB, os.path.join(EMU, "emu_driver.cpp")])
L = C.CDLL(LIB)
L.madsim_emu_run_batch.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64, C.POINTER(A.Limits), C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
which = sys.argv[1] if len(sys.argv) > 1 else "topo"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
fire_share = float(sys.argv[3]) if len(sys.argv) > 3 else {"topo": 0.33, "raft": 0.38}[which]
w, lim, _ = workload.bench_case(which)
cfg = A.Config.default()
out = np.zeros(count, dtype=A.RESULT_DTYPE)
dump = tempfile.mktemp(suffix=".bin")
os.environ["MADSIM_EMU_DUMP"] = dump; os.environ["MADSIM_EMU_DUMP_IDS"] = "20,21,19,10,11"
assert L.madsim_emu_run_batch(w.ref(), C.byref(cfg), 0, count, C.byref(lim), out.ctypes.data_as(C.c_void_p), 1, None, 0, None) == 0
raw = np.fromfile(dump, dtype=np.uint8); os.unlink(dump)
waves, p = [], 0
while p < len(raw):
    iters, lanes = np.frombuffer(raw[p:p + 8].tobytes(), dtype=np.uint32); p += 8
    n = int(iters) * int(lanes) * 5
    waves.append(raw[p:p + n].reshape(int(iters), int(lanes), 5).astype(np.int64)); p += n
tot_it = sum(len(x) for x in waves)
pops_wave = sum(x[:, :, 0].max(axis=1).sum() for x in waves) / tot_it
pops_lane = sum(x[:, :, 0].sum() for x in waves) / (tot_it * waves[0].shape[1])
sift_wave = sum(x[:, :, 1].max(axis=1).sum() for x in waves) / tot_it
print(f"{which}: {count} seeds, {len(waves)} waves, {tot_it} wave-passes; pops per pass: wave {pops_wave:.2f}, lane {pops_lane:.2f}; sift-down trips per pass (wave) {sift_wave:.2f}")
hist = np.bincount(np.concatenate([x[:, :, 0].ravel() for x in waves]), minlength=8)
print("pops per lane-pass histogram:", (hist / hist.sum()).round(4).tolist()[:10])
# cost model: a pass costs (1 - fire_share) for its poll part + fire_share * (sift trips of the pass / today's average); a lane's excess pops
# carry over (it skips the poll of the next pass: its step sequence shifts by one pass)
for K in (0, 4, 3, 2, 1):
    passes = cost = 0.0
    for x in waves:
        S, lanes = x.shape[0], x.shape[1]
        step = np.zeros(lanes, dtype=np.int64); debt_p = np.zeros(lanes, dtype=np.int64); debt_s = np.zeros(lanes, dtype=np.float64)
        idx = np.arange(lanes)
        while (step < S).any() or (debt_p > 0).any():
            live = step < S
            fresh = live & (debt_p == 0)                       # lanes that poll this pass and get their new due timers
            s_ = np.minimum(step, S - 1)
            newp = np.where(fresh, x[s_, idx, 0], 0); news = np.where(fresh, x[s_, idx, 1], 0).astype(np.float64)
            havep = debt_p + newp; haves = debt_s + news
            dop = np.minimum(havep, K) if K else havep
            frac = np.divide(dop, havep, out=np.zeros(lanes), where=havep > 0)
            dos = haves * frac
            cost += (1 - fire_share) * (1.0 if fresh.any() else 0.3) + fire_share * dos.max() / sift_wave
            debt_p = havep - dop; debt_s = haves - dos
            step = np.where(fresh, step + 1, step)
            passes += 1
    print(f"cap {K or 'none':>4}: wave-passes x{passes / tot_it:.3f}, time x{cost / tot_it:.3f}")
