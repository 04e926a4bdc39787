#!/usr/bin/env python3
"""What would a per-lane FIFO of xoshiro outputs buy the base-op kernel?  (CPU only; a model, nothing of it ships.)

Today every rejection loop runs at the pace of its slowest lane: a wave executes max-over-lanes generator steps at each draw
site.  A lane's output STREAM does not depend on who consumes it, so generation could run lane-uniformly — R passes per
executor pass, every lane whose ring is not full appends one output — while the draw sites pop from the lane's ring (and fall
back to stepping the generator on the spot when it runs dry).  This script takes the host-compiled kernel's per-lane, per-pass
attempt counts of the 4-node ping-pong (tests/emu, MADSIM_EMU_DUMP: gen_index / gen_bool / latency / gen_range / advance
attempts) and replays them against ring depths D and refill passes R.

    python tools/rng_fifo_model.py [seeds=4096]
"""
import ctypes as C, os, subprocess, sys, tempfile
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
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = workload.pingpong(4, 64)
lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots, lim.mbox_regs, lim.mbox_msgs = 4, 0, 1, A.LIMIT_NONE
cfg = A.Config.default()
out = np.zeros(count, dtype=A.RESULT_DTYPE)
dump = tempfile.mktemp(suffix=".bin")
os.environ["MADSIM_EMU_DUMP"] = dump
rc = L.madsim_emu_run_batch(w.ref(), C.byref(cfg), 0, count, C.byref(lim), out.ctypes.data_as(C.c_void_p), 1, None, 0, None)
assert rc == 0
raw = np.fromfile(dump, dtype=np.uint8); os.unlink(dump)
waves, p = [], 0
while p < len(raw):
    iters, lanes = np.frombuffer(raw[p:p + 8].tobytes(), dtype=np.uint32); p += 8
    n = int(iters) * int(lanes) * 5
    waves.append(raw[p:p + n].reshape(int(iters), int(lanes), 5).astype(np.int32)); p += n
GEN, ACC, POP, PUSH = 16, 3, 4, 3          # VALU per generator step / accept test + loop / ring pop (address, count) / ring push
tot_iters = sum(len(x) for x in waves)
# today: a wave pays max-over-lanes attempts at each of the five sites, each attempt = generator step + accept test
today_trips = sum(x.max(axis=1).sum() for x in waves) / tot_iters
need = sum(x.sum() for x in waves) / (tot_iters * waves[0].shape[1])
print(f"4-node ping-pong, {count} seeds, {len(waves)} waves, {tot_iters} wave-passes")
print(f"today: {today_trips:.2f} wave attempts per pass for {need:.2f} outputs a lane consumes -> {today_trips * (GEN + ACC):.0f} VALU per pass in the draw loops")
print(f"{'ring depth':>10s} {'refill passes':>14s} {'generator passes/pass':>22s} {'dry attempts/pass':>18s} {'VALU per pass':>14s} {'vs today':>9s}")
for D in (4, 6, 8, 12, 16):
    for R in (5, 6, 7, 8):
        gen_passes = dry_trips = 0.0
        for x in waves:
            lanes = x.shape[1]
            avail = np.full(lanes, D, dtype=np.int64)
            for it in range(len(x)):
                for _ in range(R):                       # lane-uniform refill: a pass runs if any ring has room
                    room = avail < D
                    if not room.any(): break
                    avail[room] += 1; gen_passes += 1
                dry_site = np.zeros(5, dtype=np.int64)
                for s in range(5):                       # the sites in program order; pops first, then on-the-spot steps
                    c = x[it, :, s]
                    take = np.minimum(c, avail); avail -= take
                    dry_site[s] = (c - take).max()
                dry_trips += dry_site.sum()
        gp, dt = gen_passes / tot_iters, dry_trips / tot_iters
        valu = gp * (GEN + PUSH) + today_trips * (ACC + POP) + dt * (GEN + ACC)
        print(f"{D:10d} {R:14d} {gp:22.2f} {dt:18.2f} {valu:14.0f} {valu / (today_trips * (GEN + ACC)):9.2f}")

# Policy B: no refill passes at all — a lane banks outputs while it idles in SOMEBODY ELSE'S rejection trips (the generator step of a trip is issued
# for the whole wave anyway), and every draw pops its own ring first.  A trip then costs the step + accept test + a push for the idle lanes.
print()
print("policy B: lanes bank outputs during the other lanes' rejection trips (no refill passes)")
print(f"{'ring depth':>10s} {'generator trips/pass':>21s} {'pop trips/pass':>15s} {'VALU per pass':>14s} {'vs today':>9s}")
for D in (1, 2, 3, 4, 5, 6, 8, 12):
    gen_trips = pop_trips = 0.0
    for x in waves:
        lanes = x.shape[1]
        avail = np.zeros(lanes, dtype=np.int64)
        for it in range(len(x)):
            for s in range(5):
                c = x[it, :, s]
                take = np.minimum(c, avail); avail -= take
                pop_trips += take.max()                       # the pop loop runs at its slowest lane too
                r = c - take
                T = int(r.max())
                gen_trips += T
                if T:
                    idle = T - r                                # trips in which the lane has nothing of its own to draw
                    avail += np.minimum(idle, D - avail)
    gt, pt = gen_trips / tot_iters, pop_trips / tot_iters
    valu = gt * (GEN + ACC + PUSH + 2) + pt * (ACC + POP)
    print(f"{D:10d} {gt:21.2f} {pt:15.2f} {valu:14.0f} {valu / (today_trips * (GEN + ACC)):9.2f}")
