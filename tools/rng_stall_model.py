#!/usr/bin/env python3
"""What would CAPPING the rejection loops buy the base-op kernel?  (CPU only; a model — the measurement is the A/B on a GPU box.)

A wave runs every rejection loop to its slowest lane: 7.3 trips at the ready-queue draw for the 2.0 a lane needs.  But the lanes of a wave are
independent simulations — nothing says they must take the same executor step in the same pass.  Cap the loop at K trips: a lane that has not
accepted by then keeps its generator state, skips the rest of the pass (no poll, no clock advance) and goes on drawing in the same loop when
the wave comes round again.  The wave's loop is short; the straggler loses a pass.  Per lane nothing changes (same outputs consumed in the
same order); what changes is how many passes the wave needs until its slowest lane is through.

Replays the host-compiled kernel's per-lane, per-step attempt counts of the 4-node ping-pong (tests/emu, MADSIM_EMU_DUMP: ready-queue draw,
gen_bool, latency, gen_range(rand_delay), advance) under caps on the ready-queue draw and the advance draw.

    python tools/rng_stall_model.py [seeds=2048]

MEASURED (round 5, tools/experiment/r5_draw_caps.patch, gpurun_out/r5k, bit-exact, 1 280 verified seeds per line): the model's best point
(caps 4 / 2: 0.885 of the VALU instructions) runs 1.43 ms per batch against 1.19 for the uncapped loops (+20 %); caps 6 / 3: 1.36; 3 / 2: 1.51.
Two things the model does not price: a pass costs ~57 rejection trips' worth of time, not the 22 its instruction count suggests (the
always-accept build of the same round: all 12.8 excess trips gone = -22.5 % of the time), so 15 % more passes cost more than the trips save;
and lanes that leave lock-step stop sharing the poll handlers' instructions.  Dropped."""
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
count = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w = workload.pingpong(4, 64)
lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots, lim.mbox_regs, lim.mbox_msgs = 4, 0, 1, A.LIMIT_NONE
cfg = A.Config.default()
out = np.zeros(count, dtype=A.RESULT_DTYPE)
dump = tempfile.mktemp(suffix=".bin")
os.environ["MADSIM_EMU_DUMP"] = dump
assert L.madsim_emu_run_batch(w.ref(), C.byref(cfg), 0, count, C.byref(lim), out.ctypes.data_as(C.c_void_p), 1, None, 0, None) == 0
raw = np.fromfile(dump, dtype=np.uint8); os.unlink(dump)
waves, p = [], 0
while p < len(raw):
    iters, lanes = np.frombuffer(raw[p:p + 8].tobytes(), dtype=np.uint32); p += 8
    n = int(iters) * int(lanes) * 5
    waves.append(raw[p:p + n].reshape(int(iters), int(lanes), 5).astype(np.int64)); p += n
TRIP, FIXED = 19, 83                      # VALU per rejection trip; VALU per pass outside the draw loops (421 - 17.8 x 19)
POP, ADV = 0, 4                           # columns of the dump: ready-queue draw, advance draw (1..3 = the draws inside the poll)

def replay(kpop, kadv):
    """-> (wave-passes, VALU) summed over the waves.  kpop / kadv = trip caps (0 = none)."""
    passes = valu = 0
    for x in waves:
        S, lanes = x.shape[0], x.shape[1]
        step = np.zeros(lanes, dtype=np.int64)          # executor step each lane is at
        rem_pop = x[0, :, POP].copy(); rem_adv = np.zeros(lanes, dtype=np.int64)
        stage = np.zeros(lanes, dtype=np.int64)         # 0 = at the ready-queue draw, 1 = at the advance draw
        idx = np.arange(lanes)
        while (step < S).any():
            live = step < S
            cost = FIXED
            # ready-queue draw
            at_pop = live & (stage == 0)
            t = np.where(at_pop, np.minimum(rem_pop, kpop) if kpop else rem_pop, 0)
            cost += TRIP * t.max()
            rem_pop -= t
            polled = at_pop & (rem_pop == 0)
            # the poll's own draws: uncapped, paid at the slowest polling lane per site
            s_ = np.minimum(step, S - 1)
            for col in (1, 2, 3):
                cost += TRIP * np.where(polled, x[s_, idx, col], 0).max()
            rem_adv = np.where(polled, x[s_, idx, ADV], rem_adv)
            stage = np.where(polled, 1, stage)
            # advance draw (lanes that polled this pass, or that stalled here before)
            at_adv = live & (stage == 1)
            t = np.where(at_adv, np.minimum(rem_adv, kadv) if kadv else rem_adv, 0)
            cost += TRIP * t.max()
            rem_adv -= t
            done = at_adv & (rem_adv == 0)
            step = np.where(done, step + 1, step)
            stage = np.where(done, 0, stage)
            nxt = np.minimum(step, S - 1)
            rem_pop = np.where(done, x[nxt, idx, POP], rem_pop)
            passes += 1; valu += cost
    return passes, valu

base_p, base_v = replay(0, 0)
print(f"4-node ping-pong, {count} seeds, {len(waves)} waves; today: {base_p} wave-passes, {base_v / base_p:.0f} VALU per pass (model)")
print(f"{'cap: ready-queue draw':>22s} {'advance draw':>13s} {'wave-passes':>12s} {'VALU per pass':>14s} {'total VALU vs today':>20s}")
for kpop in (0, 6, 5, 4, 3, 2):
    for kadv in (0, 3, 2, 1):
        pz, vz = replay(kpop, kadv)
        print(f"{kpop or 'none':>22} {kadv or 'none':>13} {pz / base_p:12.3f} {vz / pz:14.0f} {vz / base_v:20.3f}")
