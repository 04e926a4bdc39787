#!/usr/bin/env python3
"""Per-phase cycle breakdown from an EXP_PROF build (MADSIM_HIP_LIB=.../libmadsim_hip_prof.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madsim_amd import runtime, workload, _abi as A
import torch
runtime.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "pingpong"       # any bench.py --workload name
w, lim, _ = workload.bench_case(which)
lim.state_mem |= int(os.environ.get("MADSIM_BENCH_STATE_FLAGS", "0"), 0)      # e.g. 0x100 = MADSIM_STATE_DEDUP_TIMERS
buf = torch.empty(65536 * 48, dtype=torch.uint8, device="cuda")
out = (C.c_uint64 * 16)()
L = runtime.lib()
runtime.run_batch_device(w, 0, 65536, buf.data_ptr(), 0, None, lim)
L.madsim_hip_debug_counters(out)
ns = int(os.environ.get("PROF_STREAMS", "0"))          # > 0: that many batches in flight, PROF_ROUNDS rounds (the loaded regime of bench.py)
if ns:
    import time
    streams = [torch.cuda.Stream() for _ in range(ns)]
    bufs = [torch.empty(65536 * 48, dtype=torch.uint8, device="cuda") for _ in range(ns)]
    rounds = int(os.environ.get("PROF_ROUNDS", "3"))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(rounds):
        for i, st in enumerate(streams):
            runtime.run_batch_async(w, (2 + r * ns + i) * 65536, 65536, bufs[i].data_ptr(), 0, st.cuda_stream, None, lim)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    class S: kernel_ms = dt * 1e3 / (rounds * ns)
    s = S()
    print(f"{ns} batches in flight, {rounds} rounds: {s.kernel_ms:.3f} ms wall time per batch")
else:
    s = runtime.run_batch_device(w, 65536, 65536, buf.data_ptr(), 0, None, lim)
L.madsim_hip_debug_counters(out)
v = list(out)
names = ["loop top/seed init+result", "idx draw + ready pop + task load", "poll_task exit (writeback u1)", "writeback + advance draw", "fire/idle loop",
         "poll: entry, u1 load, insn fetch", "poll [A] await check + completion (try_send)", "poll [B] light ops", "poll [C] begin op (mailbox / rare switch)", "behind the poll loop: wait for other lanes' further rounds",
         "poll [C] rand_delay draw + timer_flush (pushes); first round: poll entry"]
waves, iters = v[13], v[12]
tot = sum(v[:12])
if os.environ.get("PROF2"):
    names = ["everything else"] + ["-"] * 9 + ["timer_add (heap push)", "timer_pop (heap pop)"]
print(f"kernel {s.kernel_ms:.3f} ms, waves {waves}, iterations/wave {iters / waves:.0f}, cycles/wave {tot / waves:.0f}")
for i, n in enumerate(names):
    if i >= len(v) or n == "-": continue
    print(f"  {n:36s} {v[i] / waves:12.0f} cycles/wave  {100 * v[i] / tot:5.1f} %   {v[i] / iters:8.0f} cycles/iteration")
