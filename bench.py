#!/usr/bin/env python3
"""bench.py — BASELINE.json metric on BASELINE.json's config.

One "step" = one pass of the hot path over one batch: 65 536 seeds of the 4-node ping-pong
(R = 64 rounds per pair, Config::default()) per GPU, i.e. BASELINE.json configs[1].  With N GPUs each
rank runs its own contiguous block of 65 536 seeds (weak scaling, no data-path collective) and the
ranks exchange one 64-byte report all-gather per step (RCCL).

`python bench.py --gpus N` with N > 1 and no launcher environment re-executes itself as N ranks under
torch.distributed.run (one process per GPU); under a launcher, WORLD_SIZE must equal --gpus.  `n_gpus` in the
output is the size of the process group that actually ran, never the flag.

Prints ONE JSON line (rank 0).  `value` = simulated seconds per wall second summed over every seed of
every rank; seeds/s and executor-steps/s ride along in `extra`.  After the timed region (never inside it) the
line gains `verified_seeds` (sampled seeds of the last batch on every stream compared bit-for-bit with the CPU
oracle), a first-failing-seed measurement on the packet-loss variant, and the CPU baseline.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from madsim_amd import launch  # noqa: E402  (pure host logic, no torch)

ALGO_BYTES_PER_STEP = 120      # pop 16 + push 16 + rng 32r+32w + clock 8r+8w + ready 4r+4w
IO_BYTES_PER_SEED = 8 + 48     # seed in, madsim_result_t out
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
REPORT_WORDS = 8               # int64 per step: 4 from the library's reduction kernel + rank, device, 2 spare


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)      # ~1.5 s timed at 1.46 ms per batch (65.5 M seeds)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--seeds", type=int, default=0, help="seeds per GPU per step (default 65 536)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed regions of exactly --steps steps each, back to back, every one bracketed by barrier + synchronize; the "
                         "line's ms_per_step / value are those of the MEDIAN region, all of them are listed in extra.regions "
                         "(default: 21 for short regions, fewer for long ones — about 2 000 steps in total, at least 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of sampled seeds after the timed region")
    ap.add_argument("--no-first-fail", action="store_true", help="skip the first-failing-seed measurements (loss variants)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip extra.workloads (a few timed steps each of the raft / kv / topo / timers workloads after the headline)")
    ap.add_argument("--rare-loss", type=float, default=1.2e-6,
                    help="packet_loss_rate of the rare-failure search: 256 datagrams per seed => ~3e-4 of the seeds deadlock")
    ap.add_argument("--very-rare-loss", type=float, default=2e-8,
                    help="packet_loss_rate of the very-rare-failure search: ~5e-6 of the seeds deadlock, i.e. one per ~3 batches")
    ap.add_argument("--measure-traffic", action="store_true", default=None,
                    help="collect FETCH_SIZE / WRITE_SIZE of this same command with two short rocprofv3 --pmc passes after the "
                         "timed region (default at one GPU when rocprofv3 is on PATH; a few seconds)")
    ap.add_argument("--no-measure-traffic", dest="measure_traffic", action="store_false")
    ap.add_argument("--loss", type=float, default=0.0, help="packet_loss_rate of the timed batches (0 = Config::default())")
    ap.add_argument("--first-fail-loss", type=float, default=0.01, help="packet_loss_rate of the first-fail leg (SURVEY 8d)")
    ap.add_argument("--sched", type=int, default=0, help="0 = static seed striding, 1 = per-launch atomic work queue (madsim_limits_t.sched)")
    ap.add_argument("--state-mem", type=int, default=0, help="madsim_limits_t.state_mem: 0 auto, 1 LDS, 2 global-memory state block")
    ap.add_argument("--lpw", type=int, default=0, help="seed-carrying lanes per wave (0 = library auto)")
    ap.add_argument("--nodes", type=int, default=4, help="ping-pong nodes (experiments; the bench line is quoted on 4)")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the steps are spread over (0 = pick 3 or 2 by a short trial during warm-up). One 65 536-seed "
                         "batch is 1 024 waves = one per SIMD; the 200 B of LDS per seed admit three waves per SIMD, so three "
                         "batches are kept in flight (measured: 3.66 / 1.96 / 1.47 / 2.03 ms per batch with 1 / 2 / 3 / 4 streams); "
                         "on a box whose HIP streams share hardware queues two can be better, hence the trial")
    ap.add_argument("--heap-lds", type=int, default=4, help="timer-heap entries kept in LDS (the rest spill to HBM)")
    ap.add_argument("--generic", action="store_true", help="force the generic kernel variant (HBM heap spill enabled)")
    ap.add_argument("--workload", default="pingpong", choices=["pingpong", "raft", "kv", "timers", "topo"],
                    help="pingpong = BASELINE configs[1] (the headline); raft / kv / topo = configs[2] / [3] / [4]-shaped extras")
    return ap.parse_args(argv)


PMC_ISSUE = "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"


def measure_counters(argv, passes=("TRACE", "FETCH_SIZE", "WRITE_SIZE", None), extra_args=()):
    """Per-launch hardware counters of sim_kernel for this same command, from short rocprofv3 passes (--kernel-trace only, never
    combined with other traces): "TRACE" = no counters, the kernel trace alone — the launches overlap as in the timed region, and
    the average sim_kernel duration of that pass (`launch_ms_rocprof`) is what `rocprofv3 --kernel-trace --stats` reports for the
    same command; FETCH_SIZE and WRITE_SIZE (separate --pmc passes, MI355X_MICROARCH.md §PMC slots) give the HBM bytes; None = the
    pass of SQ counters (instruction counts, VALU lane utilisation).  `extra_args` are appended to the child command (another
    --workload).  Returns (dict, None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    base = [a for a in argv if a != "--measure-traffic"] + ["--no-measure-traffic"]
    for flag in ("--steps", "--warmup", "--workload"):
        if flag in base:
            k = base.index(flag)
            del base[k:k + 2]
    base += ["--no-cpu-baseline", "--no-verify", "--no-first-fail", "--no-extras", *extra_args]
    vals = {}
    tmp = tempfile.mkdtemp(prefix="madsim_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, ctrs in enumerate(passes):
            d = os.path.join(tmp, f"pass{i}")
            if ctrs == "TRACE":
                # the kernel trace alone: launches overlap as in the timed region (a --pmc pass serialises them)
                cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                       sys.executable, os.path.join(ROOT, "bench.py")] + base + ["--steps", "40", "--warmup", "10", "--repeats", "1"]
                try:
                    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
                except (OSError, subprocess.TimeoutExpired) as e:
                    return None, f"rocprofv3 kernel-trace pass failed: {e}"
                durs = []
                for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if "sim_kernel" in row.get("Kernel_Name", ""):
                            durs.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
                if durs:
                    vals["launch_ms_rocprof"] = sum(durs) / len(durs) * 1e-6
                    vals["launches_traced"] = len(durs)
                continue
            ctrs = ctrs or PMC_ISSUE
            child = base + ["--steps", "6", "--warmup", "2"]
            cmd = [exe, "--kernel-trace", "--pmc", *ctrs.split(), "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "bench.py")] + child
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, f"rocprofv3 pass {i} ({ctrs}) failed: {e}"
            xs = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "sim_kernel" in row.get("Kernel_Name", ""):
                        xs.setdefault(row.get("Counter_Name"), []).append(float(row["Counter_Value"]))
            for c in ctrs.split():
                if c not in xs:
                    if c in ("FETCH_SIZE", "WRITE_SIZE"):
                        return None, f"no {c} rows for sim_kernel"
                    continue                      # an SQ counter this rocprofv3 does not know: the issue roofline degrades, traffic stays
                vals[c] = sum(xs[c]) / len(xs[c])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    vals["source"] = ("live: rocprofv3 --kernel-trace [--pmc ..], one pass per counter group of this command (6 steps; the trace-only pass 40), "
                      "mean per sim_kernel dispatch")
    return vals, None


def issue_ceiling():
    """The chip's sustained VALU issue rate for the executor's instruction mix (wave-instructions per second by wall time):
    tools/ubench_issue --quick run live when the binary is there (build() compiles it), else the committed measurement
    (profiles/r3_issue_ceiling.json)."""
    exe = os.path.join(ROOT, "tools", "ubench_issue")
    if os.path.exists(exe):
        try:
            out = subprocess.run([exe, "--quick"], capture_output=True, text=True, timeout=60)
            j = json.loads(out.stdout.strip().splitlines()[-1])
            j["source"] = "live: tools/ubench_issue --quick on this GPU"
            return j
        except (OSError, subprocess.TimeoutExpired, ValueError, IndexError):
            pass
    path = os.path.join(ROOT, "profiles", "r3_issue_ceiling.json")
    if os.path.exists(path):
        j = json.load(open(path))
        j["source"] = "profiles/r3_issue_ceiling.json (tools/ubench_issue on an MI355X, not this run)"
        return j
    return None


def main():
    args = parse_args()
    try:
        mode, cmd = launch.plan(args.gpus, os.environ, sys.argv[1:], os.path.abspath(__file__))
    except launch.LaunchError as e:
        print(f"bench.py: {e}", file=sys.stderr)
        return 2
    if mode == "spawn":                      # python bench.py --gpus N: become N ranks, one per GPU
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return subprocess.call(cmd, env=env)

    # Three batches in flight need three hardware queues: ROCclr multiplexes HIP streams onto GPU_MAX_HW_QUEUES (default 4)
    # of them, and on some boxes two of this process's streams ended up sharing one (3 streams slower than 2).  More queues
    # make that less likely; the warm-up trial below still decides between 3 and 2 by measurement.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # torch's streams of the timed loop (up to 5) + the library's own of the campaign legs (up to 5)
    import numpy as np
    import torch
    import torch.distributed as dist
    from madsim_amd import _abi as A
    from madsim_amd import dist as mdist
    from madsim_amd import runtime, workload

    rank, local_rank, world = launch.rank_env(os.environ)
    # MADSIM_BENCH_BACKEND=gloo is a functional-test hook (several ranks sharing one GPU on a 1-GPU box);
    # the real multi-GPU run is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("MADSIM_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        print("bench.py: no GPU visible (there is no CPU path to benchmark)", file=sys.stderr)
        return 2
    if world > 1 and backend == "nccl" and n_dev < world:
        print(f"bench.py: {world} ranks but only {n_dev} GPUs visible", file=sys.stderr)
        return 2
    gpu = local_rank % n_dev
    torch.cuda.set_device(gpu)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            try:     # RCCL kernels on a high-priority stream: they take the first CU slot a finishing simulation wave frees
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group("nccl", device_id=torch.device("cuda", gpu), pg_options=opts)
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus, world)
    n_ranks = dist.get_world_size() if world > 1 else 1
    dev = torch.device("cuda", gpu)
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where the report tensors live
    runtime.init(gpu)

    w, lim, wname = workload.bench_case(args.workload, args.nodes, workload.BENCH_ROUNDS, args.heap_lds)
    headline = args.workload == "pingpong" and args.nodes == workload.BENCH_NODES
    lim.lanes_per_wave = args.lpw or int(os.environ.get("MADSIM_BENCH_LPW", 0)) or lim.lanes_per_wave       # (a bench case may bring its own: the election loop runs 32 seed lanes per wave)
    lim.state_mem = args.state_mem or lim.state_mem
    if "MADSIM_BENCH_STATE_FLAGS" in os.environ:         # experiments: OR into / clear from state_mem (e.g. 0x100 = MADSIM_STATE_DEDUP_TIMERS; "-0x100" clears it)
        v = os.environ["MADSIM_BENCH_STATE_FLAGS"]
        lim.state_mem = (lim.state_mem & ~int(v[1:], 0)) if v.startswith("-") else (lim.state_mem | int(v, 0))
    if "MADSIM_BENCH_CLEAR_FLAGS" in os.environ:         # experiments: clear bits of state_mem (beside MADSIM_BENCH_STATE_FLAGS, which sets some)
        lim.state_mem &= ~int(os.environ["MADSIM_BENCH_CLEAR_FLAGS"], 0)
    if "MADSIM_BENCH_HEAP_LDS" in os.environ:            # experiments: move timer-heap entries between LDS and the HBM spill region
        n = int(os.environ["MADSIM_BENCH_HEAP_LDS"])
        lim.heap_spill_slots, lim.heap_lds_slots = lim.heap_spill_slots + max(0, lim.heap_lds_slots - n), n
    lim.sched = int(os.environ.get("MADSIM_BENCH_SCHED", args.sched))     # work distribution inside a launch (experiments)
    if args.generic:
        lim.heap_spill_slots = max(lim.heap_spill_slots, 8)
    cfg = A.Config.default(packet_loss_rate=args.loss) if args.loss else None
    per_gpu = args.seeds or workload.BENCH_SEEDS_PER_GPU
    total = per_gpu * n_ranks
    seed0, count = mdist.shard_range(0, total, rank, n_ranks)
    g0 = runtime.geometry(w, lim)
    waves_cu = g0.blocks_per_cu * g0.block_threads // 64
    # batches in flight: one wave per SIMD each, so as many as the workload's LDS admits waves per SIMD — and one more, whose
    # launch queues behind them and fills the gaps their tails leave (measured: tools/experiment/exp_compact.sh)
    # (global-state builds with a heap-spill region — long launches that end with their slowest wave: three resident and a fourth
    # behind them, +5 % on the election loop and the topology; the KV build, whose heap sits in LDS, gains nothing from it:
    # tools/experiment/exp_gstreams.sh)
    def flights(g):
        wcu = g.blocks_per_cu * g.block_threads // 64
        # (round 6, profiles/r6_ab_launches_in_flight.txt: the election loop at three waves per SIMD +2 % with a fifth launch queued, the topology
        #  at two waves per SIMD -1 %; sub-launch sizes from 16 384 to 131 072 seeds read the same within 1 %)
        # (the round's last kernels, tools/experiment/exp_r6_streams_final.sh: the topology — two waves per SIMD — reads 5.91-5.93 G steps/s with three
        #  launches in flight, 5.87 with four, 5.79-5.82 with five; the election loop 10.28-10.30 with five against 10.14 with four; the KV three)
        return 5 if wcu >= 16 else (5 if wcu >= 12 else 3) if (g.variant & 16) and g.heap_spill_slots else 3
    max_streams = args.streams if args.streams > 0 else flights(g0)
    n_streams = max_streams
    d_outs = [torch.empty(count * 48, dtype=torch.uint8, device=dev) for _ in range(max_streams)]   # results stay in HBM
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(max_streams - 1)]
    report_stream = torch.cuda.Stream() if world > 1 else None

    # Fully asynchronous steps: the simulation kernel, the summary reduction and (N > 1) the RCCL all-gather of the
    # report are all queued on streams; the host never waits inside the timed region.
    use_device_report = world == 1 or backend == "nccl"
    repeats = args.repeats if args.repeats > 0 else max(3, min(21, 2000 // max(1, args.steps)))
    n_rows = args.steps * repeats + args.warmup
    ring = torch.zeros((n_rows, REPORT_WORDS), dtype=torch.int64, device=dev)   # one report per step
    ring[:, 4] = rank                      # identity words: the gathered report must hold one row per rank
    ring[:, 5] = gpu
    gathered = torch.zeros((n_rows, n_ranks, REPORT_WORDS), dtype=torch.int64, device=dev) if world > 1 else None
    last_on_stream = {}

    def step(k, timed):
        # a fresh block of seeds every step so nothing is cached between steps
        if use_device_report:
            si = k % n_streams
            with torch.cuda.stream(streams[si]):
                runtime.run_batch_async(w, seed0 + k * total, count, d_outs[si].data_ptr(), ring[k].data_ptr(),
                                        streams[si].cuda_stream, cfg, lim, timing_slot=(k % 64) if timed else -1)
            last_on_stream[si] = k
            if world > 1:
                # the RCCL exchange rides its own stream behind an event: the simulation streams never wait
                # for a collective kernel to find room on a chip whose LDS the simulation keeps full
                ev = torch.cuda.Event()
                ev.record(streams[si])
                report_stream.wait_event(ev)
                with torch.cuda.stream(report_stream):
                    mdist.gather_report_device(ring[k], gathered[k])     # ONE all-gather of 64 bytes per step
        else:   # functional-test hook (gloo on a 1-GPU box): host-side report
            sm = runtime.run_batch_device(w, seed0 + k * total, count, d_outs[0].data_ptr(), streams[0].cuda_stream, cfg, lim)
            last_on_stream[0] = k
            rep = mdist.reduce_report(sm.first_failing_seed, sm.n_failed, sm.total_steps, sm.total_clock_ns, cdev)
            if timed:
                host_tot[0] += rep[1]; host_tot[1] += rep[2]; host_tot[2] += rep[3]; host_tot[3] += sm.kernel_ms

    host_tot = [0, 0, 0, 0.0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stream_trial = None
    if args.streams == 0 and args.warmup >= 2:
        # untimed trial: how many batches in flight does this box reward?  (same kernel, same seeds; only the overlap differs)
        stream_trial = {}
        scratch_rep = torch.zeros(REPORT_WORDS, dtype=torch.int64, device=dev)

        def trial(cand, n):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(n):
                si = k % cand
                with torch.cuda.stream(streams[si]):
                    runtime.run_batch_async(w, seed0 + (1 << 50) + k * total, count, d_outs[si].data_ptr(), scratch_rep.data_ptr(),
                                            streams[si].cuda_stream, cfg, lim, timing_slot=-1)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n * 1e3

        trial(min(3, max_streams), 6)                # first launches pay for table uploads, module load, stream set-up
        for cand in ((3, 4, 5, 3, 4, 5) if max_streams >= 5 else (3, 4, 3, 4) if max_streams == 4 else (2, 3, 2, 3)):      # alternate, keep the better of two rounds each
            ms = trial(cand, 12)
            stream_trial[cand] = min(ms, stream_trial.get(cand, ms))
        n_streams = min(stream_trial, key=stream_trial.get)
        if world > 1:            # every rank must use the same count (the report ring is indexed by step)
            t = torch.tensor([n_streams], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            n_streams = int(t[0])
    for k in range(args.warmup):
        step(k, False)
    # `repeats` timed regions of EXACTLY args.steps steps each (a 20-step region is 24 ms: one of them is not a measurement).
    # Every region is bracketed by barrier + synchronize on both sides; fresh seeds throughout.  The line reports the median region.
    regions = []
    for r in range(repeats):
        base = args.warmup + r * args.steps
        sync()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(base + k, True)
        sync()
        rdt = time.perf_counter() - t0
        rk_ms = None
        if use_device_report:                 # (outside the region) this region's launches, start to end, from their HIP events
            nslots = min(args.steps, 64)
            rk_ms = sum(runtime.timing_ms((base + args.steps - 1 - i) % 64) for i in range(nslots)) * args.steps / nslots
        if world > 1:
            t = torch.tensor([rdt, rk_ms or 0.0], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rdt, rk_ms = float(t[0]), float(t[1])
        regions.append({"first_row": base, "dt": rdt, "kernel_ms": rk_ms})
    order = sorted(range(repeats), key=lambda i: regions[i]["dt"])
    med = regions[order[(repeats - 1) // 2]]        # the median region (the lower one of an even count)
    dt = med["dt"]
    row0 = med["first_row"]
    # ---------------- everything below is outside the timed region ----------------
    # for reference beside the overlapped figure: the same step with nothing else in flight (one stream)
    single_ms = None
    if use_device_report and n_streams > 1 and world == 1:
        ns = min(10, n_rows)
        scratch = torch.empty_like(d_outs[0])
        t1 = time.perf_counter()
        for k in range(ns):
            runtime.run_batch_async(w, seed0 + k * total, count, scratch.data_ptr(), 0, streams[0].cuda_stream, cfg, lim, timing_slot=-1)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / ns * 1e3
    rccl_ranks = None
    if use_device_report:
        if world > 1:
            gath_all = gathered[args.warmup:].cpu()
            for row in gath_all:                   # every step's gathered report holds exactly one row per rank
                launch.check_ranks([(int(r[4]), int(r[5])) for r in row], n_ranks)
            devs = {int(r[5]) for r in gath_all[-1]}
            rccl_ranks = n_ranks if backend == "nccl" and len(devs) == n_ranks else None
            rows = mdist.combine_gathered(gath_all[row0 - args.warmup: row0 - args.warmup + args.steps])
        else:
            rows = ring[row0: row0 + args.steps].cpu()
        nfail, steps_total, clock_total = (int(x) for x in rows[:, 1:4].sum(dim=0).tolist())
        kernel_ms = med["kernel_ms"]
    else:       # functional-test hook (host-side reports): totals over ALL regions, scaled to one
        nfail, steps_total, clock_total, kernel_ms = host_tot
        steps_total //= repeats; clock_total //= repeats; kernel_ms /= repeats; nfail //= repeats
    ms_list = [r["dt"] / args.steps * 1e3 for r in regions]
    region_stats = {"n": repeats, "steps_each": args.steps, "ms_per_step_median": med["dt"] / args.steps * 1e3,
                    "ms_per_step_min": min(ms_list), "ms_per_step_max": max(ms_list), "ms_per_step_first": ms_list[0],
                    "ms_per_step_all": [round(x, 5) for x in ms_list],
                    "note": "each region = exactly `steps` steps between barrier + synchronize; the line's ms_per_step / value / "
                            "kernel_ms_per_step are the median region's"}

    # oracle check of the batches that were just timed: the last batch on every stream, sampled k*257 mod count
    verified = 0
    if not args.no_verify:
        import oracle
        n_samp = min(256, count)
        for si, k in sorted(last_on_stream.items()):
            got = np.frombuffer(d_outs[si].cpu().numpy().tobytes(), dtype=A.RESULT_DTYPE)
            base = seed0 + k * total
            for j in range(n_samp):
                i = (j * 257) % count
                want, _ = oracle.run_batch(w, base + i, 1, cfg, lim)
                if int(got[i]["verdict"]) == A.OVERFLOW:      # a runner verdict (counted in failed_seeds; Builder::run re-runs
                    continue                                   # such seeds with larger capacities), not a different answer
                if got[i] != want[0]:
                    print(f"bench.py: VERIFY FAILED rank {rank} stream {si} seed {base + i}: gpu {got[i]} != oracle {want[0]}", file=sys.stderr)
                    return 3
                verified += 1
        if world > 1:
            t = torch.tensor([verified], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            verified = int(t[0])

    # first-failing-seed leg (SURVEY 8d fault variant): wall time from launch to the failing seed being known on the host
    first_fail = None
    if not args.no_first_fail and world == 1 and args.workload == "pingpong":
        import oracle
        fcfg = A.Config.default(packet_loss_rate=args.first_fail_loss)
        fbuf = d_outs[0]
        runtime.run_batch_device(w, 1 << 40, count, fbuf.data_ptr(), streams[0].cuda_stream, fcfg, lim)   # warm (tables, variant)
        torch.cuda.synchronize()
        reps, wall, ksum, sm = 5, 0.0, 0.0, None
        for r in range(reps):
            fs0 = (1 << 41) + r * count
            t1 = time.perf_counter()
            sm = runtime.run_batch_device(w, fs0, count, fbuf.data_ptr(), streams[0].cuda_stream, fcfg, lim)   # returns once the 32-byte report is on the host
            wall += time.perf_counter() - t1
            ksum += sm.kernel_ms
        # the last repetition against the oracle: everything up to and including the first failing seed
        fs0 = (1 << 41) + (reps - 1) * count
        n_chk = min(count, max(256, int(sm.first_failing_seed - fs0) + 1 if sm.n_failed else 256))
        want, osm = oracle.run_batch(w, fs0, n_chk, fcfg, lim)
        got = np.frombuffer(fbuf.cpu().numpy().tobytes(), dtype=A.RESULT_DTYPE)[:n_chk]
        if (got != want).any() or (sm.n_failed and osm.first_failing_seed != sm.first_failing_seed):
            print(f"bench.py: FIRST-FAIL VERIFY FAILED: gpu {sm.first_failing_seed} oracle {osm.first_failing_seed}", file=sys.stderr)
            return 3
        # the same search with batches kept in flight (what a campaign does: the report of batch k is read while k + 1 .. run):
        # n_streams streams, one device report row per batch, the host looks at the rows once at the end
        n_search = 10 * n_streams
        srows = torch.zeros((n_search, REPORT_WORDS), dtype=torch.int64, device=dev)
        fs1 = 1 << 42
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(n_search):
            si = k % n_streams
            with torch.cuda.stream(streams[si]):
                runtime.run_batch_async(w, fs1 + k * count, count, d_outs[si].data_ptr(), srows[k].data_ptr(),
                                        streams[si].cuda_stream, fcfg, lim, timing_slot=-1)
        torch.cuda.synchronize()
        search_dt = time.perf_counter() - t1
        srows_h = srows.cpu()
        found = mdist.decode_first_fail(int(srows_h[:, 0].min()))
        _, osm1 = oracle.run_batch(w, fs1, 64, fcfg, lim)            # the first batch's first failing seed is the search's answer
        if osm1.n_failed and found != osm1.first_failing_seed:
            print(f"bench.py: FIRST-FAIL SEARCH VERIFY FAILED: gpu {found} oracle {osm1.first_failing_seed}", file=sys.stderr)
            return 3
        first_fail = {"packet_loss_rate": args.first_fail_loss, "seeds_per_batch": count,
                      "search_batches_in_flight": n_streams, "search_seeds_per_hour": n_search * count / search_dt * 3600.0,
                      "search_ms_per_batch": search_dt / n_search * 1e3,
                      "time_to_first_fail_ms": wall / reps * 1e3, "kernel_ms": ksum / reps,
                      "first_failing_seed_offset": int(sm.first_failing_seed - fs0) if sm.n_failed else None,
                      "failed_fraction": sm.n_failed / count, "oracle_checked_seeds": n_chk,
                      "seeds_per_hour": count / (wall / reps) * 3600.0,
                      "note": "time_to_first_fail_ms / seeds_per_hour: launch -> kernel -> device reduction -> 32-byte D2H, "
                              "host-synchronous, one batch at a time; search_*: the same batches kept in flight on the bench's streams"}


    # Rare-failure searches: time-to-first-failure only means something when failures are rare.  Batches of `count` seeds are
    # kept in flight on the bench's streams; the host reads each batch's 32-byte device report as it completes and stops
    # launching as soon as one reports a failure (seeds are contiguous, so the first failing batch holds the minimum failing
    # seed).  Afterwards — outside the measured time — EVERY seed from the start of the search up to and including the one
    # found is run through the oracle: all before it must pass, and the failing batch's GPU results are compared bit for bit.
    def rare_search(loss, fs, max_batches=96):
        """One madsim_hip_run_campaign call: the LIBRARY keeps the batches in flight on its own streams and stops launching at the
        first batch that reports a genuine failure — what a Rust / C host gets, no torch streams involved."""
        import threading
        import oracle
        fcfg = A.Config.default(packet_loss_rate=loss)
        # time to the first failure is a latency: a batch among five in flight takes 5.6 ms, among three 4.1 ms — the search keeps
        # three in flight whatever the throughput legs use
        n_search_flights = min(n_streams, 3)
        runtime.run_campaign(w, fs - 4 * count, 4 * count, count, n_search_flights, False, fcfg, lim)      # warm (streams, buffers, tables)
        torch.cuda.synchronize()
        rep = runtime.run_campaign(w, fs, max_batches * count, count, n_search_flights, True, fcfg, lim)
        found = rep.first_failing_seed != (1 << 64) - 1
        res = {"packet_loss_rate": loss, "seeds_per_batch": count, "batches_in_flight": n_search_flights,
               "batches_launched": int(rep.batches_launched), "found": found, "entry_point": "madsim_hip_run_campaign(STOP_AT_FAILURE)"}
        if not found:
            res["note"] = f"no failing seed in {int(rep.batches_run)} batches"
            return res
        seed = int(rep.first_failing_seed)
        offset, j = seed - fs, int(rep.batches_run) - 1
        got, _ = runtime.run_batch(w, fs + j * count, count, fcfg, lim)          # the failing batch again, per-seed results this time
        # oracle: every seed of [fs, seed], threads over disjoint blocks (ctypes releases the GIL)
        n_chk = offset + 1
        n_thr = max(1, min(os.cpu_count() or 1, 32, (n_chk + 4095) // 4096))
        per = (n_chk + n_thr - 1) // n_thr
        parts = [None] * n_thr

        def work(i):
            lo = i * per
            n = max(0, min(per, n_chk - lo))
            parts[i] = oracle.run_batch(w, fs + lo, n, fcfg, lim)[0] if n else np.zeros(0, dtype=A.RESULT_DTYPE)
        thr = [threading.Thread(target=work, args=(i,)) for i in range(n_thr)]
        t2 = time.perf_counter()
        for t in thr:
            t.start()
        for t in thr:
            t.join()
        want = np.concatenate(parts)
        o_first = np.nonzero(want["verdict"] != A.PASS)[0]
        in_batch = want[j * count:]
        ok = len(o_first) == 1 and int(o_first[0]) == offset and (got[:len(in_batch)] == in_batch).all()
        res.update({"batches_until_found": j + 1, "first_failing_seed_offset": offset, "failed_in_that_batch": int(rep.n_failed),
                    "time_to_first_fail_ms": rep.wall_s * 1e3, "seeds_searched": int(rep.seeds_run),
                    "seeds_per_hour": rep.seeds_run / rep.wall_s * 3600.0,
                    "oracle_checked_seeds": n_chk, "oracle_check_s": time.perf_counter() - t2, "oracle_agrees": bool(ok),
                    "note": "wall time of the call: first launch to the failing seed known on the host; then every seed up to and including "
                            "it oracle-checked (all earlier ones pass, the failing batch re-run and compared bit for bit)"})
        return res

    # the same overlap without torch: ONE library call keeps the batches in flight on the library's own streams — the rate a Rust / C
    # host gets from madsim_hip_run_campaign (no per-seed results: report only)
    campaign = None
    if not args.no_first_fail and world == 1 and headline and not args.loss:
        nb = 60
        runtime.run_campaign(w, 1 << 47, 6 * count, count, n_streams, False, cfg, lim)               # warm
        rep = runtime.run_campaign(w, (1 << 47) + 6 * count, nb * count, count, n_streams, False, cfg, lim)
        campaign = {"entry_point": "madsim_hip_run_campaign", "batches": nb, "batches_in_flight": n_streams, "seeds_per_batch": count,
                    "ms_per_batch": rep.wall_s / nb * 1e3, "seeds_per_sec": rep.seeds_run / rep.wall_s,
                    "executor_steps_per_sec": rep.total_steps / rep.wall_s, "failed_seeds": int(rep.n_failed), "runner_verdicts": int(rep.n_runner)}

    # extra.first_fail_over_ranks: the seed search with ONE PROCESS PER GPU (runtime.run_campaign_over_ranks -> madsim_amd/dist.py): every rank
    # runs chunks of the seed range as pipelined madsim_hip_run_campaign calls on its own GPU, one all-gather of the 56-byte reports per
    # round (RCCL over xGMI under "nccl").  The half of BASELINE's metric that `--gpus N` measures at N > 1: seeds searched per hour when
    # nothing fails (the steady rate of the search), and the wall time to a very rare first failure.  At N = 1 the same path, no collective.
    ranks_search = None
    if not args.no_first_fail and headline and not args.loss:
        rb = 4 * n_streams
        per_round = rb * count * n_ranks
        runtime.run_campaign_over_ranks(w, 1 << 52, per_round, count, False, cfg, lim, device_tensors=cdev, round_batches=rb)       # warm (streams, buffers, tables, the communicator)
        sync()
        t1 = time.perf_counter()
        rep = runtime.run_campaign_over_ranks(w, (1 << 52) + per_round, 3 * per_round, count, True, cfg, lim, device_tensors=cdev, round_batches=rb)
        sync()
        sdt = time.perf_counter() - t1
        fcfg2 = A.Config.default(packet_loss_rate=args.very_rare_loss)
        sync()
        t1 = time.perf_counter()
        rep2 = runtime.run_campaign_over_ranks(w, 1 << 53, 6 * per_round, count, True, fcfg2, lim, device_tensors=cdev, round_batches=rb)
        sync()
        fdt = time.perf_counter() - t1
        ranks_search = {"entry_point": "runtime.run_campaign_over_ranks (madsim_hip_run_campaign per rank and round, one all-gather per round)",
                        "world": n_ranks, "round_batches_per_rank": rb, "seeds_per_batch": count,
                        "seeds_searched": int(rep["seeds_run"]), "rounds": int(rep["rounds"]), "failed": int(rep["n_failed"]),
                        "seeds_per_sec": rep["seeds_run"] / sdt, "seeds_per_hour": rep["seeds_run"] / sdt * 3600.0,
                        "very_rare": {"packet_loss_rate": args.very_rare_loss, "found": rep2["first_failing_seed"] != (1 << 64) - 1,
                                      "first_failing_seed_offset": (int(rep2["first_failing_seed"]) - (1 << 53)) if rep2["first_failing_seed"] != (1 << 64) - 1 else None,
                                      "seeds_searched": int(rep2["seeds_run"]), "rounds": int(rep2["rounds"]), "time_to_first_fail_ms": fdt * 1e3}}

    # extra.run_batch_262144: the SURVEY 8b entry point itself — one plain madsim_hip_run_batch call with host buffers for four batches
    # of seeds (Builder::run hands over all its seeds at once, runtime/builder.rs:121-162).  The library cuts the call into sub-launches
    # it keeps in flight and overlaps their device-to-host copies (run_pipelined); the figure includes the copy of all 12.6 MB of
    # per-seed results into pageable host memory.  Sampled seeds are compared with the oracle, the whole array with the summary.
    run_batch_big = None
    if not args.no_first_fail and world == 1 and headline and not args.loss:
        import oracle
        nbig = 4 * count
        runtime.run_batch(w, 1 << 48, nbig, cfg, lim)                                   # warm (flights, staging buffers)
        walls = []
        import ctypes as C
        bcfg = cfg or A.Config.default()
        keep = []                           # (the caller's arrays are allocated, and later freed, by the caller: outside the timed call)
        for r in range(7):
            big, bsum = np.empty(nbig, dtype=A.RESULT_DTYPE), A.Summary()
            keep.append(big)
            t1 = time.perf_counter()
            rc = runtime.lib().madsim_hip_run_batch(w.ref(), C.byref(bcfg), (1 << 48) + (r + 1) * nbig, nbig, C.byref(lim),
                                                    big.ctypes.data_as(C.c_void_p), C.byref(bsum))
            walls.append(time.perf_counter() - t1)
            if rc != 0:
                print(f"bench.py: madsim_hip_run_batch({nbig}) failed: {runtime.lib().madsim_hip_last_error().decode()}", file=sys.stderr)
                return 3
        bbase = (1 << 48) + 7 * nbig
        bver = 0
        for jj in range(256):
            i = (jj * 1031) % nbig
            want, _ = oracle.run_batch(w, bbase + i, 1, cfg, lim)
            if big[i] != want[0]:
                print(f"bench.py: VERIFY FAILED run_batch({nbig}) seed {bbase + i}: gpu {big[i]} != oracle {want[0]}", file=sys.stderr)
                return 3
            bver += 1
        if int(big["steps"].sum()) != bsum.total_steps or int((big["verdict"] != A.PASS).sum()) != bsum.n_failed:
            print("bench.py: VERIFY FAILED run_batch summary does not match its own per-seed array", file=sys.stderr)
            return 3
        walls.sort()
        run_batch_big = {"entry_point": "madsim_hip_run_batch (host result array, PCIe-inclusive)", "seeds": nbig, "calls": len(walls),
                         "wall_ms_median": walls[len(walls) // 2] * 1e3, "wall_ms_min": walls[0] * 1e3, "wall_ms_max": walls[-1] * 1e3,
                         "seeds_per_sec": nbig / walls[len(walls) // 2], "kernel_ms": bsum.kernel_ms, "verified_seeds": bver,
                         "failed_seeds": int(bsum.n_failed)}

    # extra.plain_run: the same step without the per-seed fingerprint of the determinism log (madsim_limits_t.no_trace_hash) — the
    # reference's own Builder::run computes log bytes only under check_determinism (rand.rs:67), so this is the mode a drop-in test
    # run needs; the headline keeps the fingerprint (all 48 result bytes oracle-checked).  Same streams, same batch size, its own
    # timed region and its own oracle check (the oracle honours the same flag).
    plain = None
    if not args.no_extras and world == 1 and headline and not args.loss and use_device_report:
        import copy
        plim = copy.copy(lim); plim.no_trace_hash = 1
        psteps, pwu = max(args.steps, 12), 4
        pring = torch.zeros((psteps + pwu, REPORT_WORDS), dtype=torch.int64, device=dev)
        plast = {}

        def pstep(k, timed):
            si = k % n_streams
            with torch.cuda.stream(streams[si]):
                runtime.run_batch_async(w, (1 << 45) + k * count, count, d_outs[si].data_ptr(), pring[k].data_ptr(),
                                        streams[si].cuda_stream, cfg, plim, timing_slot=(k % 64) if timed else -1)
            plast[si] = k
        for k in range(pwu):
            pstep(k, False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(psteps):
            pstep(pwu + k, True)
        torch.cuda.synchronize()
        pdt = time.perf_counter() - t1
        prows = pring[pwu:].cpu()
        pfail, psteps_total, pclock = (int(x) for x in prows[:, 1:4].sum(dim=0).tolist())
        pver = 0
        if not args.no_verify:
            import oracle
            for si, k in sorted(plast.items()):
                got = np.frombuffer(d_outs[si].cpu().numpy().tobytes(), dtype=A.RESULT_DTYPE)
                for jj in range(128):
                    i = (jj * 509) % count
                    want, _ = oracle.run_batch(w, (1 << 45) + k * count + i, 1, cfg, plim)
                    if got[i] != want[0] or int(got[i]["trace_hash"]) != 0:
                        print(f"bench.py: VERIFY FAILED plain run seed {(1 << 45) + k * count + i}: gpu {got[i]} != oracle {want[0]}", file=sys.stderr)
                        return 3
                    pver += 1
        pg = runtime.geometry(w, plim)
        plain = {"what": "madsim_limits_t.no_trace_hash = 1: results without the determinism-log fingerprint (the reference logs only under "
                         "check_determinism, rand.rs:67); every other result field as in the headline run",
                 "steps": psteps, "warmup": pwu, "concurrent_batches": n_streams, "ms_per_step": pdt / psteps * 1e3,
                 "seeds_per_sec": psteps * count / pdt, "executor_steps_per_sec": psteps_total / pdt,
                 "sim_seconds_per_sec": pclock / 1e9 / pdt, "failed_seeds": pfail, "verified_seeds": pver,
                 "kernel": runtime.variant_name(pg)}

    first_fail_rare = first_fail_very_rare = None
    if not args.no_first_fail and world == 1 and args.workload == "pingpong":
        first_fail_rare = rare_search(args.rare_loss, 1 << 43)
        first_fail_very_rare = rare_search(args.very_rare_loss, 1 << 44)
        for ff in (first_fail_rare, first_fail_very_rare):
            if ff.get("found") and not ff["oracle_agrees"]:
                print(f"bench.py: RARE FIRST-FAIL VERIFY FAILED: {ff}", file=sys.stderr)
                return 3

    # extra.workloads: the configs[2] / [3] / [4]-shaped workloads and the timer storm, two timed steps per stream each on the same streams,
    # every line with sampled seeds of its timed batches checked against the oracle
    extras = None
    ceil_all = issue_ceiling() if world == 1 and rank == 0 else None      # (tools/ubench_issue --quick: issue ceilings + the attainable HBM copy rate)
    if not args.no_extras and world == 1 and headline and not args.loss:
        import oracle
        extras = {}
        # a step of these = the per-GPU batch BASELINE.json quotes the config on (configs[2]: 262 144 seeds on one GPU; configs[3]:
        # 1 048 576 over 8 GPUs; configs[4]: 4 194 304 over 8 GPUs), issued as sub-launches of `count` seeds kept in flight on the streams
        baseline_batch = {"raft": 262144, "kv": 131072, "topo": 524288, "timers": count}
        for name in ("raft", "kv", "topo", "timers"):
            xw, xlim, xname = workload.bench_case(name)
            xg0 = runtime.geometry(xw, xlim)
            # batches in flight by THIS workload's occupancy (flights(), above): five only where four waves per SIMD fit, four for
            # the global-state builds with a heap-spill region, else three
            xn = min(max_streams, flights(xg0))
            nsub = max(1, baseline_batch[name] // count)
            xtimed = max(4, -(-8 * xn // nsub))      # timed steps: at least eight sub-launches per stream (the region starts on an empty
                                                     # chip and ends with a drain — a launch lasts 3-4 sub-launch periods — so three per
                                                     # stream, as in round 3, measured ramp and tail more than the steady state)
            xs, xwu = xtimed * nsub, xn   # timed sub-launches after one untimed per stream (and a synchronize: ramp and drain are inside the timed region)
            xring = torch.zeros((xs + xwu, REPORT_WORDS), dtype=torch.int64, device=dev)
            last = {}

            def xstep(k, timed):
                si = k % xn
                with torch.cuda.stream(streams[si]):
                    runtime.run_batch_async(xw, (1 << 46) + k * count, count, d_outs[si].data_ptr(), xring[k].data_ptr(),
                                            streams[si].cuda_stream, None, xlim, timing_slot=(k % 64) if timed else -1)
                last[si] = k
            for k in range(xwu):
                xstep(k, False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(xs):
                xstep(xwu + k, True)
            torch.cuda.synchronize()
            xdt = time.perf_counter() - t1
            rows = xring[xwu:].cpu()
            xfail, xsteps, xclock = (int(x) for x in rows[:, 1:4].sum(dim=0).tolist())
            xk_ms = sum(runtime.timing_ms((xwu + xs - 1 - i) % 64) for i in range(min(xs, 48))) / min(xs, 48)
            xver = 0
            for si, k in sorted(last.items()):
                got = np.frombuffer(d_outs[si].cpu().numpy().tobytes(), dtype=A.RESULT_DTYPE)
                for jj in range(64):
                    i = (jj * 1021) % count
                    want, _ = oracle.run_batch(xw, (1 << 46) + k * count + i, 1, None, xlim)
                    if int(got[i]["verdict"]) == A.OVERFLOW:
                        continue
                    if got[i] != want[0]:
                        print(f"bench.py: VERIFY FAILED workload {name} seed {(1 << 46) + k * count + i}: gpu {got[i]} != oracle {want[0]}", file=sys.stderr)
                        return 3
                    xver += 1
            xg = runtime.geometry(xw, xlim)
            xalgo = xsteps / xs * ALGO_BYTES_PER_STEP + count * IO_BYTES_PER_SEED
            # what these kernels really move — FETCH_SIZE x 2 + WRITE_SIZE per launch and the launches' own duration from
            # live rocprofv3 passes of `bench.py --workload <name>` — against the copy rate this chip sustains (tools/ubench_issue)
            xpmc, xpmc_note = None, "not measured"
            if args.measure_traffic is not False:
                xpmc, xpmc_note = measure_counters(sys.argv[1:], passes=("TRACE", "FETCH_SIZE", "WRITE_SIZE"), extra_args=["--workload", name])
            xtraffic = (2.0 * xpmc["FETCH_SIZE"] + xpmc["WRITE_SIZE"]) * 1024.0 if xpmc else None
            xlaunch_ms = (xpmc or {}).get("launch_ms_rocprof") or xk_ms
            copy_peak = (ceil_all or {}).get("hbm_copy_gbps")
            extras[name] = {"workload": xname, "seeds_per_step": nsub * count, "steps": xtimed, "sub_launches_per_step": nsub,
                            "seeds_per_sub_launch": count, "warmup_sub_launches": xwu, "concurrent_batches": xn,
                            "ms_per_step": xdt / xtimed * 1e3, "ms_per_sub_launch": xdt / xs * 1e3, "kernel_ms_per_sub_launch": xk_ms,
                            "steps_per_sec": xsteps / xdt, "seeds_per_sec": xs * count / xdt, "sim_seconds_per_sec": xclock / 1e9 / xdt,
                            "failed_seeds": xfail, "verified_seeds": xver, "kernel": runtime.variant_name(xg),
                            "lds_bytes_per_seed": xg.lds_bytes_per_seed, "global_bytes_per_seed": xg.global_bytes_per_seed,
                            "frac": xalgo / (xlaunch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            "chip_frac": xalgo * xs / xdt / 1e9 / HBM_PEAK_GBPS,
                            "algorithmic_bytes_per_launch": xalgo, "launch_ms": xlaunch_ms,
                            "launch_ms_source": "rocprofv3 --kernel-trace, live" if (xpmc or {}).get("launch_ms_rocprof") else "HIP events",
                            "traffic_bytes_per_launch": xtraffic,
                            "traffic_over_algorithmic": (xtraffic / xalgo) if xtraffic else None,
                            "measured_gbps": (xtraffic / (xlaunch_ms * 1e-3) / 1e9) if xtraffic else None,
                            "measured_chip_gbps": (xtraffic * xs / xdt / 1e9) if xtraffic else None,
                            "copy_peak_gbps": copy_peak,
                            "measured_over_copy_peak": (xtraffic * xs / xdt / 1e9 / copy_peak) if xtraffic and copy_peak else None,
                            "traffic_detail": ({"FETCH_SIZE_KB": xpmc["FETCH_SIZE"], "WRITE_SIZE_KB": xpmc["WRITE_SIZE"], "source": xpmc["source"]} if xpmc else xpmc_note),
                            "frac_note": "frac = algorithmic bytes per launch (120 B per executor step + 56 B per seed, SURVEY 8d) / launch_ms / 8 TB/s; "
                                         "chip_frac = the same bytes of all overlapping launches / wall time; measured_gbps = (FETCH_SIZE x 2 + WRITE_SIZE) "
                                         "per launch / launch_ms (SURVEY 8d iii: the HBM-bound figure), measured_chip_gbps over wall time, against "
                                         "copy_peak_gbps (the float4 copy rate tools/ubench_issue measures on this GPU)"}

    if rank == 0:
        seeds_total = total * args.steps
        sim_s = clock_total / 1e9
        # Roofline of the dominant kernel (sim_kernel).  The executor state lives in registers and LDS, so the kernel's binding
        # bound is VALU instruction issue, not HBM: `roofline` prices the VALU wave-instructions the launches executed (PMC,
        # live) against the chip's sustained issue rate for the same instruction mix (tools/ubench_issue, live); the nominal
        # SURVEY 8d HBM yardstick (120 algorithmic bytes per executor step) rides along in roofline.hbm_nominal.
        k_avg_ms = kernel_ms / args.steps
        ms_step = dt / args.steps * 1e3
        steps_per_launch = steps_total / args.steps / n_ranks
        algo_bytes = steps_per_launch * ALGO_BYTES_PER_STEP + count * IO_BYTES_PER_SEED
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9
        g = runtime.geometry(w, lim)
        kname = runtime.variant_name(g)
        pmc, pmc_note = None, None
        if world == 1 and args.measure_traffic is not False:
            pmc, pmc_note = measure_counters(sys.argv[1:])
        traffic, tdetail = None, pmc_note
        if pmc:
            # rocprofv3 reports both in KB; gfx950 FETCH_SIZE counts 128-B requests as 64 B: x2 (MI355X_MICROARCH.md §HBM)
            traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            tdetail = {"FETCH_SIZE_KB": pmc["FETCH_SIZE"], "WRITE_SIZE_KB": pmc["WRITE_SIZE"], "source": pmc["source"]}
        elif world == 1 and per_gpu == workload.BENCH_SEEDS_PER_GPU and headline and not args.loss:
            # the committed rocprofv3 PMC passes of this same command (tools/prof_workload.sh): FETCH_SIZE x2 + WRITE_SIZE
            for name in ("r6_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
                tpath = os.path.join(ROOT, "profiles", name)
                if os.path.exists(tpath):
                    tj = json.load(open(tpath))
                    traffic = (2 * tj["FETCH_SIZE_KB"] + tj["WRITE_SIZE_KB"]) * 1024.0
                    tdetail = {"FETCH_SIZE_KB": tj["FETCH_SIZE_KB"], "WRITE_SIZE_KB": tj["WRITE_SIZE_KB"],
                               "source": f"profiles/{name}: rocprofv3 PMC passes of this command, not this run"
                                         + ("; live attempt: " + str(pmc_note) if pmc_note else "")}
                    break
        hbm_nominal = {"bound": "hbm (nominal yardstick, NOT the binding bound)", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": algo_bytes,
                       "chip_achieved": algo_bytes / (ms_step * 1e-3) / 1e9,
                       "chip_frac": algo_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                       "measured_hbm_gbps": (traffic / (k_avg_ms * 1e-3) / 1e9) if traffic else None,
                       "note": "ALGORITHMIC bytes (120 B per executor step, SURVEY 8d) / one launch's HIP-event duration: an LDS-equivalent "
                               "rate — those bytes never reach HBM on this LDS-resident path (measured_hbm_gbps is the real HBM rate from "
                               "FETCH/WRITE_SIZE).  chip_* = all overlapping launches / wall time; chip_frac > 1 means exactly that: "
                               "HBM is not a bound for this kernel"}
        ceil = ceil_all
        # `roofline` says what SURVEY 8(d) says: frac = ALGORITHMIC bytes per launch (120 B per executor step x the steps
        # one launch runs + 56 B per seed) / that launch's duration / 8 TB/s.  The duration is the average sim_kernel duration rocprofv3
        # reports for this same command with the launches overlapping as in the timed region (the live trace-only pass: the figure
        # `rocprofv3 --kernel-trace --stats` prints, profiles/r6_pingpong_profile.txt), or — no rocprofv3 — the launches' own HIP-event
        # durations (launch_ms_hip_events, always beside it).  On this LDS-resident path those bytes never reach HBM (`traffic` = what does,
        # FETCH_SIZE x 2 + WRITE_SIZE per launch), so the bound is nominal and the compute-side readings ride beside it as top-level keys:
        # frac_hw = VALU wave-instructions per second / the hardware's issue rate (a wave64 VALU instruction every 2 cycles x 1 024 SIMDs x
        # 2.4 GHz), frac_hw_useful_lanes = the same with the masked lanes taken out, frac_own_mix_ceiling = against the issue ceiling
        # measured for this kernel's own instruction mix (tools/ubench_issue; rounds 3-5 called it `frac`).
        PEAK_HW_GINST = 1228.8
        launch_ms = pmc.get("launch_ms_rocprof") if pmc else None
        launch_src = "rocprofv3 --kernel-trace of this command, live: mean sim_kernel duration over %d overlapping launches" % pmc["launches_traced"] if launch_ms else \
                     "HIP events around each timed launch (no rocprofv3 trace pass)"
        launch_ms = launch_ms or k_avg_ms
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        roof = {"bound": "hbm-nominal (LDS-resident: not binding)", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_bytes_per_executor_step": ALGO_BYTES_PER_STEP,
                "executor_steps_per_launch": steps_per_launch, "launch_ms": launch_ms, "launch_ms_source": launch_src,
                "launch_ms_hip_events": k_avg_ms, "frac_hip_events": hbm_nominal["frac"],
                "frac_hw": None, "frac_hw_useful_lanes": None, "frac_own_mix_ceiling": None, "peak_hw_ginst_s": PEAK_HW_GINST,
                "hbm_frac": hbm_nominal["frac"], "hbm_chip_frac": hbm_nominal["chip_frac"],
                "hbm_measured_gbps": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None,
                "traffic_over_algorithmic": (traffic / algo_bytes) if traffic else None,
                "traffic_detail": tdetail, "kernel": kname, "concurrent_launches": n_streams, "hbm_nominal": hbm_nominal}
        issue = {"unit": "G wave-inst/s", "peak_hw": PEAK_HW_GINST}
        if pmc and "SQ_INSTS_VALU" in pmc:
            valu = pmc["SQ_INSTS_VALU"]
            issue.update({"valu_inst_per_launch": valu, "salu_inst_per_launch": pmc.get("SQ_INSTS_SALU"),
                          "lds_inst_per_launch": pmc.get("SQ_INSTS_LDS"), "waves_per_launch": pmc.get("SQ_WAVES"),
                          "achieved_ginst_s": valu / (ms_step * 1e-3) / 1e9,
                          "per_launch_ginst_s": valu / (launch_ms * 1e-3) / 1e9})
            if pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("SQ_THREAD_CYCLES_VALU"):
                issue["lane_util"] = pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64.0)
        elif world == 1 and headline and not args.loss:
            # no live counters (rocprofv3 missing or refused): the committed per-executor-step instruction counts of this kernel
            cpath = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r6_issue_counters.json", "r3_issue_counters.json")) if os.path.exists(q)), "")
            if cpath:
                cj = json.load(open(cpath))
                valu = cj["valu_inst_per_executor_step"] * steps_per_launch
                issue.update({"valu_inst_per_launch": valu, "achieved_ginst_s": valu / (ms_step * 1e-3) / 1e9,
                              "per_launch_ginst_s": valu / (launch_ms * 1e-3) / 1e9, "lane_util": cj.get("lane_util"),
                              "counters_source": "profiles/" + os.path.basename(cpath) + " (rocprofv3 PMC of this command on an MI355X, not this run"
                                                 + ("; live attempt: " + str(pmc_note) if pmc_note else "") + ")"})
        if ceil:
            issue.update({"own_mix_ceiling_ginst_s": ceil["valu_mix_ceiling_ginst_s"], "ceiling_detail": ceil})
        if issue.get("achieved_ginst_s"):
            roof["frac_hw"] = issue["achieved_ginst_s"] / PEAK_HW_GINST
            if issue.get("lane_util"):
                roof["frac_hw_useful_lanes"] = roof["frac_hw"] * issue["lane_util"]      # lane-operations that do simulation work
            if ceil:
                roof["frac_own_mix_ceiling"] = issue["achieved_ginst_s"] / ceil["valu_mix_ceiling_ginst_s"]
        issue["note"] = ("achieved_ginst_s = VALU wave-instructions per sim_kernel launch (rocprofv3 SQ_INSTS_VALU, live) / wall time per batch "
                         f"({n_streams} launches overlap: the chip-level rate); own_mix_ceiling = the chip's sustained issue rate for the executor's "
                         "own VALU mix (tools/ubench_issue, best over 1-8 waves per SIMD) — a kernel is near 1.0 against THAT whatever its "
                         "quality, which is why it is no longer `frac`; lane_util = active lanes per issued VALU instruction")
        roof["valu_issue"] = issue
        roof["note"] = ("frac = algorithmic_bytes_per_launch / launch_ms / 8 TB/s (SURVEY 8d).  One division reproduces it from profiles/r6_pingpong_profile.txt: "
                        "12.229 GB / the sim_kernel AverageNs of the stats table / 8e12.  Those bytes stay in LDS and registers: `traffic` is what "
                        "reaches HBM per launch, and hbm_chip_frac (all overlapping launches / wall time) exceeds 1 for exactly that reason.")
        line = {
            "metric": "sim_seconds_per_sec", "value": sim_s / dt, "unit": "sim-s/s",
            "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{wname}, {per_gpu} seeds per GPU per step"
                                   + (f", packet_loss_rate={args.loss}" if args.loss else "")
                                   + (" (BASELINE configs[1])" if headline and not args.loss else ""),
                       "seeds_per_step": total, "parallelism": f"seed-shard x{n_ranks}" + (f", {n_streams} concurrent batches per GPU" if n_streams > 1 else "")},
            "verified_seeds": verified,
            # the SURVEY 8b call itself in the headline's shadow: madsim_hip_run_batch(262 144) into a pageable host array (PCIe-inclusive)
            "value_run_batch_host": (run_batch_big["seeds_per_sec"] * (sim_s / seeds_total)) if run_batch_big else None,
            "value_run_batch_host_seeds_per_sec": run_batch_big["seeds_per_sec"] if run_batch_big else None,
            "extra": {"seeds_per_sec": seeds_total / dt,
                      "executor_steps_per_sec": steps_total / dt,
                      "failed_seeds": nfail, "kernel_ms_per_step": k_avg_ms, "single_stream_ms_per_step": single_ms,
                      "lds_bytes_per_seed": g.lds_bytes_per_seed, "waves_per_cu": g.blocks_per_cu * g.block_threads // 64,
                      "lanes_per_wave": g.lanes_per_wave, "rccl_ranks": rccl_ranks, "first_fail": first_fail,
                      "first_fail_rare": first_fail_rare, "first_fail_very_rare": first_fail_very_rare,
                      "workloads": extras, "campaign": campaign, "plain_run": plain, "run_batch_262144": run_batch_big,
                      "regions": region_stats,
                      "stream_trial_ms_per_step": stream_trial,
                      "first_fail_over_ranks": ranks_search,
                      "first_fail_seeds_per_hour": ((ranks_search or {}).get("seeds_per_hour") if n_ranks > 1 else None)
                                                   or (first_fail_rare or {}).get("seeds_per_hour") or (first_fail["seeds_per_hour"] if first_fail else None)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle
            sample = 4 * workload.BENCH_SEEDS_PER_GPU if args.workload == "pingpong" else 16384
            t1 = time.perf_counter()
            _, osum = oracle.run_batch(w, 0, sample, cfg, lim)
            cdt = time.perf_counter() - t1
            line["cpu_baseline"] = {"value": osum.total_clock_ns / 1e9 / cdt, "unit": "sim-s/s", "cores": 1,
                                    "kind": "port",
                                    "sample": f"{sample} seeds of the same workload, single thread, "
                                              f"{cdt:.1f} s ({sample / cdt:.0f} seeds/s, "
                                              f"{osum.total_steps / cdt / 1e6:.1f} M steps/s)"}
            # the whole host beside it: one oracle thread per core, disjoint seed blocks (ctypes releases the GIL)
            import threading
            n_thr = min(os.cpu_count() or 1, 64)
            per = max(2048, sample // 4)
            res = [None] * n_thr

            def work(i):
                res[i] = oracle.run_batch(w, (1 << 45) + i * per, per, cfg, lim)[1]
            thr = [threading.Thread(target=work, args=(i,)) for i in range(n_thr)]
            t1 = time.perf_counter()
            for t in thr:
                t.start()
            for t in thr:
                t.join()
            hdt = time.perf_counter() - t1
            line["cpu_baseline_host"] = {"value": sum(r.total_clock_ns for r in res) / 1e9 / hdt, "unit": "sim-s/s", "cores": n_thr,
                                         "kind": "port",
                                         "sample": f"{n_thr} threads x {per} seeds, {hdt:.1f} s ({n_thr * per / hdt:.0f} seeds/s, "
                                                   f"{sum(r.total_steps for r in res) / hdt / 1e6:.0f} M steps/s)"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    runtime.shutdown()
    return 0


if __name__ == "__main__":
    sys.exit(main())
