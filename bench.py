#!/usr/bin/env python3
"""bench.py — BASELINE.json metric on BASELINE.json's config.

One "step" = one pass of the hot path over one batch: 65 536 seeds of the 4-node ping-pong
(R = 64 rounds per pair, Config::default()) per GPU, i.e. BASELINE.json configs[1].  With N GPUs each
rank runs its own contiguous block of 65 536 seeds (weak scaling, no data-path collective) and the
ranks exchange one first-failing-seed all-reduce per step (RCCL).

Prints ONE JSON line (rank 0).  `value` = simulated seconds per wall second summed over every seed of
every rank; seeds/s and executor-steps/s ride along in `extra`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEEDS_PER_GPU = 65536
N_NODES, ROUNDS = 4, 64
ALGO_BYTES_PER_STEP = 120      # pop 16 + push 16 + rng 32r+32w + clock 8r+8w + ready 4r+4w
IO_BYTES_PER_SEED = 8 + 48     # seed in, madsim_result_t out
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seeds", type=int, default=SEEDS_PER_GPU, help="seeds per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lpw", type=int, default=0, help="seed-carrying lanes per wave (0 = library auto)")
    ap.add_argument("--nodes", type=int, default=N_NODES, help="ping-pong nodes (experiments; the bench line is quoted on 4)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are spread over. One 65 536-seed batch is 1 024 waves = one per SIMD; "
                         "a second batch in flight on another stream fills the second wave slot the per-seed LDS allows")
    ap.add_argument("--heap-lds", type=int, default=4, help="timer-heap entries kept in LDS (the rest spill to HBM)")
    ap.add_argument("--generic", action="store_true", help="force the generic kernel variant (HBM heap spill enabled)")
    ap.add_argument("--workload", default="pingpong", choices=["pingpong", "raft", "kv", "timers", "topo"],
                    help="pingpong = BASELINE configs[1] (the headline); raft / kv = configs[2] / configs[3]-shaped extras")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from madsim_amd import _abi as A
    from madsim_amd import dist as mdist
    from madsim_amd import runtime, workload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MADSIM_BENCH_BACKEND=gloo is a functional-test hook (several ranks sharing one GPU on a 1-GPU box);
    # the real multi-GPU run is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("MADSIM_BENCH_BACKEND", "nccl")
    gpu = local_rank % torch.cuda.device_count()
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
    dev = torch.device("cuda", gpu)
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where the 32-byte report tensors live
    runtime.init(gpu)

    if args.workload == "pingpong":
        w = workload.pingpong(args.nodes, ROUNDS)
        wname = f"{args.nodes}-node ping-pong, R={ROUNDS}, Config::default()"
        lim = A.Limits()
        # tight capacities for this workload (high-water marks: 4 timers, 1 pending recv, never a queued message);
        # exceeding one would show up as failed seeds (verdict MADSIM_OVERFLOW), never as a different answer
        lim.heap_lds_slots, lim.heap_spill_slots = args.heap_lds, 4 - args.heap_lds
        lim.mbox_regs, lim.mbox_msgs = 1, A.LIMIT_NONE
    elif args.workload == "raft":
        w, lim = workload.raft_election(), workload.raft_election_limits()
        wname = "5-node election loop with partition injection (configs[2] shape)"
    elif args.workload == "timers":
        w, lim = workload.timer_storm(), workload.timer_storm_limits(args.heap_lds)
        wname = f"timer storm: 24 tasks x sleep(gen_range(0..2 s)), heap_lds={args.heap_lds} (HBM heap-spill path)"
    elif args.workload == "topo":
        w, lim = workload.streaming_topology(), workload.streaming_topology_limits()
        wname = "16-node streaming topology: KV meta + typed-RPC brokers + 12 compute nodes (configs[4] shape)"
    else:
        w, lim = workload.kv_rpc(), workload.kv_rpc_limits()
        wname = "etcd-style KV ops over connect1/accept1 (configs[3] shape)"
    lim.lanes_per_wave = args.lpw
    if args.generic:
        lim.heap_spill_slots = max(lim.heap_spill_slots, 8)
    per_gpu = args.seeds
    total = per_gpu * world
    seed0, count = mdist.shard_range(0, total, rank, world)
    n_streams = max(1, args.streams)
    d_outs = [torch.empty(count * 48, dtype=torch.uint8, device=dev) for _ in range(n_streams)]   # results stay in HBM
    d_out = d_outs[0]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n_streams - 1)]
    stream = streams[0].cuda_stream
    report_stream = torch.cuda.Stream() if world > 1 else None

    # Fully asynchronous steps: the simulation kernel, the summary reduction and (N > 1) the RCCL all-reduce of the
    # 32-byte report are all queued on the stream; the host never waits inside the timed region.
    use_device_report = world == 1 or backend == "nccl"
    ring = torch.zeros((args.steps + args.warmup, 4), dtype=torch.int64, device=dev)   # one 32-byte report per step
    gathered = torch.zeros((args.steps + args.warmup, world, 4), dtype=torch.int64, device=dev) if world > 1 else None

    def step(k, timed):
        # a fresh block of seeds every step so nothing is cached between steps
        if use_device_report:
            si = k % n_streams
            with torch.cuda.stream(streams[si]):
                runtime.run_batch_async(w, seed0 + k * total, count, d_outs[si].data_ptr(), ring[k].data_ptr(),
                                        streams[si].cuda_stream, None, lim, timing_slot=(k % 64) if timed else -1)
            if world > 1:
                # the 32-byte RCCL exchange rides its own stream behind an event: the simulation streams never wait
                # for a collective kernel to find room on a chip whose LDS the simulation keeps full
                ev = torch.cuda.Event()
                ev.record(streams[si])
                report_stream.wait_event(ev)
                with torch.cuda.stream(report_stream):
                    mdist.gather_report_device(ring[k], gathered[k])     # ONE all-gather of 32 bytes per step
        else:   # functional-test hook (gloo on a 1-GPU box): host-side report
            sm = runtime.run_batch_device(w, seed0 + k * total, count, d_out.data_ptr(), stream, None, lim)
            rep = mdist.reduce_report(sm.first_failing_seed, sm.n_failed, sm.total_steps, sm.total_clock_ns, cdev)
            if timed:
                host_tot[0] += rep[1]; host_tot[1] += rep[2]; host_tot[2] += rep[3]; host_tot[3] += sm.kernel_ms

    host_tot = [0, 0, 0, 0.0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k, False)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, True)
    sync()
    dt = time.perf_counter() - t0
    # for reference beside the overlapped figure: the same step with nothing else in flight (one stream), untimed region
    single_ms = None
    if use_device_report and n_streams > 1 and world == 1:
        ns = min(10, args.warmup + args.steps)
        t1 = time.perf_counter()
        for k in range(ns):
            runtime.run_batch_async(w, seed0 + k * total, count, d_outs[0].data_ptr(), ring[k].data_ptr(),
                                    streams[0].cuda_stream, None, lim, timing_slot=-1)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / ns * 1e3
    if use_device_report:
        rows = (mdist.combine_gathered(gathered[args.warmup:]) if world > 1 else ring[args.warmup:]).cpu()
        nfail, steps_total, clock_total = (int(x) for x in rows[:, 1:4].sum(dim=0).tolist())
        nslots = min(args.steps, 64)
        kernel_ms = sum(runtime.timing_ms((args.warmup + args.steps - 1 - i) % 64) for i in range(nslots)) * args.steps / nslots
    else:
        nfail, steps_total, clock_total, kernel_ms = host_tot
    if world > 1:
        t = torch.tensor([dt, kernel_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kernel_ms = float(t[0]), float(t[1])

    if rank == 0:
        seeds_total = total * args.steps
        sim_s = clock_total / 1e9
        # roofline of the dominant kernel (sim_kernel), per launch, from the library's HIP events
        k_avg_ms = kernel_ms / args.steps
        steps_per_launch = steps_total / args.steps / world
        algo_bytes = steps_per_launch * ALGO_BYTES_PER_STEP + count * IO_BYTES_PER_SEED
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9
        g = runtime.geometry(w, lim)
        b = lambda x: "true" if x else "false"
        kname = (f"sim_kernel<Variant<false,true,{g.lanes_per_wave.bit_length() - 1},true,false>>" if g.variant & 8 else
                 f"sim_kernel<Variant<false,{b(g.variant & 1)},6,{b(g.variant & 2)},{b(g.variant & 4)}>>")
        # HBM traffic per launch from the rocprofv3 PMC passes of this same command (tools/prof_pmc.sh ->
        # profiles/r1_traffic.json): FETCH_SIZE (x2, the gfx950 correction of MI355X_MICROARCH.md §HBM) + WRITE_SIZE
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")
        if world == 1 and per_gpu == SEEDS_PER_GPU and args.workload == "pingpong" and args.nodes == N_NODES and os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = (2 * tj["FETCH_SIZE_KB"] + tj["WRITE_SIZE_KB"]) * 1024.0
        line = {
            "metric": "sim_seconds_per_sec", "value": sim_s / dt, "unit": "sim-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{wname}, {per_gpu} seeds per GPU per step"
                                   + (" (BASELINE configs[1])" if args.workload == "pingpong" and args.nodes == N_NODES else ""),
                       "seeds_per_step": total, "parallelism": f"seed-shard x{world}" + (f", {n_streams} concurrent batches per GPU" if n_streams > 1 else "")},
            "extra": {"seeds_per_sec": seeds_total / dt, "first_fail_seeds_per_hour": seeds_total / dt * 3600.0,
                      "executor_steps_per_sec": steps_total / dt,
                      "failed_seeds": nfail, "kernel_ms_per_step": k_avg_ms, "single_stream_ms_per_step": single_ms,
                      "lds_bytes_per_seed": g.lds_bytes_per_seed, "waves_per_cu": g.blocks_per_cu * g.block_threads // 64,
                      "lanes_per_wave": g.lanes_per_wave},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": kname, "algorithmic_bytes_per_launch": algo_bytes,
                         "concurrent_launches": n_streams, "chip_achieved": algo_bytes * args.steps / dt / 1e9,
                         "chip_frac": algo_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBPS,
                         "note": "LDS-resident path: algorithmic bytes (120 B/executor step) never touch HBM; "
                                 "achieved/frac are per launch, chip_* = all launches' bytes / wall time of the timed region"},
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle
            sample = 4 * SEEDS_PER_GPU          # ~14 s of single-thread CPU work
            t1 = time.perf_counter()
            _, osum = oracle.run_batch(w, 0, sample)
            cdt = time.perf_counter() - t1
            line["cpu_baseline"] = {"value": osum.total_clock_ns / 1e9 / cdt, "unit": "sim-s/s", "cores": 1,
                                    "kind": "port",
                                    "sample": f"{sample} seeds of the same workload, single thread, "
                                              f"{cdt:.1f} s ({sample / cdt:.0f} seeds/s, "
                                              f"{osum.total_steps / cdt / 1e6:.1f} M steps/s)"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    runtime.shutdown()


if __name__ == "__main__":
    main()
