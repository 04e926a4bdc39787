// k_main.h — seed_init and the sim_kernel template: the executor loop of one lane.
// Part of sim_kernel.hip (included after k_mem, k_state, k_rng, k_timer, k_net, k_lifecycle, k_channel, k_poll).
#ifndef MADSIM_K_MAIN_H
#define MADSIM_K_MAIN_H

namespace madsim_k {

// ---- per-seed init: Runtime::with_seed_and_config (runtime/mod.rs:53-69) ------------------------
template <class K>
__device__ __forceinline__ void seed_init(const Ctx& c, Lane& L, uint64_t seed) {
    const KParams& P = c.P;
    if (K::G) {
        for (uint32_t w = 0; w < P.gs_plane_words; w++) gs_store32(c.gs, gs_addr_word(c, P.gs_planes + w * 4), 0);
        for (uint32_t w = 0; w < (P.max_tasks + 31) / 32; w++) AMASK(w) = 0;
        OMASK(0) = 0; OMASK(1) = 0;
        if (K::NH) for (uint32_t w = 0; w < P.pool_n / 32; w++) PMASK(w) = 0;
    }
    else for (uint32_t w = 0; w < P.lane_words; w++) RW(w) = 0;     // plane 0 is the ready queue: RW spans all planes
    if (K::DEDUP) { for (uint32_t b = 0; b < P.dedup_n; b++) gs_store32(c.gs, gs_addr_uword(c, P.dedup_off + b * 16u + 12u), 0); L.hazard = 0; L.dd_occ = 0; }   // empty buckets
    for (uint32_t t = 0; t < P.max_tasks; t++) { TWORD(c, t, 0, 0) = 0; if (!K::LIFE) TWORD(c, t, 1, 1) = 0; }
    if (K::FA && P.ipvs_dyn)                               // the services as the table declares them: the ipvs calls made before the first task runs
        for (uint32_t k = 0; k < P.n_services; k++) {
            const uint32_t w0 = SMEM[c.nodet0 + P.svc_off + 2 * k], w1 = SMEM[c.nodet0 + P.svc_off + 2 * k + 1];
            IPVSW(2 * k) = (w0 >> 16) | (w1 << 16);
            IPVSW(2 * k + 1) = (w1 >> 16) | (((w0 >> 8) & 7u) << 16) | ((w0 & 0x8000u) ? 0u : 1u << 24);
        }
    // GlobalRng::new_with_seed -> Xoshiro256PlusPlus::seed_from_u64: SplitMix64 [DEP A.1]
    uint64_t x = seed, z;
#define SPLITMIX(dst) x += 0x9e3779b97f4a7c15ull; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; dst = z ^ (z >> 31)
    SPLITMIX(L.s0); SPLITMIX(L.s1); SPLITMIX(L.s2); SPLITMIX(L.s3);
#undef SPLITMIX
    L.peek = rng_out(L);                                   // (k_rng.h Peek: the first with() has no log entry before it)
    L.rng_calls = 0; L.clock = 0; L.msg_count = 0; L.steps = 0;
    L.pq_n = 0; L.ready_len = 0; L.rq = 0; L.heap_len = 0; L.top_dl = ~0ull; L.top_meta = 0; L.verdict = MADSIM_RUNNING; L.ovf = 0; L.main_done = 0;
    L.loss_pint = P.loss_pint; L.loss_always = P.loss_always;
    // TimeRuntime::new (time/mod.rs:26-38): base_time draw, before logging is enabled
    { uint64_t h = L.trace_hash, n = L.log_len; uint32_t bt = gen_range_small<Variant<false, false, K::LWS, 0, K::RQ>, 31536000u>(c, L); L.trace_hash = h; L.log_len = n;
      if (K::LIFE) NODEW(4 + ((P.n_nodes + 4) >> 2)) = bt; }      // seconds into 2022: SystemTime::now() of MS_OP_TRACE_TIME
    L.trace_hash = FNV_OFFSET; L.obs_hash = FNV_OFFSET; L.log_len = 0;
    // tasks spawned before block_on, then the main task (task/mod.rs:222-235)
    for (uint32_t p = 1; p < P.n_progs; p++) {
        uint32_t fl = (PROGW(c, p) >> 8) & 0xff;
        if (fl & MADSIM_PROG_PRE) spawn_task<K>(c, L, p, !(fl & MADSIM_PROG_INIT));
    }
    spawn_task<K>(c, L, 0, true);
}

// Global-state builds wait on memory most of the time (rocprofv3: 60-70 % of wave cycles), so they trade registers for
// resident waves: MADSIM_G_WAVES_PER_EU waves per SIMD (the second __launch_bounds__ argument caps the VGPR budget) — the single-class
// builds, which fit 168 registers.  The every-class builds do not (24 spilled registers at three waves): two waves per SIMD, 215
// registers, no scratch, and the LDS a third workgroup would have taken holds 15 timer-heap entries per seed instead of 8 —
// the topology 4.65 against 4.13 G steps/s (round 4, gpurun_out/r4v).
#ifndef MADSIM_G_WAVES_PER_EU
#define MADSIM_G_WAVES_PER_EU 3
#endif
#ifndef MADSIM_ALLG_WAVES_PER_EU          // the every-class global-state builds (experiments: tools/build_variant.sh -DMADSIM_ALLG_WAVES_PER_EU=3)
#define MADSIM_ALLG_WAVES_PER_EU 2
#endif
// Priority of a wave that has done `pass` of about `est` passes: who is behind is served first.  Four levels, one per quarter of the
// work, looked at every 16th pass — measured against finer resolution near the end (the last 1/2, 1/4, 1/8: equal on finite sets, 1 %
// worse in long regions; the last 1/4, 1/8, 1/16: no gain at all) and against every 4th / 64th pass (+0.7 % / +0.6 %): profiles/r5_experiments.md.
constexpr uint32_t MADSIM_PRIO_EVERY_MASK = 15u;
__device__ __forceinline__ uint32_t progress_priority(uint32_t pass, uint32_t est) {
    const uint32_t q = (pass < (1u << 30) ? pass : (1u << 30) - 1u) * 4u / est;      // (saturating: pass * 4 wrapped once a wave had run 2^30 passes; one s_min_u32)
    return q >= 3 ? 0u : 3u - q;
}

template <class K>
__global__ __launch_bounds__(256, K::CMP ? 4 : !K::G ? 1 : (K::FEAT & (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR)) == (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR) ? MADSIM_ALLG_WAVES_PER_EU : MADSIM_G_WAVES_PER_EU) void sim_kernel(const KParams P) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // workgroup-shared tables
    uint32_t* sh = SMEM;
    const uint32_t cp0 = table_copy_first(), cps = table_copy_stride(P.waves_per_block);
    for (uint32_t i = cp0; i < P.n_insns * 4; i += cps) sh[P.sh_insns + i] = ((const uint32_t*)P.insns)[i];
    for (uint32_t i = cp0; i < P.n_progs; i += cps) sh[P.sh_progs + i] = P.progs[i];
    for (uint32_t i = cp0; i < P.n_socks; i += cps) sh[P.sh_socks + i] = P.socks[i];
    for (uint32_t i = cp0; i < P.n_nodetab; i += cps) sh[P.sh_nodes + i] = P.nodes[i];
    __syncthreads();

    Ctx c(P);
    c.insn0 = P.sh_insns / 4;
    c.prog0 = P.sh_progs;
    c.sockt0 = P.sh_socks;
    c.nodet0 = P.sh_nodes;
    const uint32_t wbase = wv * P.wave_words;       // this wave's slice of the workgroup's LDS
    if (K::NH) c.heap0 = (P.sh_heap + wbase) / 2 + lane;          // 8-byte entries: a uint2 index
    else if (K::LIFE) c.heap0 = (P.sh_heap + wbase) / 4 + lane;
    else if (K::CMP) { c.heap0 = (P.sh_heap + wbase) / 2 + lane; c.heapm0 = 0; }        // 8-byte entries 1 .. heap_lds - 1 (entry 0: registers)
    else { c.heap0 = (P.sh_heap + wbase) / 2 + lane; c.heapm0 = P.sh_heap + wbase + ((P.heap_lds * 2) << P.lw_shift) + lane; }
    c.lws = P.lw_shift;
    const uint32_t pl = P.sh_planes + wbase + lane;
    c.ready0 = pl + (P.off_ready << P.lw_shift);
    c.amask0 = pl + (P.off_amask << P.lw_shift);
    c.omask0 = pl + (P.off_omask << P.lw_shift);
    c.pmask0 = pl + (P.off_pmask << P.lw_shift);
    if (K::G) {                                      // byte offsets inside the lane's global state block
        c.task0 = 0;
        c.sock0 = P.gs_planes + P.off_socks * 4; c.hand0 = P.gs_planes + P.off_handles * 4; c.node0 = P.gs_planes + P.off_nodes * 4;
        c.clog0 = P.gs_planes + P.off_clog * 4; c.pause0 = P.gs_planes + P.off_pause * 4; c.greg0 = P.gs_planes + P.off_greg * 4;
        c.conn0 = P.gs_planes + P.off_conn * 4; c.hook0 = P.gs_planes + P.off_hooks * 4; c.ipvs0 = P.gs_planes + P.off_ipvs * 4;
    } else {
        c.task0 = (P.sh_tasks + wbase) / 4 + lane;
        c.task1 = (P.sh_tasks + wbase + ((P.max_tasks * 4) << P.lw_shift)) / 2 + lane;      // base-op builds: behind the unit0 array
        if (K::CMP) {          // slots 1 .. max_tasks - 1 in LDS (slot 0 = the main task: global memory), the arrays indexed by slot - 1
            c.task1 = (P.sh_tasks + wbase + (((P.max_tasks - 1) * 4) << P.lw_shift)) / 2 + lane - (1u << P.lw_shift);
            c.task0 -= 1u << P.lw_shift;
        }
        c.sock0 = pl + (P.off_socks << P.lw_shift);
        c.hand0 = pl + (P.off_handles << P.lw_shift);
        c.node0 = pl + (P.off_nodes << P.lw_shift);
        c.clog0 = pl + (P.off_clog << P.lw_shift);
        c.pause0 = pl + (P.off_pause << P.lw_shift);
        c.greg0 = pl + (P.off_greg << P.lw_shift);
        c.conn0 = pl + (P.off_conn << P.lw_shift);
        c.hook0 = pl + (P.off_hooks << P.lw_shift);
        c.ipvs0 = pl + (P.off_ipvs << P.lw_shift);
    }
    if (lane >= (1u << P.lw_shift)) return;      // sub-wave occupancy: only lw = 2^lw_shift lanes carry seeds
    if (EXP_LANE_DIV > 1 && (lane % EXP_LANE_DIV)) return;          // (timing experiments only: tools/experiment/k_experiment.h EXP_HALF_LANES)
    const uint32_t glane = (((blockIdx.x * P.waves_per_block + wv) << P.lw_shift) + lane) / EXP_LANE_DIV;
    c.spill_off = glane * (K::NH ? 8u : 16u);
    c.spill = buf_make(P.spill, (uint64_t)P.heap_spill * P.total_lanes * (K::NH ? 8u : 16u));
    c.gs_lane = glane;
    c.gs = buf_make(P.gstate, (uint64_t)P.gs_stride * P.total_lanes);
    c.tlog = P.trace_log;

    Lane L;
#ifdef MADSIM_K_PROF
    for (int i = 0; i < 12; i++) L.prof_acc[i] = 0;
    L.prof_t = __builtin_readcyclecounter(); uint64_t prof_iters = 0;
#endif
    L.exact = 0; L.hazard = 0;
    uint64_t next = glane;          // first unit of lane g; then g+G, g+2G, ... or the work queue (below)
    bool have = false;
    // Progress-based issue priority (round 5).  The SIMD's arbiter serves priority first, then the OLDEST wave: launches resident
    // together do not share a SIMD evenly, the oldest runs at nearly its solo speed and the youngest gets the rest — so a finite set
    // of launches (the sub-batches of one madsim_hip_run_batch call, a campaign's last batches, a short timed region) ends 0.5-0.9 ms
    // apart and the last launch finishes on SIMDs it has to itself, at a third of their issue rate (profiles/r5_experiments.md).
    // A wave that knows how far through its work it is can undo that: priority 3 in the first quarter of its passes, 2 / 1 / 0 in
    // the next ones — whoever is behind is served first, co-resident launches finish together.  "How many passes" is what the longest
    // finished wave of this workload and launch shape counted (one word per (workload, limits, seeds per lane), written below); until one
    // has finished: no priority.
    uint32_t pass_est = (K::TRACE || !P.iter_est) ? 0u : wave_uniform(*P.iter_est);
    uint32_t pass = 0;
    for (;;) {
        if (!K::TRACE && P.iter_est) {
            const uint32_t pu = wave_uniform(pass);
            if ((pu & MADSIM_PRIO_EVERY_MASK) == 0) {
                // (the first launches of a workload start without an estimate: they look again until a wave has finished — the drain, where
                //  the priorities matter, comes after that)
                if (!pass_est && (pu & 63u) == 0) pass_est = wave_uniform(__atomic_load_n(P.iter_est, __ATOMIC_RELAXED));
                if (pass_est) wave_set_priority(progress_priority(pu, pass_est));
            }
        }
        pass++;
        if (!have) {
            if (next >= P.count) break;
            seed_init<K>(c, L, P.seed_list ? P.seed_list[next] : P.seed0 + next);
            have = true;
        }
        // One iteration = one pass of the block_on loop body (task/mod.rs:239-259):
        //   [poll]  ready queue non-empty: one run_all_ready iteration (pop, poll, 50..100 ns advance)
        //   [fire]  Timer::expire up to `now`; while the queue stays empty: is_finished / deadlock checks
        //           and advance_to_next_event, firing again — the ONLY place timers fire.
        // A lane that polls and then runs dry fires its next timer in the same pass, so in steady state
        // every lane does one poll and one timer fire per iteration and the wave stays in phase.
#ifdef MADSIM_K_PROF
        prof_iters++;
#endif
        uint64_t now = L.clock;
        PROBE(0);
        REG(0);
        if (L.ready_len > 0) {
            // Latency hiding: the queue usually holds exactly one task, so ready[0] and its two state units are
            // loaded BEFORE the draw loop (whose rejection retries take hundreds of cycles) and used if idx == 0.
            const uint32_t slot0 = K::RQ ? (uint32_t)(L.rq & 0xff) : rq_get<K>(c, 0);
            const uint4 pu0 = TU(c, slot0, 0), pu1 = load_u1<K>(c, slot0);
            PollPrefetch pp = poll_prefetch<K>(c, slot0);     // (global-state builds: more of the slot's granule, same cache line)
            // try_recv_random (utils/mpsc.rs:73-83): idx drawn even when len == 1
            uint32_t idx = gen_index<K>(c, L, L.ready_len);
            L.ready_len--;
            uint32_t slot = slot0;
            uint4 u0 = pu0, u1 = pu1;
            if (K::RQ) {
                if (idx != 0) { slot = (uint32_t)(L.rq >> (8 * idx)) & 0xff; u0 = TU(c, slot, 0); u1 = load_u1<K>(c, slot); pp = poll_prefetch<K>(c, slot); }
                uint64_t last = (L.rq >> (8 * L.ready_len)) & 0xff;      // swap_remove on bytes
                L.rq = (L.rq & ~(0xffull << (8 * idx))) | (last << (8 * idx));
                L.rq &= ~(0xffull << (8 * L.ready_len));
            } else {
                if (idx != 0) { slot = rq_get<K>(c, idx); u0 = TU(c, slot, 0); u1 = load_u1<K>(c, slot); pp = poll_prefetch<K>(c, slot); }
                if (idx != L.ready_len) rq_set<K>(c, idx, rq_get<K>(c, L.ready_len));       // swap_remove
            }
            L.steps++;
            bool panicked = false;
            PROBE(1);
            REG(26);
            bool parked = false;
            if (K::FN && (u0.x & (TF_CANCEL | TF_KILLED))) {   // task/mod.rs:269-273: drop(runnable)
                task_finish<K>(c, L, slot, H_CANCELLED);
            } else if (K::FN && P.uses_pause && ((NODEW(1) >> (PROGW(c, u0.x >> 24) & 0xff)) & 1)) {
                uint32_t n = PAUSEW(0);                       // :274-277: park the Runnable; no poll, no time advance
                PAUSEW(1 + n) = slot; PAUSEW(0) = n + 1;
                L.steps--;
                parked = true;
            } else {
                u0.x = (u0.x & ~TF_SCHED) | TF_RUN;          // async-task run(): SCHEDULED -> RUNNING
                if (K::FN) L.panic_code = MADSIM_PANIC_CODE_OTHER;
                panicked = poll_task<K>(c, L, slot, u0, u1, pp);
                if (!panicked && (u0.x & TF_ALIVE)) {
                    if (u0.x & TF_SCHED) ready_push<K>(c, L, slot);   // woken while running: re-queue after the poll
                    u0.x &= ~TF_RUN;
                    TU(c, slot, 0) = u0;
                }
            }
            PROBE(2);
            if (K::FN && panicked && P.has_restart_on_panic) {   // task/mod.rs:289-314
                uint32_t node = PROGW(c, u0.x >> 24) & 0xff;
                // restart_on_panic || restart_on_panic_matching.iter().any(|s| error_msg.contains(s)) (task/mod.rs:297-300)
                // (the host evaluated `contains` for every message code: one bit of the node's 256-bit row, geometry.h build_tables)
                const uint32_t nw = NODET(c, node);
                const bool matches = (nw & MADSIM_NODE_RESTART_MATCHING) &&
                                     ((SMEM[c.nodet0 + P.pm_off + node * 8 + (L.panic_code >> 5)] >> (L.panic_code & 31)) & 1);
                if ((nw & MADSIM_NODE_RESTART_ON_PANIC) || matches) {
                    // async-task's panic guard already dropped the future and notified the awaiter
                    TU(c, slot, 0) = u0;
                    task_finish<K>(c, L, slot, H_CANCELLED);
                    // delay = gen_range(1 s..10 s) in ONE with(): UniformDuration Medium path [DEP A.3]
                    const uint64_t range = 9000000000ull, zone = ~0ull - ((~0ull - range + 1) % range);
                    uint64_t v;
                    do { v = rng_next(L); } while (v * range > zone);       // (an extended-build path: no peek, k_rng.h)
                    rng_log<K>(c, L);
                    uint64_t delay = NS_PER_S + __umul64hi(v, range);
                    node_kill<K>(c, L, node);                 // self.kill(node_id)
                    if (!timer_add<K>(c, L, L.clock + delay, (EV_RESTART << EV_SHIFT) | node, 0)) OVF_SET(L, OVF_CAP);
                    panicked = false;
                }
            }
            if (panicked) L.verdict = MADSIM_PANIC;          // resume_unwind (:315): no advance, no expire
            else if (!parked) L.clock += 50 + gen_range_small<K, 50>(c, L);   // :319-321, then Timer::expire (time/mod.rs:103-106)
            now = L.clock;
            PROBE(3);
        }
        // One exit (every `break` out of a hot loop is an exit edge whose phis cost register copies): the checks decide a
        // verdict — or that the run queue has work again — in the reference's order, the loop leaves when anything was decided.
        bool idle_jump = false;
        while (L.verdict == MADSIM_RUNNING) {
            REG(19);
            timer_expire<K>(c, L, now);
            uint32_t v = MADSIM_RUNNING;
            if (idle_jump) {
                L.clock = now;                                // time/mod.rs:55: after the callbacks
                idle_jump = false;
                if (P.time_limit && L.clock >= P.time_limit) v = MADSIM_TIME_LIMIT;     // task/mod.rs:253-258
            }
            const bool more = L.ready_len > 0;                // back to run_all_ready
            if (v == MADSIM_RUNNING) {
                if (L.steps >= P.max_steps) v = MADSIM_STEP_LIMIT;
                else if (more) { }
                else if (L.main_done) v = MADSIM_PASS;                                   // :241-243
                else if (L.heap_len == 0) v = MADSIM_DEADLOCK;                           // :250
            }
            L.verdict = v;
            if (v != MADSIM_RUNNING || more) break;
            now = L.top_dl + 50;                              // advance_to_next_event (time/mod.rs:47-53)
            idle_jump = true;
        }
        PROBE(4);
        // MADSIM_STATE_DEDUP_TIMERS: two different events tied on a deadline — this seed once more from the start, every timer a
        // heap entry (k_timer.h dedup_note); nothing of the abandoned attempt is reported
        if (K::DEDUP && L.hazard) { L.exact = 1; have = false; continue; }
        if (L.ovf) L.verdict = (L.ovf & OVF_CAP) ? MADSIM_OVERFLOW : (L.ovf & OVF_BUG) ? MADSIM_INTERNAL : MADSIM_UNSUPPORTED;
        if (L.verdict != MADSIM_RUNNING) {
            REG(25);
            madsim_result_t r;
            r.verdict = L.verdict; r.steps = L.steps; r.clock_ns = L.clock; r.msg_count = L.msg_count;
            r.rng_calls = L.rng_calls; r.trace_hash = (K::NOLOG || (K::LOGSW && P.no_log)) ? 0 : L.trace_hash; r.obs_hash = L.obs_hash;
            if (r.verdict >= MADSIM_UNSUPPORTED) { r.steps = 0; r.clock_ns = 0; r.msg_count = 0; r.rng_calls = 0; r.trace_hash = 0; r.obs_hash = 0; }   // the verdict is the whole answer
            P.out[next] = r;
            if (K::TRACE) *P.trace_len = L.log_len;
            have = false;
            if (K::DEDUP) L.exact = 0;
            // next unit: static striding, or the per-launch work queue (a lane whose seeds end early — deadlocks under
            // packet loss — then keeps pulling work instead of idling behind the slowest lane of its stride)
            if (P.work_ctr) next = P.total_lanes + atomicAdd(P.work_ctr, 1ull);
            else next += P.total_lanes / EXP_LANE_DIV;
        }
    }
    // the estimate = the longest wave of this (workload, launch shape) so far: the maximum over the wave's lanes (lane 0 may have left long
    // before the others when seeds differ in length or come from the work queue), one atomic per wave into the word the host keyed by
    // workload, limits and seeds per lane (madsim_hip.cpp upload_workload)
    // (the base-op builds — every seed of theirs runs the same program at nearly the same pace — keep round 5's form: lane 0's own count)
    if (!K::TRACE && P.iter_est) {
        if (K::LIFE) {
            const uint32_t wmax = wave_max_u32(pass);
            if (wave_first_lane() && wmax > 16) atomic_max_u32(P.iter_est, wmax);
        } else if (lane == 0 && pass > 16) *P.iter_est = pass;
    }
#ifdef MADSIM_K_PROF
    PROBE2(0);
    if (lane == 0 && P.prof) { for (int i = 0; i < 12; i++) atomicAdd((unsigned long long*)&P.prof[i], (unsigned long long)L.prof_acc[i]); atomicAdd((unsigned long long*)&P.prof[12], (unsigned long long)prof_iters); atomicAdd((unsigned long long*)&P.prof[13], 1ull); }
#endif
}

}  // namespace madsim_k

#endif
