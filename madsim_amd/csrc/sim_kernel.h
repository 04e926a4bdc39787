// sim_kernel.h — kernel parameter block shared by the host side (madsim_hip.cpp) and sim_kernel.hip.
#ifndef MADSIM_SIM_KERNEL_H
#define MADSIM_SIM_KERNEL_H

#include <stdint.h>

#include "../../include/madsim_hip.h"

#define MADSIM_RUNNING 0xffu /* lane-internal: seed still executing */

/* classes of extended ops a kernel variant may carry (Variant<..., FEAT, ...>, KParams.features) */
#define MADSIM_FEAT_TIME 1  /* timeout(recv_from), t0 family (mark / sleep_until / assert_elapsed), advance, trace_time */
#define MADSIM_FEAT_CHAN 2  /* connect1 / accept1 / channel send + recv                                               */
#define MADSIM_FEAT_RPC  4  /* typed RPC call / reply                                                                   */
#define MADSIM_FEAT_NODE 8  /* kill / restart / pause / resume / abort / is_exit, init programs, restart_on_panic       */
#define MADSIM_FEAT_ADDR 16 /* general address resolution (network.rs:272-313): 0.0.0.0 / 127.0.0.1 entries, IP-less nodes, several
                               table entries naming one address.  Builds without it take every address for a distinct node IP.   */
#define MADSIM_FEAT_ALL  31
/* Variant<..., FEAT, ...> only, never in KParams.features: a base-op build without the determinism-log fold (rng_log), selected
   by KParams.no_log.  The other builds test KParams.no_log at run time (a wave-uniform branch); the base-op builds on full waves
   are issue-bound, so they get a twin compiled without the code instead. */
#define MADSIM_FEAT_NOLOG 32
/* Variant<..., FEAT, ...> only: the COMPACT base-op build (KParams.compact) — a fourth wave per SIMD for small workloads.
   A 4-wave workgroup may hold 40 960 bytes of LDS if four are to share a CU; the 4-node ping-pong's 200 bytes per seed make
   it 51 200.  The compact layout gets it to 152: timer-heap entries of 8 bytes (the low 32 bits of the deadline + meta: exact
   while every live deadline lies within 2^31 ns of the clock, which the host checks against the workload's longest sleep), the
   heap's root in registers, and the main task's state — polled a handful of times per run — in global memory. */
#define MADSIM_FEAT_COMPACT 64
/* Variant<..., FEAT, ...> only: global-state builds with NARROW timer-heap entries (KParams.narrow, madsim_limits_t.state_mem |
   MADSIM_STATE_NARROW_HEAP): every heap entry — in LDS and in the spill region — is 8 bytes {low deadline word, meta} instead of 16
   {deadline, meta, payload}; the 32-bit payload and the tag / sender fields of a datagram delivery (a fifth of the entries) live in a
   per-seed record pool in global memory, written once by the push and read once when the entry is the root (k_timer.h).  Twice
   the heap levels per LDS byte, half the bytes per spilled level.  Exact while every live deadline lies within 2^31 ns of the clock:
   checked per push on the device (a violation is a capacity verdict: the re-run uses the wide entries). */
#define MADSIM_FEAT_NARROW 128
/* Global-state builds: identical wake-ups are fired as a batch (k_net.h timer_expire).  Sleep::poll registers ANOTHER timer with the same
   deadline and waker on every not-elapsed poll (time/sleep.rs:51-53), so more than half of the topology's heap entries are copies of an
   earlier one; copies leave the heap back to back, and every one after the first finds its task SCHEDULED already (or gone): a step
   each and nothing else.  The batch pops them without their callback's loads.  (The part of VERDICT r5 #1c that removes work.) */
#ifndef MADSIM_FIRE_COPIES
#define MADSIM_FIRE_COPIES 2        /* 1: a loop of its own behind the wake-up; 2: popped by the fire loop itself, in the other lanes' trips */
#endif
/* Builds with a spill region: BinaryHeap::pop written top-down (k_timer.h timer_pop) — the array sift_down_to_bottom + sift_up leave, without the
   levels below the moved entry's final slot. */
/* Base-op builds that keep the determinism log: the ready-queue draw starts from the number the previous with()'s log entry computed (k_rng.h gen_index). */
#ifndef MADSIM_RNG_PEEK
#define MADSIM_RNG_PEEK 2        /* 1: the ready-queue draw only; 2: every with() whose accepted output is not needed behind its loop */
#endif
#ifndef MADSIM_RNG_PEEK_LIFE
#define MADSIM_RNG_PEEK_LIFE 0   /* ... in the extended builds too */
#endif
/* MADSIM_STATE_DEDUP_TIMERS builds: the occupancy of the re-registration table mirrored in a register (k_timer.h dedup_note). */
#ifndef MADSIM_DEDUP_OCC
#define MADSIM_DEDUP_OCC 1
#endif
/* Extended builds: whether any clog exists is mirrored in the lane, a send loads the clog masks only then (k_channel.h net_try_send). */
#ifndef MADSIM_CLOG_MIRROR
#define MADSIM_CLOG_MIRROR 1
#endif
#ifndef MADSIM_POP_TOPDOWN
#define MADSIM_POP_TOPDOWN 1
#endif
/* ... with the LDS-resident levels walked before the heap's last entry (a spill-region load) has arrived (k_timer.h timer_pop). */
/* Narrow-heap global-state builds: a poll round requests the parent of the slot its first push will start from when it begins (k_timer.h
   timer_push_prefetch), beside its handler's own reads. */
#ifndef MADSIM_PUSH_PREFETCH
#define MADSIM_PUSH_PREFETCH 0        /* measured: topology 5.75 with, 5.76 without (profiles/r6_ab_chains.txt): the three registers cost what the round trip saves */
#endif
#ifndef MADSIM_POP_LDS_FIRST
#define MADSIM_POP_LDS_FIRST 1
#endif

namespace madsim_k {

struct KParams {
    // workload tables in device memory (copied into LDS by every workgroup)
    const uint4*    insns;     // {op|a<<8|b<<16, imm, fused assert value, fused post-chain word} (geometry.h build_tables)
    const uint32_t* progs;     // node | flags<<8 | entry<<16
    const uint32_t* socks;     // node | kind<<8 | port<<16 (a SocketAddr as one word)
    const uint32_t* nodes;     // n_nodetab words: per node its flags word; then (pm_off != 0) 8 words per node: the 256-bit row "a panic
                               // with message code c restarts this node"; then (n_services != 0) 2 words per IPVS service:
                               // vaddr | n_servers<<8 (| 0x80<<8: declared absent) | servers[0]<<16 | servers[1]<<24, servers[2..5]
    const uint64_t* dur_table; // per MS_OP_SLEEP_RAND (its `a` is rewritten to an index): {mode, low, range, zone}
    uint32_t n_insns, n_progs, n_socks, n_nodes;
    // net config (Bernoulli p_int, UniformDuration parameters precomputed on the host)
    uint64_t loss_pint; uint32_t loss_always; uint32_t buggify; uint64_t bug_pint;
    uint32_t lat_mode; uint32_t has_clog_link; uint32_t has_clog; uint32_t pad0; uint64_t lat_low, lat_range, lat_zone;
    uint64_t loss_table_pint[4]; uint32_t loss_table_always[4];
    // MS_OP_SET_LATENCY (extended builds): madsim_config_t.lat_table as UniformDuration parameters; a lane keeps the index of its
    // current entry (Lane::loss_always bits 4-6: 0 = lat_* above) and sample_latency selects
    uint32_t uses_set_lat; uint32_t lat_tab_mode[4]; uint64_t lat_tab_low[4], lat_tab_range[4], lat_tab_zone[4];
    // limits
    uint64_t time_limit; uint32_t max_steps;
    // capacities
    uint32_t heap_lds, heap_spill, max_tasks, mbox_regs, mbox_msgs;
    uint32_t task_units, sock_words, lane_words, uniq_addr;
    uint32_t lw_shift;         // log2(seed-carrying lanes per wave): lane stride of every per-lane LDS array
    // LDS layout, in 32-bit words: workgroup-shared tables, heap units and task units (16-byte
    // aligned, [unit][lane]), then the 32-bit planes ([word][lane])
    uint32_t sh_insns, sh_progs, sh_socks, sh_nodes, sh_heap, sh_tasks, sh_planes;
    // a workgroup is waves_per_block independent waves (one per SIMD): the tables once, then one
    // [heap][tasks][planes] slice of wave_words per wave; sh_heap/sh_tasks/sh_planes are wave 0's
    uint32_t waves_per_block, wave_words;
    // per-lane plane offsets (in words)
    uint32_t off_ready, off_socks, off_handles, off_nodes, off_clog, off_pause, off_greg, off_conn, off_hooks;
    uint32_t n_nodetab, pm_off, svc_off, n_services;   // layout of the node table (word offsets from its start; 0 = absent)
    uint32_t ipvs_dyn;         // MS_OP_IPVS present: the services' server lists are per-seed state (k_state.h IPVSW), initialised from the table
    uint32_t panic_dyn_max;    // largest code a run-time formatted panic message may have (madsim_workload_t.panic_dyn_max)
    uint32_t off_ipvs;         // per-seed plane: one round-robin counter per IPVS service (net/ipvs.rs rr_index)
    uint32_t uses_eph;         // ephemeral Endpoint handles in the socket table (geometry.h device_socks)
    uint32_t uses_hooks;       // MS_OP_HOOK_REQ / MS_OP_HOOK_RSP present: one hook word per node
    uint32_t uses_chan, max_conns, chan_queue, conn_words;   // reliable channel (connect1/accept1) state, if used
    uint32_t chan_unit;            // index of the task unit holding the (tx, rx) pair state
    uint32_t uses_rpc, rpc_unit;   // typed RPC: index of the task unit holding the response tags
    uint32_t rq_in_reg;        // the ready queue needs no LDS region (register variant)
    // global-state builds (Variant::G): a lane's task table + planes are gs_stride logical bytes — [task units: max_tasks x
    // task_units x 16 B][plane words (off_* count from gs_planes, the ready queue stays in LDS)] — stored across the launch
    // as [unit][global lane] then [word][global lane] (k_state.h gs_addr_*): logical byte `at` of lane g lives at
    // at * total_lanes + g * 16 (units) or + g * 4 (words)
    uint32_t gstate_mode, gs_stride, gs_planes, gs_plane_words;
    uint32_t gs_gran_sh;         // log2 of a task slot's granule in the global block: [slot][lane][task_units x 16 bytes, padded to 2^gs_gran_sh]
    uint32_t off_amask, off_omask;     // LDS plane words (after the ready queue) of the alive-task / owned-socket masks
    uint8_t* gstate;
    uint32_t lifecycle;        // any extended op: the extended LDS layout (features != 0)
    uint32_t features;         // MADSIM_FEAT_* classes the workload needs
    uint32_t uses_pause, has_restart_on_panic, restart_nodes;   // restart_nodes: bit n = NodeBuilder::restart_on_panic
    // batch
    uint64_t seed0, count;
    const uint64_t* seed_list; // non-null: unit i runs seed_list[i] instead of seed0 + i (compacted re-run of overflowed seeds)
    madsim_result_t* out;
    unsigned long long* work_ctr; // non-null: a lane that finishes unit i pulls unit total_lanes + atomicAdd(work_ctr, 1) next
    uint4* spill;
    uint32_t total_lanes;
    // trace mode (single seed)
    uint8_t* trace_log; uint64_t trace_cap; uint64_t* trace_len;
    uint32_t compact;          // base-op builds: the compact LDS layout (MADSIM_FEAT_COMPACT); gstate then holds the main tasks' records
    uint32_t no_log;           // madsim_limits_t.no_trace_hash: skip rng_log, report trace_hash = 0 (trace launches ignore it)
    // MADSIM_STATE_DEDUP_TIMERS (global-state timeout-only build): dedup_n (a power of two, 0 = off) 16-byte buckets
    // {deadline lo, hi, wake meta, count} behind the task units, at logical byte dedup_off of the lane's block (k_timer.h dedup_note)
    uint32_t dedup_n, dedup_off;
    // MADSIM_STATE_NARROW_HEAP (Variant::NH): 8-byte heap entries; pool_n (a multiple of 32) delivery records of 8 bytes {meta, payload}
    // behind the planes of the lane's global block, at logical byte pool_off ([record][global lane]); free / used mask = pool_n / 32 LDS
    // plane words from off_pmask
    uint32_t narrow, pool_n, pool_off, off_pmask;
    uint64_t* prof;            // profiling builds (tools/experiment): per-phase cycle accumulators
    uint32_t* iter_est;        // one word per workload: the passes a wave of it runs, as the last finished wave counted them (0 = nothing
                               // finished yet); null = no progress-based priority (k_main.h wave_progress_priority)
};

// Kernel variants (Variant<TRACE, SPILL, LWS, FEAT, RQ>): the trace build; for base-op workloads on full 64-lane waves one
// build per (heap spill, register ready queue) combination plus a runtime-lane-stride build; single-class builds for
// workloads that only use timeouts (FEAT_TIME) or only the reliable channel (FEAT_CHAN) — a third of the code and
// fewer registers than the full build; the full build for every lane stride (64/32/16/8 seed lanes per wave, runtime);
// and the global-state builds (G: task table + planes in global memory) of the three extended classes.
#ifndef MADSIM_FOR_EACH_VARIANT      // (tools/ may compile a subset: -D'MADSIM_FOR_EACH_VARIANT(X)=X(false,false,6,0,true,false)')
#define MADSIM_FOR_EACH_VARIANT(X)                     \
    X(true, true, -1, MADSIM_FEAT_ALL, false, false)   \
    X(false, false, 6, 0, false, false)                \
    X(false, false, 6, 0, true, false)                 \
    X(false, false, 6, MADSIM_FEAT_NOLOG, false, false) \
    X(false, false, 6, MADSIM_FEAT_NOLOG, true, false)  \
    X(false, false, 6, MADSIM_FEAT_COMPACT, true, false) \
    X(false, false, 6, MADSIM_FEAT_COMPACT | MADSIM_FEAT_NOLOG, true, false) \
    X(false, true, 6, 0, true, false)                  \
    X(false, true, 6, 0, false, false)                 \
    X(false, true, -1, 0, false, false)                \
    X(false, true, -1, MADSIM_FEAT_TIME, false, false) \
    X(false, true, -1, MADSIM_FEAT_CHAN, false, false) \
    X(false, false, 6, MADSIM_FEAT_ALL, false, false)  \
    X(false, true, 6, MADSIM_FEAT_ALL, false, false)   \
    X(false, true, 5, MADSIM_FEAT_ALL, false, false)   \
    X(false, true, 4, MADSIM_FEAT_ALL, false, false)   \
    X(false, true, 3, MADSIM_FEAT_ALL, false, false)   \
    X(false, true, -1, MADSIM_FEAT_ALL, false, false)  \
    X(false, true, 6, MADSIM_FEAT_TIME, false, true)   \
    X(false, true, 5, MADSIM_FEAT_TIME, false, true)   \
    X(false, true, 6, MADSIM_FEAT_CHAN, false, true)   \
    X(false, false, 6, MADSIM_FEAT_CHAN, false, true)  \
    X(false, true, 6, MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR, false, true) \
    X(false, true, 6, MADSIM_FEAT_ALL, false, true)    \
    X(false, true, 6, MADSIM_FEAT_TIME | MADSIM_FEAT_NARROW, false, true) \
    X(false, true, 5, MADSIM_FEAT_TIME | MADSIM_FEAT_NARROW, false, true) \
    X(false, true, 6, (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR) | MADSIM_FEAT_NARROW, false, true)
#endif

// Which compiled specialisation of sim_kernel a parameter block runs on (one rule for the launcher and for
// madsim_hip_geometry's report).  Compiled set = MADSIM_FOR_EACH_VARIANT in sim_kernel.hip.
struct VariantSel { int trace, spill, lws, feat, rq, g; };
inline VariantSel select_variant(const KParams& P, bool trace) {
    const int spill = P.heap_spill > 0, lw = (int)P.lw_shift, feat = (int)P.features;
    if (trace) return {1, 1, -1, MADSIM_FEAT_ALL, 0, 0};
    if (feat == 0) {                                                    // base ops only
        if (lw == 6 && P.compact) return {0, 0, 6, MADSIM_FEAT_COMPACT | (P.no_log ? MADSIM_FEAT_NOLOG : 0), 1, 0};
        if (lw == 6) return {0, spill, 6, P.no_log && !spill ? MADSIM_FEAT_NOLOG : 0, (int)P.rq_in_reg, 0};
        return {0, 1, -1, 0, 0, 0};                                     // sub-wave lane stride: runtime-stride build
    }
    // single-class workloads: a build without the other classes' code
    // (general address resolution only exists in the full build: rare, and it would cost the lean builds ~5 %)
    const int cls = (feat & ~MADSIM_FEAT_TIME) == 0 ? MADSIM_FEAT_TIME : (feat & ~MADSIM_FEAT_CHAN) == 0 ? MADSIM_FEAT_CHAN : MADSIM_FEAT_ALL;
    if (P.gstate_mode) {                                                // task table + planes in global memory: full waves
        const int nh = P.narrow ? MADSIM_FEAT_NARROW : 0;               // 8-byte heap entries (make_geometry sets it only where such a build exists)
        if (cls == MADSIM_FEAT_ALL && !(feat & MADSIM_FEAT_ADDR)) return {0, 1, 6, (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR) | nh, 0, 1};   // plain addresses
        // (connection workloads keep short heaps — a handful of backoff / timeout timers — that sit in LDS whole: a build
        // without the spill path, like the base-op builds have)
        // 32 seed lanes per wave: the timeout-only build has the variant.  (Round 4 also tried 16 lanes there — the election loop fell
        // from 9.2 to 6.5 G steps/s — and 32 lanes on the every-class build — the topology: 4.10 G steps/s with 20 heap entries in LDS
        // against 4.30 on full waves with 8: neither is compiled, make_geometry refuses both.)
        if (lw == 5 && cls == MADSIM_FEAT_TIME) return {0, 1, 5, cls | nh, 0, 1};
        return {0, cls == MADSIM_FEAT_CHAN ? spill : 1, 6, cls | nh, 0, 1};
    }
    if (cls != MADSIM_FEAT_ALL) return {0, 1, -1, cls, 0, 0};
    if (lw == 6) return {0, spill, 6, MADSIM_FEAT_ALL, 0, 0};
    if (lw >= 3 && lw <= 5) return {0, 1, lw, MADSIM_FEAT_ALL, 0, 0};
    return {0, 1, -1, MADSIM_FEAT_ALL, 0, 0};
}

// Is the build `v` names one of the compiled set?  (Host-side twin of the dispatch chain in sim_kernel.hip: same macro, no kernels.)
inline bool variant_compiled(const VariantSel& v) {
    bool hit = false;
#define MADSIM_VARIANT_HIT(T_, S_, L_, F_, R_, G_) \
    hit = hit || (v.trace == (int)(T_) && v.spill == (int)(S_) && v.lws == (L_) && v.feat == (F_) && v.rq == (int)(R_) && v.g == (int)(G_));
    MADSIM_FOR_EACH_VARIANT(MADSIM_VARIANT_HIT)
#undef MADSIM_VARIANT_HIT
    return hit;
}

// The contract between a parameter block and the build that will run it — checked by make_geometry before anything can be launched
// (round 4 ran the 64-lane build of the every-class kernel on a 32-lane geometry once: a selection-order slip that hung a GPU box;
// with this check such a pair is MADSIM_E_LIMITS, never a launch) and walked by tests/test_geometry_consistency.py over every
// workload class x state_mem x lanes_per_wave.  Returns nullptr when consistent, else what is wrong.
inline const char* variant_mismatch(const KParams& P, const VariantSel& v, bool trace) {
    if (!variant_compiled(v)) return "select_variant names a build that is not compiled";
    if (v.lws >= 0 && v.lws != (int)P.lw_shift) return "the build's compile-time lane stride differs from the geometry's";
    if (P.lw_shift < 3 || P.lw_shift > 6) return "lane stride outside 8..64 seed lanes per wave";
    if ((v.g != 0) != (P.gstate_mode != 0)) return "global-state build on an LDS-resident layout (or the reverse)";
    if ((v.rq != 0) != (P.rq_in_reg != 0)) return "register ready queue build on a layout with an LDS ready queue (or the reverse)";
    if (v.rq && (P.max_tasks > 8 || P.lw_shift != 6 || P.lifecycle)) return "register ready queue needs <= 8 tasks, full waves, base ops";
    if (!v.spill && P.heap_spill) return "a build without the spill path on a geometry with spilled heap levels";
    const int classes = v.feat & MADSIM_FEAT_ALL;
    if (((int)P.features & ~classes) != 0) return "the build lacks an op class the workload uses";
    if ((classes != 0) != (P.lifecycle != 0)) return "extended-op build on the base-op LDS layout (or the reverse)";
    if (((v.feat & MADSIM_FEAT_COMPACT) != 0) != (P.compact != 0)) return "compact build on a plain layout (or the reverse)";
    if (P.compact && (P.heap_spill || P.lifecycle || P.lw_shift != 6 || !P.rq_in_reg || P.max_tasks > 8)) return "compact layout outside its conditions";
    if ((v.feat & MADSIM_FEAT_NOLOG) && (!P.no_log || trace)) return "a build without the determinism-log fold for a run that wants it";
    if (P.dedup_n && !(v.g && classes == MADSIM_FEAT_TIME)) return "re-registration counts on a build that does not carry them";
    if (P.dedup_n & (P.dedup_n - 1)) return "dedup_n must be a power of two";
    if (((v.feat & MADSIM_FEAT_NARROW) != 0) != (P.narrow != 0)) return "narrow-heap build on a wide-entry layout (or the reverse)";
    if (P.narrow && (!v.g || (P.pool_n & 31) || !P.pool_n)) return "narrow heap entries outside their conditions";
    if (P.waves_per_block != 1 && P.waves_per_block != 2 && P.waves_per_block != 4) return "a workgroup is 1, 2 or 4 waves";
    if (trace != (v.trace != 0)) return "trace build / trace launch mismatch";
    return nullptr;
}

}  // namespace madsim_k

extern "C" {
int  madsim_k_launch_sim(const madsim_k::KParams* P, uint32_t grid, uint32_t lds_bytes, void* stream, int trace);
void madsim_k_launch_summary(const madsim_result_t* out, uint64_t count, uint64_t seed0, unsigned long long* acc, void* stream);
void madsim_k_launch_summary6(const madsim_result_t* out, uint64_t count, uint64_t seed0, unsigned long long* acc6, void* stream);
int  madsim_k_set_max_lds(uint32_t lds_bytes);
int  madsim_k_variant_vgprs(const madsim_k::VariantSel* v);
void madsim_k_launch_keyflip(unsigned long long* acc, void* stream);
}

#endif
