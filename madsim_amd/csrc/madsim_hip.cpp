// madsim_hip.cpp — host side of libmadsim_hip.so: the C-ABI entry points of include/madsim_hip.h.
//
// Replaces the seed fan-out of madsim::runtime::Builder::run (madsim/src/sim/runtime/builder.rs:121-162):
// where the reference spawns one OS thread per seed, this picks an LDS geometry for the workload,
// launches the gfx950 executor kernel (sim_kernel.hip) with one lane per seed, and reduces the
// per-seed verdicts to "first failing seed" on the device.  No CPU execution path exists here: when
// HIP is unavailable every entry point returns MADSIM_E_HIP / MADSIM_E_NOINIT.
//
// State lives in per-device contexts (madsim_hip_ctx_t): a process may hold one context per GPU and drive them all from
// one host thread (madsim_hip_run_batch_multi), which is what a `cargo test` process on an 8-GPU node needs.  The v1
// entry points (madsim_hip_init / madsim_hip_run_batch ...) are wrappers on a process-default context.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

#include "sim_kernel.h"
#include "geometry.h"

using madsim_k::KParams;
using madsim_geo::Geo;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(MADSIM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

uint64_t fnv(const void* p, size_t n, uint64_t h) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

// VGPRs of a kernel build, asked of the loaded code object once per build and process (same on every device of a node).
int variant_vgprs_cached(const madsim_k::VariantSel* v) {
    static std::mutex mu;
    static std::unordered_map<uint32_t, int> cache;
    const uint32_t key = (uint32_t)v->trace | (uint32_t)v->spill << 1 | (uint32_t)v->rq << 2 | (uint32_t)v->g << 3 | (uint32_t)(v->lws & 0xf) << 4 | (uint32_t)v->feat << 8;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int r = madsim_k_variant_vgprs(v);
    cache[key] = r;
    return r;
}

double since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

// One GPU's worth of runner state.  Re-entrant per handle: distinct contexts never share mutable state; one context
// serialises its callers on `mu`.
struct madsim_hip_ctx {
    std::mutex mu;
    int device = -1;
    int num_cus = 0;
    size_t lds_per_cu = 160 * 1024;
    // Device copies of workload tables.  Looked up by content hash, confirmed by comparing the host bytes kept beside
    // them; entries are immutable once uploaded, so launches still in flight on other streams keep valid pointers when a
    // different workload comes along.
    struct Tables {
        uint64_t hash = 0;
        std::vector<uint32_t> host;      // insns | progs | socks | durs (as 32-bit words) | n_insns
        uint4* insns = nullptr; uint32_t* progs = nullptr; uint32_t* socks = nullptr; uint32_t* nodes = nullptr; uint64_t* durs = nullptr;
        uint32_t* iter_est = nullptr;    // ITER_KEYS words: how many passes a wave of this workload runs, one per launch shape (limits, seeds per lane:
                                         // upload_workload picks the word; written by finishing waves, k_main.h)
    };
    static constexpr uint32_t ITER_KEYS = 16;
    std::vector<Tables> tables;
    // per-stream scratch: launches on one stream run in order, launches on different streams may overlap and must
    // not share the timer-heap spill region or the work-queue counter
    struct Scratch { uint4* spill = nullptr; size_t spill_bytes = 0; unsigned long long* work_ctr = nullptr; uint8_t* gstate = nullptr; size_t gstate_bytes = 0; };
    std::unordered_map<hipStream_t, Scratch> scratch;
    // madsim_hip_ctx_run_campaign: batches in flight on the context's own streams
    static constexpr int CAMPAIGN_MAX = 8;
    struct Flight { hipStream_t stream = nullptr; madsim_result_t* d_out = nullptr; size_t cap = 0; unsigned long long* d_acc6 = nullptr;
                    unsigned long long* h_acc6 = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr, done = nullptr;
                    madsim_result_t* h_out = nullptr; size_t h_cap = 0; };      // h_out: page-locked staging of run_pipelined
    Flight flights[CAMPAIGN_MAX];
    unsigned long long* d_acc = nullptr;      // 4 x u64 summary accumulators
    madsim_result_t* d_out = nullptr; size_t out_cap = 0;
    uint64_t* d_seeds = nullptr; size_t seeds_cap = 0;        // seed list of a compacted re-run
    uint8_t* d_tlog = nullptr; size_t tlog_cap = 0; uint64_t* d_tlen = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t pipe_begin = nullptr;          // run_pipelined: before the first sub-batch of a call
    hipEvent_t tev[2 * 64] = {};              // timing slots of madsim_hip_run_batch_async
    uint32_t lds_attr = 0;
    uint64_t* d_prof = nullptr;               // debug counters (profiling kernel builds)

    madsim_geo::Device dev() const {
        madsim_geo::Device d; d.num_cus = num_cus > 0 ? num_cus : 256; d.lds_per_cu = lds_per_cu; d.vgprs = variant_vgprs_cached;
        static const int cap = [] { const char* e = getenv("MADSIM_HIP_WAVES_PER_SIMD"); return e ? atoi(e) : 0; }();    // (experiments: geometry.h Device)
        d.max_waves_per_simd = cap;
        return d;
    }
    int bind() { HIP_TRY(hipSetDevice(device)); return 0; }
    static void free_tables(Tables& t) {
        if (t.insns) (void)hipFree(t.insns);
        if (t.progs) (void)hipFree(t.progs);
        if (t.socks) (void)hipFree(t.socks);
        if (t.nodes) (void)hipFree(t.nodes);
        if (t.durs) (void)hipFree(t.durs);
        if (t.iter_est) (void)hipFree(t.iter_est);
        t = Tables();
    }
    int open(int dev_index);
    void close();
    int upload_workload(const madsim_workload_t* w, KParams& P);
    int ensure_scratch(KParams& P, hipStream_t stream, bool work_queue);
    int ensure_out(size_t count);
    int launch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count, const uint64_t* d_seed_list,
               const madsim_limits_t* lim, madsim_result_t* d_out, hipStream_t stream);
    int reduce(const madsim_result_t* d_out, uint64_t count, uint64_t seed0, unsigned long long* d_acc4, hipStream_t stream);
    int run_device(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                   const madsim_limits_t* lim, madsim_result_t* d_out, hipStream_t stream, madsim_summary_t* summary);
    int run_host(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                 const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary);
    int run_list(const madsim_workload_t* w, const madsim_config_t* cfg, const std::vector<uint64_t>& seeds,
                 const madsim_limits_t* lim, std::vector<madsim_result_t>& res, double* kernel_ms);
    int ensure_flights(uint32_t n, uint64_t batch, bool staging);
    uint32_t flights_for(const madsim_workload_t* w, const madsim_config_t* cfg, const madsim_limits_t* lim, uint64_t batch);
};

// The library keeps up to five sub-batches of a call in flight, each on its own HIP stream (run_pipelined, campaigns).  ROCclr maps
// the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4): with more streams than queues two sub-batches share a
// queue and run one after the other — measured on an MI355X (round 5, tools/experiment/exp_r5_runbatch.py): madsim_hip_run_batch(262 144)
// 8.3 ms with the default, ~5 ms with 16 queues.  The variable is read when the HIP runtime initialises, and a library has no business
// changing its host's environment behind its back (setenv races with getenv on other threads, and the value leaks into child
// processes: round 5 did exactly that).  So the HOST sets it — `GPU_MAX_HW_QUEUES=16` in the environment, as madsim_amd/runtime.py and
// bench.py do before they load HIP, or madsim_hip_prefer_hw_queues() as the first thing in main() — and the library only says so,
// once, when it is about to use more streams than the process has queues.
static void hint_hw_queues(uint32_t streams) {
    static bool said = false;
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    const long q = e ? atol(e) : 4;
    if (said || (long)streams <= q || getenv("MADSIM_HIP_QUIET")) return;
    said = true;
    fprintf(stderr, "madsim_hip: note: %u batches in flight but GPU_MAX_HW_QUEUES=%ld hardware queues: streams will share queues (about 1.6x slower "
                    "madsim_hip_run_batch).  Set GPU_MAX_HW_QUEUES=16 before the process first touches HIP, or call madsim_hip_prefer_hw_queues(16) "
                    "at start-up.  (MADSIM_HIP_QUIET=1 silences this.)\n", streams, q);
}

int madsim_hip_ctx::open(int dev_index) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(MADSIM_E_HIP, "no HIP device visible (this library has no CPU fallback)");
    if (dev_index < 0 || dev_index >= n) return fail(MADSIM_E_ARG, "device index out of range");
    device = dev_index;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    num_cus = prop.multiProcessorCount;
    lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 160 * 1024;
    if (lds_per_cu > 160 * 1024) lds_per_cu = 160 * 1024;
    HIP_TRY(hipMalloc(&d_acc, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&d_tlen, sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&d_prof, 16 * sizeof(uint64_t)));
    HIP_TRY(hipMemset(d_prof, 0, 16 * sizeof(uint64_t)));
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventCreate(&pipe_begin));
    return 0;
}

void madsim_hip_ctx::close() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    for (auto& t : tables) free_tables(t);
    tables.clear();
    for (auto& kv : scratch) { if (kv.second.spill) (void)hipFree(kv.second.spill); if (kv.second.work_ctr) (void)hipFree(kv.second.work_ctr); if (kv.second.gstate) (void)hipFree(kv.second.gstate); }
    scratch.clear();
    if (d_acc) (void)hipFree(d_acc);
    if (d_out) (void)hipFree(d_out);
    if (d_seeds) (void)hipFree(d_seeds);
    if (d_tlog) (void)hipFree(d_tlog);
    if (d_tlen) (void)hipFree(d_tlen);
    if (d_prof) (void)hipFree(d_prof);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (pipe_begin) (void)hipEventDestroy(pipe_begin);
    for (auto& f : flights) {
        if (f.stream) (void)hipStreamDestroy(f.stream);
        if (f.d_out) (void)hipFree(f.d_out);
        if (f.d_acc6) (void)hipFree(f.d_acc6);
        if (f.h_acc6) (void)hipHostFree(f.h_acc6);
        if (f.h_out) (void)hipHostFree(f.h_out);
        if (f.e0) (void)hipEventDestroy(f.e0);
        if (f.e1) (void)hipEventDestroy(f.e1);
        if (f.done) (void)hipEventDestroy(f.done);
        f = Flight();
    }
    for (auto& e : tev) if (e) (void)hipEventDestroy(e);
    d_acc = nullptr; d_out = nullptr; d_seeds = nullptr; d_tlog = nullptr; d_tlen = nullptr; d_prof = nullptr;
    ev0 = ev1 = nullptr; pipe_begin = nullptr; out_cap = seeds_cap = tlog_cap = 0;
    for (auto& e : tev) e = nullptr;
    device = -1;
}

int madsim_hip_ctx::upload_workload(const madsim_workload_t* w, KParams& P) {
    madsim_geo::DeviceTables T;
    int rc = madsim_geo::build_tables(w, &T, &g_err);
    if (rc) return rc;
    std::vector<uint32_t> host;
    host.reserve(T.insns.size() + w->n_progs + T.socks.size() + T.nodes.size() + 2 * T.durs.size() + 1);
    host.insert(host.end(), T.insns.begin(), T.insns.end());
    host.insert(host.end(), T.progs.begin(), T.progs.begin() + w->n_progs);
    host.insert(host.end(), T.socks.begin(), T.socks.end());
    host.insert(host.end(), T.nodes.begin(), T.nodes.end());
    for (uint64_t d : T.durs) { host.push_back((uint32_t)d); host.push_back((uint32_t)(d >> 32)); }
    host.push_back(w->n_insns);
    const uint64_t h = fnv(host.data(), host.size() * 4, 14695981039346656037ull);
    const Tables* hit = nullptr;
    for (auto& t : tables) if (t.hash == h && t.host == host) hit = &t;     // hash first, then the bytes themselves
    if (!hit) {
        if (tables.size() >= 16) {                       // bounded cache: drop everything once nothing is in flight
            HIP_TRY(hipDeviceSynchronize());
            for (auto& t : tables) free_tables(t);
            tables.clear();
        }
        Tables t;
        t.hash = h;
        auto up = [&]() -> int {
            HIP_TRY(hipMalloc(&t.insns, T.insns.size() * 4 + 16));
            HIP_TRY(hipMalloc(&t.progs, T.progs.size() * 4 + 16));
            HIP_TRY(hipMalloc(&t.socks, T.socks.size() * 4 + 16));
            HIP_TRY(hipMalloc(&t.durs, T.durs.size() * 8 + 16));
            HIP_TRY(hipMalloc(&t.nodes, T.nodes.size() * 4 + 16));
            HIP_TRY(hipMalloc(&t.iter_est, ITER_KEYS * sizeof(uint32_t)));
            HIP_TRY(hipMemset(t.iter_est, 0, ITER_KEYS * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(t.nodes, T.nodes.data(), T.nodes.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t.insns, T.insns.data(), T.insns.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t.progs, T.progs.data(), T.progs.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t.socks, T.socks.data(), T.socks.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t.durs, T.durs.data(), T.durs.size() * 8, hipMemcpyHostToDevice));
            return 0;
        };
        if ((rc = up())) { free_tables(t); return rc; }  // no partial allocation survives an error
        t.host = std::move(host);
        tables.push_back(std::move(t));
        hit = &tables.back();
    }
    P.insns = hit->insns; P.progs = hit->progs; P.socks = hit->socks; P.nodes = hit->nodes; P.dur_table = hit->durs;
    // (MADSIM_HIP_NO_PRIO=1 in the environment: launches without the progress-based wave priorities of k_main.h — a diagnostic / A-B switch,
    //  results are the same either way)
    // The pass estimate is a property of the launch SHAPE, not of the workload alone: the capacities change the kernel build and the spilled
    // levels, the seeds per lane (a 65 536-seed campaign batch against a compacted re-run list) the passes a wave runs.  One word per
    // shape, picked by a hash of both (a collision costs fairness between co-resident launches, never a result).
    {
        static const bool off = getenv("MADSIM_HIP_NO_PRIO") != nullptr;
        const uint64_t per_lane = P.total_lanes ? (P.count + P.total_lanes - 1) / P.total_lanes : 1;
        const uint32_t shape[8] = {P.heap_lds, P.heap_spill, P.max_tasks, P.lw_shift, P.gstate_mode | P.narrow << 1 | P.compact << 2 | (P.dedup_n ? 8u : 0u),
                                   P.mbox_regs, P.max_steps, (uint32_t)std::min<uint64_t>(per_lane, 0xffffffffu)};
        P.iter_est = off ? nullptr : hit->iter_est + (fnv(shape, sizeof shape, 14695981039346656037ull) % ITER_KEYS);
    }
    return 0;
}

int madsim_hip_ctx::ensure_scratch(KParams& P, hipStream_t stream, bool work_queue) {
    P.spill = nullptr; P.work_ctr = nullptr; P.gstate = nullptr;
    if (!P.heap_spill && !work_queue && !P.gstate_mode && !P.compact) return 0;
    Scratch& sc = scratch[stream];
    if (P.gstate_mode || P.compact) {    // per-lane state blocks of the global-state builds / main-task records of the compact ones (seed_init writes what it reads)
        size_t need = (size_t)P.gs_stride * P.total_lanes;
        if (need >= (1ull << 32)) return fail(MADSIM_E_LIMITS, "global state region exceeds 4 GiB: lower the capacities");
        if (need > sc.gstate_bytes) {
            if (sc.gstate) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(sc.gstate); }
            sc.gstate = nullptr; sc.gstate_bytes = 0;
            HIP_TRY(hipMalloc(&sc.gstate, need));
            sc.gstate_bytes = need;
        }
        P.gstate = sc.gstate;
    }
    if (P.heap_spill) {
        size_t need = (size_t)P.heap_spill * P.total_lanes * (P.narrow ? 8 : sizeof(uint4));
        if (need >= (1ull << 32)) return fail(MADSIM_E_LIMITS, "heap spill region exceeds 4 GiB: lower heap_spill_slots");
        if (need > sc.spill_bytes) {
            if (sc.spill) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(sc.spill); }
            sc.spill = nullptr; sc.spill_bytes = 0;
            HIP_TRY(hipMalloc(&sc.spill, need));
            sc.spill_bytes = need;
        }
        P.spill = sc.spill;
    }
    if (work_queue) {
        if (!sc.work_ctr) HIP_TRY(hipMalloc(&sc.work_ctr, sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(sc.work_ctr, 0, sizeof(unsigned long long), stream));
        P.work_ctr = sc.work_ctr;
    }
    return 0;
}

int madsim_hip_ctx::ensure_out(size_t count) {
    if (count > out_cap) {
        if (d_out) (void)hipFree(d_out);
        d_out = nullptr; out_cap = 0;
        HIP_TRY(hipMalloc(&d_out, count * sizeof(madsim_result_t)));
        out_cap = count;
    }
    return 0;
}

// Queue the simulation kernel for `count` units on `stream`: unit i runs seed d_seed_list[i] (device memory) when a list
// is given, else seed0 + i.  Nothing is synchronised.
int madsim_hip_ctx::launch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count, const uint64_t* d_seed_list,
                           const madsim_limits_t* lim, madsim_result_t* d_res, hipStream_t stream) {
    Geo G;
    int rc;
    if ((rc = madsim_geo::make_geometry(dev(), w, cfg, lim, count, &G, &g_err))) return rc;
    G.P.count = count;                              // (upload_workload keys the pass estimate by the launch shape: seeds per lane)
    if ((rc = upload_workload(w, G.P))) return rc;
    // Work distribution: static striding (lane g runs units g, g+G, ...) or a per-launch atomic counter from which a
    // finished lane pulls its next unit.  They only differ when a launch holds more units than resident lanes.
    const bool work_queue = lim && lim->sched == MADSIM_SCHED_QUEUE && count > G.P.total_lanes;
    if ((rc = ensure_scratch(G.P, stream, work_queue))) return rc;
    G.P.seed0 = seed0; G.P.count = count; G.P.seed_list = d_seed_list; G.P.out = d_res; G.P.prof = d_prof;
    if (G.lds_bytes > lds_attr) {
        int e = madsim_k_set_max_lds((uint32_t)lds_per_cu);
        if (e) return fail(MADSIM_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        lds_attr = (uint32_t)lds_per_cu;
    }
    if (madsim_k_launch_sim(&G.P, G.grid, G.lds_bytes, stream, 0)) return fail(MADSIM_E_LIMITS, "no kernel build for this geometry (select_variant)");
    HIP_TRY(hipGetLastError());
    return 0;
}

// Queue the report reduction of d_out[0..count) into d_acc4 = {first failing seed, n_failed, steps, clock} on `stream`.
int madsim_hip_ctx::reduce(const madsim_result_t* d_res, uint64_t count, uint64_t seed0, unsigned long long* d_acc4, hipStream_t stream) {
    HIP_TRY(hipMemsetAsync(d_acc4, 0xff, 8, stream));
    HIP_TRY(hipMemsetAsync((char*)d_acc4 + 8, 0, 24, stream));
    madsim_k_launch_summary(d_res, count, seed0, d_acc4, stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int madsim_hip_ctx::run_device(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                               const madsim_limits_t* lim, madsim_result_t* d_res, hipStream_t stream, madsim_summary_t* summary) {
    auto t0 = std::chrono::steady_clock::now();
    int rc = madsim_geo::validate(w, cfg, &g_err);
    if (rc) return rc;
    if (count == 0) {
        if (summary) { memset(summary, 0, sizeof *summary); summary->first_failing_seed = UINT64_MAX; }
        return 0;
    }
    if (!d_res) return fail(MADSIM_E_ARG, "null result buffer");
    if (summary) HIP_TRY(hipEventRecord(ev0, stream));
    if ((rc = launch(w, cfg, seed0, count, nullptr, lim, d_res, stream))) return rc;
    if (summary) {
        HIP_TRY(hipEventRecord(ev1, stream));
        if ((rc = reduce(d_res, count, seed0, d_acc, stream))) return rc;
        unsigned long long acc[4];
        HIP_TRY(hipMemcpyAsync(acc, d_acc, sizeof acc, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        summary->first_failing_seed = acc[0]; summary->n_failed = acc[1];
        summary->total_steps = acc[2]; summary->total_clock_ns = acc[3];
        summary->kernel_ms = ms;
        summary->wall_s = since(t0);
    }
    return 0;
}

constexpr uint64_t PIPE_BATCH_HOST = 65536 + 32768;     // up to one and a half batches stay one launch
int run_pipelined_fwd(madsim_hip_ctx* const* ctxs, int n_ctx, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                      const madsim_limits_t* lim, madsim_result_t* out, unsigned long long* acc6, double* kernel_ms);
int madsim_hip_ctx::run_host(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                             const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary) {
    auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (count > PIPE_BATCH_HOST) {             // more than one batch: sub-launches kept in flight, copies overlapped (run_pipelined)
        if ((rc = madsim_geo::validate(w, cfg, &g_err))) return rc;
        unsigned long long a[6];                     // (`out` may be NULL: summary only — no staging buffers, no copies)
        double ms = 0.0;
        madsim_hip_ctx* one[1] = {this};
        if ((rc = run_pipelined_fwd(one, 1, w, cfg, seed0, count, lim, out, a, &ms))) return rc;
        if (summary) {
            summary->first_failing_seed = a[0]; summary->n_failed = a[1]; summary->total_steps = a[2]; summary->total_clock_ns = a[3];
            summary->kernel_ms = ms; summary->wall_s = since(t0);
        }
        return 0;
    }
    if ((rc = ensure_out(count))) return rc;
    madsim_summary_t tmp;
    if ((rc = run_device(w, cfg, seed0, count, lim, d_out, nullptr, &tmp))) return rc;
    if (out && count) HIP_TRY(hipMemcpy(out, d_out, count * sizeof(madsim_result_t), hipMemcpyDeviceToHost));
    if (summary) { *summary = tmp; summary->wall_s = since(t0); }
    return 0;
}

// One compacted launch over an arbitrary list of seeds (the re-run of seeds that outgrew a capacity).
int madsim_hip_ctx::run_list(const madsim_workload_t* w, const madsim_config_t* cfg, const std::vector<uint64_t>& seeds,
                             const madsim_limits_t* lim, std::vector<madsim_result_t>& res, double* kernel_ms) {
    const size_t n = seeds.size();
    res.resize(n);
    if (!n) return 0;
    int rc;
    if ((rc = ensure_out(n))) return rc;
    if (n > seeds_cap) {
        if (d_seeds) (void)hipFree(d_seeds);
        d_seeds = nullptr; seeds_cap = 0;
        HIP_TRY(hipMalloc(&d_seeds, n * sizeof(uint64_t)));
        seeds_cap = n;
    }
    HIP_TRY(hipMemcpy(d_seeds, seeds.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipEventRecord(ev0, nullptr));
    if ((rc = launch(w, cfg, 0, n, d_seeds, lim, d_out, nullptr))) return rc;
    HIP_TRY(hipEventRecord(ev1, nullptr));
    HIP_TRY(hipMemcpy(res.data(), d_out, n * sizeof(madsim_result_t), hipMemcpyDeviceToHost));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    if (kernel_ms) *kernel_ms += ms;
    return 0;
}

// Streams, report words, events and result buffers of the first n flights (campaigns and run_pipelined share them; the
// context's mutex serialises the two).  `staging`: also a page-locked host buffer of `batch` results per flight.
int madsim_hip_ctx::ensure_flights(uint32_t n, uint64_t batch, bool staging) {
    if (n > (uint32_t)CAMPAIGN_MAX) return fail(MADSIM_E_ARG, "at most 8 batches in flight");
    hint_hw_queues(n);
    for (uint32_t i = 0; i < n; i++) {
        Flight& f = flights[i];
        if (!f.stream) {
            HIP_TRY(hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking));
            HIP_TRY(hipMalloc(&f.d_acc6, 6 * sizeof(unsigned long long)));
            HIP_TRY(hipHostMalloc((void**)&f.h_acc6, 6 * sizeof(unsigned long long), hipHostMallocDefault));
            HIP_TRY(hipEventCreate(&f.e0)); HIP_TRY(hipEventCreate(&f.e1)); HIP_TRY(hipEventCreate(&f.done));
        }
        if (batch > f.cap) {
            if (f.d_out) { HIP_TRY(hipStreamSynchronize(f.stream)); (void)hipFree(f.d_out); }
            f.d_out = nullptr; f.cap = 0;
            HIP_TRY(hipMalloc(&f.d_out, batch * sizeof(madsim_result_t)));
            f.cap = batch;
        }
        if (staging && batch > f.h_cap) {
            if (f.h_out) { HIP_TRY(hipStreamSynchronize(f.stream)); (void)hipHostFree(f.h_out); }
            f.h_out = nullptr; f.h_cap = 0;
            HIP_TRY(hipHostMalloc((void**)&f.h_out, batch * sizeof(madsim_result_t), hipHostMallocDefault));
            f.h_cap = batch;
        }
    }
    return 0;
}

namespace { uint32_t flights_of(madsim_hip_ctx* c, const madsim_workload_t* w, const madsim_config_t* cfg, const madsim_limits_t* lim, uint64_t batch); }
uint32_t madsim_hip_ctx::flights_for(const madsim_workload_t* w, const madsim_config_t* cfg, const madsim_limits_t* lim, uint64_t batch) {
    return flights_of(this, w, cfg, lim, batch);
}

namespace {

madsim_hip_ctx* g_default = nullptr;       // the context behind the v1 entry points
std::mutex g_default_mu;

// Seeds that came back with a RUNNER verdict (a device capacity or the step cap, neither a reference concept) are run
// again, all of them in ONE compacted launch per round, with doubled capacities / a 16x step cap.
// The step cap never grows beyond madsim_limits_t.max_steps_ceiling (default 1 << 28 = the first pass's 1 << 24 and one 16x
// round): a livelocked workload — a yield loop, a 1 ms timer loop without a time limit — must come back as MADSIM_STEP_LIMIT
// after seconds, not hold the GPU (and the context mutex) in one non-preemptible kernel for an hour.
uint64_t step_ceiling(const madsim_limits_t& L) {
    uint64_t c = L.max_steps_ceiling ? L.max_steps_ceiling : (1u << 28);
    const uint64_t first = L.max_steps ? L.max_steps : (1u << 24);
    return c < first ? first : c;
}
void grow(madsim_limits_t& L, const madsim_workload_t* w, bool any_ovf, bool any_steps) {
    auto dbl = [](uint32_t v, uint32_t dflt, uint32_t cap) { uint32_t x = (v == 0 || v == MADSIM_LIMIT_NONE) ? dflt : v; x *= 2; return x > cap ? cap : x; };
    L.lanes_per_wave = 0;
    if (any_ovf) {
        // the compact base-op layout admits no heap spill and at most eight task slots: a grown re-run must be free to leave
        // it (an explicit MADSIM_STATE_COMPACT would fail make_geometry with MADSIM_E_LIMITS and take the whole call with it)
        if ((L.state_mem & 0xffu) == MADSIM_STATE_COMPACT) L.state_mem = (L.state_mem & ~0xffu) | MADSIM_STATE_AUTO;
        // ... and the 8-byte heap entries: a deadline beyond their 2^31 ns horizon is reported as a capacity verdict (k_timer.h timer_add)
        L.state_mem &= ~MADSIM_STATE_NARROW_HEAP;
        L.heap_lds_slots = L.heap_lds_slots ? L.heap_lds_slots : 8;
        L.heap_spill_slots = dbl(L.heap_spill_slots, 32, 1u << 20);
        L.max_tasks = dbl(L.max_tasks, w->n_progs + 8, 254);
        L.mbox_regs = dbl(L.mbox_regs, 2, 255);
        L.mbox_msgs = dbl(L.mbox_msgs, 2, 255);
        L.max_conns = dbl(L.max_conns, 4, 127);
        L.chan_queue = dbl(L.chan_queue, 2, 15);
    }
    if (any_steps) {
        uint64_t s = L.max_steps ? L.max_steps : (1u << 24);
        s *= 16;
        const uint64_t ceil = step_ceiling(L);
        L.max_steps = s > ceil ? (uint32_t)ceil : (uint32_t)s;
    }
}

// `ctxs` = the contexts whose locks the caller holds: round r runs on ctxs[r % n] (the shards' overflow is spread, not piled
// on the first device).
int rerun_runner_verdicts(madsim_hip_ctx* const* ctxs, int n_ctx, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                          const madsim_limits_t* lim, madsim_result_t* out, int max_rounds, double* kernel_ms) {
    madsim_limits_t L{};
    if (lim) L = *lim;
    for (int round = 0; round < max_rounds; round++) {
        std::vector<uint64_t> idx;
        bool any_ovf = false, any_steps = false;
        const bool steps_maxed = (L.max_steps ? L.max_steps : (1u << 24)) >= step_ceiling(L);
        for (uint64_t i = 0; i < count; i++) {
            if (out[i].verdict == MADSIM_OVERFLOW) { idx.push_back(i); any_ovf = true; }
            else if (out[i].verdict == MADSIM_STEP_LIMIT && !steps_maxed) { idx.push_back(i); any_steps = true; }
        }
        if (idx.empty()) break;
        grow(L, w, any_ovf, any_steps);
        std::vector<uint64_t> seeds(idx.size());
        for (size_t k = 0; k < idx.size(); k++) seeds[k] = seed0 + idx[k];
        std::vector<madsim_result_t> res;
        madsim_hip_ctx* c = ctxs[round % n_ctx];
        int rc = c->bind();
        if (rc) return rc;
        if ((rc = c->run_list(w, cfg, seeds, &L, res, kernel_ms))) return rc;
        for (size_t k = 0; k < idx.size(); k++) out[idx[k]] = res[k];
    }
    return 0;
}

void host_summary(const madsim_result_t* out, uint64_t seed0, uint64_t count, madsim_summary_t* f) {
    memset(f, 0, sizeof *f);
    f->first_failing_seed = UINT64_MAX;
    for (uint64_t i = 0; i < count; i++) {
        if (out[i].verdict != MADSIM_PASS) { if (!f->n_failed) f->first_failing_seed = seed0 + i; f->n_failed++; }
        f->total_steps += out[i].steps; f->total_clock_ns += out[i].clock_ns;
    }
}

// Batches in flight by the workload's occupancy: a batch of 65 536 seeds is one wave per SIMD, so as many batches as the
// workload's LDS admits waves per SIMD, and — when that is four (the compact base-op layout) — a fifth, whose launch queues behind
// them and fills the gaps their tails leave.
uint32_t flights_of(madsim_hip_ctx* c, const madsim_workload_t* w, const madsim_config_t* cfg, const madsim_limits_t* lim, uint64_t batch) {
    madsim_geo::Geo G;
    uint32_t n = 3;
    if (!madsim_geo::make_geometry(c->dev(), w, cfg, lim, batch, &G, &g_err)) {
        if (G.blocks_per_cu * G.waves_per_block >= 16) n = 5;
        else if (G.P.gstate_mode && G.P.heap_spill) n = 4;      // long launches that end with their slowest wave: one more behind them
    }
    return n;
}

// Per-seed results at the overlapped rate (one launch of 65 536 seeds is one wave per SIMD and leaves two
// thirds of the issue slots idle).  Builder::run hands over ALL its seeds in one call (runtime/builder.rs:121-162), so the call
// itself cuts [seed0, seed0 + count) into sub-batches: every context runs its contiguous share (context g: the block
// g * ceil(count / n) — the rule of madsim_amd/dist.py shard_range), keeping `flights` sub-batches in flight on its own streams;
// each sub-batch is kernel -> report reduction (accumulating in the flight's six words) -> device-to-host copy into the flight's
// page-locked staging buffer, and the host moves a finished sub-batch into the caller's array while the next ones run.  Every
// launch is queued from the calling thread; once the first kernel is in flight no error returns before every launched stream
// has been drained.  `acc6` (optional): the folded device reports {first failing seed, failed, steps, clock, first genuine, runner}.
constexpr uint64_t PIPE_BATCH = 65536;
int run_pipelined(madsim_hip_ctx* const* ctxs, int n_ctx, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                  const madsim_limits_t* lim, madsim_result_t* out, unsigned long long* acc6, double* kernel_ms) {
    struct Pipe { madsim_hip_ctx* c; uint64_t lo = 0, n = 0, nb = 0, launched = 0, harvested = 0; uint32_t F = 0; double ms = 0.0; };
    std::vector<Pipe> pp(n_ctx);
    const uint64_t chunk = (count + (uint64_t)n_ctx - 1) / (uint64_t)n_ctx;
    int first_err = 0;
    std::string first_msg;
    auto note = [&](int e) { if (e && !first_err) { first_err = e; first_msg = g_err; } return e; };
    for (int g = 0; g < n_ctx; g++) {
        Pipe& p = pp[g];
        p.c = ctxs[g];
        p.lo = std::min((uint64_t)g * chunk, count);
        p.n = std::min(p.lo + chunk, count) - p.lo;
        p.nb = (p.n + PIPE_BATCH - 1) / PIPE_BATCH;
        if (!p.nb) continue;
        int e;
        if (note(e = p.c->bind())) break;
        p.F = (uint32_t)std::min<uint64_t>(p.nb, p.c->flights_for(w, cfg, lim, std::min(p.n, PIPE_BATCH)));
        if (note(e = p.c->ensure_flights(p.F, std::min(p.n, PIPE_BATCH), out != nullptr))) break;
    }
    if (first_err) return fail(first_err, first_msg);
    auto queue = [&](Pipe& p) -> int {
        const uint64_t k = p.launched;
        madsim_hip_ctx::Flight& f = p.c->flights[k % p.F];
        const uint64_t lo = p.lo + k * PIPE_BATCH, n = std::min(PIPE_BATCH, p.lo + p.n - lo);
        int e;
        if ((e = p.c->bind())) return e;
        if (k < p.F) {                                           // the flight's first sub-batch of this call: a fresh report
            HIP_TRY(hipMemsetAsync(f.d_acc6, 0xff, 8, f.stream));
            HIP_TRY(hipMemsetAsync((char*)f.d_acc6 + 8, 0, 24, f.stream));
            HIP_TRY(hipMemsetAsync((char*)f.d_acc6 + 32, 0xff, 8, f.stream));
            HIP_TRY(hipMemsetAsync((char*)f.d_acc6 + 40, 0, 8, f.stream));
        }
        if (k == 0) HIP_TRY(hipEventRecord(p.c->pipe_begin, f.stream));
        p.launched++;                                            // from here on the flight must be drained before returning
        if ((e = p.c->launch(w, cfg, seed0 + lo, n, nullptr, lim, f.d_out, f.stream))) return e;
        HIP_TRY(hipEventRecord(f.e1, f.stream));
        madsim_k_launch_summary6(f.d_out, n, seed0 + lo, f.d_acc6, f.stream);
        HIP_TRY(hipGetLastError());
        if (out) HIP_TRY(hipMemcpyAsync(f.h_out, f.d_out, n * sizeof(madsim_result_t), hipMemcpyDeviceToHost, f.stream));
        HIP_TRY(hipEventRecord(f.done, f.stream));
        return 0;
    };
    auto harvest = [&](Pipe& p) -> int {
        const uint64_t k = p.harvested++;
        madsim_hip_ctx::Flight& f = p.c->flights[k % p.F];
        int e;
        if ((e = p.c->bind())) return e;
        HIP_TRY(hipEventSynchronize(f.done));
        if (first_err) return 0;                                 // (draining after an error: nothing is moved any more)
        const uint64_t lo = p.lo + k * PIPE_BATCH, n = std::min(PIPE_BATCH, p.lo + p.n - lo);
        if (out) memcpy(out + lo, f.h_out, n * sizeof(madsim_result_t));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.c->pipe_begin, f.e1));
        p.ms = std::max(p.ms, (double)ms);
        return 0;
    };
    for (;;) {
        bool any = false;
        for (Pipe& p : pp)                                       // 1. fill every device's flights before waiting for any of them
            while (!first_err && p.launched < p.nb && p.launched - p.harvested < p.F) { note(queue(p)); any = true; }
        for (Pipe& p : pp)                                       // 2. then the oldest sub-batch of every device
            if (p.harvested < p.launched) { note(harvest(p)); any = true; }
        if (!any) break;
    }
    if (first_err) {
        // A queue() that failed between its kernel launch and the record of `done` leaves that event stale (or never recorded): the
        // harvest above returned at once.  No kernel may stay in flight behind an error return — it would still be writing the
        // flight's buffers — so every stream this call used is drained for real.
        for (Pipe& p : pp) {
            if (!p.F || p.c->bind()) continue;
            for (uint32_t i = 0; i < p.F; i++) (void)hipStreamSynchronize(p.c->flights[i].stream);
        }
        return fail(first_err, first_msg);
    }
    unsigned long long a[6] = {~0ull, 0, 0, 0, ~0ull, 0};
    double ms = 0.0;
    for (Pipe& p : pp) {
        if (!p.nb) continue;
        int e;
        if ((e = p.c->bind())) return e;
        for (uint32_t i = 0; i < p.F; i++) {
            madsim_hip_ctx::Flight& f = p.c->flights[i];
            HIP_TRY(hipMemcpyAsync(f.h_acc6, f.d_acc6, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, f.stream));
            HIP_TRY(hipStreamSynchronize(f.stream));
            a[0] = std::min(a[0], f.h_acc6[0]); a[4] = std::min(a[4], f.h_acc6[4]);
            a[1] += f.h_acc6[1]; a[2] += f.h_acc6[2]; a[3] += f.h_acc6[3]; a[5] += f.h_acc6[5];
        }
        ms = std::max(ms, p.ms);                                 // devices ran concurrently
    }
    if (acc6) memcpy(acc6, a, sizeof a);
    if (kernel_ms) *kernel_ms += ms;
    return 0;
}

}  // namespace
int run_pipelined_fwd(madsim_hip_ctx* const* ctxs, int n_ctx, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                      const madsim_limits_t* lim, madsim_result_t* out, unsigned long long* acc6, double* kernel_ms) {
    return run_pipelined(ctxs, n_ctx, w, cfg, seed0, count, lim, out, acc6, kernel_ms);
}
namespace {

#define CTX_ENTER(c)                                                                              \
    if (!(c) || (c)->device < 0) return fail(MADSIM_E_NOINIT, "no context (madsim_hip_init / madsim_hip_ctx_create has not been called)"); \
    std::lock_guard<std::mutex> lk((c)->mu);                                                      \
    { int brc_ = (c)->bind(); if (brc_) return brc_; }

}  // namespace

extern "C" {

uint32_t madsim_hip_version(void) { return MADSIM_HIP_ABI_VERSION; }

// Explicit opt-in for hosts that cannot set their own environment: asks ROCclr for `n` hardware queues (GPU_MAX_HW_QUEUES) unless the
// variable is already set.  Only effective before the process's first HIP call, and — like every setenv — only safe while no other
// thread reads the environment: call it first thing in main().  Returns 1 when it set the variable, 0 when the host's own value stands.
int madsim_hip_prefer_hw_queues(int n) {
    if (n < 1 || n > 64) return fail(MADSIM_E_ARG, "hardware queues: 1..64");
    if (getenv("GPU_MAX_HW_QUEUES")) return 0;
    char buf[16];
    snprintf(buf, sizeof buf, "%d", n);
    return setenv("GPU_MAX_HW_QUEUES", buf, 0) == 0 ? 1 : fail(MADSIM_E_ARG, "setenv failed");
}

#define MADSIM_STR2(x) #x
#define MADSIM_STR(x) MADSIM_STR2(x)
#define MADSIM_COUNT_VARIANT(...) +1
const char* madsim_hip_build_info(void) {
    static const std::string info = std::string("madsim_hip abi=") + MADSIM_STR(MADSIM_HIP_ABI_VERSION) + " arch=gfx950 kernels="
        + std::to_string(0 MADSIM_FOR_EACH_VARIANT(MADSIM_COUNT_VARIANT)) + " backend=hip-rocm"
#ifdef MADSIM_EXPERIMENT_BUILD
        + " experiment-build"
#endif
        ;
    return info.c_str();
}

const char* madsim_hip_strerror(int code) {
    switch (code) {
    case 0: return "ok";
    case MADSIM_E_ARG: return "invalid argument";
    case MADSIM_E_HIP: return "HIP runtime error";
    case MADSIM_E_NOINIT: return "library not initialised (no GPU bound)";
    case MADSIM_E_WORKLOAD: return "malformed workload";
    case MADSIM_E_LIMITS: return "limits do not fit the device";
    default: return "unknown error";
    }
}

const char* madsim_hip_last_error(void) { return g_err.c_str(); }

// ---- per-device contexts ---------------------------------------------------------------------------------------------

int madsim_hip_ctx_create(int device, madsim_hip_ctx_t** out) {
    if (!out) return fail(MADSIM_E_ARG, "null context pointer");
    *out = nullptr;
    madsim_hip_ctx* c = new madsim_hip_ctx();
    int rc = c->open(device);
    if (rc) { c->close(); delete c; return rc; }
    *out = c;
    return 0;
}

int madsim_hip_ctx_destroy(madsim_hip_ctx_t* c) {
    if (!c) return 0;
    { std::lock_guard<std::mutex> lk(c->mu); c->close(); }
    delete c;
    return 0;
}

int madsim_hip_ctx_device(const madsim_hip_ctx_t* c) { return c ? c->device : -1; }

int madsim_hip_ctx_run_batch(madsim_hip_ctx_t* c, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                             const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary) {
    CTX_ENTER(c);
    return c->run_host(w, cfg, seed0, count, lim, out, summary);
}

int madsim_hip_ctx_run_batch_device(madsim_hip_ctx_t* c, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                                    const madsim_limits_t* lim, void* d_out, void* stream, madsim_summary_t* summary) {
    CTX_ENTER(c);
    return c->run_device(w, cfg, seed0, count, lim, (madsim_result_t*)d_out, (hipStream_t)stream, summary);
}

int madsim_hip_ctx_run_batch_async(madsim_hip_ctx_t* c, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                                   const madsim_limits_t* lim, void* d_out, void* d_summary4, void* stream, int timing_slot) {
    CTX_ENTER(c);
    if (timing_slot >= 64) return fail(MADSIM_E_ARG, "timing_slot must be < 64");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t* ev = nullptr;
    if (timing_slot >= 0) {
        ev = &c->tev[2 * timing_slot];
        if (!ev[0]) { HIP_TRY(hipEventCreate(&ev[0])); HIP_TRY(hipEventCreate(&ev[1])); }
        HIP_TRY(hipEventRecord(ev[0], st));
    }
    int rc = c->run_device(w, cfg, seed0, count, lim, (madsim_result_t*)d_out, st, nullptr);
    if (rc) return rc;
    if (ev) HIP_TRY(hipEventRecord(ev[1], st));
#ifdef MADSIM_EXP_NO_SUMMARY      // (timing experiment, tools/build_variant.sh only: what the report reduction behind every launch costs the stream)
    d_summary4 = nullptr;
#endif
    if (d_summary4 && count) {
        // {UINT64_MAX, 0, 0, 0}, accumulated by the reduction kernel, then word 0 is flipped into its
        // order-preserving int64 form (seed ^ 1<<63) so a signed all-reduce(MIN) yields the unsigned minimum
        if ((rc = c->reduce((const madsim_result_t*)d_out, count, seed0, (unsigned long long*)d_summary4, st))) return rc;
        madsim_k_launch_keyflip((unsigned long long*)d_summary4, st);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int madsim_hip_ctx_timing_ms(madsim_hip_ctx_t* c, int timing_slot, double* ms) {
    CTX_ENTER(c);
    if (timing_slot < 0 || timing_slot >= 64 || !c->tev[2 * timing_slot] || !ms) return fail(MADSIM_E_ARG, "bad timing slot");
    float f = 0.f;
    HIP_TRY(hipEventSynchronize(c->tev[2 * timing_slot + 1]));
    HIP_TRY(hipEventElapsedTime(&f, c->tev[2 * timing_slot], c->tev[2 * timing_slot + 1]));
    *ms = f;
    return 0;
}

int madsim_hip_ctx_run_batch_auto(madsim_hip_ctx_t* c, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                                  const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary, int max_rounds) {
    if (!out && count) return fail(MADSIM_E_ARG, "run_batch_auto needs the result array");
    CTX_ENTER(c);
    auto t0 = std::chrono::steady_clock::now();
    madsim_summary_t s{};
    int rc = c->run_host(w, cfg, seed0, count, lim, out, &s);
    if (rc) return rc;
    double kernel_ms = s.kernel_ms;
    madsim_hip_ctx* one[1] = {c};
    if ((rc = rerun_runner_verdicts(one, 1, w, cfg, seed0, count, lim, out, max_rounds, &kernel_ms))) return rc;
    if (summary) { host_summary(out, seed0, count, summary); summary->kernel_ms = kernel_ms; summary->wall_s = since(t0); }
    return 0;
}

int64_t madsim_hip_ctx_trace_seed(madsim_hip_ctx_t* c, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed,
                                  const madsim_limits_t* lim, uint8_t* log, uint64_t cap, madsim_result_t* out) {
    CTX_ENTER(c);
    int rc = madsim_geo::validate(w, cfg, &g_err);
    if (rc) return rc;
    Geo G;
    if ((rc = madsim_geo::make_geometry(c->dev(), w, cfg, lim, 1, &G, &g_err, true))) return rc;
    if ((rc = c->upload_workload(w, G.P))) return rc;
    if ((rc = c->ensure_scratch(G.P, nullptr, false))) return rc;
    if (cap > c->tlog_cap) {
        if (c->d_tlog) (void)hipFree(c->d_tlog);
        c->d_tlog = nullptr; c->tlog_cap = 0;
        HIP_TRY(hipMalloc(&c->d_tlog, cap));
        c->tlog_cap = cap;
    }
    if ((rc = c->ensure_out(64))) return rc;
    G.P.seed0 = seed; G.P.count = 1; G.P.out = c->d_out;
    G.P.trace_log = cap ? c->d_tlog : nullptr; G.P.trace_cap = cap; G.P.trace_len = c->d_tlen;
    if (G.lds_bytes > c->lds_attr) {
        if (madsim_k_set_max_lds((uint32_t)c->lds_per_cu)) return fail(MADSIM_E_HIP, "hipFuncSetAttribute failed");
        c->lds_attr = (uint32_t)c->lds_per_cu;
    }
    if (madsim_k_launch_sim(&G.P, 1, G.lds_bytes, nullptr, 1)) return fail(MADSIM_E_LIMITS, "no trace kernel build");
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    uint64_t n = 0;
    HIP_TRY(hipMemcpy(&n, c->d_tlen, sizeof n, hipMemcpyDeviceToHost));
    if (log && cap) HIP_TRY(hipMemcpy(log, c->d_tlog, n < cap ? n : cap, hipMemcpyDeviceToHost));
    if (out) HIP_TRY(hipMemcpy(out, c->d_out, sizeof *out, hipMemcpyDeviceToHost));
    return (int64_t)n;
}

// ---- one process, several GPUs -----------------------------------------------------------------------------------------
// Builder::run drives every seed from one process (builder.rs:129-150): shard [seed0, seed0 + count) contiguously over
// the contexts (context g gets [g * ceil(count / n), ...), the rule of madsim_amd/dist.py::shard_range) and run every share
// as sub-batches kept in flight on that device's own streams (run_pipelined: kernels and device-to-host copies of all devices
// queued from this one host thread, page-locked staging so the copies overlap), then fold the reports on the host.
// Why a host fold and not the RCCL gather BASELINE.json's north_star names: the per-seed results travel to the caller's host
// array anyway (Builder::run needs the failing seed's result), so the only thing a collective could carry is n <= 8 reports
// of 32 bytes that the host already holds; a single-process communicator (ncclCommInitAll) plus one collective launch per
// batch would add a dependency and a synchronisation, not remove one.  The RCCL gather lives where ranks really are
// separate processes: bench.py / madsim_amd/dist.py (one process per GPU, one 64-byte all-gather per batch).
// Robustness: context locks are taken in address order (two threads passing the same contexts in different orders cannot
// deadlock); once the first kernel is in flight no error returns before EVERY launched stream has been synchronised; the
// compacted re-runs of runner verdicts go round-robin over the contexts.
int madsim_hip_run_batch_multi(madsim_hip_ctx_t* const* ctxs, int n_ctx, const madsim_workload_t* w, const madsim_config_t* cfg,
                               uint64_t seed0, uint64_t count, const madsim_limits_t* lim, madsim_result_t* out,
                               madsim_summary_t* summary, int max_rounds) {
    auto t0 = std::chrono::steady_clock::now();
    if (!ctxs || n_ctx < 1) return fail(MADSIM_E_ARG, "run_batch_multi needs at least one context");
    if (!out && count) return fail(MADSIM_E_ARG, "run_batch_multi needs the result array");
    for (int g = 0; g < n_ctx; g++) {
        if (!ctxs[g]) return fail(MADSIM_E_NOINIT, "null context");
        for (int h = 0; h < g; h++) if (ctxs[h] == ctxs[g]) return fail(MADSIM_E_ARG, "the same context appears twice");
    }
    int rc = madsim_geo::validate(w, cfg, &g_err);
    if (rc) return rc;
    std::vector<madsim_hip_ctx*> order(ctxs, ctxs + n_ctx);
    std::sort(order.begin(), order.end(), [](madsim_hip_ctx* a, madsim_hip_ctx* b) { return std::less<madsim_hip_ctx*>()(a, b); });
    std::vector<std::unique_lock<std::mutex>> locks;
    for (madsim_hip_ctx* c : order) locks.emplace_back(c->mu);
    for (int g = 0; g < n_ctx; g++) if (ctxs[g]->device < 0) return fail(MADSIM_E_NOINIT, "closed context");
    // 1 + 2. every device's sub-batches and their device-to-host copies are queued from this thread, each device keeping as many
    //        in flight as the workload's occupancy rewards; finished sub-batches move into the caller's array while the rest run
    double kernel_ms = 0.0;
    if ((rc = run_pipelined(ctxs, n_ctx, w, cfg, seed0, count, lim, out, nullptr, &kernel_ms))) return rc;
    // 3. runner verdicts (capacity / step cap) from every shard: one compacted re-launch per round, round-robin over the devices
    if ((rc = rerun_runner_verdicts(ctxs, n_ctx, w, cfg, seed0, count, lim, out, max_rounds, &kernel_ms))) return rc;
    // 4. fold
    if (summary) { host_summary(out, seed0, count, summary); summary->kernel_ms = kernel_ms; summary->wall_s = since(t0); }
    return 0;
}

// ---- campaigns -------------------------------------------------------------------------------------------------------------
// One implementation for one context and for several (madsim_hip_run_campaign_multi): batch k of the range runs on context k % n,
// on that context's flight (k / n) % in_flight — the devices advance through the seed space TOGETHER, so with
// MADSIM_CAMPAIGN_STOP_AT_FAILURE the search ends within one round of batches of the first genuine failure (contiguous blocks per
// device — the rule of madsim_hip_run_batch_multi — would leave the devices that hold the smaller seeds running to their end).
// Reports are read in batch order, so `first_failing_seed` is the smallest failing seed of the prefix [seed0, seed0 + seeds_run)
// whatever the number of devices: the report of n contexts equals the report of one.  Builder::run's fan-out with an early exit
// (runtime/builder.rs:129-160: every thread's result is joined in seed order, the first failure is re-raised).
namespace {
int run_campaign_impl(madsim_hip_ctx* const* ctxs, int n_ctx, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t total,
                      uint64_t batch, uint32_t in_flight, uint32_t flags, const madsim_limits_t* lim, madsim_campaign_t* out) {
    auto t0 = std::chrono::steady_clock::now();
    memset(out, 0, sizeof *out);
    out->first_failing_seed = UINT64_MAX;
    int rc = madsim_geo::validate(w, cfg, &g_err);
    if (rc) return rc;
    if (batch == 0) batch = 65536;
    if (in_flight > (uint32_t)madsim_hip_ctx::CAMPAIGN_MAX) return fail(MADSIM_E_ARG, "at most 8 batches in flight per context");
    if (seed0 + total < seed0) return fail(MADSIM_E_ARG, "seed0 + total wraps");
    if (total == 0) return 0;
    const uint64_t n_batches = (total + batch - 1) / batch, N = (uint64_t)n_ctx;
    std::vector<uint32_t> F(n_ctx);                             // flights per context: what its occupancy rewards, no more than it has batches
    for (int g = 0; g < n_ctx; g++) {
        if ((rc = ctxs[g]->bind())) return rc;
        const uint64_t mine = (n_batches + N - 1 - (uint64_t)g) / N;      // batches g, g + n, g + 2n, ...
        uint32_t f = in_flight ? in_flight : ctxs[g]->flights_for(w, cfg, lim, batch);
        if ((uint64_t)f > mine) f = (uint32_t)mine;
        F[g] = f;
        if (f && (rc = ctxs[g]->ensure_flights(f, batch, false))) return rc;
    }
    int first_err = 0;
    std::string first_msg;
    auto flight_of = [&](uint64_t k) -> madsim_hip_ctx::Flight& { const int g = (int)(k % N); return ctxs[g]->flights[(k / N) % F[g]]; };
    auto queue = [&](uint64_t k) -> int {
        madsim_hip_ctx* c = ctxs[k % N];
        madsim_hip_ctx::Flight& f = flight_of(k);
        const uint64_t lo = k * batch, n = std::min(batch, total - lo);
        int e;
        if ((e = c->bind())) return e;
        HIP_TRY(hipMemsetAsync(f.d_acc6, 0xff, 8, f.stream));
        HIP_TRY(hipMemsetAsync((char*)f.d_acc6 + 8, 0, 24, f.stream));
        HIP_TRY(hipMemsetAsync((char*)f.d_acc6 + 32, 0xff, 8, f.stream));
        HIP_TRY(hipMemsetAsync((char*)f.d_acc6 + 40, 0, 8, f.stream));
        HIP_TRY(hipEventRecord(f.e0, f.stream));
        if ((e = c->launch(w, cfg, seed0 + lo, n, nullptr, lim, f.d_out, f.stream))) return e;
        HIP_TRY(hipEventRecord(f.e1, f.stream));
        madsim_k_launch_summary6(f.d_out, n, seed0 + lo, f.d_acc6, f.stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(f.h_acc6, f.d_acc6, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, f.stream));
        HIP_TRY(hipEventRecord(f.done, f.stream));
        return 0;
    };
    bool stop = false;
    auto harvest = [&](uint64_t k) -> int {                     // wait for batch k, fold its report (batches are read in order)
        madsim_hip_ctx* c = ctxs[k % N];
        madsim_hip_ctx::Flight& f = flight_of(k);
        int e;
        if ((e = c->bind())) return e;
        HIP_TRY(hipEventSynchronize(f.done));
        if (first_err) return 0;                                // (draining after an error: nothing is folded any more)
        const uint64_t lo = k * batch, n = std::min(batch, total - lo);
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, f.e0, f.e1));
        if (!stop) {                                            // batches launched beyond the failing one are not part of the answer
            out->kernel_ms += ms;
            out->batches_run++; out->seeds_run += n;
            out->n_runner += f.h_acc6[5];
            out->n_failed += f.h_acc6[1] - f.h_acc6[5];
            out->total_steps += f.h_acc6[2]; out->total_clock_ns += f.h_acc6[3];
            if (f.h_acc6[4] < out->first_failing_seed) out->first_failing_seed = f.h_acc6[4];
            if ((flags & MADSIM_CAMPAIGN_STOP_AT_FAILURE) && f.h_acc6[4] != UINT64_MAX) stop = true;
        }
        return 0;
    };
    auto note = [&](int e) { if (e && !first_err) { first_err = e; first_msg = g_err; } };
    // batch k may be queued once batch k - n * F[k % n] (the previous user of its flight) has been harvested
    uint64_t launched = 0, harvested = 0;
    while (harvested < launched || (launched < n_batches && !stop && !first_err)) {
        const bool room = launched < n_batches && launched < harvested + N * (uint64_t)F[launched % N];
        if (room && !stop && !first_err) { note(queue(launched)); launched++; continue; }
        note(harvest(harvested)); harvested++;                  // the oldest batch in flight: its stream takes the next launch
    }
    out->batches_launched = launched;
    out->wall_s = since(t0);
    if (first_err) {                                             // (as in run_pipelined: a stale `done` must not let a kernel outlive the call)
        for (int g = 0; g < n_ctx; g++) {
            if (ctxs[g]->bind()) continue;
            for (uint32_t i = 0; i < F[g]; i++) (void)hipStreamSynchronize(ctxs[g]->flights[i].stream);
        }
        return fail(first_err, first_msg);
    }
    return 0;
}
}  // namespace

int madsim_hip_ctx_run_campaign(madsim_hip_ctx_t* c, const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t total,
                                uint64_t batch, uint32_t in_flight, uint32_t flags, const madsim_limits_t* lim, madsim_campaign_t* out) {
    if (!out) return fail(MADSIM_E_ARG, "null campaign report");
    CTX_ENTER(c);
    madsim_hip_ctx* one[1] = {c};
    return run_campaign_impl(one, 1, w, cfg, seed0, total, batch, in_flight, flags, lim, out);
}

int madsim_hip_run_campaign_multi(madsim_hip_ctx_t* const* ctxs, int n_ctx, const madsim_workload_t* w, const madsim_config_t* cfg,
                                  uint64_t seed0, uint64_t total, uint64_t batch, uint32_t in_flight, uint32_t flags,
                                  const madsim_limits_t* lim, madsim_campaign_t* out) {
    if (!out) return fail(MADSIM_E_ARG, "null campaign report");
    if (!ctxs || n_ctx < 1) return fail(MADSIM_E_ARG, "run_campaign_multi needs at least one context");
    for (int g = 0; g < n_ctx; g++) {
        if (!ctxs[g]) return fail(MADSIM_E_NOINIT, "null context");
        for (int h = 0; h < g; h++) if (ctxs[h] == ctxs[g]) return fail(MADSIM_E_ARG, "the same context appears twice");
    }
    std::vector<madsim_hip_ctx*> order(ctxs, ctxs + n_ctx);      // locks in address order (see madsim_hip_run_batch_multi)
    std::sort(order.begin(), order.end(), [](madsim_hip_ctx* a, madsim_hip_ctx* b) { return std::less<madsim_hip_ctx*>()(a, b); });
    std::vector<std::unique_lock<std::mutex>> locks;
    for (madsim_hip_ctx* c : order) locks.emplace_back(c->mu);
    for (int g = 0; g < n_ctx; g++) if (ctxs[g]->device < 0) return fail(MADSIM_E_NOINIT, "closed context");
    return run_campaign_impl(ctxs, n_ctx, w, cfg, seed0, total, batch, in_flight, flags, lim, out);
}


// ---- v1 entry points: wrappers on the process-default context ------------------------------------------------------------
// The default context is reference-counted by its users: a wrapper pins it under g_default_mu for the duration of its call,
// and madsim_hip_shutdown waits until no call is inside before destroying it — a concurrent run_batch and shutdown is a
// clean "not initialised" for whichever comes second, never a use-after-free.

namespace {
int g_default_users = 0;
std::condition_variable g_default_cv;
struct DefaultPin {
    madsim_hip_ctx* c;
    DefaultPin() { std::lock_guard<std::mutex> lk(g_default_mu); c = g_default; if (c) g_default_users++; }
    ~DefaultPin() { if (c) { std::lock_guard<std::mutex> lk(g_default_mu); if (--g_default_users == 0) g_default_cv.notify_all(); } }
};
}  // namespace

int madsim_hip_init(int device) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (g_default) {
        if (g_default->device == device) { HIP_TRY(hipSetDevice(device)); return 0; }
        return fail(MADSIM_E_ARG, "the default context is bound to another GPU: call madsim_hip_shutdown first, or hold one "
                                  "madsim_hip_ctx_t per GPU (madsim_hip_ctx_create)");
    }
    return madsim_hip_ctx_create(device, &g_default);
}

int madsim_hip_shutdown(void) {
    madsim_hip_ctx* c;
    {
        std::unique_lock<std::mutex> lk(g_default_mu);
        c = g_default;
        g_default = nullptr;                                   // later calls see "not initialised"
        g_default_cv.wait(lk, [] { return g_default_users == 0; });
    }
    return madsim_hip_ctx_destroy(c);
}

madsim_hip_ctx_t* madsim_hip_default_ctx(void) { std::lock_guard<std::mutex> lk(g_default_mu); return g_default; }

int madsim_hip_run_batch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                         const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary) {
    DefaultPin p;
    return madsim_hip_ctx_run_batch(p.c, w, cfg, seed0, count, lim, out, summary);
}

int madsim_hip_run_batch_auto(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                              const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary, int max_rounds) {
    DefaultPin p;
    return madsim_hip_ctx_run_batch_auto(p.c, w, cfg, seed0, count, lim, out, summary, max_rounds);
}

int madsim_hip_run_batch_device(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                                const madsim_limits_t* lim, void* d_out, void* stream, madsim_summary_t* summary) {
    DefaultPin p;
    return madsim_hip_ctx_run_batch_device(p.c, w, cfg, seed0, count, lim, d_out, stream, summary);
}

int madsim_hip_run_batch_async(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                               const madsim_limits_t* lim, void* d_out, void* d_summary4, void* stream, int timing_slot) {
    DefaultPin p;
    return madsim_hip_ctx_run_batch_async(p.c, w, cfg, seed0, count, lim, d_out, d_summary4, stream, timing_slot);
}

int madsim_hip_timing_ms(int timing_slot, double* ms) { DefaultPin p; return madsim_hip_ctx_timing_ms(p.c, timing_slot, ms); }

int64_t madsim_hip_trace_seed(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed,
                              const madsim_limits_t* lim, uint8_t* log, uint64_t cap, madsim_result_t* out) {
    DefaultPin p;
    return madsim_hip_ctx_trace_seed(p.c, w, cfg, seed, lim, log, cap, out);
}

int madsim_hip_run_campaign(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t total, uint64_t batch,
                            uint32_t in_flight, uint32_t flags, const madsim_limits_t* lim, madsim_campaign_t* out) {
    DefaultPin p;
    return madsim_hip_ctx_run_campaign(p.c, w, cfg, seed0, total, batch, in_flight, flags, lim, out);
}

int madsim_hip_geometry(const madsim_workload_t* w, const madsim_limits_t* lim, madsim_geometry_t* out) {
    if (!out) return fail(MADSIM_E_ARG, "null geometry");
    madsim_config_t cfg = madsim_geo::probe_config();
    int rc = madsim_geo::validate(w, &cfg, &g_err);
    if (rc) return rc;
    Geo G;
    madsim_geo::Device d;
    { std::lock_guard<std::mutex> lk(g_default_mu); if (g_default) d = g_default->dev(); }
    if ((rc = madsim_geo::make_geometry(d, w, &cfg, lim, UINT64_MAX / 2, &G, &g_err))) return rc;
    out->lds_bytes_per_seed = G.lds_per_seed; out->lds_bytes_per_block = G.lds_bytes; out->block_threads = 64 * G.waves_per_block;
    out->blocks_per_cu = G.blocks_per_cu; out->grid_blocks = G.grid; out->heap_lds_slots = G.P.heap_lds;
    out->heap_spill_slots = G.P.heap_spill; out->max_tasks = G.P.max_tasks; out->lanes_per_wave = G.lanes_per_wave;
    const madsim_k::VariantSel v = madsim_k::select_variant(G.P, false);
    out->variant = (v.spill ? 1u : 0u) | ((v.feat & MADSIM_FEAT_ALL) ? 2u : 0u) | (v.rq ? 4u : 0u) | (v.lws < 0 ? 8u : 0u) | (v.g ? 16u : 0u) | ((uint32_t)v.feat << 8) | ((uint32_t)(v.lws & 0xf) << 16);
    out->global_bytes_per_seed = G.P.gstate_mode ? G.P.gs_stride : 0;
    return 0;
}

// Debug: read and clear the per-phase cycle accumulators a profiling kernel build fills (zeros otherwise).
int madsim_hip_debug_counters(uint64_t* out16) {
    DefaultPin pin;
    madsim_hip_ctx* c = pin.c;
    CTX_ENTER(c);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out16, c->d_prof, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(c->d_prof, 0, 16 * sizeof(uint64_t)));
    return 0;
}

// SURVEY.md §8d workload.  Mirrors madsim_amd/workload.py::pingpong instruction for instruction.
int madsim_workload_pingpong(uint32_t n_nodes, uint32_t rounds, madsim_node_t* nodes, madsim_prog_t* progs,
                             madsim_sock_t* socks, madsim_insn_t* insns, uint32_t cap_insns, madsim_workload_t* w) {
    if (!nodes || !progs || !socks || !insns || !w || n_nodes < 2 || (n_nodes & 1) || n_nodes > 31 || rounds == 0 || rounds > 0xffff)
        return fail(MADSIM_E_ARG, "pingpong: need an even node count in 2..30 and 1..65535 rounds");
    const uint32_t PING = 0x676E6970u, PONG = 0x676E6F70u;
    uint32_t need = 2 * n_nodes + 1 + (n_nodes / 2) * (8 + 7);
    if (cap_insns < need) return fail(MADSIM_E_ARG, "pingpong: instruction buffer too small");
    uint32_t n = 0;
    auto emit = [&](uint8_t op, uint8_t a, uint16_t b, uint32_t imm) { insns[n++] = madsim_insn_t{op, a, b, imm}; };
    memset(nodes, 0, (n_nodes + 1) * sizeof *nodes);
    for (uint32_t i = 0; i < n_nodes; i++) socks[i] = madsim_sock_t{(uint8_t)(i + 1), 0, 1};
    // main: spawn every task in node order, then await every JoinHandle in order
    progs[0] = madsim_prog_t{0, 0, 0};
    for (uint32_t i = 0; i < n_nodes; i++) emit(MS_OP_SPAWN, (uint8_t)(i + 1), 0, 0);
    for (uint32_t i = 0; i < n_nodes; i++) emit(MS_OP_JOIN, (uint8_t)(i + 1), 0, 0);
    emit(MS_OP_DONE, 0, 0, 0);
    for (uint32_t i = 0; i < n_nodes; i++) {
        progs[i + 1] = madsim_prog_t{(uint8_t)(i + 1), 0, (uint16_t)n};
        emit(MS_OP_BIND, (uint8_t)i, 0, 0);
        if ((i & 1) == 0) {            // pinger
            emit(MS_OP_SLEEP, 0, 1, 0);
            emit(MS_OP_SET, 0, 0, rounds);
            uint16_t top = (uint16_t)n;
            emit(MS_OP_SEND, (uint8_t)i, (uint16_t)((1u << 8) | (i + 1)), PING);
            emit(MS_OP_RECV, (uint8_t)i, 1u << 8, 0);
            emit(MS_OP_ASSERT_VAL, 0, 0, PONG);
            emit(MS_OP_DJNZ, 0, top, 0);
        } else {                       // ponger
            emit(MS_OP_SET, 0, 0, rounds);
            uint16_t top = (uint16_t)n;
            emit(MS_OP_RECV, (uint8_t)i, 1u << 8, 0);
            emit(MS_OP_ASSERT_VAL, 0, 0, PING);
            emit(MS_OP_REPLY, (uint8_t)i, 1u << 8, PONG);
            emit(MS_OP_DJNZ, 0, top, 0);
        }
        emit(MS_OP_DONE, 0, 0, 0);
    }
    memset(w, 0, sizeof *w);              // no services, no restart rows: every field of the struct is defined
    w->n_nodes = n_nodes; w->n_progs = n_nodes + 1; w->n_socks = n_nodes; w->n_insns = n;
    w->nodes = nodes; w->progs = progs; w->socks = socks; w->insns = insns;
    return (int)n;
}

}  // extern "C"
