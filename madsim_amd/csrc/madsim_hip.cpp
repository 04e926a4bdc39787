// madsim_hip.cpp — host side of libmadsim_hip.so: the C-ABI entry points of include/madsim_hip.h.
//
// Replaces the seed fan-out of madsim::runtime::Builder::run (madsim/src/sim/runtime/builder.rs:121-162):
// where the reference spawns one OS thread per seed, this picks an LDS geometry for the workload,
// launches the gfx950 executor kernel (sim_kernel.hip) with one lane per seed, and reduces the
// per-seed verdicts to "first failing seed" on the device.  No CPU execution path exists here: when
// HIP is unavailable every entry point returns MADSIM_E_HIP / MADSIM_E_NOINIT.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "sim_kernel.h"
#include "geometry.h"

using madsim_k::KParams;

namespace {

struct State {
    bool inited = false;
    int device = -1;
    int num_cus = 0;
    size_t lds_per_cu = 160 * 1024;
    size_t max_lds_block = 64 * 1024;
    // Device copies of workload tables, keyed by content hash.  Entries are immutable once uploaded, so launches that
    // are still in flight on other streams keep valid pointers when a different workload comes along.
    struct Tables { uint64_t hash = 0; uint4* insns = nullptr; uint32_t* progs = nullptr; uint32_t* socks = nullptr; uint64_t* durs = nullptr; };
    std::vector<Tables> tables;
    // timer-heap spill regions, one per stream: launches on one stream run in order, launches on different streams
    // may overlap and must not share scratch
    struct Spill { uint4* p = nullptr; size_t bytes = 0; };
    std::unordered_map<hipStream_t, Spill> spill;
    unsigned long long* d_acc = nullptr;      // 4 x u64 summary accumulators
    madsim_result_t* d_out = nullptr; size_t out_cap = 0;
    uint8_t* d_tlog = nullptr; size_t tlog_cap = 0; uint64_t* d_tlen = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t tev[2 * 64] = {};              // timing slots of madsim_hip_run_batch_async
    uint32_t lds_attr = 0;
    uint64_t* d_prof = nullptr;               // debug counters (EXP_PROF kernel builds)
};

State g;
std::mutex g_mu;
thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }
madsim_geo::Device dev() { madsim_geo::Device d; d.num_cus = g.num_cus > 0 ? g.num_cus : 256; d.lds_per_cu = g.lds_per_cu; return d; }

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

using madsim_geo::Geo;

void free_tables(State::Tables& t) {
    if (t.insns) (void)hipFree(t.insns);
    if (t.progs) (void)hipFree(t.progs);
    if (t.socks) (void)hipFree(t.socks);
    if (t.durs) (void)hipFree(t.durs);
    t = State::Tables();
}

int upload_workload(const madsim_workload_t* w, KParams& P) {
    madsim_geo::DeviceTables T;
    int rc = madsim_geo::build_tables(w, &T, &g_err);
    if (rc) return rc;
    uint64_t h = fnv(T.insns.data(), T.insns.size() * 4, 14695981039346656037ull);
    h = fnv(T.progs.data(), w->n_progs * 4, h);
    h = fnv(T.socks.data(), w->n_socks * 4, h);
    h = fnv(T.durs.data(), T.durs.size() * 8, h);
    h = fnv(&w->n_insns, 4, h);
    const State::Tables* hit = nullptr;
    for (auto& t : g.tables) if (t.hash == h) hit = &t;
    if (!hit) {
        if (g.tables.size() >= 16) {                       // bounded cache: drop everything once nothing is in flight
            HIP_TRY(hipDeviceSynchronize());
            for (auto& t : g.tables) free_tables(t);
            g.tables.clear();
        }
        State::Tables t;
        t.hash = h;
        HIP_TRY(hipMalloc(&t.insns, T.insns.size() * 4 + 16));
        HIP_TRY(hipMalloc(&t.progs, T.progs.size() * 4 + 16));
        HIP_TRY(hipMalloc(&t.socks, T.socks.size() * 4 + 16));
        HIP_TRY(hipMalloc(&t.durs, T.durs.size() * 8 + 16));
        HIP_TRY(hipMemcpy(t.insns, T.insns.data(), T.insns.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t.progs, T.progs.data(), T.progs.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t.socks, T.socks.data(), T.socks.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t.durs, T.durs.data(), T.durs.size() * 8, hipMemcpyHostToDevice));
        g.tables.push_back(t);
        hit = &g.tables.back();
    }
    P.insns = hit->insns; P.progs = hit->progs; P.socks = hit->socks; P.dur_table = hit->durs;
    return 0;
}

int ensure_spill(KParams& P, hipStream_t stream) {
    P.spill = nullptr;
    if (!P.heap_spill) return 0;
    size_t need = (size_t)P.heap_spill * P.total_lanes * sizeof(uint4);
    if (need >= (1ull << 32)) return fail(MADSIM_E_LIMITS, "heap spill region exceeds 4 GiB: lower heap_spill_slots");
    State::Spill& sp = g.spill[stream];
    if (need > sp.bytes) {
        if (sp.p) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(sp.p); }
        sp.p = nullptr; sp.bytes = 0;
        HIP_TRY(hipMalloc(&sp.p, need));
        sp.bytes = need;
    }
    P.spill = sp.p;
    return 0;
}

int run_device(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
               const madsim_limits_t* lim, madsim_result_t* d_out, hipStream_t stream, madsim_summary_t* summary) {
    auto t0 = std::chrono::steady_clock::now();
    if (!g.inited) return fail(MADSIM_E_NOINIT, "madsim_hip_init has not been called");
    int rc = madsim_geo::validate(w, cfg, &g_err);
    if (rc) return rc;
    if (count == 0) {
        if (summary) { memset(summary, 0, sizeof *summary); summary->first_failing_seed = UINT64_MAX; }
        return 0;
    }
    if (!d_out) return fail(MADSIM_E_ARG, "null result buffer");
    Geo G;
    if ((rc = madsim_geo::make_geometry(dev(), w, cfg, lim, count, &G, &g_err))) return rc;
    if ((rc = upload_workload(w, G.P))) return rc;
    if ((rc = ensure_spill(G.P, stream))) return rc;
    G.P.seed0 = seed0; G.P.count = count; G.P.out = d_out; G.P.prof = g.d_prof;
    if (G.lds_bytes > g.lds_attr) {
        int e = madsim_k_set_max_lds((uint32_t)g.lds_per_cu);
        if (e) return fail(MADSIM_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        g.lds_attr = (uint32_t)g.lds_per_cu;
    }
    if (summary) HIP_TRY(hipEventRecord(g.ev0, stream));
    madsim_k_launch_sim(&G.P, G.grid, G.lds_bytes, stream, 0);
    HIP_TRY(hipGetLastError());
    if (summary) {
        HIP_TRY(hipEventRecord(g.ev1, stream));
        HIP_TRY(hipMemsetAsync(g.d_acc, 0xff, 8, stream));
        HIP_TRY(hipMemsetAsync((char*)g.d_acc + 8, 0, 24, stream));
        madsim_k_launch_summary(d_out, count, seed0, g.d_acc, stream);
        HIP_TRY(hipGetLastError());
        unsigned long long acc[4];
        HIP_TRY(hipMemcpyAsync(acc, g.d_acc, sizeof acc, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g.ev0, g.ev1));
        summary->first_failing_seed = acc[0]; summary->n_failed = acc[1];
        summary->total_steps = acc[2]; summary->total_clock_ns = acc[3];
        summary->kernel_ms = ms;
        summary->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

}  // namespace

extern "C" {

uint32_t madsim_hip_version(void) { return MADSIM_HIP_ABI_VERSION; }

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

int madsim_hip_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(MADSIM_E_HIP, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(MADSIM_E_ARG, "device index out of range");
    if (g.inited && g.device == device) { HIP_TRY(hipSetDevice(device)); return 0; }
    if (g.inited) return fail(MADSIM_E_ARG, "already bound to another GPU: one process per GPU (call madsim_hip_shutdown first)");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    g.device = device;
    g.num_cus = prop.multiProcessorCount;
    g.lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 160 * 1024;
    if (g.lds_per_cu > 160 * 1024) g.lds_per_cu = 160 * 1024;
    if (!g.d_acc) HIP_TRY(hipMalloc(&g.d_acc, 4 * sizeof(unsigned long long)));
    if (!g.d_tlen) HIP_TRY(hipMalloc(&g.d_tlen, sizeof(uint64_t)));
    if (!g.d_prof) { HIP_TRY(hipMalloc(&g.d_prof, 16 * sizeof(uint64_t))); HIP_TRY(hipMemset(g.d_prof, 0, 16 * sizeof(uint64_t))); }
    if (!g.ev0) HIP_TRY(hipEventCreate(&g.ev0));
    if (!g.ev1) HIP_TRY(hipEventCreate(&g.ev1));
    g.inited = true;
    return 0;
}

int madsim_hip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.inited) return 0;
    (void)hipDeviceSynchronize();
    for (auto& t : g.tables) free_tables(t);
    for (auto& kv : g.spill) if (kv.second.p) (void)hipFree(kv.second.p);
    if (g.d_acc) (void)hipFree(g.d_acc);
    if (g.d_out) (void)hipFree(g.d_out);
    if (g.d_tlog) (void)hipFree(g.d_tlog);
    if (g.d_tlen) (void)hipFree(g.d_tlen);
    if (g.d_prof) (void)hipFree(g.d_prof);
    if (g.ev0) (void)hipEventDestroy(g.ev0);
    if (g.ev1) (void)hipEventDestroy(g.ev1);
    for (auto& e : g.tev) if (e) (void)hipEventDestroy(e);
    g = State();
    return 0;
}

int madsim_hip_run_batch_device(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                                const madsim_limits_t* lim, void* d_out, void* stream, madsim_summary_t* summary) {
    std::lock_guard<std::mutex> lk(g_mu);
    return run_device(w, cfg, seed0, count, lim, (madsim_result_t*)d_out, (hipStream_t)stream, summary);
}

int madsim_hip_run_batch_async(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                               const madsim_limits_t* lim, void* d_out, void* d_summary4, void* stream, int timing_slot) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.inited) return fail(MADSIM_E_NOINIT, "madsim_hip_init has not been called");
    if (timing_slot >= 64) return fail(MADSIM_E_ARG, "timing_slot must be < 64");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t* ev = nullptr;
    if (timing_slot >= 0) {
        ev = &g.tev[2 * timing_slot];
        if (!ev[0]) { HIP_TRY(hipEventCreate(&ev[0])); HIP_TRY(hipEventCreate(&ev[1])); }
        HIP_TRY(hipEventRecord(ev[0], st));
    }
    int rc = run_device(w, cfg, seed0, count, lim, (madsim_result_t*)d_out, st, nullptr);
    if (rc) return rc;
    if (ev) HIP_TRY(hipEventRecord(ev[1], st));
    if (d_summary4 && count) {
        // {UINT64_MAX, 0, 0, 0}, accumulated by the reduction kernel, then word 0 is flipped into its
        // order-preserving int64 form (seed ^ 1<<63) so a signed all-reduce(MIN) yields the unsigned minimum
        HIP_TRY(hipMemsetAsync(d_summary4, 0xff, 8, st));
        HIP_TRY(hipMemsetAsync((char*)d_summary4 + 8, 0, 24, st));
        madsim_k_launch_summary((const madsim_result_t*)d_out, count, seed0, (unsigned long long*)d_summary4, st);
        HIP_TRY(hipGetLastError());
        madsim_k_launch_keyflip((unsigned long long*)d_summary4, st);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int madsim_hip_timing_ms(int timing_slot, double* ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (timing_slot < 0 || timing_slot >= 64 || !g.tev[2 * timing_slot] || !ms) return fail(MADSIM_E_ARG, "bad timing slot");
    float f = 0.f;
    HIP_TRY(hipEventSynchronize(g.tev[2 * timing_slot + 1]));
    HIP_TRY(hipEventElapsedTime(&f, g.tev[2 * timing_slot], g.tev[2 * timing_slot + 1]));
    *ms = f;
    return 0;
}

int madsim_hip_run_batch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                         const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto t0 = std::chrono::steady_clock::now();
    if (!g.inited) return fail(MADSIM_E_NOINIT, "madsim_hip_init has not been called");
    if (count > g.out_cap) {
        if (g.d_out) (void)hipFree(g.d_out);
        g.d_out = nullptr; g.out_cap = 0;
        HIP_TRY(hipMalloc(&g.d_out, count * sizeof(madsim_result_t)));
        g.out_cap = count;
    }
    madsim_summary_t tmp;
    int rc = run_device(w, cfg, seed0, count, lim, g.d_out, nullptr, &tmp);
    if (rc) return rc;
    if (out && count) HIP_TRY(hipMemcpy(out, g.d_out, count * sizeof(madsim_result_t), hipMemcpyDeviceToHost));
    if (summary) { *summary = tmp; summary->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    return 0;
}

int madsim_hip_run_batch_auto(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                              const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary, int max_rounds) {
    if (!out && count) { std::lock_guard<std::mutex> lk(g_mu); return fail(MADSIM_E_ARG, "run_batch_auto needs the result array"); }
    auto t0 = std::chrono::steady_clock::now();
    madsim_summary_t s{};
    int rc = madsim_hip_run_batch(w, cfg, seed0, count, lim, out, &s);
    if (rc) return rc;
    madsim_limits_t L{};
    if (lim) L = *lim;
    double kernel_ms = s.kernel_ms;
    for (int round = 0; round < max_rounds; round++) {
        bool any = false;
        for (uint64_t i = 0; i < count; i++) any |= out[i].verdict == MADSIM_OVERFLOW;
        if (!any) break;
        // double every capacity (defaults spelled out first); the lane geometry follows the new per-seed footprint
        auto dbl = [](uint32_t v, uint32_t dflt, uint32_t cap) { uint32_t x = (v == 0 || v == MADSIM_LIMIT_NONE) ? dflt : v; x *= 2; return x > cap ? cap : x; };
        L.lanes_per_wave = 0;
        L.heap_lds_slots = L.heap_lds_slots ? L.heap_lds_slots : 8;
        L.heap_spill_slots = dbl(L.heap_spill_slots, 32, 1u << 20);
        L.max_tasks = dbl(L.max_tasks, w->n_progs + 8, 254);
        L.mbox_regs = dbl(L.mbox_regs, 2, 255);
        L.mbox_msgs = dbl(L.mbox_msgs, 2, 255);
        L.max_conns = dbl(L.max_conns, 4, 127);
        L.chan_queue = dbl(L.chan_queue, 2, 15);
        for (uint64_t i = 0; i < count;) {                  // contiguous runs of overflowed seeds
            if (out[i].verdict != MADSIM_OVERFLOW) { i++; continue; }
            uint64_t j = i;
            while (j + 1 < count && out[j + 1].verdict == MADSIM_OVERFLOW) j++;
            madsim_summary_t part{};
            rc = madsim_hip_run_batch(w, cfg, seed0 + i, j - i + 1, &L, out + i, &part);
            if (rc) return rc;
            kernel_ms += part.kernel_ms;
            i = j + 1;
        }
    }
    if (summary) {
        madsim_summary_t f{};
        f.first_failing_seed = UINT64_MAX;
        for (uint64_t i = 0; i < count; i++) {
            if (out[i].verdict != MADSIM_PASS) { if (!f.n_failed) f.first_failing_seed = seed0 + i; f.n_failed++; }
            f.total_steps += out[i].steps; f.total_clock_ns += out[i].clock_ns;
        }
        f.kernel_ms = kernel_ms;
        f.wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        *summary = f;
    }
    return 0;
}

int64_t madsim_hip_trace_seed(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed,
                              const madsim_limits_t* lim, uint8_t* log, uint64_t cap, madsim_result_t* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.inited) return fail(MADSIM_E_NOINIT, "madsim_hip_init has not been called");
    int rc = madsim_geo::validate(w, cfg, &g_err);
    if (rc) return rc;
    Geo G;
    if ((rc = madsim_geo::make_geometry(dev(), w, cfg, lim, 1, &G, &g_err, true))) return rc;
    if ((rc = upload_workload(w, G.P))) return rc;
    if ((rc = ensure_spill(G.P, nullptr))) return rc;
    if (cap > g.tlog_cap) {
        if (g.d_tlog) (void)hipFree(g.d_tlog);
        g.d_tlog = nullptr; g.tlog_cap = 0;
        HIP_TRY(hipMalloc(&g.d_tlog, cap));
        g.tlog_cap = cap;
    }
    if (1 > g.out_cap) { HIP_TRY(hipMalloc(&g.d_out, 64 * sizeof(madsim_result_t))); g.out_cap = 64; }
    G.P.seed0 = seed; G.P.count = 1; G.P.out = g.d_out;
    G.P.trace_log = cap ? g.d_tlog : nullptr; G.P.trace_cap = cap; G.P.trace_len = g.d_tlen;
    if (G.lds_bytes > g.lds_attr) {
        if (madsim_k_set_max_lds((uint32_t)g.lds_per_cu)) return fail(MADSIM_E_HIP, "hipFuncSetAttribute failed");
        g.lds_attr = (uint32_t)g.lds_per_cu;
    }
    madsim_k_launch_sim(&G.P, 1, G.lds_bytes, nullptr, 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    uint64_t n = 0;
    HIP_TRY(hipMemcpy(&n, g.d_tlen, sizeof n, hipMemcpyDeviceToHost));
    if (log && cap) HIP_TRY(hipMemcpy(log, g.d_tlog, n < cap ? n : cap, hipMemcpyDeviceToHost));
    if (out) HIP_TRY(hipMemcpy(out, g.d_out, sizeof *out, hipMemcpyDeviceToHost));
    return (int64_t)n;
}

int madsim_hip_geometry(const madsim_workload_t* w, const madsim_limits_t* lim, madsim_geometry_t* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out) return fail(MADSIM_E_ARG, "null geometry");
    madsim_config_t cfg{}; cfg.lat_lo_ns = 1000000; cfg.lat_hi_ns = 10000000;
    int rc = madsim_geo::validate(w, &cfg, &g_err);
    if (rc) return rc;
    Geo G;
    if ((rc = madsim_geo::make_geometry(dev(), w, &cfg, lim, UINT64_MAX / 2, &G, &g_err))) return rc;
    out->lds_bytes_per_seed = G.lds_per_seed; out->lds_bytes_per_block = G.lds_bytes; out->block_threads = 64 * G.waves_per_block;
    out->blocks_per_cu = G.blocks_per_cu; out->grid_blocks = G.grid; out->heap_lds_slots = G.P.heap_lds;
    out->heap_spill_slots = G.P.heap_spill; out->max_tasks = G.P.max_tasks; out->lanes_per_wave = G.lanes_per_wave;
    out->variant = G.P.lw_shift != 6 ? 8u | 3u : (G.P.heap_spill ? 1u : 0u) | (G.P.lifecycle ? 2u : 0u) | (G.P.rq_in_reg ? 4u : 0u);
    return 0;
}

// Debug: read and clear the per-phase cycle accumulators an EXP_PROF kernel build fills (zeros otherwise).
int madsim_hip_debug_counters(uint64_t* out16) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.inited) return fail(MADSIM_E_NOINIT, "madsim_hip_init has not been called");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out16, g.d_prof, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(g.d_prof, 0, 16 * sizeof(uint64_t)));
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
    w->n_nodes = n_nodes; w->n_progs = n_nodes + 1; w->n_socks = n_nodes; w->n_insns = n;
    w->nodes = nodes; w->progs = progs; w->socks = socks; w->insns = insns;
    return (int)n;
}

}  // extern "C"
