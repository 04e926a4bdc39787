// emu_driver.cpp — runs madsim_amd/csrc/sim_kernel.hip's kernel body on the host, one emulated
// thread after another (see emu_shim.h: debugging aid, not product, not a fallback).
#include "emu_shim.h"
thread_local emu_dim3 threadIdx, blockIdx;
thread_local uint32_t* emu_smem;

#include <string>
#include <vector>
#include <cstring>

// the device code, host-compiled: the same headers sim_kernel.hip includes (emu_shim.h, included first, replaces k_mem.h's body)
#include "../../madsim_amd/csrc/kernel/k_mem.h"
#include "../../madsim_amd/csrc/sim_kernel.h"
#include "../../madsim_amd/csrc/kernel/k_state.h"
#include "../../madsim_amd/csrc/kernel/k_rng.h"
#include "../../madsim_amd/csrc/kernel/k_timer.h"
#include "../../madsim_amd/csrc/kernel/k_net.h"
#include "../../madsim_amd/csrc/kernel/k_lifecycle.h"
#include "../../madsim_amd/csrc/kernel/k_channel.h"
#include "../../madsim_amd/csrc/kernel/k_poll.h"
#include "../../madsim_amd/csrc/kernel/k_main.h"
#include "../../madsim_amd/csrc/geometry.h"

static thread_local std::string emu_err;

#ifdef MADSIM_EMU_SITES
// Memory-site model (tools/mem_site_model.py; needs MADSIM_EMU_REGIONS for the iteration marks and a -O0 -fno-omit-frame-pointer
// build): every global access is keyed by the chain of return addresses above it — on the GPU everything is inlined, so one
// chain = one machine instruction site — and counted per lane and main-loop iteration; a wave issues the site's instruction
// max-over-lanes times per iteration.
#include <unordered_map>
#include <map>
static constexpr int SITE_DEPTH = 7;
struct SiteKey { uintptr_t ra[SITE_DEPTH]; int kind; bool operator<(const SiteKey& o) const { return memcmp(this, &o, sizeof *this) < 0; } };
static std::map<SiteKey, uint32_t> site_ids;
static std::vector<SiteKey> site_keys;
static thread_local std::vector<std::unordered_map<uint32_t, uint16_t>>* emu_site_log = nullptr;
static std::vector<double> site_trips, site_visits;
static std::vector<uint64_t> site_zero;      // loads whose (first) word read as zero: candidates for a mirror bit (tools/mem_site_model.py)
__attribute__((noinline)) void madsim_k::emu_site(int kind, const void* base, uint32_t off) {
    if (!emu_site_log || emu_site_log->empty()) return;
    SiteKey k; memset(&k, 0, sizeof k); k.kind = kind;
    void** fp = (void**)__builtin_frame_address(0);
    fp = (void**)fp[0];                                     // skip emu_site's caller frame (buf_load/store itself)
    for (int i = 0; i < SITE_DEPTH && fp; i++) {
        k.ra[i] = (uintptr_t)fp[1];
        void** up = (void**)fp[0];
        if (up <= fp || (uintptr_t)up - (uintptr_t)fp > (1u << 20) || ((uintptr_t)up & 7)) break;      // left the frames built with frame pointers
        fp = up;
    }
    auto it = site_ids.find(k);
    uint32_t id;
    if (it == site_ids.end()) { id = (uint32_t)site_keys.size(); site_ids.emplace(k, id); site_keys.push_back(k); } else id = it->second;
    (*emu_site_log).back()[id]++;
    if (kind == 0 || kind == 2) {
        if (site_zero.size() <= id) site_zero.resize(id + 1);
        const uint32_t* w = (const uint32_t*)((const uint8_t*)base + off);
        if (kind == 0 ? w[0] == 0 : (w[0] | w[1] | w[2] | w[3]) == 0) site_zero[id]++;
    }
}
extern "C" uint64_t madsim_emu_site_zero(uint32_t id) { return id < site_zero.size() ? site_zero[id] : 0; }
extern "C" uint32_t madsim_emu_site_count(void) { return (uint32_t)site_keys.size(); }
extern "C" void madsim_emu_site(uint32_t id, uintptr_t* ra, int* kind, double* trips, double* visits) {
    for (int i = 0; i < SITE_DEPTH; i++) ra[i] = site_keys[id].ra[i];
    *kind = site_keys[id].kind; *trips = id < site_trips.size() ? site_trips[id] : 0; *visits = id < site_visits.size() ? site_visits[id] : 0;
}
extern "C" void madsim_emu_anchor(void) {}
#endif

#ifdef MADSIM_EMU_REGIONS
// Divergence model: every lane logs how often it visits each marked code region in each main-loop iteration; a
// wave executes a region max-over-lanes times per iteration (lanes re-converge at the loop top), so
// sum(max) = wave trips and sum(visits) / (64 * trips) = lane utilisation of that region.
#include <array>
#include <stdio.h>
#include <stdlib.h>
static constexpr int NREG = 32;
static thread_local std::vector<std::array<uint16_t, NREG>>* emu_lane_log = nullptr;
void emu_region(int id) {
    if (!emu_lane_log) return;
    if (id == 0) {
        emu_lane_log->push_back({});
#ifdef MADSIM_EMU_SITES
        if (emu_site_log) emu_site_log->push_back({});
#endif
    }
    if (emu_lane_log->empty()) return;
    emu_lane_log->back()[id]++;
}
static double reg_trips[NREG], reg_visits[NREG], reg_iters;
extern "C" void madsim_emu_region_stats(double* trips, double* visits, double* iters) {
    for (int i = 0; i < NREG; i++) { trips[i] = reg_trips[i]; visits[i] = reg_visits[i]; }
    *iters = reg_iters;
}
#endif

#ifdef MADSIM_EMU_GSTAT
// per byte offset WITHIN a lane's state block: access counts by kind, summed over all lanes
static std::vector<uint64_t> gstat[4];
static uint32_t gstat_stride = 1, gstat_planes = 0, gstat_lanes = 1;
void emu_gstat(uint32_t byte_off, int kind) {
    // buffer offset -> logical offset inside the lane's state ([unit][lane] then [word][lane], k_state.h gs_addr_*)
    uint32_t o;
    if (byte_off < gstat_planes * gstat_lanes) o = byte_off / (gstat_lanes * 16u) * 16u + byte_off % 16u;
    else o = gstat_planes + (byte_off - gstat_planes * gstat_lanes) / (gstat_lanes * 4u) * 4u;
    o %= gstat_stride;
    if (gstat[kind].size() <= o / 4) gstat[kind].resize(o / 4 + 1);
    gstat[kind][o / 4]++;
}
extern "C" uint64_t madsim_emu_gstat(int kind, uint32_t word) { return word < gstat[kind].size() ? gstat[kind][word] : 0; }
extern "C" void madsim_emu_gstat_reset(uint32_t stride) { for (auto& g : gstat) g.clear(); gstat_stride = stride ? stride : 1; }
#endif

// k_state.h OVF_SET: which site raised a verdict bit (MADSIM_EMU_OVF_DEBUG=1 -> stderr)
namespace madsim_k {
void madsim_emu_ovf_note(const char* file, int line, uint32_t bits) {
    static const bool on = getenv("MADSIM_EMU_OVF_DEBUG") != nullptr;
    if (on) { const char* b = strrchr(file, '/'); fprintf(stderr, "[emu] verdict bit %u raised at %s:%d\n", bits, b ? b + 1 : file, line); }
}
}  // namespace madsim_k

extern "C" const char* madsim_emu_last_error(void) { return emu_err.c_str(); }

extern "C" int madsim_emu_run_batch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                                    const madsim_limits_t* lim, madsim_result_t* out, int num_cus, uint8_t* tlog,
                                    uint64_t tcap, uint64_t* tlen) {
    int rc = madsim_geo::validate(w, cfg, &emu_err);
    if (rc) return rc;
    madsim_geo::Device dev; dev.num_cus = num_cus > 0 ? num_cus : 2;
    madsim_geo::Geo G;
    if ((rc = madsim_geo::make_geometry(dev, w, cfg, lim, count, &G, &emu_err, tlog != nullptr))) return rc;
    madsim_k::KParams& P = G.P;
    madsim_geo::DeviceTables T;
    if ((rc = madsim_geo::build_tables(w, &T, &emu_err))) return rc;
    P.insns = (const uint4*)T.insns.data(); P.progs = T.progs.data(); P.socks = T.socks.data(); P.nodes = T.nodes.data(); P.dur_table = T.durs.data();
    std::vector<uint4> spill((size_t)P.heap_spill * P.total_lanes + 1);
    P.spill = P.heap_spill ? spill.data() : nullptr;
    std::vector<uint4> gstate((size_t)P.gs_stride * P.total_lanes / 16 + 4);
    P.gstate = (P.gstate_mode || P.compact) ? (uint8_t*)gstate.data() : nullptr;
#ifdef MADSIM_EMU_GSTAT
    gstat_planes = P.gs_planes; gstat_lanes = P.total_lanes ? P.total_lanes : 1;
#endif
    P.seed0 = seed0; P.count = count; P.out = out;
    uint64_t dummy_len = 0;
    P.trace_log = tlog; P.trace_cap = tcap; P.trace_len = tlen ? tlen : &dummy_len;
    std::vector<uint32_t> lds(G.lds_bytes / 4 + 4);
    uint32_t* base = lds.data();
    while (((uintptr_t)base) & 15) base++;
#ifdef MADSIM_EMU_REGIONS
    for (int i = 0; i < NREG; i++) reg_trips[i] = reg_visits[i] = 0;
    reg_iters = 0;
#endif
    for (uint32_t b = 0; b < G.grid * G.waves_per_block; b++) {       // b = wave index: waves are independent
#ifdef MADSIM_EMU_REGIONS
        std::vector<std::vector<std::array<uint16_t, NREG>>> logs(G.lanes_per_wave);
#endif
#ifdef MADSIM_EMU_SITES
        std::vector<std::vector<std::unordered_map<uint32_t, uint16_t>>> slogs(G.lanes_per_wave);
#endif
        for (uint32_t t = 0; t < G.lanes_per_wave; t++) {
#ifdef MADSIM_EMU_REGIONS
            emu_lane_log = &logs[t];
#endif
#ifdef MADSIM_EMU_SITES
            emu_site_log = &slogs[t];
#endif
            blockIdx.x = b / G.waves_per_block; threadIdx.x = (b % G.waves_per_block) * 64 + t; emu_smem = base;
            using namespace madsim_k;
            const VariantSel v = select_variant(P, tlog != nullptr);      // the same build the GPU launcher would pick
            bool ran = false;
#define EMU_TRY(T_, S_, L_, F_, R_, G_) \
            if (!ran && v.trace == (int)(T_) && v.spill == (int)(S_) && v.lws == (L_) && v.feat == (F_) && v.rq == (int)(R_) && v.g == (int)(G_)) { sim_kernel<Variant<T_, S_, L_, F_, R_, G_>>(P); ran = true; }
            MADSIM_FOR_EACH_VARIANT(EMU_TRY)
#undef EMU_TRY
            if (!ran) { emu_err = "select_variant named a build that is not compiled"; return MADSIM_E_LIMITS; }
        }
#ifdef MADSIM_EMU_SITES
        emu_site_log = nullptr;
        {
            size_t its = 0;
            for (auto& l : slogs) its = l.size() > its ? l.size() : its;
            std::unordered_map<uint32_t, uint32_t> mx;
            for (size_t i = 0; i < its; i++) {
                mx.clear();
                for (auto& l : slogs) if (i < l.size()) for (auto& kv : l[i]) {
                    if (site_visits.size() <= kv.first) { site_visits.resize(kv.first + 1); site_trips.resize(kv.first + 1); }
                    site_visits[kv.first] += kv.second;
                    uint32_t& m = mx[kv.first]; if (kv.second > m) m = kv.second;
                }
                for (auto& kv : mx) site_trips[kv.first] += kv.second;
            }
        }
#endif
#ifdef MADSIM_EMU_REGIONS
        emu_lane_log = nullptr;
        size_t iters = 0;
        for (auto& l : logs) iters = l.size() > iters ? l.size() : iters;
        reg_iters += (double)iters;
        if (const char* dp = getenv("MADSIM_EMU_DUMP")) {      // raw RNG-attempt counts: [wave][iter][lane][5] bytes
            FILE* f = fopen(dp, b == 0 ? "wb" : "ab");
            static int ids[5] = {1, 7, 8, 16, 18};               // (MADSIM_EMU_DUMP_IDS=a,b,c,d,e: five other region ids)
            if (const char* e = getenv("MADSIM_EMU_DUMP_IDS")) sscanf(e, "%d,%d,%d,%d,%d", &ids[0], &ids[1], &ids[2], &ids[3], &ids[4]);
            uint32_t hdr[2] = {(uint32_t)iters, (uint32_t)logs.size()};
            fwrite(hdr, 4, 2, f);
            for (size_t i = 0; i < iters; i++)
                for (auto& l : logs) { uint8_t v[5]; for (int k = 0; k < 5; k++) v[k] = i < l.size() ? (uint8_t)l[i][ids[k]] : 0; fwrite(v, 1, 5, f); }
            fclose(f);
        }
        for (size_t i = 0; i < iters; i++)
            for (int r = 0; r < NREG; r++) {
                uint32_t mx = 0, sum = 0;
                for (auto& l : logs) if (i < l.size()) { mx = l[i][r] > mx ? l[i][r] : mx; sum += l[i][r]; }
                reg_trips[r] += mx; reg_visits[r] += sum;
            }
#endif
    }
    return 0;
}

extern "C" int madsim_emu_geometry(const madsim_workload_t* w, const madsim_limits_t* lim, madsim_geometry_t* out) {
    madsim_config_t cfg = madsim_geo::probe_config();
    int rc = madsim_geo::validate(w, &cfg, &emu_err);
    if (rc) return rc;
    madsim_geo::Geo G; madsim_geo::Device dev;
    if ((rc = madsim_geo::make_geometry(dev, w, &cfg, lim, UINT64_MAX / 2, &G, &emu_err))) return rc;
    out->lds_bytes_per_seed = G.lds_per_seed; out->lds_bytes_per_block = G.lds_bytes; out->block_threads = 64 * G.waves_per_block;
    out->blocks_per_cu = G.blocks_per_cu; out->grid_blocks = G.grid; out->heap_lds_slots = G.P.heap_lds;
    out->heap_spill_slots = G.P.heap_spill; out->max_tasks = G.P.max_tasks; out->lanes_per_wave = G.lanes_per_wave;
    out->variant = G.P.lw_shift != 6 ? 8u | 3u : (G.P.heap_spill ? 1u : 0u) | (G.P.lifecycle ? 2u : 0u) | (G.P.rq_in_reg ? 4u : 0u);
    return 0;
}

// selected KParams of the geometry a workload gets (tools/gstate_access_model.py)
extern "C" int madsim_emu_geometry_params(const madsim_workload_t* w, const madsim_limits_t* lim, uint32_t* out32) {
    madsim_config_t cfg = madsim_geo::probe_config();
    madsim_geo::Device dev; dev.num_cus = 1;
    if (const char* e = getenv("MADSIM_HIP_WAVES_PER_SIMD")) dev.max_waves_per_simd = atoi(e);
    madsim_geo::Geo G;
    int rc = madsim_geo::make_geometry(dev, w, &cfg, lim, 64, &G, &emu_err);
    if (rc) return rc;
    const madsim_k::KParams& P = G.P;
    const uint32_t v[] = {P.gs_stride, P.gs_planes, P.max_tasks, P.task_units, P.n_socks, P.sock_words, P.mbox_regs, P.mbox_msgs, P.off_socks,
                          P.off_handles, P.off_nodes, P.off_clog, P.off_pause, P.off_greg, P.off_conn, P.gs_plane_words, P.n_progs,
                          P.features, P.gstate_mode, P.dedup_n, P.dedup_off, P.narrow, P.pool_n, P.heap_lds, P.heap_spill, G.lds_per_seed};
    for (size_t i = 0; i < sizeof v / sizeof *v; i++) out32[i] = v[i];
    return 0;
}
