// k_state.h — Per-lane state, LDS layout accessors and compile-time variants of sim_kernel.
// Part of sim_kernel.hip (included in this order: k_state, k_rng, k_timer, k_net, k_lifecycle, k_channel, k_poll).
#ifndef MADSIM_K_STATE_H
#define MADSIM_K_STATE_H

namespace madsim_k {

// Timing-experiment switches live outside the product tree (tools/experiment/k_experiment.h) and are reachable only
// through tools/build_variant.sh; the product build refuses them.
#ifdef MADSIM_EXPERIMENT_BUILD
#include <type_traits>
#include "../../../tools/experiment/k_experiment.h"
#else
#if defined(EXP_NOLOG) || defined(EXP_ALWAYS_ACCEPT) || defined(EXP_PROF) || defined(EXP_PROF2) || defined(EXP_NO_LWS_VARIANTS) || defined(EXP_HALF_LANES)
#error "EXP_* switches make a non-bit-exact kernel: build experiment variants with tools/build_variant.sh, never the product Makefile"
#endif
#define EXP_ACCEPT(x) (x)
#define EXP_LANE_DIV 1u
#define MADSIM_K_LOG_ENABLED 1
#define PROBE(i) do { } while (0)
#define PROBE2(i) do { } while (0)
#define PROBE_FLUSH() do { } while (0)
#endif
#define FNV_OFFSET 14695981039346656037ull
#define FNV_PRIME 1099511628211ull
#define NS_PER_S 1000000000ull
#define NS_PER_MS 1000000ull

// Task state = 16-byte units [unit][lane] (ds_read/write_b128, conflict-free).  Base-op builds (Variant::LIFE false) keep
// unit1 as 8 bytes {x, y} in its own [slot][lane] array: a task of theirs that awaits a Sleep is only ever polled again by
// that Sleep's own timer (no timeouts, no kills), so the deadline need not be remembered — 24 bytes per task instead of 32.
//   unit0 {x: flags:8 | gen:16 | prog:8,  y: pc:16 | sub:8 | from:8,  z: cnt0:16 | cnt1:16,  w: val}
//   unit1 {x: rxseq:8 | joiner:8 | joiner_gen:16,  y: -,  z: deadline lo,  w: deadline hi}
//   unit2 {x: t0 lo, y: t0 hi, z/w: timeout()'s deadline}   (only when the workload uses MS_OP_MARK / timeouts)
//   unit[P.chan_unit] {x: conn:8 | side:1 | backoff ms:16, y: staged payload, z/w: arrive}   (reliable channel)
//   unit[P.rpc_unit]  {x: rsp_tag in hand, y: rsp_tag staged with the oneshot value}        (typed RPC)
enum : uint32_t { TF_ALIVE = 1, TF_SCHED = 2, TF_RUN = 4, TF_KILLED = 8, TF_CANCEL = 16, TF_INBOX = 32,
                  TF_RXWRAP = 64 /* this task's 8-bit receive sequence number has wrapped at least once */,
                  TF_OWNER = 128 /* this task has bound an Endpoint: its finish must look for sockets to close */ };
// `sub` values of a task parked in MS_OP_JOIN (bit 7 set: stage [A] of poll_task ignores them, stage [C] owns them)
enum : uint32_t { SUB_JOIN_WAIT = 0x80, SUB_JOIN_COMPLETED = 0x81, SUB_JOIN_CANCELLED = 0x82 };
enum : uint32_t { EV_WAKE = 1, EV_DELIVER = 2, EV_RESTART = 3,
                  EV_NOP = 4 /* a delivery timer whose message a response hook drops: fires, delivers nothing */ };
// timer meta word: kind << 29 | ...;  EV_WAKE: gen << 8 | slot;  EV_RESTART: node;  EV_DELIVER (extended builds):
// socket gen << 21 | tag << 13 | from (source socket | dst-was-loopback << 6) << 6 | destination socket;  EV_DELIVER (base-op
// builds): socket gen << 21 | pc of the sending instruction << 6 | destination socket
#define EV_SHIFT 29
enum : uint32_t { H_NONE = 0, H_RUNNING = 1, H_COMPLETED = 2, H_CANCELLED = 3 };

// Compile-time kernel variant: TRACE = also emit the raw determinism log (single-seed trace mode);
// SPILL = the timer heap may overflow from LDS into the HBM spill region.
// LWS = log2(lane stride) when known at compile time (6: full 64-lane waves), or -1: read it from KParams.
// FEAT = which classes of extended ops are compiled in (MADSIM_FEAT_* bits, geometry.h picks the mask from the ops a
// workload uses): FT timeouts / t0 family / advance, FC reliable channel, FR typed RPC, FN node lifecycle (kill /
// restart / pause / abort, init programs, restart_on_panic), FA general address resolution.  0 = the fast variant: none of that cold code in the hot
// loop.  LIFE = any extended op: selects the extended LDS layout (handle plane, node region, whole-unit task stores).
// RQ = the ready queue (<= 8 tasks) lives in a 64-bit register, one byte per queued task, instead of LDS.
// G = the per-seed task table and planes live in a per-lane block of global memory (L2 / Infinity Cache / HBM) instead of
// LDS; only the timer-heap top and the ready queue stay in LDS.  Extended-op workloads carry 1-4 KB of state per seed, which
// in LDS caps a CU at 32-128 seeds; their wave-iterations cost tens of thousands of cycles (the wave executes the union of
// its lanes' op handlers), so a few hundred cycles of global latency per access are cheap next to 4-16x the seeds in flight.
template <bool TRACE_, bool SPILL_, int LWS_, int FEAT_, bool RQ_ = false, bool G_ = false> struct Variant {
    static constexpr bool TRACE = TRACE_, SPILL = SPILL_, RQ = RQ_, G = G_;
    static constexpr int LWS = LWS_, FEAT = FEAT_ & MADSIM_FEAT_ALL;
    static constexpr bool LIFE = FEAT != 0;
    // the determinism-log fold (rng_log): compiled out (a base-op twin selected by KParams.no_log), or compiled in behind a run-time
    // test of KParams.no_log — every build except the base-op full-wave ones without a spill region, whose twin takes those launches
    // the compact base-op layout (sim_kernel.h MADSIM_FEAT_COMPACT): 8-byte heap entries + root in registers + main task in global memory
    static constexpr bool CMP = (FEAT_ & MADSIM_FEAT_COMPACT) != 0;
    static constexpr bool NOLOG = (FEAT_ & MADSIM_FEAT_NOLOG) != 0;
    // 8-byte timer-heap entries + delivery record pool (sim_kernel.h MADSIM_FEAT_NARROW; k_timer.h)
    static constexpr bool NH = (FEAT_ & MADSIM_FEAT_NARROW) != 0;
    static constexpr bool LOGSW = !NOLOG && !TRACE_ && (LIFE || SPILL_ || LWS_ != 6);
    // MADSIM_STATE_DEDUP_TIMERS (KParams.dedup_n): only the global-state build of timeout-only workloads carries the code
    static constexpr bool DEDUP = G_ && !TRACE_ && (FEAT_ & MADSIM_FEAT_ALL) == MADSIM_FEAT_TIME;
    static constexpr bool FT = (FEAT_ & MADSIM_FEAT_TIME) != 0, FC = (FEAT_ & MADSIM_FEAT_CHAN) != 0,
                          FR = (FEAT_ & MADSIM_FEAT_RPC) != 0, FN = (FEAT_ & MADSIM_FEAT_NODE) != 0,
                          FA = (FEAT_ & MADSIM_FEAT_ADDR) != 0;
};

// REG(id): divergence-model markers, compiled in only by tools/divergence_model.py's host emulation build
#ifndef REG
#define REG(id) do { } while (0)
#endif

// Lane::ovf: OVF_CAP = a device capacity was exceeded (MADSIM_OVERFLOW: run again with larger limits); OVF_MODEL = the seed
// did what the workload model cannot say (MADSIM_UNSUPPORTED; the oracle reports the same); OVF_BUG = an invariant of this code
// broke (MADSIM_INTERNAL: the parity tests assert it never shows).  The FIRST bit raised is the seed's verdict: what runs
// after it inside the same round runs on state the event already spoiled (a dropped message, a stale handle), so a later bit says
// nothing — a capacity verdict raised first is re-run with larger limits and decides the rest then; a model verdict raised first
// is what the oracle reports at that very instruction.
enum : uint32_t { OVF_CAP = 1, OVF_MODEL = 2, OVF_BUG = 4 };
// OVF_SET(L, bits): the one way a verdict bit is raised.  The host-compiled test harness (tests/emu) can name the site that raised
// it (MADSIM_EMU_OVF_DEBUG=1: file:line on stderr) — "which capacity was it" is the first question behind every MADSIM_OVERFLOW.
#ifdef MADSIM_EMU
void madsim_emu_ovf_note(const char* file, int line, uint32_t bits);
#define OVF_SET(L, bits) ((L).ovf = (L).ovf ? (L).ovf : (bits), madsim_emu_ovf_note(__FILE__, __LINE__, (bits)))
#else
#define OVF_SET(L, bits) ((L).ovf = (L).ovf ? (L).ovf : (bits))
#endif
struct Lane {
    // GlobalRng
    uint64_t s0, s1, s2, s3;
    uint64_t peek;           // rng_out of the current state where the build keeps the determinism log (k_rng.h Peek, rng_log, gen_index)
    uint64_t rng_calls;
    uint64_t trace_hash;
    uint64_t log_len;
    // Clock
    uint64_t clock;
    // Timer: write-through mirror of heap[0]'s deadline (UINT64_MAX when empty)
    uint64_t top_dl;
    uint32_t top_meta;   // compact builds: the meta word of heap[0] (the root entry lives in registers)
    // accounting
    uint64_t obs_hash;
    uint32_t msg_count;
    uint32_t steps;
    uint32_t ready_len;
    uint64_t rq;         // K::RQ variants: the ready queue itself, byte i = i-th queued task slot
    uint32_t heap_len;
    uint32_t verdict;
#ifdef MADSIM_K_PROF
    uint64_t prof_acc[12]; uint64_t prof_t;
#endif
    uint32_t main_done;  // handle[0] left H_RUNNING: block_on's task.is_finished()
    uint32_t panic_code; // message code of the panic being unwound (MS_OP_PANIC), MADSIM_PANIC_CODE_OTHER for the rest
    uint32_t ovf;        // sticky OVF_* bits: the seed ends with a runner verdict (k_main.h), whatever its state says by then
    // runtime-mutable net config (MS_OP_SET_LOSS, MS_OP_SET_LATENCY)
    uint64_t loss_pint;
    uint32_t loss_always;   // bit 0: packet_loss_rate == 1; bits 4-6 (extended builds): current send_latency = 0 the launch's, k + 1 = lat_table[k]; bit 8: a node is clogged, bit 9: a link has been clogged (k_channel.h)
    // Global-state builds: the Timer::add calls of one poll round, held back and performed at ONE site (poll_task's round
    // head, k_timer.h timer_flush) in their original order: at most one delivery event, then up to three wake-ups of the
    // polled task.  Inlined at each of its ~20 call sites the push's sift-up ran once per site, for the few lanes that
    // were there, every spilled level a round trip to global memory; at one site all lanes of the wave share those trips.
    uint64_t pq_deliv_dl, pq_w0, pq_w1, pq_w2;
    uint32_t pq_deliv_meta, pq_deliv_val;
    uint32_t pq_n;       // bit 7: a delivery is pending; bits 0-2: pending wake-ups; bits 3-5: wake-up i re-registers a pending Sleep (DEDUP)
    // MADSIM_STATE_DEDUP_TIMERS: `hazard` = this iteration popped two different events with one deadline (their order is the
    // heap shape's, which the de-duplicated heap does not share with the reference's): the seed starts over with `exact` set —
    // every timer a heap entry, as everywhere else
    uint32_t exact, hazard;
    uint64_t dd_occ;         // which buckets of the re-registration table hold a count (k_timer.h dedup_note)
};

#define LDS128(i) (reinterpret_cast<uint4*>(SMEM)[(i)])
#define LDS64(i) (reinterpret_cast<uint2*>(SMEM)[(i)])

struct Ctx {
    const KParams& P;
    uint32_t lws;        // log2(lane stride) (runtime copy; K::LWS overrides when >= 0)
    uint32_t ready0, hand0, node0, clog0, pause0, greg0, conn0, hook0, ipvs0;   // word indices of this lane's plane regions
    uint32_t sock0;      // word index of this lane's socket region
    uint32_t heap0;      // uint4 index of heap entry 0: entry i = LDS128(heap0 + (i << lws)); base-op builds: uint2 index
    uint32_t heapm0;     // base-op builds: word index of the meta word of heap entry 0 (k_timer.h)
    uint32_t task0;      // uint4 index of task unit 0
    uint32_t task1;      // base-op builds: uint2 index of the 8-byte unit1 array (see "Task state")
    uint32_t insn0;      // uint4 index of the workgroup-shared instruction table
    uint32_t prog0, sockt0, nodet0;   // word indices of the shared prog / socket-address / node tables
    BufRef spill;      // the HBM spill region: entry (slot, this lane) at byte (slot * P.total_lanes) * 16 + spill_off
    uint32_t spill_off;  // this lane's column: global lane * 16
    // K::G builds: the lane's state is P.gs_stride bytes — task units, then the plane words — laid out across the launch as
    // [unit][global lane] (16-byte units) followed by [word][global lane] (4-byte words), so lanes that touch the same unit
    // or word share cache lines.  task0 and the plane bases (sock0, hand0, node0, clog0, pause0, greg0, conn0) are BYTE
    // offsets inside the lane's logical block; gs_addr_* turn such an offset into a byte offset in the buffer.
    BufRef gs;
    uint32_t gs_lane;   // global lane index
    // K::G builds keep two small indexes in LDS so the common scans never walk global memory: bit t of the alive mask =
    // task slot t holds a live task (spawn's free-slot search), bit s of the owner mask = socket s was bound by a task that
    // has not finished yet (task_finish's "which endpoints did this task own" search)
    uint32_t amask0, omask0;
    uint32_t pmask0;     // Variant::NH: word index of the delivery-record pool's used mask (bit r of word r / 32 = record r holds a message in flight)
    uint8_t* tlog;       // trace mode only
    __device__ Ctx(const KParams& p) : P(p) {}
};

template <class K> __device__ __forceinline__ uint32_t LWSH(const Ctx& c) { return K::LWS >= 0 ? (uint32_t)K::LWS : c.lws; }
#define RW(i) SMEM[c.ready0 + ((i) << LWSH<K>(c))]
// Entry i of the ready Vec (a task slot).  LDS-resident builds: one plane word per entry.  Global-state builds, where the
// ready queue is most of what is left in LDS: four entries per word.
template <class K> __device__ __forceinline__ uint32_t rq_get(const Ctx& c, uint32_t i) {
    if (!K::G) return RW(i);
    return (RW(i >> 2) >> ((i & 3u) * 8u)) & 0xffu;
}
template <class K> __device__ __forceinline__ void rq_set(const Ctx& c, uint32_t i, uint32_t slot) {
    if (!K::G) { RW(i) = slot; return; }
    const uint32_t sh = (i & 3u) * 8u, w = RW(i >> 2);
    RW(i >> 2) = (w & ~(0xffu << sh)) | (slot << sh);
}
#define AMASK(i) SMEM[c.amask0 + ((i) << LWSH<K>(c))]
#define OMASK(i) SMEM[c.omask0 + ((i) << LWSH<K>(c))]
#define PMASK(i) SMEM[c.pmask0 + ((i) << LWSH<K>(c))]

// ---- the lane's state block in global memory (K::G builds) -----------------------------------------------------------
// Reached through a buffer resource like the heap spill region (k_mem.h).  A block belongs to one lane for the whole launch.
// `off` = byte offset in the state buffer (the lane's gs_off already added).
__device__ __forceinline__ uint32_t gs_load32(const BufRef& gs, uint32_t off) { EMU_GSTAT(off, 0); return buf_load32(gs, off); }
__device__ __forceinline__ void gs_store32(const BufRef& gs, uint32_t off, uint32_t v) { EMU_GSTAT(off, 1); buf_store32(gs, off, v); }
__device__ __forceinline__ void gs_add32(const BufRef& gs, uint32_t off, uint32_t v) { EMU_GSTAT(off, 1); buf_add32(gs, off, v); }
__device__ __forceinline__ uint4 gs_load128(const BufRef& gs, uint32_t off) { EMU_GSTAT(off, 2); return buf_load128(gs, off); }
__device__ __forceinline__ void gs_store128(const BufRef& gs, uint32_t off, const uint4& e) { EMU_GSTAT(off, 3); buf_store128(gs, off, e); }

// One 32-bit word / one 16-byte unit of per-seed state, in LDS ([word][lane] planes) or in the lane's global block.
// The accessor macros below return these, so the executor code reads and writes state the same way in both layouts.
// They carry the buffer resource by value (no reference back into Ctx / KParams: an escaping address would make the
// compiler copy the kernel-argument block to scratch).
template <bool G> struct WRef;
template <> struct WRef<false> {
    uint32_t at;                      // LDS word index
    __device__ __forceinline__ void add(uint32_t v) const { SMEM[at] += v; }
    __device__ __forceinline__ operator uint32_t() const { return SMEM[at]; }
    __device__ __forceinline__ uint32_t operator=(uint32_t v) const { SMEM[at] = v; return v; }
    __device__ __forceinline__ uint32_t operator=(const WRef& o) const { return *this = (uint32_t)o; }
    __device__ __forceinline__ uint32_t operator|=(uint32_t v) const { return *this = (uint32_t)*this | v; }
    __device__ __forceinline__ uint32_t operator&=(uint32_t v) const { return *this = (uint32_t)*this & v; }
    __device__ __forceinline__ uint32_t operator+=(uint32_t v) const { return *this = (uint32_t)*this + v; }
};
template <> struct WRef<true> {
    BufRef gs; uint32_t at;         // byte offset in the state buffer
    __device__ __forceinline__ void add(uint32_t v) const { gs_add32(gs, at, v); }      // (nobody waits: k_mem.h buf_add32)
    __device__ __forceinline__ operator uint32_t() const { return gs_load32(gs, at); }
    __device__ __forceinline__ uint32_t operator=(uint32_t v) const { gs_store32(gs, at, v); return v; }
    __device__ __forceinline__ uint32_t operator=(const WRef& o) const { return *this = (uint32_t)o; }
    __device__ __forceinline__ uint32_t operator|=(uint32_t v) const { return *this = (uint32_t)*this | v; }
    __device__ __forceinline__ uint32_t operator&=(uint32_t v) const { return *this = (uint32_t)*this & v; }
    __device__ __forceinline__ uint32_t operator+=(uint32_t v) const { return *this = (uint32_t)*this + v; }
};
template <bool G> struct URef;
template <> struct URef<false> {
    uint32_t at;                      // LDS uint4 index
    __device__ __forceinline__ operator uint4() const { return LDS128(at); }
    __device__ __forceinline__ void operator=(const uint4& v) const { LDS128(at) = v; }
    __device__ __forceinline__ void operator=(const URef& o) const { *this = (uint4)o; }
};
template <> struct URef<true> {
    BufRef gs; uint32_t at;
    __device__ __forceinline__ operator uint4() const { return gs_load128(gs, at); }
    __device__ __forceinline__ void operator=(const uint4& v) const { gs_store128(gs, at, v); }
    __device__ __forceinline__ void operator=(const URef& o) const { *this = (uint4)o; }
};
// Compact base-op builds: task slot 0 (the main task) lives in global memory, the other slots in LDS.  `glob` is rarely true
// (the main task is polled a handful of times per run), so the global side sits behind a branch that whole waves skip.
struct HWRef {
    BufRef gs; uint32_t at; bool glob;   // at: LDS word index, or byte offset in the main-task buffer
    __device__ __forceinline__ operator uint32_t() const { uint32_t v; if (glob) v = gs_load32(gs, at); else v = SMEM[at]; return v; }
    __device__ __forceinline__ uint32_t operator=(uint32_t v) const { if (glob) gs_store32(gs, at, v); else SMEM[at] = v; return v; }
    __device__ __forceinline__ uint32_t operator=(const HWRef& o) const { return *this = (uint32_t)o; }
    __device__ __forceinline__ uint32_t operator|=(uint32_t v) const { return *this = (uint32_t)*this | v; }
    __device__ __forceinline__ uint32_t operator&=(uint32_t v) const { return *this = (uint32_t)*this & v; }
    __device__ __forceinline__ uint32_t operator+=(uint32_t v) const { return *this = (uint32_t)*this + v; }
};
struct HURef {
    BufRef gs; uint32_t at; bool glob;   // at: LDS uint4 index, or byte offset in the main-task buffer
    __device__ __forceinline__ operator uint4() const { uint4 v; if (glob) v = gs_load128(gs, at); else v = LDS128(at); return v; }
    __device__ __forceinline__ void operator=(const uint4& v) const { if (glob) gs_store128(gs, at, v); else LDS128(at) = v; }
    __device__ __forceinline__ void operator=(const HURef& o) const { *this = (uint4)o; }
};
template <class K> struct TaskRef {      // what TU / TWORD / HW hand out
    typedef typename std::conditional<K::CMP, HURef, URef<K::G>>::type U;
    typedef typename std::conditional<K::CMP, HWRef, WRef<K::G>>::type W;
};
template <bool G> __device__ __forceinline__ WRef<G> make_wref(const Ctx& c, uint32_t lds_at, uint32_t gs_at);
template <> __device__ __forceinline__ WRef<false> make_wref<false>(const Ctx&, uint32_t lds_at, uint32_t) { return WRef<false>{lds_at}; }
// logical offset `at` inside the lane's block -> byte offset in the state buffer (both regions: at * total_lanes + this lane)
__device__ __forceinline__ uint32_t gs_addr_unit(const Ctx& c, uint32_t at) { return __umul24(at, c.P.total_lanes) + c.gs_lane * 16u; }            // at % 16 == 0
__device__ __forceinline__ uint32_t gs_addr_uword(const Ctx& c, uint32_t at) { return __umul24(at & ~15u, c.P.total_lanes) + c.gs_lane * 16u + (at & 15u); }   // a word of a unit
__device__ __forceinline__ uint32_t gs_addr_word(const Ctx& c, uint32_t at) { return __umul24(at, c.P.total_lanes) + c.gs_lane * 4u; }              // a plane word
template <> __device__ __forceinline__ WRef<true> make_wref<true>(const Ctx& c, uint32_t, uint32_t gs_at) { return WRef<true>{c.gs, gs_addr_word(c, gs_at)}; }
__device__ __forceinline__ WRef<true> make_uword_ref(const Ctx& c, uint32_t gs_at) { return WRef<true>{c.gs, gs_addr_uword(c, gs_at)}; }
template <bool G> __device__ __forceinline__ URef<G> make_uref(const Ctx& c, uint32_t lds_at, uint32_t gs_at);
template <> __device__ __forceinline__ URef<false> make_uref<false>(const Ctx&, uint32_t lds_at, uint32_t) { return URef<false>{lds_at}; }
template <> __device__ __forceinline__ URef<true> make_uref<true>(const Ctx& c, uint32_t, uint32_t gs_at) { return URef<true>{c.gs, gs_addr_unit(c, gs_at)}; }
// plane word `i` of the region starting at `base`
template <class K> __device__ __forceinline__ WRef<K::G> plane_ref(const Ctx& c, uint32_t base, uint32_t i) {
    return make_wref<K::G>(c, base + (i << LWSH<K>(c)), base + i * 4u);
}
#define NODEW(i) plane_ref<K>(c, c.node0, (i))
#define CLOGW(i) plane_ref<K>(c, c.clog0, (i))
#define PAUSEW(i) plane_ref<K>(c, c.pause0, (i))   /* [0] = length, [1..] = paused Runnables in pop order */
#define GREGW(i) plane_ref<K>(c, c.greg0, (i))
// NetSim message hooks of node n (net/mod.rs:250-284): request hook valid:1 | all:1<<1 | code:8<<2 | tag:8<<10,
//                                                      response hook valid:1<<18 | all:1<<19 | code:8<<20
#define HOOKW(n_) plane_ref<K>(c, c.hook0, (n_))
// round-robin counter of IPVS service k (net/ipvs.rs Service::rr_index); workloads that change services at run time
// (KParams.ipvs_dyn): two words per service — [2k] servers[0..3], a socket-table entry per byte; [2k + 1] servers[4] |
// servers[5] << 8 | n << 16 | rr_index << 20 | present << 24
#define IPVSW(k_) plane_ref<K>(c, c.ipvs0, (k_))
// connection id_: [0] alive:1 | c_ep:6<<1 | d_ep:6<<7 (the address dialled) | tx0:1<<13 rx0<<14 tx1<<15 rx1<<16 | qn0:4<<17 | qn1:4<<21
//                 [1 + dir] parked receiver: valid:1 | slot:8<<1 | gen:16<<9;  [3 + (dir * Q + i) * 3 ..] {val, arrive lo, arrive hi}
#define CONNW(id_, f_) plane_ref<K>(c, c.conn0, (id_) * c.P.conn_words + (f_))
// node region: [0] killed mask, [1] paused mask, [2] gen0_killed mask, [3] spawn counter, [4 + n/4] info_gen bytes,
//              then one word: the seed's base time in seconds into 2022 (time/mod.rs:26-33)
#define NODE_INFO_GEN(n_) (((uint32_t)NODEW(4 + ((n_) >> 2)) >> (((n_) & 3) * 8)) & 0xff)
// Socket s, field f: [0] header, [1] owner (slot | gen << 16), [2 ..] registrations, then queued messages (2 words each),
// then (channel users) accept queue + parked acceptor.  Header = bound:1 | gen:8 <<1 | nreg:8 <<9 | nmsg <<17.
// Base-op builds have no owner word: the owner's task slot rides in header bits 24-31 (nmsg keeps 7 bits) — a bound
// socket's owner is alive (its finish unbinds it, nothing else can kill it), so the slot alone identifies it.
template <class K> __device__ __forceinline__ uint32_t sock_field(uint32_t f) { return K::LIFE ? f : (f ? f - 1 : 0); }
// (Round 6 built the words one delivery / one receive touches together — header, owner, first registration, first message word — as ONE
// 16-byte unit [socket][lane] (VERDICT r5 #1b), bit-exact; A/B on one MI355X, three rounds: topology 5.03 against 5.32 G steps/s (-5.5 %),
// election loop 9.38 / 9.85 (-4.8 %), KV 11.16 / 12.80 (-13 %).  As with round 4's per-socket granules: lanes of a wave that touch the same
// word of the same socket share the [word][lane] rows' lines, a 16-byte unit per lane shares a quarter as much.  profiles/r6_ab_socket_unit.txt.)
#define SW(c_, s_, f_) plane_ref<K>((c_), (c_).sock0, (s_) * (c_).P.sock_words + sock_field<K>(f_))
template <class K> struct SockHdr { static constexpr uint32_t NMSG_MASK = K::LIFE ? 0xffu : 0x7fu; };
#define HDR_NMSG(h_) (((h_) >> 17) & SockHdr<K>::NMSG_MASK)
#define HDR_SET_NMSG(h_, n_) (((h_) & ~(SockHdr<K>::NMSG_MASK << 17)) | ((n_) << 17))
// header of a fresh Endpoint bound by task (slot, gen): bound, socket gen + 1, empty mailbox
template <class K> __device__ __forceinline__ void sock_bind(const Ctx& c, uint32_t s, uint32_t slot, uint32_t gen) {
    uint32_t h = SW(c, s, 0);
    uint32_t nh = 1u | ((((h >> 1) + 1) & 0xff) << 1);
    if (K::LIFE) SW(c, s, 1) = slot | (gen << 16); else nh |= slot << 24;
    SW(c, s, 0) = nh;
}
// is the task (slot, gen) the one whose BindGuard holds socket s?  (base-op builds: of a BOUND socket)
template <class K> __device__ __forceinline__ bool sock_owned_by(const Ctx& c, uint32_t s, uint32_t h, uint32_t slot, uint32_t gen) {
    if (K::LIFE) return SW(c, s, 1) == (slot | (gen << 16));
    return (h & 1) && (h >> 24) == slot;
}
// Global-state builds: a task slot's units sit together in ONE granule per lane — [slot][global lane][unit 0 .. task_units - 1], the
// granule padded to a power of two (32 / 64 / 128 bytes) — so the two or three units a poll reads and writes back share a
// 32- or 64-byte sector instead of a sector each.  (The plane words stay [word][lane]: lanes that read the same word share lines.)
// byte offset in the state buffer of byte `b` of slot's granule:
__device__ __forceinline__ uint32_t gs_addr_task(const Ctx& c, uint32_t slot, uint32_t b) {
    return __umul24(slot << c.P.gs_gran_sh, c.P.total_lanes) + (c.gs_lane << c.P.gs_gran_sh) + b;
}
template <bool G> __device__ __forceinline__ URef<G> tu_gs(const Ctx& c, uint32_t slot, uint32_t u);
template <> __device__ __forceinline__ URef<false> tu_gs<false>(const Ctx&, uint32_t, uint32_t) { return URef<false>{0}; }
template <> __device__ __forceinline__ URef<true> tu_gs<true>(const Ctx& c, uint32_t slot, uint32_t u) { return URef<true>{c.gs, gs_addr_task(c, slot, u * 16u)}; }
template <bool G> __device__ __forceinline__ WRef<G> tword_gs2(const Ctx& c, uint32_t slot, uint32_t u, uint32_t k);
template <> __device__ __forceinline__ WRef<false> tword_gs2<false>(const Ctx&, uint32_t, uint32_t, uint32_t) { return WRef<false>{0}; }
template <> __device__ __forceinline__ WRef<true> tword_gs2<true>(const Ctx& c, uint32_t slot, uint32_t u, uint32_t k) { return WRef<true>{c.gs, gs_addr_task(c, slot, u * 16u + k * 4u)}; }
template <class K> __device__ __forceinline__ URef<K::G> tu_ref_plain(const Ctx& c, uint32_t slot, uint32_t u) {
    if (!K::LIFE) return make_uref<K::G>(c, c.task0 + (slot << LWSH<K>(c)), 0);             // unit 0 (unit 1: load_u1 / TWORD)
    if (K::G) return tu_gs<K::G>(c, slot, u);
    return make_uref<K::G>(c, c.task0 + ((slot * c.P.task_units + u) << LWSH<K>(c)), 0);
}
template <class K> __device__ __forceinline__ WRef<K::G> tword_ref_plain(const Ctx& c, uint32_t slot, uint32_t u, uint32_t k) {
    if (!K::LIFE) return make_wref<K::G>(c, u == 0 ? (c.task0 + (slot << LWSH<K>(c))) * 4u + k : (c.task1 + (slot << LWSH<K>(c))) * 2u + k, 0);
    if (K::G) return tword_gs2<K::G>(c, slot, u, k);
    return make_wref<K::G>(c, (c.task0 + ((slot * c.P.task_units + u) << LWSH<K>(c))) * 4u + k, 0);
}
// Compact builds: the main task's record in the global buffer is unit0 at [lane * 16], then unit1 {x, y} at
// [total_lanes * 16 + lane * 8]; c.task0 / c.task1 are biased so that slot s >= 1 indexes LDS entry s - 1.
template <class K> __device__ __forceinline__ typename TaskRef<K>::U tu_ref(const Ctx& c, uint32_t slot, uint32_t u) {
    if constexpr (K::CMP) return HURef{c.gs, slot == 0 ? c.gs_lane * 16u : c.task0 + (slot << LWSH<K>(c)), slot == 0};
    else return tu_ref_plain<K>(c, slot, u);
}
template <class K> __device__ __forceinline__ typename TaskRef<K>::W tword_ref(const Ctx& c, uint32_t slot, uint32_t u, uint32_t k) {
    if constexpr (K::CMP) {
        const uint32_t lds = u == 0 ? (c.task0 + (slot << LWSH<K>(c))) * 4u + k : (c.task1 + (slot << LWSH<K>(c))) * 2u + k;
        const uint32_t glb = u == 0 ? c.gs_lane * 16u + k * 4u : c.P.total_lanes * 16u + c.gs_lane * 8u + k * 4u;
        return HWRef{c.gs, slot == 0 ? glb : lds, slot == 0};
    } else return tword_ref_plain<K>(c, slot, u, k);
}
// unit1 as a uint4 in registers; base-op builds hold {x, y} only
template <class K> __device__ __forceinline__ uint4 load_u1(const Ctx& c, uint32_t slot) {
    if (K::LIFE) return tu_ref<K>(c, slot, 1);
    uint2 t;
    if (K::CMP && slot == 0) t = buf_load64(c.gs, c.P.total_lanes * 16u + c.gs_lane * 8u);
    else t = LDS64(c.task1 + (slot << LWSH<K>(c)));
    return make_uint4(t.x, t.y, 0, 0);
}
// Global-state builds: words of the polled task's granule beyond units 0 and 1 that its poll is likely to want, loaded with
// them before the ready-queue draw (k_main.h) — the same 64- or 128-byte line, so they arrive with unit 0 instead of costing the
// poll a dependent round trip of their own:  d2 = timeout()'s deadline (unit 2 z/w; k_poll.h recv_timeout_poll, rpc_call_poll).
// poll_task keeps the copies current through its own writes; nobody else writes these words of a task that is being polled.
// cu = the connection unit {conn | side | backoff, staged payload, arrive} (k_poll.h: every channel op starts from it).
// Load hoists that cost a register each across a handler (k_channel.h, k_poll.h channel ops): the channel-only global-state builds
// take them, and the every-class ones since they run two waves per SIMD (256 registers: k_main.h) — at three they spilled.
template <class K> struct Hoist {
    static constexpr bool ALLG = K::G && (K::FEAT & (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR)) == (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR);
    static constexpr bool CHAN = (K::G && K::FEAT == MADSIM_FEAT_CHAN) || ALLG;
};
// rq0 / rq1 = the rpc unit's two words {rsp_tag in hand, rsp_tag staged with the oneshot value} (typed-RPC workloads): the completion of a request's
// receive moves one to the other, rpc_reply and `spawn(async move)` read the first — each used to be a round trip of its own inside a divergent handler.
struct PollPrefetch { uint32_t d2lo, d2hi; uint4 cu; uint32_t rq0, rq1; };
template <class K> struct HoistRpc { static constexpr bool ON = K::G && K::FR; };
// Every-class global-state builds with plain addresses at compile time: stage [A] of a poll round requests the destination socket's header for every lane whose
// op sends in this round (k_poll.h), before the handlers.
// (The every-class builds — two waves per SIMD, registers to spare: topology +2.4 %.  The KV's channel build, at its 168-register cap, spilled two more
//  registers with them and read -4.4 %: profiles/r6_ab_stage_prefetch.txt.)
template <class K> struct HdrPrefetch { static constexpr bool ON = Hoist<K>::ALLG && !K::FA; };
// ... and stage [C]: the header and first queued message of the Endpoint a recv_from / timeout(recv_from) begins on, for the lanes of both at once.
#ifndef MADSIM_RECV_PREFETCH_TIME
#define MADSIM_RECV_PREFETCH_TIME 0          /* measured on the election loop's build: 10.62-10.69 with, 10.67-10.70 without (profiles/r6_ab_stage_prefetch.txt) */
#endif
template <class K> struct RecvPrefetch { static constexpr bool ON = Hoist<K>::ALLG || (MADSIM_RECV_PREFETCH_TIME && K::G && (K::FEAT & MADSIM_FEAT_ALL) == MADSIM_FEAT_TIME); };
// ... and the words a spawn (the free slot's old flag word, the node's info generation, the spawn counter, the gen-0 killed mask) and a finishing task
// (its JoinHandle word) read first.
#ifndef MADSIM_SWITCH_PREFETCH_CHAN
#define MADSIM_SWITCH_PREFETCH_CHAN 0        /* experiment: the channel-only global-state builds too (tools/build_variant.sh) */
#endif
template <class K> struct SwitchPrefetch { static constexpr bool ON = Hoist<K>::ALLG || (MADSIM_SWITCH_PREFETCH_CHAN && Hoist<K>::CHAN); };
__device__ __forceinline__ bool has_t0_unit(const KParams& P) { return P.task_units > 2 && !(P.uses_chan && P.chan_unit == 2); }   // geometry.h `t0`
template <class K> __device__ __forceinline__ PollPrefetch poll_prefetch(const Ctx& c, uint32_t slot) {
    PollPrefetch pp = {0, 0, make_uint4(0, 0, 0, 0), 0, 0};
    if (HoistRpc<K>::ON && c.P.uses_rpc) { const uint2 t = buf_load64(c.gs, gs_addr_task(c, slot, c.P.rpc_unit * 16u)); pp.rq0 = t.x; pp.rq1 = t.y; }
    if (K::G && K::FT && has_t0_unit(c.P)) { const uint2 t = buf_load64(c.gs, gs_addr_task(c, slot, 2 * 16u + 8u)); pp.d2lo = t.x; pp.d2hi = t.y; }
    // (the whole unit in the channel-only builds; the builds that carry every op class are short of registers: word 0 only, the
    // rest is read where it is wanted — k_poll.h cu_get)
    if (K::G && K::FC && c.P.uses_chan) {
        if (Hoist<K>::CHAN) pp.cu = gs_load128(c.gs, gs_addr_task(c, slot, c.P.chan_unit * 16u));
        else pp.cu.x = gs_load32(c.gs, gs_addr_task(c, slot, c.P.chan_unit * 16u));
    }
    return pp;
}
#define TU(c_, slot_, u_) tu_ref<K>((c_), (slot_), (u_))
#define TWORD(c_, slot_, u_, k_) tword_ref<K>((c_), (slot_), (u_), (k_))
template <class K> __device__ __forceinline__ WRef<K::G> hw_ref_plain(const Ctx& c, uint32_t p) {
    if (K::LIFE) return plane_ref<K>(c, c.hand0, p);
    return tword_ref_plain<K>(c, p, 1, 1);
}
// JoinHandle state of prog p.  Workloads with the extended ops keep a handle plane; the others park the word in the
// otherwise unused unit1.y of task slot p (max_tasks >= n_progs there, geometry.h) and save the plane's LDS.
template <class K> __device__ __forceinline__ typename TaskRef<K>::W hw_ref(const Ctx& c, uint32_t p) {
    if constexpr (K::CMP) return tword_ref<K>(c, p, 1, 1);
    else return hw_ref_plain<K>(c, p);
}
#define HW(p) hw_ref<K>(c, (p))
// unit1 write-back: only x when unit1.y is a handle word (see HW)
template <class K> __device__ __forceinline__ void tu1_store(const Ctx& c, uint32_t slot, const uint4& u1) {
    if (K::LIFE) { TU(c, slot, 1) = u1; return; }
    TWORD(c, slot, 1, 0) = u1.x;
}
// every socket-table entry is a distinct node-IP address (always so in builds without general address resolution)
#define PLAIN_ADDR (!K::FA || c.P.uniq_addr)
__device__ __forceinline__ uint4 INSN(const Ctx& c, uint32_t pc) { return LDS128(c.insn0 + pc); }
__device__ __forceinline__ uint32_t PROGW(const Ctx& c, uint32_t p) { return SMEM[c.prog0 + p]; }
__device__ __forceinline__ uint32_t SOCKW(const Ctx& c, uint32_t s) { return SMEM[c.sockt0 + s]; }
// Ephemeral Endpoints (network.rs:224-236; geometry.h device_socks): a table entry with kind bit 7 is a handle over the
// candidate entries base .. base+K-1 (ports 1 .. K of its node and IP); bits 25-31 of its header word remember which
// candidate its last bind took.  Every op but MS_OP_BIND works on that candidate.
template <class K> __device__ __forceinline__ uint32_t sock_resolve(const Ctx& c, uint32_t s) {
    const uint32_t w = SOCKW(c, s);
    return (w & 0x8000u) ? ((w >> 16) & 0xff) + (SW(c, s, 0) >> 25) : s;
}
// The handle word of entry s (a handle: SOCKW & 0x8000): candidate << 25 | valid << 24 | the candidate's socket gen after the
// handle's last bind << 16.  Does the handle still NAME that socket — is it in the table (bound) with the same gen?  The candidates
// of a (node, IP) are shared by its handles, so once the socket is gone the candidate may carry another handle's Endpoint: an
// op through the stale name would read or write a stranger's mailbox, where the oracle's entry keeps its own dead one.  No Rust
// program uses an Endpoint it never bound or has dropped, so such an op is MADSIM_UNSUPPORTED on both sides (k_poll.h insn_fetch).
template <class K> __device__ __forceinline__ bool handle_names_its_socket(const Ctx& c, uint32_t s) {
    const uint32_t hw = SW(c, s, 0), ch = SW(c, ((SOCKW(c, s) >> 16) & 0xff) + (hw >> 25), 0);
    return (hw & (1u << 24)) && (ch & 1) && ((ch >> 1) & 0xff) == ((hw >> 16) & 0xff);
}
__device__ __forceinline__ uint32_t NODET(const Ctx& c, uint32_t n) { return SMEM[c.nodet0 + n]; }   // the node's flags (MADSIM_NODE_*)

__device__ __forceinline__ uint64_t u64of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

}  // namespace madsim_k

#endif
