// sim_kernel.hip — the many-seed executor kernel for gfx950 (MI355X, CDNA4).
//
// One wavefront (64 lanes) per workgroup, ONE LANE = ONE SEED.  Each lane runs madsim's whole
// per-seed executor loop:
//     Executor::block_on / run_all_ready       madsim/src/sim/task/mod.rs:220-323
//     mpsc::Receiver::try_recv_random          madsim/src/sim/utils/mpsc.rs:73-83
//     TimeRuntime::advance_to_next_event       madsim/src/sim/time/mod.rs:45-60
//     TimeHandle::advance / Sleep::poll        time/mod.rs:103-124, time/sleep.rs:47-54
//     GlobalRng (xoshiro256++, gen_range ...)  madsim/src/sim/rand.rs:27-158
//     NetSim::send / Network::try_send / Mailbox   net/mod.rs:287-333, net/network.rs:261-313,
//                                                  net/endpoint.rs:331-362
// on a workload given as an actor program (include/madsim_hip.h).
//
// Data placement (per seed):
//   VGPRs : xoshiro256++ state (4 x u64), clock, counters, hashes, queue lengths.
//   LDS   : timer heap (16-byte entries, [slot][lane] => ds_read/write_b128, conflict-free),
//           lane stride = lw, the number of seed-carrying lanes per wave (8..64, see geometry.h),
//           task table, ready queue, mailboxes, handles, node/clog masks — all as 32-bit
//           "word planes" [word][lane] so that any per-lane dynamic index hits bank = lane % 32.
//           The workload tables (instructions, programs, socket addresses) sit once per
//           workgroup in front of the planes.
//   HBM   : heap entries beyond the LDS quota spill to a [slot][global lane] region
//           (adjacent lanes -> adjacent 16-byte entries: coalesced), results 48 B/seed.
//
// Integer/indexing work only: no MFMA.  No inter-lane communication: lanes only share the
// instruction stream, so there is no barrier anywhere in the kernel.
#ifdef MADSIM_EMU
#include "emu_shim.h"   // tests/emu: host emulation of this file, a debugging aid for GPU-less boxes
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "sim_kernel.h"

namespace madsim_k {

#ifdef EXP_ALWAYS_ACCEPT
#define EXP_ACCEPT(x) ((x) && false)   /* timing experiment only: breaks parity */
#else
#define EXP_ACCEPT(x) (x)
#endif
#define FNV_OFFSET 14695981039346656037ull
#define FNV_PRIME 1099511628211ull
#define NS_PER_S 1000000000ull
#define NS_PER_MS 1000000ull

// Task state = 16-byte units [unit][lane] (ds_read/write_b128, conflict-free):
//   unit0 {x: flags:8 | gen:16 | prog:8,  y: pc:16 | sub:8 | from:8,  z: cnt0:16 | cnt1:16,  w: val}
//   unit1 {x: rxseq:8 | joiner:8 | joiner_gen:16,  y: -,  z: deadline lo,  w: deadline hi}
//   unit2 {x: t0 lo, y: t0 hi, z/w: timeout()'s deadline}   (only when the workload uses MS_OP_MARK / timeouts)
//   unit[P.chan_unit] {x: conn:8 | side:1 | backoff ms:16, y: staged payload, z/w: arrive}   (reliable channel)
//   unit[P.rpc_unit]  {x: rsp_tag in hand, y: rsp_tag staged with the oneshot value}        (typed RPC)
enum : uint32_t { TF_ALIVE = 1, TF_SCHED = 2, TF_RUN = 4, TF_KILLED = 8, TF_CANCEL = 16, TF_INBOX = 32 };
enum : uint32_t { EV_WAKE = 1, EV_DELIVER = 2, EV_RESTART = 3 };
enum : uint32_t { H_NONE = 0, H_RUNNING = 1, H_COMPLETED = 2, H_CANCELLED = 3 };

// Compile-time kernel variant: TRACE = also emit the raw determinism log (single-seed trace mode);
// SPILL = the timer heap may overflow from LDS into the HBM spill region.
// LWS = log2(lane stride) when known at compile time (6: full 64-lane waves), or -1: read it from KParams.
// LIFE = the workload uses node lifecycle (kill/restart/pause/abort ops, init programs, restart_on_panic);
// the fast variant compiles that cold code out of the hot loop.
// RQ = the ready queue (<= 8 tasks) lives in a 64-bit register, one byte per queued task, instead of LDS.
template <bool TRACE_, bool SPILL_, int LWS_, bool LIFE_, bool RQ_ = false> struct Variant { static constexpr bool TRACE = TRACE_, SPILL = SPILL_, LIFE = LIFE_, RQ = RQ_; static constexpr int LWS = LWS_; };

// REG(id): divergence-model markers, compiled in only by tools/divergence_model.py's host emulation build
#ifndef REG
#define REG(id) do { } while (0)
#endif
#ifdef EXP_PROF2
#define PROBE2(i) do { uint64_t t_ = __builtin_readcyclecounter(); L.prof_acc[i] += t_ - L.prof_t; L.prof_t = t_; } while (0)
#else
#define PROBE2(i) do { } while (0)
#endif
#ifdef EXP_PROF
#define PROBE(i) do { uint64_t t_ = __builtin_readcyclecounter(); L.prof_acc[i] += t_ - L.prof_t; L.prof_t = t_; } while (0)
#else
#define PROBE(i) do { } while (0)
#endif

struct Lane {
    // GlobalRng
    uint64_t s0, s1, s2, s3;
    uint64_t rng_calls;
    uint64_t trace_hash;
    uint64_t log_len;
    // Clock
    uint64_t clock;
    // Timer: write-through mirror of heap[0]'s deadline (UINT64_MAX when empty)
    uint64_t top_dl;
    // accounting
    uint64_t obs_hash;
    uint32_t msg_count;
    uint32_t steps;
    uint32_t ready_len;
    uint64_t rq;         // K::RQ variants: the ready queue itself, byte i = i-th queued task slot
    uint32_t heap_len;
    uint32_t verdict;
#if defined(EXP_PROF) || defined(EXP_PROF2)
    uint64_t prof_acc[12]; uint64_t prof_t;
#endif
    uint32_t main_done;  // handle[0] left H_RUNNING: block_on's task.is_finished()
    uint32_t ovf;        // sticky: a device capacity was exceeded this iteration (=> MADSIM_OVERFLOW)
    // runtime-mutable net config (MS_OP_SET_LOSS)
    uint64_t loss_pint;
    uint32_t loss_always;
};

// All LDS traffic goes through the workgroup's one `extern __shared__` array, indexed by per-lane
// offsets held in VGPRs: the compiler then knows every access is LDS (ds_read/ds_write) — pointer
// members that may alias the HBM spill region degrade to flat_* instructions.
#ifdef MADSIM_EMU
#define SMEM emu_smem
#else
extern __shared__ __attribute__((aligned(16))) uint32_t madsim_smem[];
#define SMEM madsim_smem
#endif
#define LDS128(i) (reinterpret_cast<uint4*>(SMEM)[(i)])
#define LDS64(i) (reinterpret_cast<uint2*>(SMEM)[(i)])

struct Ctx {
    const KParams& P;
    uint32_t lws;        // log2(lane stride) (runtime copy; K::LWS overrides when >= 0)
    uint32_t ready0, hand0, node0, clog0, pause0, greg0, conn0;   // word indices of this lane's plane regions
    uint32_t sock0;      // word index of this lane's socket region
    uint32_t heap0;      // uint4 index of heap entry 0: entry i = LDS128(heap0 + (i << lws))
    uint32_t task0;      // uint4 index of task unit 0
    uint32_t insn0;      // uint4 index of the workgroup-shared instruction table
    uint32_t prog0, sockt0;   // word indices of the shared prog / socket-address tables
    uint4* spill;        // this lane's column of the HBM spill region, stride P.total_lanes
    uint8_t* tlog;       // trace mode only
    __device__ Ctx(const KParams& p) : P(p) {}
};

template <class K> __device__ __forceinline__ uint32_t LWSH(const Ctx& c) { return K::LWS >= 0 ? (uint32_t)K::LWS : c.lws; }
#define RW(i) SMEM[c.ready0 + ((i) << LWSH<K>(c))]
// JoinHandle state of prog p.  Workloads with the extended ops keep a handle plane; the others park the word in the
// otherwise unused unit1.y of task slot p (max_tasks >= n_progs there, geometry.h) and save the plane's LDS.
#define HW(p) (*hw_ref<K>(c, (p)))
#define NODEW(i) SMEM[c.node0 + ((i) << LWSH<K>(c))]
#define CLOGW(i) SMEM[c.clog0 + ((i) << LWSH<K>(c))]
#define PAUSEW(i) SMEM[c.pause0 + ((i) << LWSH<K>(c))]   /* [0] = length, [1..] = paused Runnables in pop order */
#define GREGW(i) SMEM[c.greg0 + ((i) << LWSH<K>(c))]
// connection id_: [0] alive:1 | c_ep:6<<1 | s_ep:6<<7 | tx0:1<<13 rx0<<14 tx1<<15 rx1<<16 | qn0:4<<17 | qn1:4<<21
//                 [1 + dir] parked receiver: valid:1 | slot:8<<1 | gen:16<<9;  [3 + (dir * Q + i) * 3 ..] {val, arrive lo, arrive hi}
#define CONNW(id_, f_) SMEM[c.conn0 + (((id_) * c.P.conn_words + (f_)) << LWSH<K>(c))]
// node region: [0] killed mask, [1] paused mask, [2] gen0_killed mask, [3] spawn counter, [4 + n/4] info_gen bytes
#define NODE_INFO_GEN(n_) ((NODEW(4 + ((n_) >> 2)) >> (((n_) & 3) * 8)) & 0xff)
#define SW(c_, s_, f_) SMEM[(c_).sock0 + (((s_) * (c_).P.sock_words + (f_)) << LWSH<K>(c_))]
#define TU(c_, slot_, u_) LDS128((c_).task0 + (((slot_) * (c_).P.task_units + (u_)) << LWSH<K>(c_)))
#define TWORD(c_, slot_, u_, k_) SMEM[((c_).task0 + (((slot_) * (c_).P.task_units + (u_)) << LWSH<K>(c_))) * 4 + (k_)]
template <class K> __device__ __forceinline__ uint32_t* hw_ref(const Ctx& c, uint32_t p) {
    return K::LIFE ? &SMEM[c.hand0 + (p << LWSH<K>(c))] : &TWORD(c, p, 1, 1);
}
// unit1 write-back: x and the deadline only when unit1.y is a handle word (see HW)
template <class K> __device__ __forceinline__ void tu1_store(const Ctx& c, uint32_t slot, const uint4& u1) {
    if (K::LIFE) { TU(c, slot, 1) = u1; return; }
    TWORD(c, slot, 1, 0) = u1.x;
    LDS64(((c.task0 + ((slot * c.P.task_units + 1) << LWSH<K>(c))) << 1) + 1) = make_uint2(u1.z, u1.w);
}
__device__ __forceinline__ uint4 INSN(const Ctx& c, uint32_t pc) { return LDS128(c.insn0 + pc); }
__device__ __forceinline__ uint32_t PROGW(const Ctx& c, uint32_t p) { return SMEM[c.prog0 + p]; }
__device__ __forceinline__ uint32_t SOCKW(const Ctx& c, uint32_t s) { return SMEM[c.sockt0 + s]; }

// 64-bit rotate as two v_alignbit_b32 (the compiler's shift/or expansion takes 3-4 VALU ops): K is a compile-time constant.
template <int K_>
__device__ __forceinline__ uint64_t rotl64(uint64_t x) {
#ifdef MADSIM_EMU
    return (x << K_) | (x >> (64 - K_));
#else
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (K_ >= 32) { uint32_t t = lo; lo = hi; hi = t; }            // rotate by 32 = swap halves
    constexpr int k = K_ & 31;
    if (k == 0) return ((uint64_t)hi << 32) | lo;
    uint32_t nhi = __builtin_amdgcn_alignbit(hi, lo, 32 - k);      // ({hi,lo} >> (32-k))[31:0] = hi<<k | lo>>(32-k)
    uint32_t nlo = __builtin_amdgcn_alignbit(lo, hi, 32 - k);
    return ((uint64_t)nhi << 32) | nlo;
#endif
}
__device__ __forceinline__ uint64_t u64of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// ---- GlobalRng ---------------------------------------------------------------------------------
// Xoshiro256PlusPlus::next_u64 [DEP rand_xoshiro 0.6]
__device__ __forceinline__ uint64_t rng_next(Lane& L) {
    uint64_t r = rotl64<23>(L.s0 + L.s3) + L.s0;
    uint64_t t = L.s1 << 17;
    L.s2 ^= L.s0; L.s3 ^= L.s1; L.s1 ^= L.s2; L.s0 ^= L.s3;
    L.s2 ^= t;
    L.s3 = rotl64<45>(L.s3);
    L.rng_calls++;
    return r;
}

// One determinism-log byte per GlobalRng::with (rand.rs:64-88): clone.gen::<u8>() ^ xor-fold(elapsed).
template <class K>
__device__ __forceinline__ void rng_log(const Ctx& c, Lane& L) {
#ifdef EXP_NOLOG
    return;
#endif
    uint64_t r = rotl64<23>(L.s0 + L.s3) + L.s0;   // what the clone's next_u64 would return
    uint32_t v = (uint32_t)(r >> 32);
    uint32_t f = (uint32_t)L.clock ^ (uint32_t)(L.clock >> 32);
    f ^= f >> 16; f ^= f >> 8;
    v = (v ^ f) & 0xff;
    L.trace_hash = (L.trace_hash ^ v) * FNV_PRIME;
    if (K::TRACE) { if (L.log_len < c.P.trace_cap) c.tlog[L.log_len] = (uint8_t)v; }
    L.log_len++;
}

// gen_range(lo..hi) on u64 [DEP rand 0.8 UniformInt::sample_single_inclusive]; one with() per call.
template <class K>
__device__ __forceinline__ uint64_t gen_range_u64(const Ctx& c, Lane& L, uint64_t lo, uint64_t range) {
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    uint64_t v;
    do { REG(16); v = rng_next(L); } while (v * range > zone);
    rng_log<K>(c, L);
    return lo + __umul64hi(v, range);
}

// ready-queue index draw: range = len <= 255, so the 128-bit product splits into two 32x32 pieces.
template <class K>
__device__ __forceinline__ uint32_t gen_index(const Ctx& c, Lane& L, uint32_t len) {
    uint64_t zone = ((uint64_t)len << __builtin_clzll((uint64_t)len)) - 1;
    uint64_t v;
    do { REG(1); v = rng_next(L); } while (EXP_ACCEPT(v * (uint64_t)len > zone));   // accept test on the low 64 bits only
    rng_log<K>(c, L);
    uint64_t mid = (uint64_t)(uint32_t)(v >> 32) * len + (((uint64_t)(uint32_t)v * len) >> 32);
    return (uint32_t)(mid >> 32);                               // high 64 bits of v * len
}

// gen_range with a compile-time range < 2^32.
template <class K, uint32_t RANGE>
__device__ __forceinline__ uint32_t gen_range_small(const Ctx& c, Lane& L) {
    constexpr uint64_t zone = ((uint64_t)RANGE << __builtin_clzll((uint64_t)RANGE)) - 1;
    uint64_t v;
    do { REG(RANGE == 50 ? 18 : 16); v = rng_next(L); } while (EXP_ACCEPT(v * (uint64_t)RANGE > zone));
    rng_log<K>(c, L);
    uint64_t mid = (uint64_t)(uint32_t)(v >> 32) * RANGE + (((uint64_t)(uint32_t)v * RANGE) >> 32);
    return (uint32_t)(mid >> 32);
}

// gen_bool through GlobalRng's RngCore impl (rand.rs:142-158): one with() per draw [DEP Bernoulli].
template <class K>
__device__ __forceinline__ bool gen_bool_pint(const Ctx& c, Lane& L, uint64_t p_int, uint32_t always) {
    if (always) return true;
    REG(7);
    uint64_t v = rng_next(L);
    rng_log<K>(c, L);
    return v < p_int;
}

// UniformDuration sample on the GlobalRng itself (network.rs:267): one with() per attempt [DEP A.3].
template <class K>
__device__ __forceinline__ uint64_t sample_latency(const Ctx& c, Lane& L) {
    const KParams& P = c.P;
    uint64_t res;
    for (;;) {
        REG(8);
        uint64_t v = rng_next(L);
        rng_log<K>(c, L);
        if (P.lat_mode == 0) {
            uint64_t m = (uint64_t)(uint32_t)(v >> 32) * (uint64_t)(uint32_t)P.lat_range;
            if ((uint32_t)m <= (uint32_t)P.lat_zone) { res = P.lat_low + (m >> 32); break; }
        } else {
            uint64_t mlo = v * P.lat_range;
            if (mlo <= P.lat_zone) { res = P.lat_low + __umul64hi(v, P.lat_range); break; }
        }
    }
    return res;
}

// ---- Timer = BinaryHeap<Event>, reversed Ord on deadline [DEP naive-timer 0.2 + alloc BinaryHeap] --
// entry: x = deadline lo, y = deadline hi, z = meta, w = payload value
__device__ __forceinline__ uint64_t ev_deadline(const uint4& e) { return u64of(e.x, e.y); }

// Entries [0, heap_lds) live in LDS; entries beyond spill to HBM as [slot][global lane] (coalesced
// across the wave).  Variants without a spill region drop the HBM path.  The LDS load is issued
// unconditionally (clamped index) and the HBM value selected afterwards, so the two address spaces
// never merge into a flat_* access.
template <class K>
__device__ __forceinline__ uint4 heap_get(const Ctx& c, uint32_t i) {
    if (!K::SPILL) return LDS128(c.heap0 + (i << LWSH<K>(c)));
    uint32_t cap = c.P.heap_lds;
    uint4 v = LDS128(c.heap0 + ((i < cap ? i : cap - 1) << LWSH<K>(c)));
    if (i >= cap) v = c.spill[(size_t)(i - cap) * c.P.total_lanes];
    return v;
}
template <class K>
__device__ __forceinline__ void heap_set(const Ctx& c, uint32_t i, const uint4& e) {
    if (!K::SPILL || i < c.P.heap_lds) LDS128(c.heap0 + (i << LWSH<K>(c))) = e;
    else c.spill[(size_t)(i - c.P.heap_lds) * c.P.total_lanes] = e;
}

// BinaryHeap::sift_up(0, pos) with `hole` as the moving element; keeps the root mirror current.
template <class K>
__device__ __forceinline__ void heap_sift_up(const Ctx& c, Lane& L, uint32_t pos, const uint4& hole) {
    uint64_t hd = ev_deadline(hole);
    while (pos > 0) {
        REG(11);
        uint32_t parent = (pos - 1) >> 1;
        if (parent == 0 && hd >= L.top_dl) break;      // root deadline is mirrored in a register
        uint4 p = heap_get<K>(c, parent);
        if (hd >= ev_deadline(p)) break;     // hole <= parent in heap order: stop
        heap_set<K>(c, pos, p);
        pos = parent;
    }
    heap_set<K>(c, pos, hole);
    if (pos == 0) L.top_dl = hd;
}

// Timer::add -> BinaryHeap::push.  Returns false on capacity overflow.
template <class K>
__device__ __forceinline__ bool timer_add(const Ctx& c, Lane& L, uint64_t deadline, uint32_t meta, uint32_t val) {
    PROBE2(0);
    REG(10);
    if (L.heap_len >= c.P.heap_lds + (K::SPILL ? c.P.heap_spill : 0u)) return false;
    uint4 e = make_uint4((uint32_t)deadline, (uint32_t)(deadline >> 32), meta, val);
    heap_sift_up<K>(c, L, L.heap_len, e);
    L.heap_len++;
    PROBE2(10);
    return true;
}

// BinaryHeap::pop: swap the last element into the root, sift_down_to_bottom(0), then sift_up.
template <class K>
__device__ __forceinline__ uint4 timer_pop(const Ctx& c, Lane& L) {
    PROBE2(0);
    REG(20);
    uint32_t end = --L.heap_len;
    uint4 item = heap_get<K>(c, end);
    if (end > 0) {
        uint4 top = heap_get<K>(c, 0);
        uint32_t pos = 0, child = 1;
        while (child + 1 < end) {
            REG(21);
            uint4 l = heap_get<K>(c, child), r = heap_get<K>(c, child + 1);
            bool right = ev_deadline(l) >= ev_deadline(r);   // left <= right in heap order: take right
            uint4 m = right ? r : l;
            heap_set<K>(c, pos, m);
            if (pos == 0) L.top_dl = ev_deadline(m);
            pos = child + (right ? 1u : 0u);
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            uint4 m = heap_get<K>(c, child);
            heap_set<K>(c, pos, m);
            if (pos == 0) L.top_dl = ev_deadline(m);
            pos = child;
        }
        heap_sift_up<K>(c, L, pos, item);
        item = top;
    } else {
        L.top_dl = ~0ull;
    }
    PROBE2(11);
    return item;
}

// ---- async-task wake / schedule [DEP A.7] -------------------------------------------------------
template <class K>
__device__ __forceinline__ void ready_push(const Ctx& c, Lane& L, uint32_t slot) {
    if (K::RQ) L.rq |= (uint64_t)slot << (8 * L.ready_len);
    else RW(L.ready_len) = slot;
    L.ready_len++;
}

template <class K>
__device__ __forceinline__ void wake(const Ctx& c, Lane& L, uint32_t slot, uint32_t gen) {
    uint32_t f = TWORD(c, slot, 0, 0);
    if (!(f & TF_ALIVE) || ((f >> 8) & 0xffff) != gen) return;   // COMPLETED | CLOSED
    if (f & TF_SCHED) return;
    TWORD(c, slot, 0, 0) = f | TF_SCHED;
    if (!(f & TF_RUN)) ready_push<K>(c, L, slot);                   // RUNNING: run() re-queues after the poll
}

// ---- Network -----------------------------------------------------------------------------------
// Network::try_send's socket lookup (network.rs:304-306): the bound socket at addr(dst), if any.
template <class K>
__device__ __forceinline__ int find_bound(const Ctx& c, uint32_t addr) {
    if (c.P.uniq_addr) return (SW(c, addr, 0) & 1) ? (int)addr : -1;
    uint32_t key = SOCKW(c, addr) & 0xffff00ffu;
    for (uint32_t i = 0; i < c.P.n_socks; i++)
        if ((SOCKW(c, i) & 0xffff00ffu) == key && (SW(c, i, 0) & 1)) return (int)i;
    return -1;
}

// Mailbox::deliver (endpoint.rs:331-351)
template <class K>
__device__ __forceinline__ void mailbox_deliver(const Ctx& c, Lane& L, uint32_t meta, uint32_t val) {
    uint32_t s = meta & 0x3f, from = (meta >> 6) & 0x3f, tag = (meta >> 12) & 0xff, sgen = (meta >> 20) & 0xff;
    uint32_t h = SW(c, s, 0);
    if (!(h & 1) || ((h >> 1) & 0xff) != sgen) return;     // that Endpoint object is gone
    uint32_t nreg = (h >> 9) & 0xff, nmsg = (h >> 17) & 0xff;
    // Typed RPC (net/rpc.rs): a response (tag 0xff) is addressed to one pending receive — the word of its registration,
    // tag | slot | rxseq | gen, rides in the payload's upper 24 bits — where the reference matches a random u64 tag.
    const bool rpc = K::LIFE && c.P.uses_rpc;
    const bool rsp = rpc && tag == 0xff;
    uint32_t i = 0;
    while (i < nreg) {
        REG(24);
        uint32_t r = SW(c, s, 2 + i);
        if ((r & 0xff) == tag && (!rsp || (r >> 8) == (val >> 8))) {
            nreg--;
            SW(c, s, 2 + i) = SW(c, s, 2 + nreg);          // swap_remove
            uint32_t slot = (r >> 8) & 0xff, rxseq = (r >> 16) & 0xff, g8 = r >> 24;
            uint4 u0 = TU(c, slot, 0);
            uint32_t link = TWORD(c, slot, 1, 0);
            if ((u0.x & TF_ALIVE) && ((u0.x >> 8) & 0xff) == g8 && (link & 0xff) == rxseq && !(u0.x & TF_INBOX)) {
                // oneshot::Sender::send Ok -> value stored, receiver task woken [DEP tokio oneshot]
                bool sched = u0.x & TF_SCHED;
                u0.x |= TF_INBOX | TF_SCHED;
                u0.y = (u0.y & 0x00ffffffu) | (from << 24);
                u0.w = val;
                if (rpc && tag >= MADSIM_TAG_RPC_FIRST) {                  // 8-bit code; a request also carries its rsp_tag
                    u0.w = val & 0xff;
                    if (!rsp) TWORD(c, slot, c.P.rpc_unit, 1) = val >> 8;  // staged with the oneshot value
                }
                TU(c, slot, 0) = u0;
                SW(c, s, 0) = (h & ~(0xffu << 9)) | (nreg << 9);
                if (!sched && !(u0.x & TF_RUN)) ready_push<K>(c, L, slot);
                return;
            }
        } else {
            i++;
        }
    }
    if (nmsg >= c.P.mbox_msgs) { L.ovf = 1; return; }
    if (rsp) tag = 0xfe;                                   // nobody holds that rsp_tag any more: it can never be received
    SW(c, s, 2 + c.P.mbox_regs + 2 * nmsg) = tag | (from << 8);
    SW(c, s, 2 + c.P.mbox_regs + 2 * nmsg + 1) = val;
    nmsg++;
    SW(c, s, 0) = (h & ~((0xffu << 9) | (0xffu << 17))) | (nreg << 9) | (nmsg << 17);
}

template <class K> __device__ void node_restart(const Ctx& c, Lane& L, uint32_t node);

// Timer::expire [DEP A.5]: fire every entry with deadline <= now
template <class K>
__device__ __forceinline__ void timer_expire(const Ctx& c, Lane& L, uint64_t now) {
    while (L.top_dl <= now) {
        uint4 e = timer_pop<K>(c, L);
        L.steps++;
        uint32_t kind = e.z >> 28;
        if (kind == EV_WAKE) { REG(22); wake<K>(c, L, e.z & 0xff, (e.z >> 8) & 0xffff); }   // time/sleep.rs:52
        else if (kind == EV_DELIVER) { REG(23); mailbox_deliver<K>(c, L, e.z, e.w); }      // net/mod.rs:323-330
        else if (K::LIFE && kind == EV_RESTART) node_restart<K>(c, L, e.z & 0xff);   // task/mod.rs:313
    }
}

// ---- task lifecycle ----------------------------------------------------------------------------
// `via_handle`: spawned through a NodeHandle captured at build() (NodeHandle::spawn from another node's task:
// that Spawner holds the ORIGINAL Arc<NodeInfo>); otherwise task::spawn / init on the node's current info.
template <class K>
__device__ __forceinline__ uint32_t spawn_task(const Ctx& c, Lane& L, uint32_t prog, bool record, bool via_handle = false) {
    uint32_t slot = 0;
    while (slot < c.P.max_tasks && (TWORD(c, slot, 0, 0) & TF_ALIVE)) slot++;
    if (slot >= c.P.max_tasks) { L.ovf = 1; return 0xffffffffu; }
    uint32_t gen = (((TWORD(c, slot, 0, 0) >> 8) & 0xffff) + 1) & 0xffff;
    uint32_t pw = PROGW(c, prog);
    uint32_t node = pw & 0xff;
    uint32_t killed = 0, info_gen = 0;
    if (K::LIFE) {
        uint32_t cur_gen = NODE_INFO_GEN(node);
        info_gen = cur_gen;
        if (via_handle && cur_gen != 0) { killed = 1; info_gen = 0; }             // stale handle: dead info
        else killed = ((via_handle ? NODEW(2) : NODEW(0)) >> node) & 1;          // task/mod.rs:632-634
    }
    uint32_t seq = 0;
    if (K::LIFE) { seq = NODEW(3); NODEW(3) = seq + 1; }      // spawn order matters only to NodeInfo::kill
    TU(c, slot, 0) = make_uint4(TF_ALIVE | TF_SCHED | (killed ? TF_KILLED : 0) | (gen << 8) | (prog << 24), pw >> 16, 0, 0);
    tu1_store<K>(c, slot, make_uint4(0xffu << 8, (seq & 0xffffff) | (info_gen << 24), 0, 0));   // rxseq 0, no awaiter; spawn order
    if (K::LIFE && c.P.uses_chan) TU(c, slot, c.P.chan_unit) = make_uint4(0xff, 0, 0, 0);                // no connection held
    if (K::LIFE && c.P.uses_rpc) TU(c, slot, c.P.rpc_unit) = make_uint4(0, 0, 0, 0);         // no request in hand
    ready_push<K>(c, L, slot);
    if (record) HW(prog) = H_RUNNING | (slot << 8) | (gen << 16);
    return slot;
}

template <class K> __device__ void conn_drop_handles(const Ctx& c, Lane& L, uint32_t id, uint32_t side);
template <class K> __device__ void sock_drop_acceptq(const Ctx& c, Lane& L, uint32_t s);

// The future is gone: completed (outcome H_COMPLETED) or dropped by the executor (H_CANCELLED).
template <class K>
__device__ __forceinline__ void task_finish(const Ctx& c, Lane& L, uint32_t slot, uint32_t outcome) {
    uint32_t f = TWORD(c, slot, 0, 0);
    uint32_t gen = (f >> 8) & 0xffff, prog = f >> 24;
    if (K::LIFE && c.P.uses_chan) {                          // the task's (Sender, Receiver) pair drops with its future
        uint32_t cx = TWORD(c, slot, c.P.chan_unit, 0);
        if ((cx & 0xff) != 0xff) { conn_drop_handles<K>(c, L, cx & 0xff, (cx >> 8) & 1); TWORD(c, slot, c.P.chan_unit, 0) = cx | 0xff; }
    }
    {
        uint32_t own = slot | (gen << 16);
        for (uint32_t i = 0; i < c.P.n_socks; i++) {
            if (SW(c, i, 1) != own) continue;
            // BindGuard::drop (net/mod.rs:483-493), skipped when the binder's NodeInfo is killed
            if (!(f & TF_KILLED) && (SW(c, i, 0) & 1)) SW(c, i, 0) &= ~1u;
            if (K::LIFE && c.P.uses_chan && (SW(c, i, 2 + c.P.mbox_regs + 2 * c.P.mbox_msgs) & 0xf)) sock_drop_acceptq<K>(c, L, i);
        }
    }
    uint32_t h = HW(prog);
    if (h == (H_RUNNING | (slot << 8) | (gen << 16))) { HW(prog) = (h & ~3u) | outcome; if (prog == 0) L.main_done = 1; }
    uint32_t link = TWORD(c, slot, 1, 0);
    TWORD(c, slot, 0, 0) = f & ~(TF_ALIVE | TF_SCHED | TF_RUN | TF_INBOX);
    uint32_t j = (link >> 8) & 0xff;
    if (j != 0xff) wake<K>(c, L, j, link >> 16);              // async-task notifies the awaiter
}

// NodeInfo::kill (task/mod.rs:133-140): mark + wake every live task holding NodeInfo `info_gen` of `node`, in
// spawn order (the order of NodeInfo.tasks).  Tasks carry their spawn sequence number, so no list is stored.
template <class K>
__device__ void info_kill(const Ctx& c, Lane& L, uint32_t node, uint32_t info_gen) {
    uint32_t last = 0xffffffffu;                            // "none yet": sequence numbers are < 2^24
    for (;;) {
        uint32_t best = 0xffffffffu, best_seq = 0xffffffffu;
        for (uint32_t t = 0; t < c.P.max_tasks; t++) {
            uint32_t f = TWORD(c, t, 0, 0);
            if (!(f & TF_ALIVE) || (PROGW(c, f >> 24) & 0xff) != node) continue;
            uint32_t sw = TWORD(c, t, 1, 1);
            uint32_t seq = sw & 0xffffff;
            if ((sw >> 24) != info_gen) continue;
            if ((last == 0xffffffffu || seq > last) && seq < best_seq) { best = t; best_seq = seq; }
        }
        if (best == 0xffffffffu) break;
        uint32_t f = TWORD(c, best, 0, 0);
        TWORD(c, best, 0, 0) = f | TF_KILLED;
        wake<K>(c, L, best, (f >> 8) & 0xffff);
        last = best_seq;
    }
}

template <class K>
__device__ __forceinline__ void task_finish(const Ctx& c, Lane& L, uint32_t slot, uint32_t outcome);

// node.paused.clear() (task/mod.rs:365,392): the parked Runnables of `node` are dropped, in order.
template <class K>
__device__ void paused_clear(const Ctx& c, Lane& L, uint32_t node) {
    if (!c.P.uses_pause) return;
    uint32_t n = PAUSEW(0), keep = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t slot = PAUSEW(1 + i);
        if ((PROGW(c, TWORD(c, slot, 0, 0) >> 24) & 0xff) == node) task_finish<K>(c, L, slot, H_CANCELLED);
        else { PAUSEW(1 + keep) = slot; keep++; }
    }
    PAUSEW(0) = keep;
}

template <class K>
__device__ void node_kill(const Ctx& c, Lane& L, uint32_t node) {        // TaskHandle::kill_id (task/mod.rs:362-371)
    paused_clear<K>(c, L, node);
    uint32_t g = NODE_INFO_GEN(node);
    NODEW(0) |= 1u << node;
    if (g == 0) NODEW(2) |= 1u << node;
    info_kill<K>(c, L, node, g);
    for (uint32_t i = 0; i < c.P.n_socks; i++)               // NetSim::reset_node (network.rs:142-147)
        if ((SOCKW(c, i) & 0xff) == node) SW(c, i, 0) &= ~1u;
}

template <class K>
__device__ void node_restart(const Ctx& c, Lane& L, uint32_t node) {     // TaskHandle::restart (task/mod.rs:374-401)
    uint32_t g = NODE_INFO_GEN(node);
    if (g == 0) NODEW(2) |= 1u << node;
    uint32_t w = NODEW(4 + (node >> 2)), sh = (node & 3) * 8;
    NODEW(4 + (node >> 2)) = (w & ~(0xffu << sh)) | (((g + 1) & 0xff) << sh);      // new_info
    NODEW(0) &= ~(1u << node);
    NODEW(1) &= ~(1u << node);
    paused_clear<K>(c, L, node);
    info_kill<K>(c, L, node, g);                             // old_info.kill()
    for (uint32_t p = 1; p < c.P.n_progs; p++) {             // init(&Spawner { new info })
        uint32_t pw = PROGW(c, p);
        if ((pw & 0xff) == node && ((pw >> 8) & MADSIM_PROG_INIT)) spawn_task<K>(c, L, p, false);
    }
}

// ---- reliable channel (NetSim::connect1 / channel, net/mod.rs:337-430) — LIFE variants only ---------------------
// Network::try_send as a function (the datagram path has it inlined in poll_task's [A] stage).
template <class K>
__device__ bool try_send_fn(const Ctx& c, Lane& L, uint32_t src_node, uint32_t dst_addr, uint64_t* latency, int* dst_sock) {
    const KParams& P = c.P;
    uint32_t dst_node = SOCKW(c, dst_addr) & 0xff;
    bool clogged = false;
    if (P.has_clog) clogged = ((CLOGW(1) >> src_node) & 1) | ((CLOGW(0) >> dst_node) & 1);
    if (P.has_clog_link) clogged |= (CLOGW(2 + src_node) >> dst_node) & 1;
    if (clogged) return false;
    if (gen_bool_pint<K>(c, L, L.loss_pint, L.loss_always)) return false;
    L.msg_count++;
    *latency = sample_latency<K>(c, L);
    int ds = find_bound<K>(c, dst_addr);
    if (ds < 0) return false;
    *dst_sock = ds;
    return true;
}

// the `test_link` closure of channel() (net/mod.rs:375-380): Some(now + latency) or None
template <class K>
__device__ uint64_t chan_test_link(const Ctx& c, Lane& L, uint32_t cw, uint32_t dir) {
    uint32_t c_ep = (cw >> 1) & 0x3f, s_ep = (cw >> 7) & 0x3f;
    uint64_t lat; int ds;
    if (!try_send_fn<K>(c, L, SOCKW(c, dir == 0 ? c_ep : s_ep) & 0xff, dir == 0 ? s_ep : c_ep, &lat, &ds)) return ~0ull;
    return L.clock + lat;
}

// drop the (Sender, Receiver) pair of one end of connection `id`
template <class K>
__device__ void conn_drop_handles(const Ctx& c, Lane& L, uint32_t id, uint32_t side) {
    uint32_t cw = CONNW(id, 0);
    if (cw & (1u << (13 + 2 * side))) {                       // my PayloadSender: last mpsc sender gone
        cw &= ~(1u << (13 + 2 * side));
        uint32_t r = CONNW(id, 1 + side);
        if (r & 1) { CONNW(id, 1 + side) = 0; CONNW(id, 0) = cw; wake<K>(c, L, (r >> 1) & 0xff, r >> 9); }   // parked receiver sees None
    }
    cw &= ~(1u << (14 + 2 * (1 - side)));                     // my PayloadReceiver
    CONNW(id, 1 + (1 - side)) = 0;
    if (!(cw & (0xfu << 13))) cw = 0;                          // all four handles gone: slot is free
    CONNW(id, 0) = cw;
}

// the listening Endpoint is dropped: connections still queued in conn_rx go with it
template <class K>
__device__ void sock_drop_acceptq(const Ctx& c, Lane& L, uint32_t s) {
    uint32_t base = 2 + c.P.mbox_regs + 2 * c.P.mbox_msgs;
    uint32_t q = SW(c, s, base);
    SW(c, s, base) = 0; SW(c, s, base + 1) = 0;
    uint32_t n = q & 0xf;
    for (uint32_t i = 0; i < n; i++) conn_drop_handles<K>(c, L, (q >> (4 + 7 * i)) & 0x7f, 1);
}

// TimeHandle::sleep_until (time/mod.rs:118-124): 1 ms floor
__device__ __forceinline__ uint64_t sleep_deadline(const Lane& L, uint64_t deadline) {
    uint64_t m = L.clock + NS_PER_MS;
    return deadline > m ? deadline : m;
}

// NetSim::rand_delay up to the creation of its Sleep (net/mod.rs:287-292): returns the Sleep's deadline.
template <class K>
__device__ __forceinline__ uint64_t rand_delay_deadline(const Ctx& c, Lane& L) {
    uint64_t delay = (uint64_t)gen_range_small<K, 5>(c, L) * 1000ull;
    if (c.P.buggify) {
        if (gen_bool_pint<K>(c, L, c.P.bug_pint, 0)) delay = (uint64_t)(1 + gen_range_small<K, 4>(c, L)) * NS_PER_S;
    }
    return sleep_deadline(L, L.clock + delay);
}

__device__ __forceinline__ bool is_light(uint32_t op) {
    return op == MS_OP_ASSERT_VAL || op == MS_OP_DJNZ || op == MS_OP_SET || op == MS_OP_JMP || op == MS_OP_TRACE || op == MS_OP_JEQ;
}

// One poll of a task's future (Runnable::run, task/mod.rs:279-283).  `u0` is the task's unit0, held
// in registers for the whole poll and written back by the caller.  Returns true if the task panicked.
//
// A poll is a sequence of rounds; one round = [A] resolve the await the task is parked on and run its
// completion action, [B] run the cheap straight-line ops that follow, [C] begin the next awaiting op.
// In steady state every poll is exactly one round, and all lanes walk A -> B -> C together, so the
// expensive primitives (RNG draws, heap pushes, link test, mailbox scan) sit at fixed points that the
// whole wave reaches at the same time.
template <class K>
__device__ __forceinline__ bool poll_task(const Ctx& c, Lane& L, const uint32_t slot, uint4& u0, uint4 u1) {
    enum : uint32_t { ST_RUN = 0, ST_PENDING = 1, ST_FINISHED = 2, ST_PANIC = 3 };
    const KParams& P = c.P;
    bool u1_dirty = false;
    uint32_t pc = u0.y & 0xffff, sub = (u0.y >> 16) & 0xff, from = u0.y >> 24;
    const uint32_t gen = (u0.x >> 8) & 0xffff;
    const uint32_t node = PROGW(c, u0.x >> 24) & 0xff;
    uint32_t st = ST_RUN;

    // timeout(d, ep.recv_from(tag)) = select_biased! { fut, sleep } (time/mod.rs:128-140): poll the recv future,
    // then the timeout's Sleep — which registers ANOTHER timer on every not-elapsed poll (time/sleep.rs:51-53).
    // Returns true when the op completed (Ok or Err(Elapsed)); otherwise the task is Pending.
    auto recv_timeout_poll = [&]() -> bool {
        bool fut_ready = false;
        if (sub == 1 && (u0.x & TF_INBOX)) {                 // oneshot ready -> rand_delay (endpoint.rs:145)
            u0.x &= ~TF_INBOX;
            from = u0.y >> 24;
            if (P.uses_rpc && (INSN(c, pc).x >> 24) >= MADSIM_TAG_RPC_FIRST) TWORD(c, slot, P.rpc_unit, 0) = TWORD(c, slot, P.rpc_unit, 1);
            uint64_t d1 = rand_delay_deadline<K>(c, L);
            u1.z = (uint32_t)d1; u1.w = (uint32_t)(d1 >> 32); u1_dirty = true;
            sub = 2;
        }
        if (sub == 2) {
            uint64_t d1 = u64of(u1.z, u1.w);
            if (L.clock >= d1) fut_ready = true;
            else if (!timer_add<K>(c, L, d1, (EV_WAKE << 28) | (gen << 8) | slot, 0)) L.ovf = 1;
        }
        if (fut_ready) return true;                          // Ok((len, from))
        uint4 u2 = TU(c, slot, 2);
        uint64_t d2 = u64of(u2.z, u2.w);
        if (L.clock >= d2) {                                 // Err(Elapsed): the recv future is dropped
            u1.x = (u1.x & ~0xffu) | (((u1.x & 0xff) + 1) & 0xff); u1_dirty = true;   // its oneshot::Receiver is gone
            u0.x &= ~TF_INBOX;
            u0.w = MADSIM_VAL_TIMEOUT;
            return true;
        }
        if (!timer_add<K>(c, L, d2, (EV_WAKE << 28) | (gen << 8) | slot, 0)) L.ovf = 1;
        st = ST_PENDING;
        return false;
    };

    // Endpoint::call / call_timeout (net/rpc.rs:96-131) from its first poll on.  sub 1: send_to_raw's rand_delay;
    // sub 2: recv_from_raw(rsp_tag)'s oneshot; sub 3: its rand_delay.  With a timeout, the timeout's Sleep is polled after
    // the call future on every poll and registers ANOTHER timer each time (select_biased!, time/sleep.rs:51-53).
    // Returns true when the op completed (Ok, Err(TimedOut)) or the task panicked (st).
    auto rpc_call_poll = [&]() -> bool {
        const uint4 ci = INSN(c, pc);
        const uint32_t ca = (ci.x >> 8) & 0xff, cb = ci.x >> 16, cimm = ci.y;
        const uint32_t dst = cb & 0xff;
        if (sub == 1) {
            uint64_t d1 = u64of(u1.z, u1.w);
            if (L.clock < d1) {
                if (!timer_add<K>(c, L, d1, (EV_WAKE << 28) | (gen << 8) | slot, 0)) L.ovf = 1;
            } else {
                // the caller's pending receive doubles as the rsp_tag: registration word >> 8 (see mailbox_deliver)
                const uint32_t rxseq = ((u1.x & 0xff) + 1) & 0xff;
                const uint32_t reg = 0xffu | (slot << 8) | (rxseq << 16) | ((gen & 0xff) << 24);
                uint64_t lat; int ds;
                if (try_send_fn<K>(c, L, SOCKW(c, ca) & 0xff, dst, &lat, &ds)) {
                    uint32_t sgen = (SW(c, ds, 0) >> 1) & 0xff;
                    uint32_t meta = (EV_DELIVER << 28) | (sgen << 20) | ((cb >> 8) << 12) | (ca << 6) | (uint32_t)ds;
                    if (!timer_add<K>(c, L, L.clock + lat, meta, (cimm & 0xff) | (reg & 0xffffff00u))) L.ovf = 1;
                }
                // recv_from_raw(rsp_tag): Mailbox::recv (endpoint.rs:353-362); no queued message can carry a fresh tag
                u1.x = (u1.x & ~0xffu) | rxseq; u1_dirty = true;
                u0.x &= ~TF_INBOX;
                uint32_t h = SW(c, ca, 0);
                uint32_t nreg = (h >> 9) & 0xff;
                for (uint32_t i = 0; i < nreg; i++) if (SW(c, ca, 2 + i) == reg) L.ovf = 1;   // 8-bit rxseq wrapped onto a dead twin
                if (nreg >= P.mbox_regs) L.ovf = 1;
                else {
                    SW(c, ca, 2 + nreg) = reg;
                    SW(c, ca, 0) = (h & ~(0xffu << 9)) | ((nreg + 1) << 9);
                }
                sub = 2;
            }
        }
        if (sub == 2 && (u0.x & TF_INBOX)) {                 // oneshot ready -> rand_delay (endpoint.rs:145)
            u0.x &= ~TF_INBOX;
            from = u0.y >> 24;
            uint64_t d1 = rand_delay_deadline<K>(c, L);
            u1.z = (uint32_t)d1; u1.w = (uint32_t)(d1 >> 32); u1_dirty = true;
            sub = 3;
        }
        if (sub == 3) {
            uint64_t d1 = u64of(u1.z, u1.w);
            if (L.clock >= d1) {
                if (from != dst) st = ST_PANIC;              // assert_eq!(from, dst) rpc.rs:126
                return true;
            }
            if (!timer_add<K>(c, L, d1, (EV_WAKE << 28) | (gen << 8) | slot, 0)) L.ovf = 1;
        }
        if (cimm >> 8) {
            uint4 u2 = TU(c, slot, 2);
            uint64_t d2 = u64of(u2.z, u2.w);
            if (L.clock >= d2) {                             // Err(Elapsed) -> TimedOut: the call future is dropped
                if (sub >= 2) { u1.x = (u1.x & ~0xffu) | (((u1.x & 0xff) + 1) & 0xff); u1_dirty = true; u0.x &= ~TF_INBOX; }
                u0.w = MADSIM_VAL_TIMEOUT;
                return true;
            }
            if (!timer_add<K>(c, L, d2, (EV_WAKE << 28) | (gen << 8) | slot, 0)) L.ovf = 1;
        }
        st = ST_PENDING;
        return false;
    };

    // accept1's conn_rx.recv() (endpoint.rs:200): take the oldest queued connection or park. true = op completed.
    auto accept_check = [&](uint32_t a) -> bool {
        uint32_t base = 2 + P.mbox_regs + 2 * P.mbox_msgs;
        uint32_t q = SW(c, a, base);
        uint32_t n = q & 0xf;
        if (n == 0) { SW(c, a, base + 1) = 1u | (slot << 1) | (gen << 9); st = ST_PENDING; return false; }
        uint32_t id = (q >> 4) & 0x7f;
        SW(c, a, base) = (n - 1) | ((q >> 11) << 4);           // pop front
        uint32_t cx = TWORD(c, slot, c.P.chan_unit, 0);
        if ((cx & 0xff) != 0xff) conn_drop_handles<K>(c, L, cx & 0xff, (cx >> 8) & 1);
        TWORD(c, slot, c.P.chan_unit, 0) = id | (1u << 8);                 // server side
        return true;
    };
    // the receiver stream of channel() (net/mod.rs:386-400) from "a payload is in hand" (sub 1): either sleep(backoff)
    // while its State is None (sub 2) or sleep_until(arrive_time) (sub 3).  Always ends Pending (1 ms floor).
    auto crecv_arm = [&]() {
        uint4 u3 = TU(c, slot, c.P.chan_unit);
        uint64_t arrive = u64of(u3.z, u3.w), d;
        if (arrive != ~0ull) { d = sleep_deadline(L, arrive); sub = 3; }
        else { d = sleep_deadline(L, L.clock + (uint64_t)(u3.x >> 16) * NS_PER_MS); sub = 2; }
        u1.z = (uint32_t)d; u1.w = (uint32_t)(d >> 32); u1_dirty = true;
        if (!timer_add<K>(c, L, d, (EV_WAKE << 28) | (gen << 8) | slot, 0)) L.ovf = 1;
        st = ST_PENDING;
    };

    while (st == ST_RUN) {
        if (pc >= P.n_insns) { st = ST_PANIC; break; }
        uint4 in = INSN(c, pc);
        uint32_t op = in.x & 0xff, a = (in.x >> 8) & 0xff, b = in.x >> 16, imm = in.y;

        PROBE(5);
        REG(2);
        // ================= [A] the task is parked on an await of this op =========================
        if (sub != 0) {
            bool completed = false;                        // this op is done: step to the next one below
            if (op == MS_OP_RECV && sub == 1) {            // oneshot::Receiver (endpoint.rs:142-144)
                REG(4);
                if (!(u0.x & TF_INBOX)) { st = ST_PENDING; break; }
                u0.x &= ~TF_INBOX;
                from = u0.y >> 24;
                if (K::LIFE && P.uses_rpc && (b >> 8) >= MADSIM_TAG_RPC_FIRST)   // (rsp_tag, req, data) = *data.downcast()
                    TWORD(c, slot, P.rpc_unit, 0) = TWORD(c, slot, P.rpc_unit, 1);
                sub = 2;                                   // -> rand_delay, begun in [C]
            } else if (op == MS_OP_YIELD) {
                completed = true;
            } else if (K::LIFE && op == MS_OP_RECV_TIMEOUT) {
                completed = recv_timeout_poll();
                if (!completed) break;
            } else if (K::LIFE && op == MS_OP_RPC_CALL) {
                completed = rpc_call_poll();
                if (!completed || st == ST_PANIC) break;
            } else if (K::LIFE && op == MS_OP_ACCEPT && sub == 2) {
                completed = accept_check(a);
                if (!completed) break;
            } else {                                       // a Sleep (time/sleep.rs:47-54)
                uint64_t deadline = u64of(u1.z, u1.w);
                REG(3);
                if (L.clock < deadline) {                  // not elapsed: register ANOTHER timer
                    REG(5);
                    if (!timer_add<K>(c, L, deadline, (EV_WAKE << 28) | (gen << 8) | slot, 0)) L.ovf = 1;
                    st = ST_PENDING;
                    break;
                }
                if (K::LIFE && op == MS_OP_ACCEPT) {       // rand_delay done -> conn_rx.recv()
                    sub = 2;
                    if (!accept_check(a)) break;
                } else if (K::LIFE && op == MS_OP_CRECV) {
                    uint4 u3 = TU(c, slot, c.P.chan_unit);
                    if (sub == 2) {                        // sleep(backoff) done: backoff = min(2 * backoff, 10 s); retry the link
                        uint32_t bo = (u3.x >> 16) * 2; if (bo > 10000) bo = 10000;
                        uint32_t cw = CONNW(u3.x & 0xff, 0);
                        uint64_t arrive = chan_test_link<K>(c, L, cw, 1 - ((u3.x >> 8) & 1));
                        u3.x = (u3.x & 0xffff) | (bo << 16); u3.z = (uint32_t)arrive; u3.w = (uint32_t)(arrive >> 32);
                        TU(c, slot, c.P.chan_unit) = u3;
                        crecv_arm();
                        break;
                    }
                    u0.w = u3.y;                           // sub 3: sleep_until(arrive_time) done -> yield value
                } else if (K::LIFE && op == MS_OP_CONNECT) {   // NetSim::connect1 (net/mod.rs:345-363)
                    uint32_t cx = TWORD(c, slot, c.P.chan_unit, 0);
                    if ((cx & 0xff) != 0xff) { conn_drop_handles<K>(c, L, cx & 0xff, (cx >> 8) & 1); TWORD(c, slot, c.P.chan_unit, 0) = cx | 0xff; }
                    uint64_t lat; int ds;
                    if (!try_send_fn<K>(c, L, SOCKW(c, a) & 0xff, b & 0xff, &lat, &ds)) {
                        u0.w = MADSIM_VAL_REFUSED;
                    } else {
                        uint32_t id = 0;
                        while (id < P.max_conns && (CONNW(id, 0) & 1)) id++;
                        uint32_t base = 2 + P.mbox_regs + 2 * P.mbox_msgs;
                        uint32_t q = SW(c, ds, base);
                        if (id >= P.max_conns || (q & 0xf) >= 4) { L.ovf = 1; }
                        else {
                            CONNW(id, 0) = 1u | (a << 1) | ((uint32_t)ds << 7) | (0xfu << 13);
                            CONNW(id, 1) = 0; CONNW(id, 2) = 0;
                            TWORD(c, slot, c.P.chan_unit, 0) = id;             // client side
                            u0.w = 0;
                            uint32_t n = q & 0xf;                  // socket.new_connection -> conn_tx.try_send
                            SW(c, ds, base) = (q & ~0xfu) | (n + 1) | (id << (4 + 7 * n));
                            uint32_t acc = SW(c, ds, base + 1);
                            if (acc & 1) { SW(c, ds, base + 1) = 0; wake<K>(c, L, (acc >> 1) & 0xff, acc >> 9); }
                        }
                    }
                } else if (op == MS_OP_BIND) {                    // Network::bind (network.rs:206-251)
                    uint32_t sw = SOCKW(c, a);
                    if ((sw & 0xff) != node || find_bound<K>(c, a) >= 0) { st = ST_PANIC; break; }
                    uint32_t h = SW(c, a, 0);
                    SW(c, a, 0) = 1u | ((((h >> 1) + 1) & 0xff) << 1);     // bound, gen+1, empty mailbox
                    SW(c, a, 1) = slot | (gen << 16);
                    if (K::LIFE && P.uses_chan) { SW(c, a, 2 + P.mbox_regs + 2 * P.mbox_msgs) = 0; SW(c, a, 3 + P.mbox_regs + 2 * P.mbox_msgs) = 0; }
                } else if (op == MS_OP_SEND || op == MS_OP_REPLY || (K::LIFE && op == MS_OP_RPC_REPLY)) {   // net/mod.rs:307-331
                    REG(6);
                    uint32_t dst = (op == MS_OP_SEND) ? (b & 0xff) : from;
                    if (K::LIFE && op == MS_OP_RPC_REPLY) {    // send_to_raw(from, rsp_tag, rsp): rpc.rs:172-175
                        b = 0xff00;
                        imm = (imm & 0xff) | (TWORD(c, slot, P.rpc_unit, 0) << 8);
                    }
                    uint32_t src_node = SOCKW(c, a) & 0xff;
                    uint32_t dst_node = SOCKW(c, dst) & 0xff;
                    // Network::try_send -> test_link (network.rs:261-269, 296-313)
                    bool clogged = false;
                    if (P.has_clog) clogged = ((CLOGW(1) >> src_node) & 1) | ((CLOGW(0) >> dst_node) & 1);
                    if (P.has_clog_link) clogged |= (CLOGW(2 + src_node) >> dst_node) & 1;
                    if (!clogged && !gen_bool_pint<K>(c, L, L.loss_pint, L.loss_always)) {
                        L.msg_count++;
                        uint64_t lat = sample_latency<K>(c, L);
                        int ds = find_bound<K>(c, dst);
                        if (ds >= 0) {
                            uint32_t sgen = (SW(c, ds, 0) >> 1) & 0xff;
                            uint32_t meta = (EV_DELIVER << 28) | (sgen << 20) | ((b >> 8) << 12) | (a << 6) | (uint32_t)ds;
                            if (!timer_add<K>(c, L, L.clock + lat, meta, imm)) L.ovf = 1;
                        }
                    }
                }
                completed = true;
            }
            if (completed) {                               // fall through to [B]/[C] with the next op: one pass per poll
                REG(12);
                sub = 0;
                // fused post-chain of this op (geometry.h build_tables): assert_eq!(val, ..), then djnz / jmp
                const uint32_t pf = in.w;
                if ((pf & 1) && u0.w != in.z) { st = ST_PANIC; break; }
                pc = (pf >> 4) & 0x3fff;
                if (pf & 2) {
                    uint32_t sh = ((pf >> 2) & 1) * 16;
                    uint32_t v = (((u0.z >> sh) & 0xffff) - 1) & 0xffff;
                    u0.z = (u0.z & ~(0xffffu << sh)) | (v << sh);
                    if (v) pc = pf >> 18;
                } else if (pf & 8) {
                    pc = pf >> 18;
                }
                if (pc >= P.n_insns) { st = ST_PANIC; break; }
                in = INSN(c, pc);
                op = in.x & 0xff; a = (in.x >> 8) & 0xff; b = in.x >> 16; imm = in.y;
            }
        }

        PROBE(6);
        // ================= [B] cheap ops that never await ========================================
        while (is_light(op)) {
            REG(13);
            if (op == MS_OP_ASSERT_VAL) {
                if (u0.w != imm) { st = ST_PANIC; break; }
                pc++;
            } else if (op == MS_OP_DJNZ) {
                uint32_t sh = (a & 1) * 16;
                uint32_t v = (((u0.z >> sh) & 0xffff) - 1) & 0xffff;
                u0.z = (u0.z & ~(0xffffu << sh)) | (v << sh);
                pc = v ? b : pc + 1;
            } else if (op == MS_OP_SET) {
                u0.z = (a & 1) ? ((u0.z & 0xffffu) | (imm << 16)) : ((u0.z & 0xffff0000u) | (imm & 0xffffu));
                pc++;
            } else if (op == MS_OP_JMP) {
                pc = b;
            } else if (op == MS_OP_JEQ) {
                pc = (u0.w == imm) ? b : pc + 1;
            } else {                                       // MS_OP_TRACE
                uint64_t v = imm;
                if (b & 1) v += (u0.z >> ((a & 1) * 16)) & 0xffff;
                L.obs_hash = (L.obs_hash ^ v) * FNV_PRIME;
                pc++;
            }
            if (pc >= P.n_insns) { st = ST_PANIC; break; }
            in = INSN(c, pc);
            op = in.x & 0xff; a = (in.x >> 8) & 0xff; b = in.x >> 16; imm = in.y;
        }
        if (st != ST_RUN) break;

        PROBE(7);
        REG(9);
        // ================= [C] begin the next op =================================================
        bool want_delay = false, want_sleep = false;
        uint64_t deadline = 0;
        if (op == MS_OP_RECV) {
            if (sub == 0) {                                // Mailbox::recv (endpoint.rs:353-362)
                REG(14);
                uint32_t tag = b >> 8;
                uint32_t rxseq = ((u1.x & 0xff) + 1) & 0xff;
                u1.x = (u1.x & ~0xffu) | rxseq; u1_dirty = true;
                u0.x &= ~TF_INBOX;
                uint32_t h = SW(c, a, 0);
                uint32_t nreg = (h >> 9) & 0xff, nmsg = (h >> 17) & 0xff;
                uint32_t idx = 0, mbase = 2 + P.mbox_regs;
                while (idx < nmsg && (SW(c, a, mbase + 2 * idx) & 0xff) != tag) idx++;
                if (idx < nmsg) {
                    uint32_t m0 = SW(c, a, mbase + 2 * idx), m1 = SW(c, a, mbase + 2 * idx + 1);
                    nmsg--;
                    SW(c, a, mbase + 2 * idx) = SW(c, a, mbase + 2 * nmsg);        // swap_remove
                    SW(c, a, mbase + 2 * idx + 1) = SW(c, a, mbase + 2 * nmsg + 1);
                    u0.w = m1;
                    if (K::LIFE && P.uses_rpc && tag >= MADSIM_TAG_RPC_FIRST) { u0.w = m1 & 0xff; TWORD(c, slot, P.rpc_unit, 0) = m1 >> 8; }
                    from = (m0 >> 8) & 0xff;
                    sub = 2;                               // oneshot already holds the value
                    SW(c, a, 0) = (h & ~(0xffu << 17)) | (nmsg << 17);
                } else {
                    if (nreg >= P.mbox_regs) { L.ovf = 1; st = ST_PENDING; break; }
                    SW(c, a, 2 + nreg) = tag | (slot << 8) | (rxseq << 16) | ((gen & 0xff) << 24);
                    SW(c, a, 0) = (h & ~(0xffu << 9)) | ((nreg + 1) << 9);
                    sub = 1;
                    st = ST_PENDING;
                }
            }
            want_delay = (sub == 2);                       // endpoint.rs:145 rand_delay
        } else if (op == MS_OP_SEND || op == MS_OP_REPLY || op == MS_OP_BIND || (K::LIFE && (op == MS_OP_CONNECT || op == MS_OP_ACCEPT || op == MS_OP_RPC_REPLY))) {
            want_delay = true;                             // net/mod.rs:306,344,457, endpoint.rs:198: rand_delay first
        } else if (K::LIFE && op == MS_OP_RECV_TIMEOUT) {
            uint32_t tag = b >> 8;
            uint64_t d2 = sleep_deadline(L, L.clock + (uint64_t)(b & 0xff) * NS_PER_S + imm);   // timeout()'s Sleep
            uint4 u2 = TU(c, slot, 2);
            u2.z = (uint32_t)d2; u2.w = (uint32_t)(d2 >> 32);
            TU(c, slot, 2) = u2;
            uint32_t rxseq = ((u1.x & 0xff) + 1) & 0xff;       // Mailbox::recv (endpoint.rs:353-362)
            u1.x = (u1.x & ~0xffu) | rxseq; u1_dirty = true;
            u0.x &= ~TF_INBOX;
            uint32_t h = SW(c, a, 0);
            uint32_t nreg = (h >> 9) & 0xff, nmsg = (h >> 17) & 0xff;
            uint32_t idx = 0, mbase = 2 + P.mbox_regs;
            while (idx < nmsg && (SW(c, a, mbase + 2 * idx) & 0xff) != tag) idx++;
            if (idx < nmsg) {
                uint32_t m0 = SW(c, a, mbase + 2 * idx), m1 = SW(c, a, mbase + 2 * idx + 1);
                nmsg--;
                SW(c, a, mbase + 2 * idx) = SW(c, a, mbase + 2 * nmsg);
                SW(c, a, mbase + 2 * idx + 1) = SW(c, a, mbase + 2 * nmsg + 1);
                u0.w = m1;
                if (P.uses_rpc && tag >= MADSIM_TAG_RPC_FIRST) { u0.w = m1 & 0xff; TWORD(c, slot, P.rpc_unit, 1) = m1 >> 8; }
                u0.y = (u0.y & 0x00ffffffu) | (((m0 >> 8) & 0xff) << 24);
                u0.x |= TF_INBOX;
                SW(c, a, 0) = (h & ~(0xffu << 17)) | (nmsg << 17);
            } else if (nreg >= P.mbox_regs) {
                L.ovf = 1;
            } else {
                SW(c, a, 2 + nreg) = tag | (slot << 8) | (rxseq << 16) | ((gen & 0xff) << 24);
                SW(c, a, 0) = (h & ~(0xffu << 9)) | ((nreg + 1) << 9);
            }
            sub = 1;
            if (recv_timeout_poll()) { sub = 0; pc++; }
        } else if (K::LIFE && op == MS_OP_RPC_CALL) {          // first poll of timeout(d, ep.call(dst, req)) / ep.call(dst, req)
            if (imm >> 8) {                                    // timeout()'s Sleep exists before the call is polled
                uint64_t d2 = sleep_deadline(L, L.clock + (uint64_t)(imm >> 8) * NS_PER_MS);
                uint4 u2 = TU(c, slot, 2);
                u2.z = (uint32_t)d2; u2.w = (uint32_t)(d2 >> 32);
                TU(c, slot, 2) = u2;
            }
            (void)rng_next(L); rng_log<K>(c, L);               // rsp_tag = random::<u64>(): one with() (rand.rs:146-148)
            uint64_t d1 = rand_delay_deadline<K>(c, L);        // send_to_raw -> NetSim::send: rand_delay first
            u1.z = (uint32_t)d1; u1.w = (uint32_t)(d1 >> 32); u1_dirty = true;
            sub = 1;
            if (rpc_call_poll()) { sub = 0; pc++; }            // (never on the first poll: 1 ms floor)
        } else if (K::LIFE && op == MS_OP_SLEEP_RAND) {        // sleep(thread_rng().gen_range(lo..hi)): [DEP A.3],
            const uint64_t* dp = P.dur_table + 4 * a;          // host-precomputed UniformDuration {mode, low, range, zone}
            uint64_t mode = dp[0], low = dp[1], range = dp[2], zone = dp[3], d;
            for (;;) {                                         // on the GlobalRng itself: one with() per attempt
                uint64_t v = rng_next(L);
                rng_log<K>(c, L);
                if (mode == 0) {
                    uint64_t m = (uint64_t)(uint32_t)(v >> 32) * (uint64_t)(uint32_t)range;
                    if ((uint32_t)m <= (uint32_t)zone) { d = low + (m >> 32); break; }
                } else if (v * range <= zone) { d = low + __umul64hi(v, range); break; }
            }
            deadline = sleep_deadline(L, L.clock + d);
            want_sleep = true;
        } else if (op == MS_OP_SLEEP || op == MS_OP_SLEEP_UNTIL) {
            uint64_t base = L.clock;
            if (K::LIFE && op == MS_OP_SLEEP_UNTIL) { uint4 u2 = TU(c, slot, 2); base = u64of(u2.x, u2.y); }
            deadline = sleep_deadline(L, base + (uint64_t)b * NS_PER_S + imm);
            want_sleep = true;
        } else {
            // ---- everything else: rare, control-plane ops ----
            switch (op) {
            case MS_OP_DONE:
                u0.y = pc | (sub << 16) | (from << 24);
                TU(c, slot, 0) = u0;
                if (u1_dirty) { tu1_store<K>(c, slot, u1); u1_dirty = false; }
                // an init task is `async { future.await; h.exit() }` (runtime/mod.rs:362-370): Spawner::exit =
                // NodeInfo::kill on the info it was spawned with (task/mod.rs:657-661), before the future drops
                if (K::LIFE && ((PROGW(c, u0.x >> 24) >> 8) & MADSIM_PROG_INIT) && (u1.y >> 24) == NODE_INFO_GEN(node)) {
                    NODEW(0) |= 1u << node;
                    if ((u1.y >> 24) == 0) NODEW(2) |= 1u << node;
                    info_kill<K>(c, L, node, u1.y >> 24);
                }
                task_finish<K>(c, L, slot, H_COMPLETED);
                u0.x = TWORD(c, slot, 0, 0);
                st = ST_FINISHED;
                break;
            case MS_OP_SPAWN: {
                uint32_t child = spawn_task<K>(c, L, a, true, (PROGW(c, a) & 0xff) != node);
                if (K::LIFE && P.uses_chan && (b & 2) && child != 0xffffffffu) {   // `async move`: the (tx, rx) pair moves
                    uint32_t cx = TWORD(c, slot, c.P.chan_unit, 0);
                    TWORD(c, child, c.P.chan_unit, 0) = cx & 0x1ff;
                    TWORD(c, slot, c.P.chan_unit, 0) = cx | 0xff;
                }
                if (K::LIFE && P.uses_rpc && (b & MADSIM_SPAWN_MOVE_REQUEST) && child != 0xffffffffu) {   // rpc.rs:170
                    TWORD(c, child, 0, 3) = u0.w;
                    TWORD(c, child, 0, 1) = (TWORD(c, child, 0, 1) & 0x00ffffffu) | (from << 24);
                    TWORD(c, child, P.rpc_unit, 0) = TWORD(c, slot, P.rpc_unit, 0);
                }
            }
                pc++;
                break;
            case MS_OP_BUILD:
                for (uint32_t p = 1; p < P.n_progs; p++) {
                    uint32_t pw = PROGW(c, p);
                    if ((pw & 0xff) == a && ((pw >> 8) & MADSIM_PROG_INIT) && !((pw >> 8) & MADSIM_PROG_PRE)) spawn_task<K>(c, L, p, false);
                }
                pc++;
                break;
            case MS_OP_JOIN: {                             // task/join.rs:59-72 + async-task poll_task
                uint32_t h = HW(a);
                uint32_t hs = h & 3;
                if (hs == H_RUNNING) {
                    uint32_t cs = (h >> 8) & 0xff;
                    uint32_t link = TWORD(c, cs, 1, 0);
                    TWORD(c, cs, 1, 0) = (link & 0xff) | (slot << 8) | (gen << 16);   // register awaiter
                    st = ST_PENDING;
                } else if (hs == H_NONE || ((hs == H_CANCELLED) != ((b & 1) != 0))) {
                    st = ST_PANIC;
                } else {
                    pc++;
                }
                break;
            }
            case MS_OP_YIELD:                              // [DEP tokio yield_now outside a runtime]
                sub = 1;
                u0.x |= TF_SCHED;                          // wake_by_ref while RUNNING
                st = ST_PENDING;
                break;
            case MS_OP_PANIC:
                st = ST_PANIC;
                break;
            case MS_OP_ABORT: {                            // AbortHandle::abort (task/join.rs:158-163)
                if (!K::LIFE) { st = ST_PANIC; break; }
                uint32_t h = HW(a);
                if ((h & 3) == H_RUNNING) {
                    uint32_t cs = (h >> 8) & 0xff;
                    TWORD(c, cs, 0, 0) |= TF_CANCEL;
                    wake<K>(c, L, cs, h >> 16);
                }
                pc++;
                break;
            }
            case MS_OP_KILL: case MS_OP_RESTART:
                if (!K::LIFE) { st = ST_PANIC; break; }
                u0.y = pc | (sub << 16) | (from << 24);     // this task may be woken/killed by the call: sync LDS first
                TU(c, slot, 0) = u0;
                if (op == MS_OP_KILL) node_kill<K>(c, L, a); else node_restart<K>(c, L, a);
                u0.x = TWORD(c, slot, 0, 0);
                pc++;
                break;
            case MS_OP_PAUSE:                               // task/mod.rs:404-410
                if (!K::LIFE) { st = ST_PANIC; break; }
                NODEW(1) |= 1u << a;
                pc++;
                break;
            case MS_OP_RESUME: {                            // task/mod.rs:413-424: parked Runnables go back, in order
                if (!K::LIFE) { st = ST_PANIC; break; }
                NODEW(1) &= ~(1u << a);
                if (P.uses_pause) {
                    uint32_t n = PAUSEW(0), keep = 0;
                    for (uint32_t i = 0; i < n; i++) {
                        uint32_t ps = PAUSEW(1 + i);
                        if ((PROGW(c, TWORD(c, ps, 0, 0) >> 24) & 0xff) == a) ready_push<K>(c, L, ps);
                        else { PAUSEW(1 + keep) = ps; keep++; }
                    }
                    PAUSEW(0) = keep;
                }
                pc++;
                break;
            }
            case MS_OP_ASSERT_EXIT:                         // Handle::is_exit (task/mod.rs:444-449)
                if (((NODEW(0) >> a) & 1) != (b & 1)) st = ST_PANIC; else pc++;
                break;
            case MS_OP_CSEND: {                            // PayloadSender::send (net/mod.rs:417-421)
                if (!K::LIFE) { st = ST_PANIC; break; }
                uint32_t cx = TWORD(c, slot, c.P.chan_unit, 0);
                if ((cx & 0xff) == 0xff) { u0.w = MADSIM_VAL_RESET; pc++; break; }
                uint32_t id = cx & 0xff, side = (cx >> 8) & 1;
                uint32_t cw = CONNW(id, 0);
                uint64_t arrive = chan_test_link<K>(c, L, cw, side);          // draws happen before the closed check
                if (!(cw & (1u << (14 + 2 * side)))) { u0.w = MADSIM_VAL_RESET; pc++; break; }   // ConnectionReset
                uint32_t qn = (cw >> (17 + 4 * side)) & 0xf;
                if (qn >= P.chan_queue) { L.ovf = 1; pc++; break; }
                uint32_t e = 3 + (side * P.chan_queue + qn) * 3;
                CONNW(id, e) = imm; CONNW(id, e + 1) = (uint32_t)arrive; CONNW(id, e + 2) = (uint32_t)(arrive >> 32);
                CONNW(id, 0) = (cw & ~(0xfu << (17 + 4 * side))) | ((qn + 1) << (17 + 4 * side));
                uint32_t r = CONNW(id, 1 + side);
                if (r & 1) { CONNW(id, 1 + side) = 0; wake<K>(c, L, (r >> 1) & 0xff, r >> 9); }   // mpsc wakes the parked receiver
                pc++;
                break;
            }
            case MS_OP_CRECV: {                            // rx.recv().await (net/mod.rs:386), sub == 0 here
                if (!K::LIFE) { st = ST_PANIC; break; }
                uint32_t cx = TWORD(c, slot, c.P.chan_unit, 0);
                if ((cx & 0xff) == 0xff) { u0.w = MADSIM_VAL_RESET; pc++; break; }
                uint32_t id = cx & 0xff, dir = 1 - ((cx >> 8) & 1);
                uint32_t cw = CONNW(id, 0);
                uint32_t qn = (cw >> (17 + 4 * dir)) & 0xf;
                if (qn == 0) {
                    if (!(cw & (1u << (13 + 2 * dir)))) { u0.w = MADSIM_VAL_RESET; pc++; break; }   // all senders gone
                    CONNW(id, 1 + dir) = 1u | (slot << 1) | (gen << 9);
                    st = ST_PENDING;
                    break;
                }
                uint32_t e0 = 3 + dir * P.chan_queue * 3;
                uint4 u3 = make_uint4((cx & 0x1ff) | (1u << 16), CONNW(id, e0), CONNW(id, e0 + 1), CONNW(id, e0 + 2));   // backoff = 1 ms
                for (uint32_t i = 1; i < qn; i++)              // VecDeque::pop_front
                    for (uint32_t k = 0; k < 3; k++) CONNW(id, e0 + (i - 1) * 3 + k) = CONNW(id, e0 + i * 3 + k);
                CONNW(id, 0) = (cw & ~(0xfu << (17 + 4 * dir))) | ((qn - 1) << (17 + 4 * dir));
                TU(c, slot, c.P.chan_unit) = u3;
                crecv_arm();
                break;
            }
            case MS_OP_CCLOSE: {
                if (!K::LIFE) { st = ST_PANIC; break; }
                uint32_t cx = TWORD(c, slot, c.P.chan_unit, 0);
                if ((cx & 0xff) != 0xff) { conn_drop_handles<K>(c, L, cx & 0xff, (cx >> 8) & 1); TWORD(c, slot, c.P.chan_unit, 0) = cx | 0xff; }
                pc++;
                break;
            }
            case MS_OP_GSET: GREGW(a & 3) = imm; pc++; break;
            case MS_OP_GADD: GREGW(a & 3) += imm; pc++; break;
            case MS_OP_ASSERT_G: if (GREGW(a & 3) != imm) st = ST_PANIC; else pc++; break;
            case MS_OP_PANIC_IF_G_LT: if (GREGW(a & 3) < imm) st = ST_PANIC; else pc++; break;
            case MS_OP_MARK:
                if (!K::LIFE) { st = ST_PANIC; break; }     // t0 family and advance(): extended variant only
                TU(c, slot, 2) = make_uint4((uint32_t)L.clock, (uint32_t)(L.clock >> 32), 0, 0);
                pc++;
                break;
            case MS_OP_ASSERT_ELAPSED: {
                if (!K::LIFE) { st = ST_PANIC; break; }
                uint4 u2 = TU(c, slot, 2);
                uint64_t el = L.clock - u64of(u2.x, u2.y), d = (uint64_t)b * NS_PER_S + imm;
                bool ok = a == 0 ? el == d : a == 1 ? el >= d : el < d;
                if (!ok) st = ST_PANIC; else pc++;
                break;
            }
            case MS_OP_ADVANCE:                            // time/mod.rs:103-106
                if (!K::LIFE) { st = ST_PANIC; break; }
                L.clock += (uint64_t)b * NS_PER_S + imm;
                pc++;
                u0.y = pc | (sub << 16) | (from << 24);
                TU(c, slot, 0) = u0;
                if (u1_dirty) { tu1_store<K>(c, slot, u1); u1_dirty = false; }
                timer_expire<K>(c, L, L.clock);
                u0 = TU(c, slot, 0); u1 = TU(c, slot, 1);
                from = u0.y >> 24;
                break;
            case MS_OP_CLOSE: {
                uint32_t h = SW(c, a, 0);
                if ((h & 1) && SW(c, a, 1) == (slot | (gen << 16)) && !(u0.x & TF_KILLED)) SW(c, a, 0) = h & ~1u;
                if (K::LIFE && P.uses_chan && SW(c, a, 1) == (slot | (gen << 16)) && (SW(c, a, 2 + P.mbox_regs + 2 * P.mbox_msgs) & 0xf)) sock_drop_acceptq<K>(c, L, a);
                pc++;
                break;
            }
            case MS_OP_CLOG_NODE:
                if (b & 1) CLOGW(0) |= 1u << a;
                if (b & 2) CLOGW(1) |= 1u << a;
                pc++;
                break;
            case MS_OP_UNCLOG_NODE:
                if (b & 1) CLOGW(0) &= ~(1u << a);
                if (b & 2) CLOGW(1) &= ~(1u << a);
                pc++;
                break;
            case MS_OP_CLOG_LINK:
                CLOGW(2 + a) |= 1u << b;
                pc++;
                break;
            case MS_OP_UNCLOG_LINK:
                CLOGW(2 + a) &= ~(1u << b);
                pc++;
                break;
            case MS_OP_RAND_BOOL:                           // thread_rng().gen_bool(p) [DEP A.4]
                if (!K::LIFE) { st = ST_PANIC; break; }
                u0.w = gen_bool_pint<K>(c, L, P.loss_table_pint[a & 3], P.loss_table_always[a & 3]) ? 1u : 0u;
                pc++;
                break;
            case MS_OP_SET_LOSS:
                L.loss_pint = P.loss_table_pint[a & 3];
                L.loss_always = P.loss_table_always[a & 3];
                pc++;
                break;
            default:
                st = ST_PANIC;
                break;
            }
        }
        PROBE(8);
        if (want_delay) {                                  // NetSim::rand_delay (net/mod.rs:287-292)
            REG(15);
            deadline = rand_delay_deadline<K>(c, L);
            want_sleep = true;
        }
        if (want_sleep) {                                  // first Sleep::poll: never elapsed (1 ms floor)
            REG(17);
            u1.z = (uint32_t)deadline; u1.w = (uint32_t)(deadline >> 32); u1_dirty = true;
            sub = (op == MS_OP_RECV) ? 3 : 1;
            if (!timer_add<K>(c, L, deadline, (EV_WAKE << 28) | (gen << 8) | slot, 0)) L.ovf = 1;
            st = ST_PENDING;
        }
    }
    PROBE(9);
    if (st != ST_FINISHED) {
        u0.y = pc | (sub << 16) | (from << 24);
        if (u1_dirty) tu1_store<K>(c, slot, u1);
    }
    return st == ST_PANIC;
}

// ---- per-seed init: Runtime::with_seed_and_config (runtime/mod.rs:53-69) ------------------------
template <class K>
__device__ void seed_init(const Ctx& c, Lane& L, uint64_t seed) {
    const KParams& P = c.P;
    for (uint32_t w = 0; w < P.lane_words; w++) RW(w) = 0;          // plane 0 is the ready queue: RW spans all planes
    for (uint32_t t = 0; t < P.max_tasks; t++) { TWORD(c, t, 0, 0) = 0; if (!K::LIFE) TWORD(c, t, 1, 1) = 0; }
    // GlobalRng::new_with_seed -> Xoshiro256PlusPlus::seed_from_u64: SplitMix64 [DEP A.1]
    uint64_t x = seed, z;
#define SPLITMIX(dst) x += 0x9e3779b97f4a7c15ull; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; dst = z ^ (z >> 31)
    SPLITMIX(L.s0); SPLITMIX(L.s1); SPLITMIX(L.s2); SPLITMIX(L.s3);
#undef SPLITMIX
    L.rng_calls = 0; L.clock = 0; L.msg_count = 0; L.steps = 0;
    L.ready_len = 0; L.rq = 0; L.heap_len = 0; L.top_dl = ~0ull; L.verdict = MADSIM_RUNNING; L.ovf = 0; L.main_done = 0;
    L.loss_pint = P.loss_pint; L.loss_always = P.loss_always;
    // TimeRuntime::new (time/mod.rs:26-38): base_time draw, before logging is enabled
    { uint64_t h = L.trace_hash, n = L.log_len; (void)gen_range_small<Variant<false, false, K::LWS, false, K::RQ>, 31536000u>(c, L); L.trace_hash = h; L.log_len = n; }
    L.trace_hash = FNV_OFFSET; L.obs_hash = FNV_OFFSET; L.log_len = 0;
    // tasks spawned before block_on, then the main task (task/mod.rs:222-235)
    for (uint32_t p = 1; p < P.n_progs; p++) {
        uint32_t fl = (PROGW(c, p) >> 8) & 0xff;
        if (fl & MADSIM_PROG_PRE) spawn_task<K>(c, L, p, !(fl & MADSIM_PROG_INIT));
    }
    spawn_task<K>(c, L, 0, true);
}

template <class K>
__global__ __launch_bounds__(256) void sim_kernel(const KParams P) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // workgroup-shared tables
    uint32_t* sh = SMEM;
#ifdef MADSIM_EMU
    const uint32_t cp0 = 0, cps = 1;      // emulated threads run one after another: each copies everything
#else
    const uint32_t cp0 = threadIdx.x, cps = 64 * P.waves_per_block;
#endif
    for (uint32_t i = cp0; i < P.n_insns * 4; i += cps) sh[P.sh_insns + i] = ((const uint32_t*)P.insns)[i];
    for (uint32_t i = cp0; i < P.n_progs; i += cps) sh[P.sh_progs + i] = P.progs[i];
    for (uint32_t i = cp0; i < P.n_socks; i += cps) sh[P.sh_socks + i] = P.socks[i];
    __syncthreads();

    Ctx c(P);
    c.insn0 = P.sh_insns / 4;
    c.prog0 = P.sh_progs;
    c.sockt0 = P.sh_socks;
    const uint32_t wbase = wv * P.wave_words;       // this wave's slice of the workgroup's LDS
    c.heap0 = (P.sh_heap + wbase) / 4 + lane;
    c.task0 = (P.sh_tasks + wbase) / 4 + lane;
    c.lws = P.lw_shift;
    const uint32_t pl = P.sh_planes + wbase + lane;
    c.ready0 = pl + (P.off_ready << P.lw_shift);
    c.sock0 = pl + (P.off_socks << P.lw_shift);
    c.hand0 = pl + (P.off_handles << P.lw_shift);
    c.node0 = pl + (P.off_nodes << P.lw_shift);
    c.clog0 = pl + (P.off_clog << P.lw_shift);
    c.pause0 = pl + (P.off_pause << P.lw_shift);
    c.greg0 = pl + (P.off_greg << P.lw_shift);
    c.conn0 = pl + (P.off_conn << P.lw_shift);
    if (lane >= (1u << P.lw_shift)) return;      // sub-wave occupancy: only lw = 2^lw_shift lanes carry seeds
    const uint32_t glane = ((blockIdx.x * P.waves_per_block + wv) << P.lw_shift) + lane;
    c.spill = P.spill ? P.spill + glane : nullptr;
    c.tlog = P.trace_log;

    Lane L;
#if defined(EXP_PROF) || defined(EXP_PROF2)
    for (int i = 0; i < 12; i++) L.prof_acc[i] = 0;
    L.prof_t = __builtin_readcyclecounter(); uint64_t prof_iters = 0;
#endif
    uint64_t next = glane;          // static striding: lane g runs seeds g, g+G, g+2G, ...
    bool have = false;
    for (;;) {
        if (!have) {
            if (next >= P.count) break;
            seed_init<K>(c, L, P.seed0 + next);
            have = true;
        }
        // One iteration = one pass of the block_on loop body (task/mod.rs:239-259):
        //   [poll]  ready queue non-empty: one run_all_ready iteration (pop, poll, 50..100 ns advance)
        //   [fire]  Timer::expire up to `now`; while the queue stays empty: is_finished / deadlock checks
        //           and advance_to_next_event, firing again — the ONLY place timers fire.
        // A lane that polls and then runs dry fires its next timer in the same pass, so in steady state
        // every lane does one poll and one timer fire per iteration and the wave stays in phase.
#if defined(EXP_PROF) || defined(EXP_PROF2)
        prof_iters++;
#endif
        uint64_t now = L.clock;
        PROBE(0);
        REG(0);
        if (L.ready_len > 0) {
            // Latency hiding: the queue usually holds exactly one task, so ready[0] and its two state units are
            // loaded BEFORE the draw loop (whose rejection retries take hundreds of cycles) and used if idx == 0.
            const uint32_t slot0 = K::RQ ? (uint32_t)(L.rq & 0xff) : RW(0);
            const uint4 pu0 = TU(c, slot0, 0), pu1 = TU(c, slot0, 1);
            // try_recv_random (utils/mpsc.rs:73-83): idx drawn even when len == 1
            uint32_t idx = gen_index<K>(c, L, L.ready_len);
            L.ready_len--;
            uint32_t slot = slot0;
            uint4 u0 = pu0, u1 = pu1;
            if (K::RQ) {
                if (idx != 0) { slot = (uint32_t)(L.rq >> (8 * idx)) & 0xff; u0 = TU(c, slot, 0); u1 = TU(c, slot, 1); }
                uint64_t last = (L.rq >> (8 * L.ready_len)) & 0xff;      // swap_remove on bytes
                L.rq = (L.rq & ~(0xffull << (8 * idx))) | (last << (8 * idx));
                L.rq &= ~(0xffull << (8 * L.ready_len));
            } else {
                if (idx != 0) { slot = RW(idx); u0 = TU(c, slot, 0); u1 = TU(c, slot, 1); }
                if (idx != L.ready_len) RW(idx) = RW(L.ready_len);       // swap_remove
            }
            L.steps++;
            bool panicked = false;
            PROBE(1);
            REG(26);
            bool parked = false;
            if (K::LIFE && (u0.x & (TF_CANCEL | TF_KILLED))) {   // task/mod.rs:269-273: drop(runnable)
                task_finish<K>(c, L, slot, H_CANCELLED);
            } else if (K::LIFE && P.uses_pause && ((NODEW(1) >> (PROGW(c, u0.x >> 24) & 0xff)) & 1)) {
                uint32_t n = PAUSEW(0);                       // :274-277: park the Runnable; no poll, no time advance
                PAUSEW(1 + n) = slot; PAUSEW(0) = n + 1;
                L.steps--;
                parked = true;
            } else {
                u0.x = (u0.x & ~TF_SCHED) | TF_RUN;          // async-task run(): SCHEDULED -> RUNNING
                panicked = poll_task<K>(c, L, slot, u0, u1);
                if (!panicked && (u0.x & TF_ALIVE)) {
                    if (u0.x & TF_SCHED) ready_push<K>(c, L, slot);   // woken while running: re-queue after the poll
                    u0.x &= ~TF_RUN;
                    TU(c, slot, 0) = u0;
                }
            }
            PROBE(2);
            if (K::LIFE && panicked && P.has_restart_on_panic) {   // task/mod.rs:289-314
                uint32_t node = PROGW(c, u0.x >> 24) & 0xff;
                if ((P.restart_nodes >> node) & 1) {
                    // async-task's panic guard already dropped the future and notified the awaiter
                    TU(c, slot, 0) = u0;
                    task_finish<K>(c, L, slot, H_CANCELLED);
                    // delay = gen_range(1 s..10 s) in ONE with(): UniformDuration Medium path [DEP A.3]
                    const uint64_t range = 9000000000ull, zone = ~0ull - ((~0ull - range + 1) % range);
                    uint64_t v;
                    do { v = rng_next(L); } while (v * range > zone);
                    rng_log<K>(c, L);
                    uint64_t delay = NS_PER_S + __umul64hi(v, range);
                    node_kill<K>(c, L, node);                 // self.kill(node_id)
                    if (!timer_add<K>(c, L, L.clock + delay, (EV_RESTART << 28) | node, 0)) L.ovf = 1;
                    panicked = false;
                }
            }
            if (panicked) L.verdict = MADSIM_PANIC;          // resume_unwind (:315): no advance, no expire
            else if (!parked) L.clock += 50 + gen_range_small<K, 50>(c, L);   // :319-321, then Timer::expire (time/mod.rs:103-106)
            now = L.clock;
            PROBE(3);
        }
        bool idle_jump = false;
        while (L.verdict == MADSIM_RUNNING) {
            REG(19);
            timer_expire<K>(c, L, now);
            if (idle_jump) {
                L.clock = now;                                // time/mod.rs:55: after the callbacks
                idle_jump = false;
                if (P.time_limit && L.clock >= P.time_limit) { L.verdict = MADSIM_TIME_LIMIT; break; }   // task/mod.rs:253-258
            }
            if (L.steps >= P.max_steps) { L.verdict = MADSIM_STEP_LIMIT; break; }
            if (L.ready_len > 0) break;                       // back to run_all_ready
            if (L.main_done) { L.verdict = MADSIM_PASS; break; }                          // :241-243
            if (L.heap_len == 0) { L.verdict = MADSIM_DEADLOCK; break; }                  // :250
            now = L.top_dl + 50;                              // advance_to_next_event (time/mod.rs:47-53)
            idle_jump = true;
        }
        PROBE(4);
        if (L.ovf) L.verdict = MADSIM_OVERFLOW;
        if (L.verdict != MADSIM_RUNNING) {
            REG(25);
            madsim_result_t r;
            r.verdict = L.verdict; r.steps = L.steps; r.clock_ns = L.clock; r.msg_count = L.msg_count;
            r.rng_calls = L.rng_calls; r.trace_hash = L.trace_hash; r.obs_hash = L.obs_hash;
            P.out[next] = r;
            if (K::TRACE) *P.trace_len = L.log_len;
            have = false;
            next += P.total_lanes;
        }
    }
#if defined(EXP_PROF) || defined(EXP_PROF2)
    PROBE2(0);
    if (lane == 0 && P.prof) { for (int i = 0; i < 12; i++) atomicAdd((unsigned long long*)&P.prof[i], (unsigned long long)L.prof_acc[i]); atomicAdd((unsigned long long*)&P.prof[12], (unsigned long long)prof_iters); atomicAdd((unsigned long long*)&P.prof[13], 1ull); }
#endif
}

#ifndef MADSIM_EMU
// ---- summary reduction over the result array (first failing seed = min) -------------------------
__global__ __launch_bounds__(256) void summary_kernel(const madsim_result_t* __restrict__ out, uint64_t count,
                                                      uint64_t seed0, unsigned long long* __restrict__ acc) {
    __shared__ unsigned long long part[4][4];
    unsigned long long first = ~0ull, nfail = 0, steps = 0, clk = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4* p = reinterpret_cast<const uint4*>(out + i);     // verdict|steps|clock_ns in the first 16 bytes
        uint4 r = p[0];
        if (r.x != MADSIM_PASS) { nfail++; unsigned long long s = seed0 + i; first = s < first ? s : first; }
        steps += r.y; clk += ((unsigned long long)r.w << 32) | r.z;
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long f2 = __shfl_xor(first, o), n2 = __shfl_xor(nfail, o), s2 = __shfl_xor(steps, o), c2 = __shfl_xor(clk, o);
        first = f2 < first ? f2 : first; nfail += n2; steps += s2; clk += c2;
    }
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[wv][0] = first; part[wv][1] = nfail; part[wv][2] = steps; part[wv][3] = clk; }
    __syncthreads();
    if (threadIdx.x == 0) {                                             // one set of atomics per workgroup
        for (int k = 1; k < 4; k++) {
            first = part[k][0] < first ? part[k][0] : first; nfail += part[k][1]; steps += part[k][2]; clk += part[k][3];
        }
        atomicMin(&acc[0], first); atomicAdd(&acc[1], nfail); atomicAdd(&acc[2], steps); atomicAdd(&acc[3], clk);
    }
}

__global__ void keyflip_kernel(unsigned long long* acc) { acc[0] ^= 0x8000000000000000ull; }

#endif  // !MADSIM_EMU

}  // namespace madsim_k

#ifndef MADSIM_EMU
// Kernel variants: the trace build, a fully generic build (runtime lane stride), for full 64-lane waves one build
// per (heap spill, extended ops) combination so that workloads only pay for what they use, and the full-featured
// build again for each sub-wave lane stride (32/16/8 seed lanes per wave).
#define MADSIM_FOR_EACH_VARIANT(X)      \
    X(true, true, -1, true)             \
    X(false, true, -1, true)            \
    X(false, false, 6, false)           \
    X(false, false, 6, false, true)     \
    X(false, true, 6, false, true)      \
    X(false, true, 6, false)            \
    X(false, false, 6, true)            \
    X(false, true, 6, true)             \
    X(false, true, 5, true)             \
    X(false, true, 4, true)             \
    X(false, true, 3, true)

extern "C" void madsim_k_launch_sim(const madsim_k::KParams* P, uint32_t grid, uint32_t lds_bytes, void* stream, int trace) {
    using namespace madsim_k;
    const bool spill = P->spill != nullptr && P->heap_spill > 0;
    const bool life = P->lifecycle != 0;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(...) hipLaunchKernelGGL((sim_kernel<Variant<__VA_ARGS__>>), dim3(grid), dim3(64 * P->waves_per_block), lds_bytes, st, *P)
    if (trace) LAUNCH(true, true, -1, true);
#ifndef EXP_NO_LWS_VARIANTS
    else if (P->lw_shift == 5) LAUNCH(false, true, 5, true);     // sub-wave occupancy (large per-seed state): the lane
    else if (P->lw_shift == 4) LAUNCH(false, true, 4, true);     // stride stays a compile-time shift
    else if (P->lw_shift == 3) LAUNCH(false, true, 3, true);
#endif
    else if (P->lw_shift != 6) LAUNCH(false, true, -1, true);
    else if (!spill && !life && P->rq_in_reg) LAUNCH(false, false, 6, false, true);
    else if (!spill && !life) LAUNCH(false, false, 6, false);
    else if (spill && !life && P->rq_in_reg) LAUNCH(false, true, 6, false, true);
    else if (spill && !life) LAUNCH(false, true, 6, false);
    else if (!spill && life) LAUNCH(false, false, 6, true);
    else LAUNCH(false, true, 6, true);
#undef LAUNCH
}

extern "C" void madsim_k_launch_summary(const madsim_result_t* out, uint64_t count, uint64_t seed0, unsigned long long* acc, void* stream) {
    uint32_t grid = (uint32_t)((count + 1023) / 1024);     // >= 4 results per thread, <= 256 workgroups (4 atomics each)
    if (grid > 256) grid = 256;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(madsim_k::summary_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, count, seed0, acc);
}

extern "C" void madsim_k_launch_keyflip(unsigned long long* acc, void* stream) {
    hipLaunchKernelGGL(madsim_k::keyflip_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc);
}

extern "C" int madsim_k_set_max_lds(uint32_t lds_bytes) {
    using namespace madsim_k;
    hipError_t e = hipSuccess;
#define SETATTR(...)                                                                                                     \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sim_kernel<Variant<__VA_ARGS__>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    MADSIM_FOR_EACH_VARIANT(SETATTR)
#undef SETATTR
    return (int)e;
}
#endif  // !MADSIM_EMU
