// k_state.h — Per-lane state, LDS layout accessors and compile-time variants of sim_kernel.
// Part of sim_kernel.hip (included in this order: k_state, k_rng, k_timer, k_net, k_lifecycle, k_channel, k_poll).
#ifndef MADSIM_K_STATE_H
#define MADSIM_K_STATE_H

namespace madsim_k {

// Timing-experiment switches live outside the product tree (tools/experiment/k_experiment.h) and are reachable only
// through tools/build_variant.sh; the product build refuses them.
#ifdef MADSIM_EXPERIMENT_BUILD
#include "../../../tools/experiment/k_experiment.h"
#else
#if defined(EXP_NOLOG) || defined(EXP_ALWAYS_ACCEPT) || defined(EXP_PROF) || defined(EXP_PROF2) || defined(EXP_NO_LWS_VARIANTS)
#error "EXP_* switches make a non-bit-exact kernel: build experiment variants with tools/build_variant.sh, never the product Makefile"
#endif
#define EXP_ACCEPT(x) (x)
#define MADSIM_K_LOG_ENABLED 1
#define PROBE(i) do { } while (0)
#define PROBE2(i) do { } while (0)
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
#ifdef MADSIM_K_PROF
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

// The spill region is reached through a buffer resource (buffer_load/store_dwordx4, byte offsets in a VGPR): with a
// plain pointer the compiler folds the LDS and HBM alternatives of heap_get/heap_set into one flat_load/flat_store,
// which is slower for both and waits on both counters.
#ifdef MADSIM_EMU
struct SpillRef { uint4* base; };
#else
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct SpillRef { __amdgpu_buffer_rsrc_t rsrc; };
#endif

struct Ctx {
    const KParams& P;
    uint32_t lws;        // log2(lane stride) (runtime copy; K::LWS overrides when >= 0)
    uint32_t ready0, hand0, node0, clog0, pause0, greg0, conn0;   // word indices of this lane's plane regions
    uint32_t sock0;      // word index of this lane's socket region
    uint32_t heap0;      // uint4 index of heap entry 0: entry i = LDS128(heap0 + (i << lws))
    uint32_t task0;      // uint4 index of task unit 0
    uint32_t insn0;      // uint4 index of the workgroup-shared instruction table
    uint32_t prog0, sockt0;   // word indices of the shared prog / socket-address tables
    SpillRef spill;      // the HBM spill region: entry (slot, this lane) at byte (slot * P.total_lanes) * 16 + spill_off
    uint32_t spill_off;  // this lane's column: global lane * 16
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
// node region: [0] killed mask, [1] paused mask, [2] gen0_killed mask, [3] spawn counter, [4 + n/4] info_gen bytes,
//              then one word: the seed's base time in seconds into 2022 (time/mod.rs:26-33)
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

}  // namespace madsim_k

#endif
