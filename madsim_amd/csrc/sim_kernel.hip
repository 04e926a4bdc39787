// sim_kernel.hip — the many-seed executor kernel for gfx950 (MI355X, CDNA4).
//
// ONE LANE = ONE SEED.  A workgroup is up to four independent 64-lane wavefronts (one per SIMD of a CU) that share
// nothing but the read-only workload tables.  Each lane runs madsim's whole per-seed executor loop:
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
#include "kernel/k_mem.h"
#include "sim_kernel.h"

#include "kernel/k_state.h"
#include "kernel/k_rng.h"
#include "kernel/k_timer.h"
#include "kernel/k_net.h"
#include "kernel/k_lifecycle.h"
#include "kernel/k_channel.h"
#include "kernel/k_poll.h"
#include "kernel/k_main.h"

namespace madsim_k {

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

// the campaign form: also tells the RUNNER verdicts (device capacity, step cap: not reference verdicts) from genuine failures —
// acc6 = {first failing seed, n_failed, steps, clock, first seed with a GENUINE verdict (panic / deadlock / time limit), n runner verdicts}
__global__ __launch_bounds__(256) void summary6_kernel(const madsim_result_t* __restrict__ out, uint64_t count,
                                                       uint64_t seed0, unsigned long long* __restrict__ acc) {
    unsigned long long first = ~0ull, nfail = 0, steps = 0, clk = 0, gfirst = ~0ull, nrun = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 r = reinterpret_cast<const uint4*>(out + i)[0];
        if (r.x != MADSIM_PASS) {
            const unsigned long long s = seed0 + i;
            nfail++; first = s < first ? s : first;
            if (MADSIM_IS_RUNNER_VERDICT(r.x)) nrun++; else gfirst = s < gfirst ? s : gfirst;
        }
        steps += r.y; clk += ((unsigned long long)r.w << 32) | r.z;
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long f2 = __shfl_xor(first, o), g2 = __shfl_xor(gfirst, o);
        first = f2 < first ? f2 : first; gfirst = g2 < gfirst ? g2 : gfirst;
        nfail += __shfl_xor(nfail, o); steps += __shfl_xor(steps, o); clk += __shfl_xor(clk, o); nrun += __shfl_xor(nrun, o);
    }
    if ((threadIdx.x & 63) == 0) {                                      // one set of atomics per wave (<= 1 024 per launch)
        atomicMin(&acc[0], first); atomicAdd(&acc[1], nfail); atomicAdd(&acc[2], steps); atomicAdd(&acc[3], clk);
        atomicMin(&acc[4], gfirst); atomicAdd(&acc[5], nrun);
    }
}

__global__ void keyflip_kernel(unsigned long long* acc) { acc[0] ^= 0x8000000000000000ull; }

}  // namespace madsim_k

extern "C" int madsim_k_launch_sim(const madsim_k::KParams* P, uint32_t grid, uint32_t lds_bytes, void* stream, int trace) {
    using namespace madsim_k;
    const VariantSel v = select_variant(*P, trace != 0);
    hipStream_t st = (hipStream_t)stream;
    bool launched = false;
#define TRY_LAUNCH(T_, S_, L_, F_, R_, G_)                                                                               \
    if (!launched && v.trace == (int)(T_) && v.spill == (int)(S_) && v.lws == (L_) && v.feat == (F_) && v.rq == (int)(R_) && v.g == (int)(G_)) { \
        hipLaunchKernelGGL((sim_kernel<Variant<T_, S_, L_, F_, R_, G_>>), dim3(grid), dim3(64 * P->waves_per_block), lds_bytes, st, *P); \
        launched = true;                                                                                                 \
    }
    MADSIM_FOR_EACH_VARIANT(TRY_LAUNCH)
#undef TRY_LAUNCH
    return launched ? 0 : -1;      // select_variant named a build that is not compiled: a bug, never a silent substitute
}

extern "C" void madsim_k_launch_summary(const madsim_result_t* out, uint64_t count, uint64_t seed0, unsigned long long* acc, void* stream) {
    uint32_t grid = (uint32_t)((count + 1023) / 1024);     // >= 4 results per thread, <= 256 workgroups (4 atomics each)
    if (grid > 256) grid = 256;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(madsim_k::summary_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, count, seed0, acc);
}

extern "C" void madsim_k_launch_summary6(const madsim_result_t* out, uint64_t count, uint64_t seed0, unsigned long long* acc6, void* stream) {
    uint32_t grid = (uint32_t)((count + 1023) / 1024);
    if (grid > 256) grid = 256;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(madsim_k::summary6_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, count, seed0, acc6);
}

extern "C" void madsim_k_launch_keyflip(unsigned long long* acc, void* stream) {
    hipLaunchKernelGGL(madsim_k::keyflip_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc);
}

// VGPRs per lane of the build select_variant names (0 = not compiled / error): the host sizes waves per CU with it.
extern "C" int madsim_k_variant_vgprs(const madsim_k::VariantSel* v) {
    using namespace madsim_k;
    int regs = 0;
#define TRY_ATTR(T_, S_, L_, F_, R_, G_)                                                                                 \
    if (!regs && v->trace == (int)(T_) && v->spill == (int)(S_) && v->lws == (L_) && v->feat == (F_) && v->rq == (int)(R_) && v->g == (int)(G_)) { \
        hipFuncAttributes a;                                                                                             \
        if (hipFuncGetAttributes(&a, (const void*)sim_kernel<Variant<T_, S_, L_, F_, R_, G_>>) == hipSuccess) regs = a.numRegs; \
    }
    MADSIM_FOR_EACH_VARIANT(TRY_ATTR)
#undef TRY_ATTR
    return regs;
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
