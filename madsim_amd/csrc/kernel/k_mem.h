// k_mem.h — the memory primitives sim_kernel is written against: the workgroup's LDS array, buffer-resource access to
// global memory, the 64-bit rotate — for gfx950, the only target.  Nothing else in kernel/*.h or sim_kernel.hip touches a
// HIP built-in directly, so a test harness can compile the executor text against its own definitions of these names
// (tests/emu does, by defining this header's include guard before it is reached); the product build has one path.
#ifndef MADSIM_K_MEM_H
#define MADSIM_K_MEM_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace madsim_k {

// All LDS traffic goes through the workgroup's one `extern __shared__` array, indexed by per-lane
// offsets held in VGPRs: the compiler then knows every access is LDS (ds_read/ds_write) — pointer
// members that may alias global memory degrade to flat_* instructions.
extern __shared__ __attribute__((aligned(16))) uint32_t madsim_smem[];
#define SMEM madsim_smem

// Global memory (the timer-heap spill region, the per-lane state blocks) is reached through buffer resources
// (buffer_load/store, byte offsets in a VGPR): with plain pointers the compiler folds the LDS and global alternatives of an
// access into one flat_load/flat_store, which is slower for both and waits on both counters.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct BufRef { __amdgpu_buffer_rsrc_t rsrc; };
// gfx9 raw buffer, 32-bit data format; num_records in bytes (the host keeps every region below 4 GiB)
__device__ __forceinline__ BufRef buf_make(const void* base, uint64_t bytes) {
    BufRef b; b.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (uint32_t)bytes, 0x00020000); return b;
}
__device__ __forceinline__ uint32_t buf_load32(const BufRef& b, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(b.rsrc, off, 0, 0); }
__device__ __forceinline__ void buf_store32(const BufRef& b, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, b.rsrc, off, 0, 0); }
// word += v with nobody waiting for it: buffer_atomic_add without a return value (the lane's own word: nothing contends; a later read by the
// same lane sees it — program order).  A load + add + store would hold the wave for the load's round trip.
__device__ __forceinline__ void buf_add32(const BufRef& b, uint32_t off, uint32_t v) { (void)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32((int)v, b.rsrc, off, 0, 0); }
__device__ __forceinline__ uint2 buf_load64(const BufRef& b, uint32_t off) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(b.rsrc, off, 0, 0);
    return make_uint2(t.x, t.y);
}
__device__ __forceinline__ void buf_store64(const BufRef& b, uint32_t off, const uint2& e) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t t = {e.x, e.y};
    __builtin_amdgcn_raw_buffer_store_b64(t, b.rsrc, off, 0, 0);
}
__device__ __forceinline__ uint4 buf_load128(const BufRef& b, uint32_t off) {
    u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(b.rsrc, off, 0, 0);
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void buf_store128(const BufRef& b, uint32_t off, const uint4& e) {
    u32x4_t t = {e.x, e.y, e.z, e.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, b.rsrc, off, 0, 0);
}

// 64-bit rotate as two v_alignbit_b32 (the compiler's shift/or expansion takes 3-4 VALU ops): K_ is a compile-time constant.
template <int K_>
__device__ __forceinline__ uint64_t rotl64(uint64_t x) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (K_ >= 32) { uint32_t t = lo; lo = hi; hi = t; }            // rotate by 32 = swap halves
    constexpr int k = K_ & 31;
    if (k == 0) return ((uint64_t)hi << 32) | lo;
    uint32_t nhi = __builtin_amdgcn_alignbit(hi, lo, 32 - k);      // ({hi,lo} >> (32-k))[31:0] = hi<<k | lo>>(32-k)
    uint32_t nlo = __builtin_amdgcn_alignbit(lo, hi, 32 - k);
    return ((uint64_t)nhi << 32) | nlo;
}

// a ^ b ^ c on 64 bits as one v_bitop3_b32 (truth table 0x96) per half: the compiler does not form it from two xors
__device__ __forceinline__ uint64_t xor3_64(uint64_t a, uint64_t b, uint64_t c) {
    uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0x96);
    uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0x96);
    return ((uint64_t)hi << 32) | lo;
}

// a + b as ONE v_lshl_add_u64: when `a` was just assembled from two 32-bit halves the compiler reassociates the sum into a
// 64-bit add of the low half plus a 32-bit add of the high half (three instructions with the zero-extension)
__device__ __forceinline__ uint64_t add64_1(uint64_t a, uint64_t b) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// x << K_ as ONE v_lshlrev_b64 (half rate, 4.3 cycles): left alone the compiler splits a 64-bit shift by a constant into
// v_lshlrev_b32 + v_alignbit_b32 — two half-rate instructions (tools/ubench_issue.hip: only add / sub / xor / mov / bitop3 are
// full rate on gfx950, every shift is half rate whatever its width)
template <int K_>
__device__ __forceinline__ uint64_t shl64(uint64_t x) {
    uint64_t r;
    asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "n"(K_), "v"(x));
    return r;
}

// (v << K_) + v as ONE v_lshl_add_u64: the low 64 bits of v * (2^K_ + 1), K_ <= 4
template <int K_>
__device__ __forceinline__ uint64_t mul_pow2p1(uint64_t v) {
    uint64_t p;
    asm("v_lshl_add_u64 %0, %1, %2, %1" : "=v"(p) : "v"(v), "n"(K_));
    return p;
}

// Issue priority of this wave (0..3; s_setprio takes an immediate, hence the ladder).  `q` must be wave-uniform.
__device__ __forceinline__ void wave_set_priority(uint32_t q) {
    q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
    if (q == 0) __builtin_amdgcn_s_setprio(0); else if (q == 1) __builtin_amdgcn_s_setprio(1);
    else if (q == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
}
__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// maximum of `v` over the ACTIVE lanes of the wave, as a wave-uniform value (a scalar loop over the exec mask: used once per wave, at its end)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    uint64_t m = __builtin_amdgcn_ballot_w64(true);
    uint32_t mx = 0;
    while (m) {
        const int l = __builtin_ctzll(m);
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
        mx = x > mx ? x : mx;
        m &= m - 1;
    }
    return mx;
}
__device__ __forceinline__ bool wave_first_lane() { return (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)); }
__device__ __forceinline__ void atomic_max_u32(uint32_t* p, uint32_t v) { atomicMax(p, v); }

// true when `p` holds in every active lane of the wave (a scalar: branches on it are uniform)
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0; }

// the threads of a workgroup share the copy of the workload tables into LDS: thread t copies words t, t + stride, ...
__device__ __forceinline__ uint32_t table_copy_first() { return threadIdx.x; }
__device__ __forceinline__ uint32_t table_copy_stride(uint32_t waves_per_block) { return 64 * waves_per_block; }
// per-access statistics hook of the emulation harness (tools/gstate_access_model.py); nothing on the GPU
#define EMU_GSTAT(off, kind) do { } while (0)

}  // namespace madsim_k

#endif
