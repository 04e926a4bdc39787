// emu_shim.h — lets madsim_amd/csrc/sim_kernel.hip compile as plain host C++ (g++), one emulated
// GPU thread at a time.  DEBUGGING AID for this GPU-less container: it exercises the device code's
// logic (LDS layout arithmetic, heap, mailbox, interpreter) against the oracle before GPU minutes
// are spent.  It is not part of the product, is never loaded by madsim_amd/, and proves nothing
// about the GPU build — the -m gpu tests do that.
#ifndef MADSIM_EMU_SHIM_H
#define MADSIM_EMU_SHIM_H
#define MADSIM_K_MEM_H      // this header stands in for madsim_amd/csrc/kernel/k_mem.h: its body is skipped when included later
#include <stdint.h>
#include <stddef.h>

struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct emu_dim3 { uint32_t x, y, z; };
extern thread_local emu_dim3 threadIdx, blockIdx;
extern thread_local uint32_t* emu_smem;
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
static inline void __syncthreads() {}
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }   // emulated threads run one after another
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
// tools/gstate_access_model.py: which words of the per-lane global state block the kernel touches, and how often
#ifdef MADSIM_EMU_GSTAT
void emu_gstat(uint32_t byte_off, int kind);      // kind: 0 load32, 1 store32, 2 load128, 3 store128
#define EMU_GSTAT(off, kind) emu_gstat((off), (kind))
#else
#define EMU_GSTAT(off, kind) do { } while (0)
#endif
// ---- the memory primitives of madsim_amd/csrc/kernel/k_mem.h, host form ---------------------------------------------------
namespace madsim_k {
#define SMEM emu_smem
struct BufRef { uint8_t* base; };
static inline BufRef buf_make(const void* base, uint64_t) { return BufRef{(uint8_t*)base}; }
#ifdef MADSIM_EMU_SITES     // tools/mem_site_model.py: which machine-level site (= inlining context) issues each global access
void emu_site(int kind, const void* base, uint32_t off);
#define EMU_SITE(kind, b, off) emu_site((kind), (b).base, (off))
#else
#define EMU_SITE(kind, b, off) do { } while (0)
#endif
static inline uint32_t buf_load32(const BufRef& b, uint32_t off) { EMU_SITE(0, b, off); return *(const uint32_t*)(b.base + off); }
static inline void buf_store32(const BufRef& b, uint32_t off, uint32_t v) { EMU_SITE(1, b, off); *(uint32_t*)(b.base + off) = v; }
static inline void buf_add32(const BufRef& b, uint32_t off, uint32_t v) { EMU_SITE(1, b, off); *(uint32_t*)(b.base + off) += v; }
static inline uint2 buf_load64(const BufRef& b, uint32_t off) { EMU_SITE(0, b, off); return *(const uint2*)(b.base + off); }
static inline void buf_store64(const BufRef& b, uint32_t off, const uint2& e) { EMU_SITE(1, b, off); *(uint2*)(b.base + off) = e; }
static inline uint4 buf_load128(const BufRef& b, uint32_t off) { EMU_SITE(2, b, off); return *(const uint4*)(b.base + off); }
static inline void buf_store128(const BufRef& b, uint32_t off, const uint4& e) { EMU_SITE(3, b, off); *(uint4*)(b.base + off) = e; }
template <int K_> static inline uint64_t rotl64(uint64_t x) { return (x << K_) | (x >> (64 - K_)); }
static inline uint64_t add64_1(uint64_t a, uint64_t b) { return a + b; }
template <int K_> static inline uint64_t shl64(uint64_t x) { return x << K_; }
template <int K_> static inline uint64_t mul_pow2p1(uint64_t v) { return (v << K_) + v; }
static inline uint64_t xor3_64(uint64_t a, uint64_t b, uint64_t c) { return a ^ b ^ c; }
static inline void wave_set_priority(uint32_t) {}
static inline uint32_t wave_uniform(uint32_t v) { return v; }
static inline uint32_t wave_max_u32(uint32_t v) { return v; }
static inline bool wave_first_lane() { return true; }
static inline void atomic_max_u32(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
static inline bool wave_all(bool p) { return p; }                        // (one emulated lane at a time: results may not depend on the vote)
static inline uint32_t table_copy_first() { return 0; }                   // emulated threads run one after another:
static inline uint32_t table_copy_stride(uint32_t) { return 1; }          // each copies everything
}
#ifdef MADSIM_EMU_REGIONS   // tools/divergence_model.py: per-iteration code-region visit counts of each emulated lane
void emu_region(int id);
#define REG(id) emu_region(id)
#endif
#endif
