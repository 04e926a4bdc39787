// Per-instruction issue-cost microbenchmark for gfx950: one wave per SIMD, long unrolled chains of one
// instruction kind (independent destinations, 8 accumulators), cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define BENCH(name, asmstr)                                                              \
  __global__ void k_##name(uint64_t* out, uint32_t seed) {                               \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 ^ 5, a3 = a0 + 7;             \
    uint64_t q0 = a0 * 11ull, q1 = a1 * 13ull, q2 = seed, q3 = a3;                       \
    uint64_t t0 = __builtin_readcyclecounter();                                          \
    for (int i = 0; i < 64; i++) { asm volatile(REP64(asmstr) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "s"(seed) : "vcc"); } \
    uint64_t t1 = __builtin_readcyclecounter();                                          \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                            \
    if (a0 + a1 + a2 + a3 + q0 + q1 + q2 + q3 == 12345) out[1] = 1;                       \
  }
BENCH(add_u32,      "v_add_u32 %0, %1, %0\n v_add_u32 %2, %3, %2\n")
BENCH(xor_b32,      "v_xor_b32 %0, %1, %0\n v_xor_b32 %2, %3, %2\n")
BENCH(alignbit,     "v_alignbit_b32 %0, %1, %0, 9\n v_alignbit_b32 %2, %3, %2, 9\n")
BENCH(mul_lo,       "v_mul_lo_u32 %0, %1, %0\n v_mul_lo_u32 %2, %3, %2\n")
BENCH(mul_hi,       "v_mul_hi_u32 %0, %1, %0\n v_mul_hi_u32 %2, %3, %2\n")
BENCH(mul_u24,      "v_mul_u32_u24 %0, %1, %0\n v_mul_u32_u24 %2, %3, %2\n")
BENCH(mad_u64_u32,  "v_mad_u64_u32 %4, vcc, %0, %1, %4\n v_mad_u64_u32 %5, vcc, %2, %3, %5\n")
BENCH(lshl_add_u64, "v_lshl_add_u64 %4, %5, 2, %4\n v_lshl_add_u64 %6, %7, 2, %6\n")
BENCH(lshlrev_b64,  "v_lshlrev_b64 %4, 5, %4\n v_lshlrev_b64 %6, 5, %6\n")
BENCH(lshrrev_b64,  "v_lshrrev_b64 %4, 5, %4\n v_lshrrev_b64 %6, 5, %6\n")
BENCH(addco,        "v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, %3, %2, vcc\n")
BENCH(cndmask,      "v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %3, %2, vcc\n")
BENCH(cmp_u64,      "v_cmp_lt_u64 vcc, %4, %5\n v_cmp_lt_u64 vcc, %6, %7\n")
BENCH(cmp_u32,      "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %2, %3\n")
BENCH(mov_b64,      "v_mov_b64 %4, %5\n v_mov_b64 %6, %7\n")
BENCH(bitop3,       "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96\n v_bitop3_b32 %3, %1, %2, %3 bitop3:0x96\n")
BENCH(salu_add,     "s_add_u32 s20, s20, %8\n s_and_b32 s21, s21, %8\n")
BENCH(salu_b64,     "s_or_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[22:23], s[22:23], s[20:21]\n")
BENCH(readlane,     "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n")
BENCH(dep_add,      "v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n")
BENCH(dep_xor_b64,  "v_lshl_add_u64 %4, %4, 1, %4\n v_lshl_add_u64 %4, %4, 1, %4\n")
#define RUN(name) do { hipMemset(d, 0, 16); hipLaunchKernelGGL(k_##name, dim3(blocks), dim3(threads), 0, 0, d, 1u); hipDeviceSynchronize(); \
  hipLaunchKernelGGL(k_##name, dim3(blocks), dim3(threads), 0, 0, d, 1u); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
  printf("%-14s waves/SIMD=%d  %6.2f cycles per instruction (clock64 ticks / (64*64*2))\n", #name, threads/256 ? threads/256 : 1, (double)h[0] / (64.0 * 64 * 2)); } while (0)
int main() {
  uint64_t *d, h[2]; hipMalloc(&d, 16);
  for (int threads : {64, 256, 512, 1024}) {
    int blocks = threads == 64 ? 1024 : 256;
    printf("---- %d threads/block, %d blocks (%s) ----\n", threads, blocks, threads == 64 ? "1 wave per SIMD" : threads == 256 ? "1 wave/SIMD" : threads == 512 ? "2 waves/SIMD" : "4 waves/SIMD");
    RUN(add_u32); RUN(xor_b32); RUN(alignbit); RUN(mul_lo); RUN(mul_hi); RUN(mul_u24); RUN(mad_u64_u32); RUN(lshl_add_u64);
    RUN(lshlrev_b64); RUN(lshrrev_b64); RUN(addco); RUN(cndmask); RUN(cmp_u64); RUN(cmp_u32); RUN(mov_b64); RUN(bitop3);
    RUN(salu_add); RUN(salu_b64); RUN(readlane); RUN(dep_add); RUN(dep_xor_b64);
  }
  return 0;
}
