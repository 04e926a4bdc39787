// tools/ubench_issue.hip — issue-rate ceilings of gfx950 (MI355X) by WALL TIME, for the roofline of sim_kernel.
//
// sim_kernel is LDS-resident integer work: its binding bound is instruction issue, not HBM.  This
// microbenchmark measures what the chip sustains, as wave-instructions per second over HIP-event time:
//   (1) VALU: long unrolled chains of one instruction kind, and of the kernel's own mix (xor / add / alignbit / bitop3 /
//       mul_lo / mul_hi / lshl_add_u64 / cndmask / cmp), 8 independent accumulators per lane, at N = 1,2,3,4,6,8 waves
//       per SIMD on every CU.  Occupancy is pinned with dynamic LDS: 256-thread workgroups (one wave per SIMD), an LDS
//       allocation of which exactly N fit in a CU's 160 KiB, grid = CUs x N, so every workgroup is resident at once.
//   (2) SALU alone and VALU + SALU interleaved in the kernel's ratio (670 k : 304 k per wave): one scalar unit per CU.
//   (3) xoshiro256++ rejection draws written in C++ exactly as kernel/k_rng.h does (a DEPENDENT chain per lane): the
//       real code's draws per second.
//   (4) attainable HBM bandwidth: a float4 copy and a float4 read of 1 GiB (SURVEY.md §8d asks for the copy peak
//       beside the 8 TB/s spec).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o tools/ubench_issue && tools/ubench_issue
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

extern __shared__ uint32_t lds_pin[];

// One loop trip = 16 x INSTS wave-instructions (+ two s_mov that set the cndmask mask: not counted, < 3 %).  Operands: %0-%3 32-bit, %4-%7 64-bit accumulators.
#define VALU_KERNEL(name, INSTS, asmstr)                                                                              \
    constexpr int insts_##name = (INSTS) * 16;                                                                        \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t trips, uint32_t seed) {                   \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55u, a3 = a0 + 7;                              \
        uint64_t q0 = a0 * 11ull + 3, q1 = a1 * 13ull + 5, q2 = (uint64_t)seed << 33 | a2, q3 = a3 * 17ull;           \
        for (uint32_t i = 0; i < trips; i++) {                                                                        \
            asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n" REP16(asmstr) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) \
                         : "s"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "scc");                       \
        }                                                                                                             \
        if (a0 + a1 + a2 + a3 + q0 + q1 + q2 + q3 == 12345) { out[0] = 1; lds_pin[threadIdx.x] = 1; }                 \
    }

VALU_KERNEL(add_u32, 4, "v_add_u32 %0, %1, %0\n v_add_u32 %2, %3, %2\n v_add_u32 %1, %0, %1\n v_add_u32 %3, %2, %3\n")
VALU_KERNEL(xor_b32, 4, "v_xor_b32 %0, %1, %0\n v_xor_b32 %2, %3, %2\n v_xor_b32 %1, %0, %1\n v_xor_b32 %3, %2, %3\n")
VALU_KERNEL(alignbit, 4, "v_alignbit_b32 %0, %1, %0, 9\n v_alignbit_b32 %2, %3, %2, 9\n v_alignbit_b32 %1, %0, %1, 7\n v_alignbit_b32 %3, %2, %3, 7\n")
VALU_KERNEL(bitop3, 4, "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96\n v_bitop3_b32 %3, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %1, %0, %3, %1 bitop3:0x96\n v_bitop3_b32 %2, %0, %3, %2 bitop3:0x96\n")
VALU_KERNEL(mul_lo, 4, "v_mul_lo_u32 %0, %1, %0\n v_mul_lo_u32 %2, %3, %2\n v_mul_lo_u32 %1, %0, %1\n v_mul_lo_u32 %3, %2, %3\n")
VALU_KERNEL(mul_hi, 4, "v_mul_hi_u32 %0, %1, %0\n v_mul_hi_u32 %2, %3, %2\n v_mul_hi_u32 %1, %0, %1\n v_mul_hi_u32 %3, %2, %3\n")
VALU_KERNEL(mad_u64_u32, 4, "v_mad_u64_u32 %4, vcc, %0, %1, %4\n v_mad_u64_u32 %5, vcc, %2, %3, %5\n v_mad_u64_u32 %6, vcc, %0, %3, %6\n v_mad_u64_u32 %7, vcc, %2, %1, %7\n")
VALU_KERNEL(lshl_add_u64, 4, "v_lshl_add_u64 %4, %5, 2, %4\n v_lshl_add_u64 %6, %7, 2, %6\n v_lshl_add_u64 %5, %4, 1, %5\n v_lshl_add_u64 %7, %6, 1, %7\n")
VALU_KERNEL(cndmask, 4, "v_cndmask_b32 %0, %1, %0, s[20:21]\n v_cndmask_b32 %2, %3, %2, s[20:21]\n v_cndmask_b32 %1, %0, %1, s[20:21]\n v_cndmask_b32 %3, %2, %3, s[20:21]\n")
VALU_KERNEL(cmp_u32, 4, "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 s[22:23], %2, %3\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 s[22:23], %3, %0\n")
VALU_KERNEL(cmp_u64, 4, "v_cmp_lt_u64 vcc, %4, %5\n v_cmp_lt_u64 s[22:23], %6, %7\n v_cmp_lt_u64 vcc, %5, %6\n v_cmp_lt_u64 s[22:23], %7, %4\n")
// more instruction kinds (which VALU ops issue at the full rate?)
VALU_KERNEL(mov_b32, 4, "v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %1, %0\n v_mov_b32 %3, %2\n")
VALU_KERNEL(mov_b64, 4, "v_mov_b64 %4, %5\n v_mov_b64 %6, %7\n v_mov_b64 %5, %4\n v_mov_b64 %7, %6\n")
VALU_KERNEL(lshlrev_b32, 4, "v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %2, 5, %3\n v_lshrrev_b32 %1, 3, %0\n v_lshrrev_b32 %3, 5, %2\n")
VALU_KERNEL(lshlrev_b64, 4, "v_lshlrev_b64 %4, 5, %5\n v_lshlrev_b64 %6, 5, %7\n v_lshrrev_b64 %5, 3, %4\n v_lshrrev_b64 %7, 3, %6\n")
VALU_KERNEL(lshl_add_u32, 4, "v_lshl_add_u32 %0, %1, 2, %0\n v_lshl_add_u32 %2, %3, 2, %2\n v_lshl_add_u32 %1, %0, 1, %1\n v_lshl_add_u32 %3, %2, 1, %3\n")
VALU_KERNEL(lshl_or_b32, 4, "v_lshl_or_b32 %0, %1, 2, %0\n v_lshl_or_b32 %2, %3, 2, %2\n v_lshl_or_b32 %1, %0, 1, %1\n v_lshl_or_b32 %3, %2, 1, %3\n")
VALU_KERNEL(add3_u32, 4, "v_add3_u32 %0, %1, %2, %0\n v_add3_u32 %3, %1, %2, %3\n v_add3_u32 %1, %0, %3, %1\n v_add3_u32 %2, %0, %3, %2\n")
VALU_KERNEL(and_or_b32, 4, "v_and_or_b32 %0, %1, %2, %0\n v_and_or_b32 %3, %1, %2, %3\n v_and_or_b32 %1, %0, %3, %1\n v_and_or_b32 %2, %0, %3, %2\n")
VALU_KERNEL(bfe_u32, 4, "v_bfe_u32 %0, %1, 3, 8\n v_bfe_u32 %2, %3, 5, 8\n v_bfe_u32 %1, %0, 1, 9\n v_bfe_u32 %3, %2, 1, 9\n")
VALU_KERNEL(perm_b32, 4, "v_perm_b32 %0, %1, %0, %2\n v_perm_b32 %3, %1, %3, %2\n v_perm_b32 %1, %0, %1, %3\n v_perm_b32 %2, %0, %2, %3\n")
VALU_KERNEL(add_co, 4, "v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, %3, %2, vcc\n v_add_co_u32 %1, vcc, %0, %1\n v_addc_co_u32 %3, vcc, %2, %3, vcc\n")
VALU_KERNEL(mul_u24, 4, "v_mul_u32_u24 %0, %1, %0\n v_mul_u32_u24 %2, %3, %2\n v_mul_u32_u24 %1, %0, %1\n v_mul_u32_u24 %3, %2, %3\n")
VALU_KERNEL(mad_u32_u24, 4, "v_mad_u32_u24 %0, %1, %2, %0\n v_mad_u32_u24 %3, %1, %2, %3\n v_mad_u32_u24 %1, %0, %3, %1\n v_mad_u32_u24 %2, %0, %3, %2\n")
VALU_KERNEL(cndmask_vcc, 4, "v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %3, %2, vcc\n v_cndmask_b32 %1, %0, %1, vcc\n v_cndmask_b32 %3, %2, %3, vcc\n")
VALU_KERNEL(cmp_vcc, 4, "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_eq_u32 vcc, %1, %2\n v_cmp_ne_u32 vcc, %3, %0\n")
VALU_KERNEL(min_u32, 4, "v_min_u32 %0, %1, %0\n v_max_u32 %2, %3, %2\n v_min_u32 %1, %0, %1\n v_max_u32 %3, %2, %3\n")
VALU_KERNEL(sub_u32, 4, "v_sub_u32 %0, %1, %0\n v_subrev_u32 %2, %3, %2\n v_sub_u32 %1, %0, %1\n v_subrev_u32 %3, %2, %3\n")
VALU_KERNEL(alignbyte, 4, "v_alignbyte_b32 %0, %1, %0, 1\n v_alignbyte_b32 %2, %3, %2, 1\n v_alignbyte_b32 %1, %0, %1, 3\n v_alignbyte_b32 %3, %2, %3, 3\n")
VALU_KERNEL(xad_u32, 4, "v_xad_u32 %0, %1, %2, %0\n v_xad_u32 %3, %1, %2, %3\n v_xad_u32 %1, %0, %3, %1\n v_xad_u32 %2, %0, %3, %2\n")
VALU_KERNEL(or3_b32, 4, "v_or3_b32 %0, %1, %2, %0\n v_or3_b32 %3, %1, %2, %3\n v_or3_b32 %1, %0, %3, %1\n v_or3_b32 %2, %0, %3, %2\n")
VALU_KERNEL(xor_sgpr, 4, "v_xor_b32 %0, s20, %0\n v_xor_b32 %2, s21, %2\n v_xor_b32 %1, s20, %1\n v_xor_b32 %3, s21, %3\n")
VALU_KERNEL(readlane, 4, "v_readlane_b32 s22, %0, 3\n v_readlane_b32 s23, %1, 5\n v_readfirstlane_b32 s24, %2\n v_readfirstlane_b32 s25, %3\n")
// the executor kernel's mix (static count of its hot loop: ~40 % logic/add, 15 % alignbit + bitop3, 12 % multiplies,
// 8 % 64-bit adds, 15 % cndmask, 10 % compares): 20 instructions
VALU_KERNEL(mix, 20,
    "v_xor_b32 %0, %1, %0\n v_add_u32 %2, %3, %2\n v_alignbit_b32 %1, %0, %1, 9\n v_bitop3_b32 %3, %1, %2, %3 bitop3:0x96\n"
    "v_mul_lo_u32 %0, %1, %0\n v_lshl_add_u64 %4, %5, 0, %4\n v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %2, %3, %2, vcc\n"
    "v_and_b32 %1, %0, %1\n v_add_u32 %3, %2, %3\n v_mul_hi_u32 %2, %3, %2\n v_cndmask_b32 %0, %1, %0, s[20:21]\n"
    "v_xor_b32 %1, %2, %1\n v_lshrrev_b32 %3, 3, %3\n v_alignbit_b32 %0, %3, %0, 19\n v_cmp_lt_u64 s[22:23], %4, %5\n"
    "v_cndmask_b32 %3, %0, %3, s[22:23]\n v_or_b32 %2, %1, %2\n v_bitop3_b32 %1, %0, %2, %1 bitop3:0x96\n v_add_u32 %0, %1, %0\n")
// SALU alone: 4 instructions
VALU_KERNEL(salu, 4, "s_add_u32 s22, s22, %8\n s_and_b32 s23, s23, %8\n s_or_b64 s[24:25], s[24:25], s[22:23]\n s_lshl_b32 s22, s22, 1\n")
// VALU + SALU in the kernel's ratio (2.2 : 1): 9 VALU + 4 SALU = 13
VALU_KERNEL(valu_salu, 13,
    "v_xor_b32 %0, %1, %0\n s_add_u32 s22, s22, %8\n v_add_u32 %2, %3, %2\n v_alignbit_b32 %1, %0, %1, 9\n s_and_b32 s23, s23, %8\n"
    "v_bitop3_b32 %3, %1, %2, %3 bitop3:0x96\n v_mul_lo_u32 %0, %1, %0\n s_or_b64 s[24:25], s[24:25], s[22:23]\n v_lshl_add_u64 %4, %5, 0, %4\n"
    "v_cndmask_b32 %2, %3, %2, s[20:21]\n s_lshl_b32 s22, s22, 1\n v_add_u32 %3, %2, %3\n v_xor_b32 %1, %2, %1\n")

// (3) the kernel's RNG as written in kernel/k_rng.h: xoshiro256++ next_u64 + rand 0.8's accept test, a dependent chain per lane
__device__ __forceinline__ uint64_t rotl64c(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__global__ __launch_bounds__(256) void k_xoshiro(uint32_t* out, uint32_t trips, uint32_t seed) {
    uint64_t s0 = seed * 0x9e3779b97f4a7c15ull + threadIdx.x, s1 = s0 ^ 0xbf58476d1ce4e5b9ull, s2 = s1 * 3 + 1, s3 = s2 ^ (s0 >> 7);
    uint32_t acc = 0;
    for (uint32_t i = 0; i < trips; i++) {
#pragma unroll 4
        for (int k = 0; k < 16; k++) {                       // one draw: the straight-line body of a rejection-loop trip
            const uint64_t r = rotl64c(s0 + s3, 23) + s0, t = s1 << 17;
            s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl64c(s3, 45);
            acc += ((uint32_t)(r >> 32) * 5u + __umulhi((uint32_t)r, 5u) > 0x9fffffffu) ? 1u : 0u;
        }
    }
    if (acc == 0xdeadbeefu) { out[0] = acc; lds_pin[threadIdx.x] = 1; }
}

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ src, float* out, size_t n) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_fill(float4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

static int g_cus = 256;
static uint32_t* g_out;

// LDS bytes per 256-thread workgroup of which exactly n fit into 160 KiB
static uint32_t lds_for(int n) { uint32_t b = 163840u / (uint32_t)(n + 1) + 1024u; return (b + 255u) & ~255u; }

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());                                  // warm
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);
// chip-wide wave-instructions per second of kernel k at n waves per SIMD
static double issue_rate(kern_t k, int insts_per_trip, uint32_t trips, int n) {
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    const uint32_t lds = lds_for(n);
    const double ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(g_cus * n), dim3(256), lds, 0, g_out, trips, 1u); });
    return (double)g_cus * n * 4 * trips * insts_per_trip / (ms * 1e-3);
}
static void run_issue(const char* name, kern_t k, int insts_per_trip, uint32_t trips, double valu_share) {
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    printf("%-14s", name);
    for (int n : {1, 2, 3, 4, 6, 8}) {
        const uint32_t lds = lds_for(n);
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k, 256, lds));
        const double ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(g_cus * n), dim3(256), lds, 0, g_out, trips, 1u); });
        const double winst = (double)g_cus * n * 4 * trips * insts_per_trip;         // wave-instructions
        const double gps = winst / (ms * 1e-3) / 1e9;
        // cycles one SIMD spends per wave-instruction if it ran at 2.4 GHz: SIMDs x 2.4e9 / rate
        printf("  N=%d%s %8.1f G/s (%.2f cyc)", n, occ == n ? "" : "!", gps, (double)g_cus * 4 * 2.4 / gps);
    }
    printf("\n");
    (void)valu_share;
}

// --quick: one JSON line for bench.py's roofline (the kernel-mix VALU ceiling, the SALU ceiling, the copy bandwidth), ~1 s
static int quick() {
    double best_mix = 0, best_salu = 0, best_vs = 0; int n_mix = 0;
    printf("{\"cus\": %d, \"mix_ginst_s\": {", g_cus);
    bool first = true;
    for (int n : {1, 2, 3, 4, 6, 8}) {
        const double r = issue_rate(k_mix, insts_mix, 4000, n) / 1e9;
        printf("%s\"%d\": %.1f", first ? "" : ", ", n, r); first = false;
        if (r > best_mix) { best_mix = r; n_mix = n; }
    }
    for (int n : {2, 4, 8}) { double r = issue_rate(k_salu, insts_salu, 4000, n) / 1e9; if (r > best_salu) best_salu = r; }
    for (int n : {3, 4, 8}) { double r = issue_rate(k_valu_salu, insts_valu_salu, 4000, n) / 1e9; if (r > best_vs) best_vs = r; }
    const size_t bytes = 1ull << 28, n = bytes / 16;
    float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    hipLaunchKernelGGL(k_fill, dim3(g_cus * 8), dim3(256), 0, 0, a, n); CK(hipDeviceSynchronize());
    double best_copy = 0;
    for (int mult : {8, 16}) {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(g_cus * mult), dim3(256), 0, 0, a, b, n); });
        const double g = 2.0 * bytes / (ms * 1e-3) / 1e9; if (g > best_copy) best_copy = g;
    }
    printf("}, \"valu_mix_ceiling_ginst_s\": %.1f, \"valu_mix_best_waves_per_simd\": %d, \"salu_ceiling_ginst_s\": %.1f, "
           "\"valu_salu_interleaved_ginst_s\": %.1f, \"hbm_copy_gbps\": %.1f}\n", best_mix, n_mix, best_salu, best_vs, best_copy);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    g_cus = p.multiProcessorCount;
    if (argc > 1 && std::string(argv[1]) == "--quick") { CK(hipMalloc(&g_out, 64)); return quick(); }
    printf("# %s, %d CUs, clockRate %d kHz, LDS/CU %zu\n", p.gcnArchName, g_cus, p.clockRate, (size_t)p.maxSharedMemoryPerMultiProcessor);
    printf("# wave-instructions per second over HIP-event time, whole chip; (cyc) = SIMD cycles per wave-instruction at 2.4 GHz\n");
    printf("# = CUs x 4 SIMDs x 2.4e9 / rate.  N = waves per SIMD (pinned with dynamic LDS; '!' = occupancy query disagrees)\n");
    CK(hipMalloc(&g_out, 64));
    const uint32_t T = 20000;
#define RUN(name) run_issue(#name, k_##name, insts_##name, T, 1.0)
    RUN(add_u32); RUN(xor_b32); RUN(alignbit); RUN(bitop3); RUN(mul_lo); RUN(mul_hi); RUN(mad_u64_u32); RUN(lshl_add_u64);
    RUN(cndmask); RUN(cmp_u32); RUN(cmp_u64); RUN(mix); RUN(salu); RUN(valu_salu);
    RUN(mov_b32); RUN(mov_b64); RUN(lshlrev_b32); RUN(lshlrev_b64); RUN(lshl_add_u32); RUN(lshl_or_b32); RUN(add3_u32); RUN(and_or_b32);
    RUN(bfe_u32); RUN(perm_b32); RUN(add_co); RUN(mul_u24); RUN(mad_u32_u24); RUN(cndmask_vcc); RUN(cmp_vcc); RUN(min_u32); RUN(sub_u32);
    RUN(alignbyte); RUN(xad_u32); RUN(or3_b32); RUN(xor_sgpr); RUN(readlane);
    printf("# xoshiro: draws (one rejection-loop trip of kernel/k_rng.h: next_u64 + accept test) per second per chip, G draws/s\n");
    {
        CK(hipFuncSetAttribute((const void*)k_xoshiro, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        printf("%-14s", "xoshiro_draw");
        for (int n : {1, 2, 3, 4, 6, 8}) {
            const uint32_t lds = lds_for(n), trips = 4000;
            const double ms = time_ms([&] { hipLaunchKernelGGL(k_xoshiro, dim3(g_cus * n), dim3(256), lds, 0, g_out, trips, 1u); });
            const double draws = (double)g_cus * n * 4 * trips * 16;           // wave-draws
            printf("  N=%d %8.2f G wave-draws/s", n, draws / (ms * 1e-3) / 1e9);
        }
        printf("\n");
    }
    // (4) attainable HBM bandwidth
    {
        const size_t bytes = 1ull << 30, n = bytes / 16;
        float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        hipLaunchKernelGGL(k_fill, dim3(g_cus * 8), dim3(256), 0, 0, a, n); CK(hipDeviceSynchronize());
        for (int mult : {4, 8, 16, 32}) {
            const double msc = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(g_cus * mult), dim3(256), 0, 0, a, b, n); });
            const double msr = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(g_cus * mult), dim3(256), 0, 0, a, (float*)g_out, n); });
            printf("hbm  grid=%5d x256  float4 copy 1 GiB -> 1 GiB: %7.1f GB/s (read+write)   float4 read 1 GiB: %7.1f GB/s\n",
                   g_cus * mult, 2.0 * bytes / (msc * 1e-3) / 1e9, (double)bytes / (msr * 1e-3) / 1e9);
        }
        const double msm = time_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
        printf("hbm  hipMemcpy D2D 1 GiB: %7.1f GB/s (read+write)\n", 2.0 * bytes / (msm * 1e-3) / 1e9);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
