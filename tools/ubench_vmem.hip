// tools/ubench_vmem.hip — what a divergent global-memory instruction costs on gfx950 (MI355X), by wall time.
//
// The global-state builds of sim_kernel (configs[2]-[4] shapes) execute ~100 buffer loads and ~60 stores per wave-iteration,
// most of them with few active lanes that each touch their own cache line.  This microbenchmark measures the rate at which a
// CU retires such instructions, so the kernel's memory-instruction count can be priced:
//   pattern   = how the 64 lanes of a wave-instruction spread over memory (coalesced 16 B per lane; one 128-B line per
//               lane; sibling pairs; ...), with K of the 64 lanes active
//   footprint = the region the addresses wrap in: 16 MiB (L2-resident: the address / tag pipeline alone) or 2 GiB (HBM)
//   mode      = independent loads (8 in flight per wave: throughput) or a dependent chain (each address from the previous
//               load: latency per round trip)
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/ubench_vmem.hip -o tools/ubench_vmem && tools/ubench_vmem
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// lane l of wave w in iteration i reads 16 (or 4) bytes at
//   ((w * wave_stride + i * iter_stride + (l / group) * lane_stride + (l % group) * 16) mod footprint)
template <bool WIDE>
__global__ __launch_bounds__(256) void k_tput(const uint8_t* base, uint64_t footprint_mask, uint32_t lane_stride, uint32_t iter_stride,
                                              uint32_t wave_stride, uint32_t active, uint32_t iters, uint32_t group, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lane % (64 / active) != 0) return;                     // `active` lanes, evenly spread over the wave
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, 0xffffffffu, 0x00020000);
    uint64_t off = (uint64_t)wv * wave_stride + (uint64_t)(lane / group) * lane_stride + (lane % group) * 16u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {                          // eight independent loads in flight
            const uint32_t a = (uint32_t)((off + (uint64_t)(i + k) * iter_stride) & footprint_mask);
            if (WIDE) { u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, a, 0, 0); acc += v.x ^ v.w; }
            else acc += __builtin_amdgcn_raw_buffer_load_b32(rs, a, 0, 0);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// The global-state layout of sim_kernel: logical unit u of global lane g lives at u * plane_bytes + g * 16.  Every lane picks
// its own pseudo-random unit in [0, units) each iteration (what a wave of seeds in different phases does); `active` lanes of the
// wave take part, either spread over the wave (one per quad ...) or packed into its first lanes.
__global__ __launch_bounds__(256) void k_plane(const uint8_t* base, uint32_t plane_bytes, uint32_t units, uint32_t active, uint32_t packed,
                                               uint32_t iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (packed ? lane >= active : lane % (64 / active) != 0) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, 0xffffffffu, 0x00020000);
    const uint32_t g = wv * 64 + lane;
    uint32_t h = g * 2654435761u + 12345u, acc = 0;
    for (uint32_t i = 0; i < iters; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            h = h * 1664525u + 1013904223u;
            const uint32_t u = (uint32_t)(((uint64_t)(h >> 8) * units) >> 24);
            // plane_bytes == 0: the wave-blocked layout [wave][unit][lane of the wave] — a wave's whole state in units x 1 KiB
            const uint32_t a = plane_bytes ? u * plane_bytes + g * 16u : wv * (units * 1024u) + u * 1024u + lane * 16u;
            u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, a, 0, 0);
            acc += v.x ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// dependent chain: the next address comes out of the loaded word (memory holds a permutation of line indices)
__global__ __launch_bounds__(256) void k_chain(const uint32_t* next, uint32_t n_lines, uint32_t active, uint32_t iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lane % (64 / active) != 0) return;
    uint32_t p = (wv * 64 + lane) * 97u % n_lines;
    for (uint32_t i = 0; i < iters; i++) p = next[(size_t)p * 32];          // one word per 128-B line
    if (p == 0xffffffffu) out[0] = p;
}

static double time_ms(void (*launch)(void*), void* arg) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(arg); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 2; r++) {
        CK(hipEventRecord(e0, 0)); launch(arg); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

struct TArgs { const uint8_t* base; uint64_t mask; uint32_t ls, is, ws, active, iters, grid; bool wide; uint32_t* out; uint32_t group; };
static void launch_t(void* p) {
    TArgs* a = (TArgs*)p;
    if (a->wide) hipLaunchKernelGGL(k_tput<true>, dim3(a->grid), dim3(256), 0, 0, a->base, a->mask, a->ls, a->is, a->ws, a->active, a->iters, a->group, a->out);
    else hipLaunchKernelGGL(k_tput<false>, dim3(a->grid), dim3(256), 0, 0, a->base, a->mask, a->ls, a->is, a->ws, a->active, a->iters, a->group, a->out);
}
struct PArgs { const uint8_t* base; uint32_t plane, units, active, packed, iters, grid; uint32_t* out; };
static void launch_p(void* p) { PArgs* a = (PArgs*)p; hipLaunchKernelGGL(k_plane, dim3(a->grid), dim3(256), 0, 0, a->base, a->plane, a->units, a->active, a->packed, a->iters, a->out); }
struct CArgs { const uint32_t* next; uint32_t n_lines, active, iters, grid; uint32_t* out; };
static void launch_c(void* p) { CArgs* a = (CArgs*)p; hipLaunchKernelGGL(k_chain, dim3(a->grid), dim3(256), 0, 0, a->next, a->n_lines, a->active, a->iters, a->out); }

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t big = 2ull << 30;
    uint8_t* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
    uint32_t* out; CK(hipMalloc(&out, 64));
    printf("# %s, %d CUs.  Rates are wave-instructions per microsecond PER CU (x %d CUs x 1e6 = per second per chip);\n", prop.gcnArchName, cus, cus);
    printf("# 'cyc' = CU cycles per wave-instruction at 2.4 GHz.  W = waves per SIMD (workgroups of 4 waves per CU).\n");
    struct Pat { const char* name; uint32_t lane_stride, iter_stride, wave_stride; bool wide; uint32_t group; };
    const Pat pats[] = {
        {"b128 coalesced: 16 B per lane, 1 KiB per wave (8 lines)", 16, 1024 * 1031, 1024, true, 1},
        {"b128 one 128-B line per lane (64 lines)", 128 * 257, 8192 * 1031, 128, true, 1},
        {"b128 lane pairs share a 32-B sector (32 lines)", 128 * 257, 8192 * 1031, 128, true, 2},
        {"b128 quads share 64 B (16 lines)", 128 * 257, 8192 * 1031, 128, true, 4},
        {"b128 [slot][lane]-like: 8 lanes per line, lanes agree", 16, 1048576 + 1024, 1024, true, 1},
        {"b32  coalesced: 4 B per lane (2 lines)", 4, 256 * 4099, 256, false, 1},
        {"b32  one line per lane (64 lines)", 128 * 257, 8192 * 1031, 128, false, 1},
    };
    for (uint64_t fp : {(uint64_t)16 << 20, (uint64_t)2 << 30}) {
        printf("== independent loads (8 in flight per wave), footprint %llu MiB\n", (unsigned long long)(fp >> 20));
        for (const Pat& p : pats) {
            for (uint32_t active : {64u, 16u, 4u}) {
                printf("%-58s K=%2u:", p.name, active);
                for (int W : {1, 3, 6}) {
                    TArgs a{buf, fp - 1, p.lane_stride, p.iter_stride, p.wave_stride, active, 4096, (uint32_t)(cus * W), p.wide, out, p.group};
                    const double ms = time_ms(launch_t, &a);
                    const double winst = (double)cus * W * 4 * a.iters;
                    const double per_cu_us = winst / cus / (ms * 1e3);
                    printf("  W=%d %6.1f/us (%5.1f cyc)", W, per_cu_us, 2400.0 / per_cu_us);
                }
                printf("\n");
            }
        }
    }
    // the kernel's own layout: does the plane stride (a power of two: 65 536 lanes x 16 B = 1 MiB) alias memory channels?
    printf("== [unit][lane] planes, each lane on its own pseudo-random unit of 168 (the election loop's state block), b128\n");
    for (uint32_t plane : {1u << 20, (1u << 20) + 640u, 0u}) {
        for (uint32_t packed : {0u, 1u}) {
            for (uint32_t active : {64u, 16u, 4u}) {
                if (packed && active == 64) continue;
                if (plane) printf("plane %7u B, %2u lanes %-6s:", plane, active, packed ? "packed" : "spread");
                else printf("[wave][unit][lane],%2u lanes %-6s:", active, packed ? "packed" : "spread");
                for (int W : {1, 3, 6}) {
                    PArgs a{buf, plane, 168, active, packed, 4096, (uint32_t)(cus * W), out};
                    const double ms = time_ms(launch_p, &a);
                    const double per_cu_us = (double)W * 4 * a.iters / (ms * 1e3);
                    printf("  W=%d %6.1f/us (%5.1f cyc)", W, per_cu_us, 2400.0 / per_cu_us);
                }
                printf("\n");
            }
        }
    }
    // dependent chains: a random cyclic permutation of lines
    for (uint64_t fp : {(uint64_t)16 << 20, (uint64_t)512 << 20, (uint64_t)2 << 30}) {
        const uint32_t n_lines = (uint32_t)(fp / 128);
        uint32_t* h = (uint32_t*)malloc((size_t)n_lines * 4);
        for (uint32_t i = 0; i < n_lines; i++) h[i] = i;
        uint64_t s = 88172645463325252ull;
        for (uint32_t i = n_lines - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; uint32_t j = (uint32_t)(s % i); uint32_t t = h[i]; h[i] = h[j]; h[j] = t; }   // Sattolo: one cycle
        uint32_t* hn = (uint32_t*)calloc((size_t)n_lines * 32, 4);
        for (uint32_t i = 0; i < n_lines; i++) hn[(size_t)i * 32] = h[i];
        CK(hipMemcpy(buf, hn, (size_t)n_lines * 128, hipMemcpyHostToDevice));
        free(h); free(hn);
        printf("== dependent chain (one load per round trip), footprint %llu MiB: ns per round trip as seen by a wave\n", (unsigned long long)(fp >> 20));
        for (uint32_t active : {64u, 16u, 1u}) {
            printf("K=%2u:", active);
            for (int W : {1, 3, 6}) {
                CArgs a{(const uint32_t*)buf, n_lines, active, 2000, (uint32_t)(cus * W), out};
                const double ms = time_ms(launch_c, &a);
                printf("  W=%d %7.1f ns/RT (%6.1f M wave-RT/s per CU)", W, ms * 1e6 / a.iters, (double)W * 4 * a.iters / (ms * 1e3));
            }
            printf("\n");
        }
    }
    return 0;
}
