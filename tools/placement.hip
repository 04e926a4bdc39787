// placement.hip — where does the MI355X workgroup dispatcher put the waves of concurrent launches?
// Each block spins ~spin_us, records HW_ID / XCC_ID and its start/end clock; the host histograms waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/placement tools/placement.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>
struct Rec { unsigned hw, xcc; unsigned long long t0, t1; };
extern __shared__ unsigned smem[];
__global__ void spin(Rec* out, unsigned long long ticks, int base) {
    unsigned long long t0 = wall_clock64();
    smem[threadIdx.x] = threadIdx.x;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if ((threadIdx.x & 63) == 0) { Rec r{hw, xcc, t0, (unsigned long long)wall_clock64()}; out[base + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r; }
}
int main(int argc, char** argv) {
    int nstreams = argc > 1 ? atoi(argv[1]) : 2, lds = argc > 2 ? atoi(argv[2]) : 19360, threads = argc > 3 ? atoi(argv[3]) : 64;
    int waves_per_launch = 1024, blocks = waves_per_launch * 64 / threads;
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Rec* d; hipMalloc(&d, sizeof(Rec) * waves_per_launch * nstreams);
    std::vector<hipStream_t> st(nstreams);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; rep++) {
        for (int i = 0; i < nstreams; i++) hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), lds * (threads / 64), st[i], d, 100000ull * 2, i * waves_per_launch);  // 100 MHz clock: 2 ms
        hipDeviceSynchronize();
    }
    std::vector<Rec> h(waves_per_launch * nstreams);
    hipMemcpy(h.data(), d, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu;
    unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
    for (auto& r : h) {
        unsigned simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7, xcc = r.xcc & 15;
        unsigned cuid = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cuid]++; per_simd[(cuid << 2) | simd]++;
        tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); smax = std::max(smax, r.t0);
    }
    std::map<int, int> hs, hc;
    for (auto& kv : per_simd) hs[kv.second]++;
    for (auto& kv : per_cu) hc[kv.second]++;
    printf("streams %d lds/wave %d threads/block %d: CUs used %zu, SIMDs used %zu, span %.3f ms, last start +%.3f ms\n", nstreams, lds, threads, per_cu.size(), per_simd.size(), (tmax - tmin) / 1e5, (smax - tmin) / 1e5);
    printf("  waves per CU histogram:"); for (auto& kv : hc) printf(" %d:%d", kv.first, kv.second); printf("\n");
    printf("  waves per SIMD histogram:"); for (auto& kv : hs) printf(" %d:%d", kv.first, kv.second); printf("\n");
    return 0;
}
