// lds_occupancy.hip -- how many one-wave workgroups with S bytes of LDS are resident per CU on gfx950 (what k_raster_c's LDS
// footprint buys): each block registers on its CU (HW_ID), spins, and the peak of the per-CU counters is reported next to the
// occupancy API's answer.   build: hipcc --offload-arch=gfx950 -O2 lds_occupancy.hip -o lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(64) void k(unsigned* active, unsigned* peak, int spin) {
    extern __shared__ float lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // HW_ID: [11:8] CU id, [14:12] SH/SE bits ... use (xcc, se, cu) as a key
    const unsigned key = ((xcc & 15u) << 8) | ((hw >> 8) & 0xffu);
    if (threadIdx.x == 0) {
        unsigned v = atomicAdd(&active[key], 1u) + 1u;
        atomicMax(&peak[key], v);
    }
    lds[threadIdx.x] = (float)spin;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) atomicSub(&active[key], 1u);
    if (lds[threadIdx.x] < 0) active[0] = 0;
}
int main() {
    unsigned *active, *peak;
    hipMalloc(&active, 4096 * 4); hipMalloc(&peak, 4096 * 4);
    for (int S = 4096; S <= 11264; S += 256) {
        hipMemset(active, 0, 4096 * 4); hipMemset(peak, 0, 4096 * 4);
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, S);
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64, S);
        hipLaunchKernelGGL(k, dim3(256 * 48), dim3(64), S, 0, active, peak, 200000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(4096);
        hipMemcpy(h.data(), peak, 4096 * 4, hipMemcpyDeviceToHost);
        unsigned mx = 0, mn = 1u << 30; int n = 0; unsigned long long sum = 0;
        for (unsigned v : h) if (v) { mx = std::max(mx, v); mn = std::min(mn, v); sum += v; ++n; }
        printf("LDS %5d B: occupancy API %2d blocks/CU; measured peak resident waves per CU: min %u max %u mean %.2f over %d CUs\n", S, nb, mn, mx, n ? (double)sum / n : 0.0, n);
    }
    return 0;
}
