// dispatch_rate.hip -- how fast gfx950 hands out workgroups: kernels that do (almost) nothing, grids of 256 .. 65,536 workgroups of
// 64 / 256 / 1024 threads, with and without static LDS; time per launch from hipEvents around 20 back-to-back launches.  What the
// front-end kernels of the renderer (2,048 - 65,536 short workgroups each) can and cannot gain from fewer, fatter workgroups.
//   build: hipcc --offload-arch=gfx950 -O2 dispatch_rate.hip -o dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_WORDS>
__global__ void k_empty(unsigned* out, int iters) {
    __shared__ unsigned s[LDS_WORDS > 0 ? LDS_WORDS : 1];
    unsigned v = threadIdx.x;
    if (LDS_WORDS > 0) { s[threadIdx.x % LDS_WORDS] = v; __syncthreads(); v = s[(threadIdx.x + 1) % LDS_WORDS]; }
    for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;         // `iters` dependent multiply-adds per thread
    if (v == 0xdeadbeefu) out[0] = v;
}
template <int LDS_WORDS>
static void run(const char* what, unsigned* out, int iters) {
    const int grids[] = { 256, 1024, 2048, 4096, 16384, 65536 };
    const int blocks[] = { 64, 256, 1024 };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int b : blocks) {
        printf("%-28s block %4d:", what, b);
        for (int g : grids) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_empty<LDS_WORDS>, dim3(g), dim3(b), 0, 0, out, iters);
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty<LDS_WORDS>, dim3(g), dim3(b), 0, 0, out, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("  %6d wg %7.2f us", g, ms * 1000.0f / 20.0f);
        }
        printf("\n");
    }
}
int main() {
    unsigned* out; hipMalloc(&out, 64);
    run<0>("no LDS, empty", out, 0);
    run<512>("2 KB LDS, empty", out, 0);
    run<2048>("8 KB LDS, empty", out, 0);
    run<0>("no LDS, 200 dependent ops", out, 200);
    run<0>("no LDS, 2000 dependent ops", out, 2000);
    return 0;
}
