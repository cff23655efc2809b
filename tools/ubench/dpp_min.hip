#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ float wave_min_c(float x) {
    const int big = __float_as_int(3.0e38f);
    int v = __float_as_int(x);
#define STEP(ctrl, rows) { const int t = __builtin_amdgcn_update_dpp(big, v, ctrl, rows, 0xf, false); v = __float_as_int(fminf(__int_as_float(v), __int_as_float(t))); }
    STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
__device__ __forceinline__ float wave_min_asm(float x, float* lanes) {
    asm volatile("s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1"
                 : "+v"(x));
    lanes[threadIdx.x] = x;
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__global__ void k(const float* in, float* out, float* lanes) {
    float x = in[threadIdx.x];
    out[0] = wave_min_c(x);
    out[1] = wave_min_asm(x, lanes);
}
int main() {
    float h[64], *d, *o, *l; hipMalloc(&d, 256); hipMalloc(&o, 8); hipMalloc(&l, 256);
    for (int t = 0; t < 4; ++t) {
        for (int i = 0; i < 64; ++i) h[i] = -(float)(rand() % 1000) - 1.0f;
        if (t == 1) for (int i = 0; i < 64; ++i) h[i] = -(float)i;
        if (t == 2) for (int i = 0; i < 64; ++i) h[i] = -(float)(63 - i);
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, l);
        float r[2], ll[64]; hipMemcpy(r, o, 8, hipMemcpyDeviceToHost); hipMemcpy(ll, l, 256, hipMemcpyDeviceToHost);
        float m = h[0]; for (int i = 1; i < 64; ++i) m = h[i] < m ? h[i] : m;
        printf("true %g  c %g  asm %g | lanes15,31,47,63: %g %g %g %g\n", m, r[0], r[1], ll[15], ll[31], ll[47], ll[63]);
    }
}
