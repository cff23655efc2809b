// valu_rate.hip -- issue cost of the VALU / LDS instructions the raster kernel is made of, on gfx950.
// One workgroup = W waves on ... (blocks of 64*W threads, 256*4/W... see main): each wave runs ITER iterations of 64
// independent instructions of one kind; cycles per wave-instruction per SIMD = elapsed shader cycles * waves sharing
// the SIMD / instructions issued.   build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 2000

#define REP8(X, a) X(a##0) X(a##1) X(a##2) X(a##3) X(a##4) X(a##5) X(a##6) X(a##7)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float seed) {
    __shared__ float4 lds[1024];
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    const float m = 1.0000001f;
    unsigned addr = (threadIdx.x & 63) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#define ONE(i) \
        if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&r[2 * ((i) & 7)]) : "v"(*(const double*)&r[2 * ((i) & 7)])); \
        else if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&r[2 * ((i) & 7)]) : "v"(*(const double*)&r[2 * ((i) & 7)])); \
        else if (KIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i])); \
        else if (KIND == 7) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i])); \
        else if (KIND == 8) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(r[i]) : "v"(m) : "vcc"); \
        else if (KIND == 9) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 10) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 11) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 12) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(r[i])); \
        else if (KIND == 13) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 14) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&r[2 * ((i) & 7)]) : "v"(*(const double*)&r[2 * ((i) & 7)])); \
        else if (KIND == 15) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(r[i]), "v"(m) : "vcc"); \
        else if (KIND == 17) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 18) asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m)); \
        else if (KIND == 19) asm volatile("v_readfirstlane_b32 s20, %0" :: "v"(r[i]) : "s20"); \
        else if (KIND == 20) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i])); \
        else if (KIND == 21) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i])); \
        else if (KIND == 22) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[0]) : "v"(m));   /* dependent chain */ \
        else if (KIND == 23) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(*(double*)&r[2 * ((i) & 7)]) : "v"(*(const double*)&r[2 * ((i) & 7)]));
#define FOUR(a) ONE(a) ONE(a + 1) ONE(a + 2) ONE(a + 3)
        FOUR(0) FOUR(4) FOUR(8) FOUR(12) FOUR(0) FOUR(4) FOUR(8) FOUR(12)
        FOUR(0) FOUR(4) FOUR(8) FOUR(12) FOUR(0) FOUR(4) FOUR(8) FOUR(12)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[addr & 1023].x;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// LDS read throughput: 64 independent ds_read_b128 / b32 per iteration
template <int KIND>
__global__ __launch_bounds__(256) void kl(float* out, unsigned long long* cyc, float seed) {
    __shared__ float4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // KIND 0: b128 consecutive (conflict-free), 1: b128 broadcast (all lanes same address), 2: b32 consecutive,
    // 3: b128 random-ish 16 distinct addresses, 4: b96-like via b128 stride 13 rows
    unsigned a;
    if (KIND == 0) a = lane * 16; else if (KIND == 1) a = 0; else if (KIND == 2) a = lane * 4;
    else if (KIND == 3) a = (lane & 15) * 16 * 13; else a = ((lane * 7) & 15) * 16;
    float4 acc = make_float4(0, 0, 0, 0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 2) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(j * 256)); asm volatile("s_waitcnt lgkmcnt(8)"); acc.x += v; }
            else { float4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(j * 1024)); asm volatile("s_waitcnt lgkmcnt(8)"); acc.x += v.x; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, int waves_per_simd, int n_inst) {
    const int threads = 64 * 4 * (waves_per_simd >= 4 ? 1 : 1);
    // one block of 256 threads = 4 waves = one wave per SIMD of a CU; waves_per_simd blocks per CU
    const int blocks = 256 * waves_per_simd;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, 8 * blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks, threads, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, threads, out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, 8 * blocks * 4, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
    // s_memtime ticks at 100 MHz "shader clock"? report both: ticks per instruction and wall ns per instruction per SIMD
    const double inst = (double)n_inst;
    printf("%-28s waves/SIMD=%d  memtime ticks/inst/wave=%.3f  wall: %.3f ns per wave-inst per SIMD (x%d waves)  => %.2f cyc @2.4GHz\n",
           name, waves_per_simd, avg / inst, ms * 1e6 / (inst * waves_per_simd), waves_per_simd, ms * 1e6 / (inst * waves_per_simd) * 2.4);
    hipFree(out); hipFree(cyc);
}

int main() {
    const char* names[] = { "v_mul_f32", "v_pk_mul_f32", "v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_cndmask_b32", "v_rcp_f32", "v_sqrt_f32",
                            "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_bcnt_u32_b32", "v_cvt_f32_ubyte0", "v_mul_lo_u32",
                            "v_pk_add_f32", "v_add_f32", "v_cmp_lt_f32", "v_max3_f32", "v_alignbit_b32", "v_readfirstlane_b32", "v_mov_b32_dpp",
                            "v_cvt_i32_f32", "v_fma_f32 dependent", "v_pk_mul_f32 op_sel" };
    for (int w : { 1, 2, 4 }) {
#define RUN(K) run(names[K], [](int b, int t, float* o, unsigned long long* c) { hipLaunchKernelGGL(k<K>, dim3(b), dim3(t), 0, 0, o, c, 1.0f); }, w, ITER * 64);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17)
        RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23)
#define RUNL(K, nm) run(nm, [](int b, int t, float* o, unsigned long long* c) { hipLaunchKernelGGL(kl<K>, dim3(b), dim3(t), 0, 0, o, c, 1.0f); }, w, (ITER / 4) * 16);
        RUNL(0, "ds_read_b128 consecutive") RUNL(1, "ds_read_b128 broadcast") RUNL(2, "ds_read_b32 consecutive") RUNL(3, "ds_read_b128 16 rows x13") RUNL(4, "ds_read_b128 16 addr perm")
    }
    return 0;
}
