// swr_device.h -- device-side structs and .NET-semantics scalar helpers for the gfx950 backend.
//
// Everything here is compiled with -ffp-contract=off: the reference's C# is JIT-compiled
// without FMA contraction, and depth words must match bit for bit.
// Division and sqrt are the correctly rounded IEEE forms (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt); f32 denormals are kept (hipcc default).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "swr.h"

// System.Numerics models.  Whether .NET 9 fuses the multiply-adds of Vector4.Transform / Vector3.TransformNormal /
// Matrix4x4.Multiply and of Vector4.Lerp cannot be settled here, and nothing says the answer is the same for both families:
//   * Lerp (Shaders.Lerp in the clipper, Renderer.cs:858 in the fragment stage) sits in the hot kernels: COMPILE-TIME, SWR_NUMERICS_FMA;
//   * Transform / TransformNormal live in the vertex stage and the frustum test only: RUN-TIME flags per context
//     (swr_set_transform_fma -> DrawParams::nm_flags), whose default is SWR_NUMERICS_FMA for both, so libswr_hip_fma.so alone still
//     models "everything fused".
// With the three dot orders (SWR_DOT_PAIRWISE) that is six libraries x four flag settings: every combination the C# start-up probe
// can observe is served.
#ifndef SWR_NUMERICS_FMA
#define SWR_NUMERICS_FMA 0   // 1 models a fused MultiplyAddEstimate inside System.Numerics Lerp (and is the default of the run-time Transform flags)
#endif
#define SWR_NM_TRANSFORM_FMA        1u    // DrawParams::nm_flags: Vector4.Transform / Vector3.Transform / Matrix4x4.Multiply fuse
#define SWR_NM_TRANSFORM_NORMAL_FMA 2u    // Vector3.TransformNormal fuses
#ifndef SWR_DOT_PAIRWISE
#define SWR_DOT_PAIRWISE 0   // summation order of Vector3.Dot / LengthSquared (Renderer.cs:835,851, Rasterizer.cs:684) on the
#endif                       // (x, y, z, 0) lanes of a Vector128: 0 sequential (xx + yy) + zz; 1 dpps order (xx + yy) + (zz + 0);
                             // 2 two shuffle-adds (Vector128.Sum) (xx + zz) + (yy + 0)

#define SWR_TILE 16                      // Rasterizer.cs:53 TileSize -- part of the numerical contract
#define SWR_FLOAT_MINVALUE (-3.40282347e+38f)   // float.MinValue, MainWindow.cs:425,434
#define SWR_EPSILON 1e-6f                // Rasterizer.cs:52
#define SWR_FLAG_INTERP 0x80000000u
#define SWR_FLAG_LINE   0x40000000u      // record is one DrawLine edge of DebugMode.Wireframe: sx/sy[0..1] = p0, p1
#define SWR_FLAG_FASTDIV 0x20000000u     // k_raster_c staging only: the pair's three clip.W and its draw's fog range are in div_operand_safe()'s range
#define SWR_FLAG_SIMPLE 0x10000000u      // k_raster_c staging only: every row of the pair's coverage mask is one run of pixels (k_cover's SWR_INFO_SIMPLE)
#define SWR_DRAW_MASK   0x0fffffffu
#define SWR_INFO_SIMPLE 0x80000000u      // k_cover's info.x: count | SWR_INFO_SIMPLE

#define SWR_TB_INVALID 0xffffffffffffffffull     // slot_tb word of a slot with nothing to rasterise

// Co-residency budget of the FRONT-END kernels (frames in flight, swr_api.hip).  The raster kernel of the previous flush fills every
// CU with 16 one-wave workgroups of 10,240 B of LDS each (all 160 KB) at <= 104 allocated VGPRs (k_raster_c<DUST2>: 103; 4 waves per
// SIMD: 96 of 512 registers left per SIMD lane).  A front-end workgroup is placed when one raster wave retires -- so it runs BESIDE
// the raster kernel only if it fits what that leaves: at most 10,240 B of LDS, 96 VGPRs per wave, 4 waves.  Measured
// (profiles/r04_frames_in_flight.md): k_vertex with 16 KB of LDS per block sat out the whole raster kernel (385-408 us), without LDS
// it ran beside it in 40 us; with the raster kernel at 106 registers (112 allocated: 64 left) k_bin (69 / 79) waited for its tail
// and, capped at 64, spilled (+29 us alone) -- three registers fewer in the raster kernel are what let the whole front end in.
// tests/test_resource_budget.py holds every kernel to this budget from the compiler's resource remarks (a register more in the wrong
// place costs the overlap, silently; this hipcc ignores amdgpu_num_vgpr, so the budget cannot be declared on the kernels themselves).
#define SWR_FRONT_MAX_LDS 10240
#define SWR_FRONT_MAX_VGPRS 96
#define SWR_RASTER_MAX_VGPRS 104
// ... and its waves issue ahead of the raster kernel's on the SIMDs they share (s_setprio 3, first statement of every front-end kernel):
// with frames in flight the front end of frame N+1, stretched across the raster kernel of frame N, is the critical path (cfg3: 547 us
// against 492), and on small frames it is starved outright.  Same-box A/B (tools/ab/r4_prio.sh): cfg2 0.138 -> 0.120 ms (-13 %), cfg3
// 0.563 -> 0.558, cfg5 +-0; priority 1 and 3 measure alike.  Alone on the chip the instruction changes nothing.
#define SWR_FRONT_ENTER() __builtin_amdgcn_s_setprio(3)
#define SWR_GEOM_BLOCK 128                       // threads per k_vertex / k_setup block (vertices / triangles of ONE draw per block)

namespace swr {

// ---------------------------------------------------------------- device records ----
// Vertex-stage output kept in HBM: only the varyings a built-in fragment program can read
// (Shaders.VertexOutput, Shaders.cs:26-47, minus Normal/ScreenCoords/Barycentric which no
// built-in FS consumes).  64 B = one 16-dword scalar load per vertex in the raster kernel.
struct VOut {
    float clip[4];     // ClipPosition
    float color[4];    // Color
    float uv[2];       // TexCoord
    float wn[3];       // Data["WorldNormal"]
    float wpos[3];     // Data["WorldPos"].xyz (PHONG_4POINT only)
};
static_assert(sizeof(VOut) == 64, "VOut must be 64 bytes");

// Triangle setup record: what RasterizeTriangle (Rasterizer.cs:401-460) has in locals when
// it enters the tile loop.  Vertex order is the reference's outputs[] order {v2, v1, v0}.
struct TriRec {
    float sx[3], sy[3];     // screen[0..2]
    float depth[3];         // depths[0..2]
    float inv_area;         // 1 / EdgeFunction(s0,s1,s2)
    uint32_t vref[3];       // index of outputs[0..2] in the VOut array (clip pool follows the VS outputs)
    uint32_t bbox_x;        // minX | maxX << 16   (pixel bbox, already clamped to the frame)
    uint32_t bbox_y;        // minY | maxY << 16
    uint32_t draw_flags;    // draw index | SWR_FLAG_LINE (wireframe edge) | SWR_FLAG_INTERP (outputs[0].Interpolate)
};
static_assert(sizeof(TriRec) == 64, "TriRec must be 64 bytes");

// Per-draw constants (one RenderMesh call), read with scalar loads.
struct DrawParams {
    float model[16], view[16], proj[16];
    swr_uniforms u;
    const swr_vertex* verts;
    const uint16_t* idx;
    const uint8_t* tex;
    int tex_w, tex_h;          // tex_h < 0: build-defined bilinear filter on a texture of height -tex_h (reference = nearest)
    int program, cull, depth_test, blend;
    uint32_t n_verts, n_tris;
    uint32_t vert_base;     // first VOut of this draw
    uint32_t tri_base;      // first global triangle number of this draw
    float fog_r1;           // rcp_refined(u.fog_end - u.fog_start), or 0 when that range is outside div_operand_safe():
                            // written on the device by k_vertex (the fragment program's fog division, Renderer.cs:855)
    float tex_wf, tex_hf;   // (float)tex_w, (float)|tex_h|: Texture.Sample's `u * Width` / `v * Height` operands (Texture.cs:50-51)
    float fog_den;          // u.fog_end - u.fog_start, written next to fog_r1 by k_vertex: one float subtraction per draw, not per fragment
    uint32_t frag_draw;     // index of the FIRST draw of the batch whose fragment-stage state (program, blend, depth test, texture,
                            // uniforms) equals this draw's (execute_batch): k_setup writes it into TriRec::draw_flags instead of the
                            // draw's own index, so k_raster_c -- which cuts a fragment chunk where the draw changes, because the
                            // per-draw constants are wave-uniform -- sees the 16 meshes of one model with one material as ONE draw
    uint32_t nm_flags;      // SWR_NM_*: the context's System.Numerics model of Transform / TransformNormal when the draw was recorded
};
static_assert(offsetof(DrawParams, fog_den) == offsetof(DrawParams, fog_r1) + 12, "k_vertex writes fog_den three floats after fog_r1");

struct Counters {           // device-side swr_stats accumulators
    unsigned long long triangles_in, triangles_setup, triangles_clipped;
    unsigned long long fragments_tested, fragments_shaded, fragments_written;
    unsigned long long tile_pairs;
    unsigned int overflow;  // set when a tile list did not fit
    unsigned int pad;
};

// Optimistic execution control block (one per context, in HBM).  A flush is launched without reading the pair
// total back: k_scan_apply compares it with the list capacity and, if it does not fit, sets `poison`; every
// kernel of that batch and of every later one that would touch the pair lists or the framebuffer returns at once
// (batch_poisoned), so the framebuffer stays exactly as it was before the first batch that did not fit.  The host looks
// at this block at its next synchronisation point, grows the buffers and replays from `first_bad`.
struct Ctrl {
    uint32_t poison;
    uint32_t first_bad;              // sequence number of the first batch that did not fit (atomicMin)
    unsigned long long need;         // largest pair total seen by a batch that did not fit (atomicMax)
    uint32_t* host_flag;             // pinned host word, set to 1 together with `poison`: the host polls it without a copy
};
// Must batch `seq` leave the pair lists and the framebuffer alone?  Yes when it, or a batch before it, did not fit.  Decided by
// `first_bad`, not by the sticky `poison` word: with frames pipelined (swr_api.hip: the front end of flush N+1 runs beside the
// raster kernel of flush N) a LATER batch may poison itself while this one's raster kernel is half way through its tiles, and
// that must not stop the tiles that have not started yet -- the host replays from `first_bad` only.
__device__ __forceinline__ bool batch_poisoned(const Ctrl* __restrict__ c, uint32_t seq) { return c->first_bad <= seq; }

struct FrameParams {
    int width, height;          // full frame
    int tiles_x, tiles_y;       // full frame, in 16x16 tiles
    int band_ty0, band_ty1;     // this context renders tile rows [band_ty0, band_ty1) ...
    int band_y0;                // = band_ty0 * 16: first pixel row stored in the buffers (contiguous band)
    int band_rows;              // pixel rows stored
    int il_k, il_world, il_rank;   // ... or, when il_k > 0, the stripes of il_k tile rows whose stripe number s has s % il_world == il_rank
    int band_tile_rows;         // tile rows stored (either form); the buffers hold them in ascending order, 16 pixel rows each
    float near_clip;
};

// Tile-row ownership of a context (multi-GPU): a contiguous band, or interleaved stripes (load balance for clustered
// scenes).  local row = position of the tile row in this context's buffers, -1 = not ours.
struct BandMap { int ty0, ty1, il_k, il_world, il_rank; };
__host__ __device__ __forceinline__ int band_local_row(const BandMap& b, int ty) {
    if (b.il_k <= 0) return (ty >= b.ty0 && ty < b.ty1) ? ty - b.ty0 : -1;
    const int stripe = ty / b.il_k;
    if (stripe % b.il_world != b.il_rank) return -1;
    return (stripe / b.il_world) * b.il_k + (ty - stripe * b.il_k);
}
__host__ __device__ __forceinline__ int band_global_row(const BandMap& b, int local) {
    if (b.il_k <= 0) return b.ty0 + local;
    const int ls = local / b.il_k;
    return (ls * b.il_world + b.il_rank) * b.il_k + (local - ls * b.il_k);
}
__host__ __device__ __forceinline__ BandMap band_map(const FrameParams& fp) {
    BandMap b; b.ty0 = fp.band_ty0; b.ty1 = fp.band_ty1; b.il_k = fp.il_k; b.il_world = fp.il_world; b.il_rank = fp.il_rank;
    return b;
}

// ---------------------------------------------------------------- .NET scalar semantics ----
__device__ __forceinline__ bool is_neg_bits(float f) { return (__float_as_uint(f) >> 31) != 0u; }
__device__ __forceinline__ bool is_nan(float f) { return f != f; }
__device__ __forceinline__ bool is_nan_or_inf(float f) { return (__float_as_uint(f) & 0x7f800000u) == 0x7f800000u; }

// (int)float, .NET 9 x64: truncating, saturating, NaN -> 0.  That is exactly v_cvt_i32_f32 on gfx9 (round toward
// zero, out-of-range and infinities saturate, NaN -> 0); C++ `(int)f` is undefined out of range, so name the instruction.
__device__ __forceinline__ int f2i(float f) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
// MathF.Min / MathF.Max: NaN-propagating, -0 < +0
__device__ __forceinline__ float mathf_min(float a, float b) {
    if (a != b) { if (!is_nan(a)) return a < b ? a : b; return a; }
    return is_neg_bits(a) ? a : b;
}
__device__ __forceinline__ float mathf_max(float a, float b) {
    if (a != b) { if (!is_nan(a)) return b < a ? a : b; return a; }
    return is_neg_bits(b) ? a : b;
}
__device__ __forceinline__ float math_clamp(float v, float lo, float hi) {   // Math.Clamp
    if (v < lo) return lo;
    else if (v > hi) return hi;
    return v;
}

// ---------------------------------------------------------------- exact division / sqrt without the scaling steps ----
// hipcc expands a correctly rounded f32 division n / d (the flag above) to
//     d' = v_div_scale(d)   n' = v_div_scale(n)          (a power-of-two rescue for extreme exponents; vcc = "rescaled")
//     r0 = v_rcp_f32(d')    e = fma(-d', r0, 1)    r1 = fma(e, r0, r0)
//     q0 = n' * r1          e2 = fma(-d', q0, n')  q1 = fma(e2, r1, q0)    e3 = fma(-d', q1, n')
//     q  = v_div_fmas(e3, r1, q1)   (= fma, times 2^+-64 if vcc)           result = v_div_fixup(q, d, n)   (NaN / Inf / 0 / denormal cases)
// When |n| and |d| lie in [2^-40, 2^40] none of v_div_scale's rescue conditions holds (they need an exponent
// difference >= 96, a denormal operand or quotient, or |n| < 2^-103), so d' = d, n' = n, vcc = 0, v_div_fmas is a
// plain fma and v_div_fixup returns q: the division IS the eight-instruction core below.  r1 depends on d only, so a
// denominator shared by many fragments (a vertex's clip.W, a draw's fog range) pays rcp_refined ONCE and each quotient
// costs 1 mul + 4 fma instead of 11 instructions -- bit for bit the same quotient, because it is the same instruction
// sequence on the same operands (fma used inside a correctly rounded division is not a contraction of reference
// arithmetic).  swr_selftest_division compares it with the compiler's `/` over random, near-midpoint and boundary
// operands on the GPU (tests/test_gpu_api.py::test_exact_division_core_matches_ieee_division).
__device__ __forceinline__ float rcp_refined(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float div_core(float n, float d, float r1) {
    const float q0 = n * r1;
    const float e2 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e2, r1, q0);
    const float e3 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e3, r1, q1);
}
// |v| in [2^-40, 2^40], by its bit pattern (NaN, Inf, zero and denormals are all outside)
#define SWR_DIV_LO_BITS 0x2B800000u      // 2^-40
#define SWR_DIV_HI_BITS 0x53800000u      // 2^40
__device__ __forceinline__ bool div_operand_safe(float v) {
    return __builtin_fabsf(v) >= 0x1p-40f && __builtin_fabsf(v) <= 0x1p40f;       // two compares with the |.| source modifier; NaN fails both
}
__device__ __forceinline__ bool div_operands_safe3(float a, float b, float c) {
    const uint32_t ua = __float_as_uint(a) & 0x7fffffffu, ub = __float_as_uint(b) & 0x7fffffffu, uc = __float_as_uint(c) & 0x7fffffffu;
    return min(min(ua, ub), uc) >= SWR_DIV_LO_BITS && max(max(ua, ub), uc) <= SWR_DIV_HI_BITS;
}
// The same guard for three values that are results of float arithmetic (canonical, so min/add need no quieting):
// 5 instructions instead of 7.  The sum of the magnitudes is >= each of them and carries NaN / Inf (a NaN is dropped by
// v_min3 but not by the sum), so `sum <= 2^40` bounds all three from above -- slightly conservative, which a guard may be.
__device__ __forceinline__ bool div_operands_safe3_arith(float a, float b, float c) {
    const float fa = __builtin_fabsf(a), fb = __builtin_fabsf(b), fc = __builtin_fabsf(c);
    const float sum = (fa + fb) + fc;
    const float lo = __builtin_fminf(__builtin_fminf(fa, fb), fc);
    return lo >= 0x1p-40f && sum <= 0x1p40f;
}
// 1 / d: the division core with n = 1 (q0 = 1 * r1 is r1 itself): 7 instructions instead of 11.  With n = 1 the range is
// wider on the large side: |d| in [2^-40, 2^83] (v_div_scale rescues only |d| >= 2^126, |d| <= 2^-96 and denormals).
// swr_selftest_division checks EVERY float of that range, both signs, against the compiler's 1.0f / d.
#define SWR_RCP_HI_BITS 0x69000000u      // 2^83
__device__ __forceinline__ float recip_core(float d) {
    const float r1 = rcp_refined(d);
    const float e2 = __builtin_fmaf(-d, r1, 1.0f);
    const float q1 = __builtin_fmaf(e2, r1, r1);
    const float e3 = __builtin_fmaf(-d, q1, 1.0f);
    return __builtin_fmaf(e3, r1, q1);
}
// hipcc's correctly rounded sqrtf is: scale x by 2^32 if x < 2^-96; s = v_sqrt_f32(x); step s down one ulp if
// fma(-(s-1ulp), s, x) <= 0, up one ulp if fma(-(s+1ulp), s, x) > 0; unscale; return x itself for +-0 / +inf.
// For x in [2^-40, 2^40] the scaling and the class test are no-ops: the five-instruction core below is the same sqrt.
__device__ __forceinline__ float sqrt_core(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    float r = rd <= 0.0f ? sd : s;
    r = ru > 0.0f ? su : r;
    return r;
}

// ---------------------------------------------------------------- System.Numerics ----
template <bool FUSED>
__device__ __forceinline__ float nm_madd(float a, float b, float c) {
    if (FUSED) return __builtin_fmaf(a, b, c);
    float p = a * b;
    return p + c;
}
// Vector4.Transform(v, M): ((x*row1 + y*row2) + z*row3) + w*row4 ; M row-major M11..M44
template <bool FUSED>
__device__ __forceinline__ void vec4_transform_t(const float v[4], const float* __restrict__ m, float out[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r = m[j] * v[0];
        r = nm_madd<FUSED>(m[4 + j], v[1], r);
        r = nm_madd<FUSED>(m[8 + j], v[2], r);
        r = nm_madd<FUSED>(m[12 + j], v[3], r);
        out[j] = r;
    }
}
template <bool FUSED>
__device__ __forceinline__ void vec3_transform_normal_t(const float n[3], const float* __restrict__ m, float out[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float r = m[j] * n[0];
        r = nm_madd<FUSED>(m[4 + j], n[1], r);
        r = nm_madd<FUSED>(m[8 + j], n[2], r);
        out[j] = r;
    }
}
// `fused` is uniform over a draw (DrawParams::nm_flags): one scalar branch per call site
__device__ __forceinline__ void vec4_transform(const float v[4], const float* __restrict__ m, float out[4], bool fused) {
    if (fused) vec4_transform_t<true>(v, m, out); else vec4_transform_t<false>(v, m, out);
}
__device__ __forceinline__ void vec3_transform_normal(const float n[3], const float* __restrict__ m, float out[3], bool fused) {
    if (fused) vec3_transform_normal_t<true>(n, m, out); else vec3_transform_normal_t<false>(n, m, out);
}
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
#if SWR_DOT_PAIRWISE == 1
    const float p = ax * bx + ay * by, q = az * bz + 0.0f;
    return p + q;
#elif SWR_DOT_PAIRWISE == 2
    const float p = ax * bx + az * bz, q = ay * by + 0.0f;
    return p + q;
#else
    return (ax * bx + ay * by) + az * bz;
#endif
}
__device__ __forceinline__ float nm_lerp(float a, float b, float t) {   // a*(1-t) + b*t
#if SWR_NUMERICS_FMA
    return __builtin_fmaf(a, 1.0f - t, b * t);
#else
    float x = a * (1.0f - t);
    float y = b * t;
    return x + y;
#endif
}

// EdgeFunction, Rasterizer.cs:562-563
__device__ __forceinline__ float edge_function(float ax, float ay, float bx, float by, float cx, float cy) {
    float p = (cx - ax) * (by - ay);
    float q = (cy - ay) * (bx - ax);
    return p - q;
}

// GetDepthTestFunction, Rasterizer.cs:543-559
__device__ __forceinline__ bool depth_func(int test, float nd, float od) {
    switch (test) {
    case SWR_DEPTH_LESSEQUAL:    return nd >= od;
    case SWR_DEPTH_LESS:         return nd > od;
    case SWR_DEPTH_GREATER:      return nd < od;
    case SWR_DEPTH_GREATEREQUAL: return nd <= od;
    case SWR_DEPTH_EQUAL:        return fabsf(nd - od) < SWR_EPSILON;
    case SWR_DEPTH_NOTEQUAL:     return fabsf(nd - od) >= SWR_EPSILON;
    default:                     return true;    // Disabled, Always, anything else
    }
}

// Blend, Rasterizer.cs:58-65
__device__ __forceinline__ float4 blend(float4 s, float4 d, int mode) {
    float4 o;
    switch (mode) {
    case SWR_BLEND_ALPHA: {
        float a = s.w, ia = 1.0f - s.w;
        float x;
        x = s.x * a; o.x = x + d.x * ia;
        x = s.y * a; o.y = x + d.y * ia;
        x = s.z * a; o.z = x + d.z * ia;
        x = s.w * a; o.w = x + d.w * ia;
        return o; }
    case SWR_BLEND_ADDITIVE:
        o.x = mathf_min(s.x + d.x, 1.0f); o.y = mathf_min(s.y + d.y, 1.0f);
        o.z = mathf_min(s.z + d.z, 1.0f); o.w = mathf_min(s.w + d.w, 1.0f);
        return o;
    case SWR_BLEND_MULTIPLY:
        o.x = s.x * d.x; o.y = s.y * d.y; o.z = s.z * d.z; o.w = s.w * d.w;
        return o;
    default:
        return s;
    }
}

// DrawLine's per-pixel test, Rasterizer.cs:296-313: parameter t of the closest point of segment p0->p1 to the
// pixel centre (x+0.5, y+0.5) and whether the centre lies within 0.5 px of the segment
__device__ __forceinline__ bool line_test(float p0x, float p0y, float p1x, float p1y, int x, int y, float& t_out) {
    const float dx = p1x - p0x, dy = p1y - p0y;                       // :257-258
    const float len_sq = dx * dx + dy * dy;                          // :259
    const float px = (float)x + 0.5f - p0x, py = (float)y + 0.5f - p0y;   // :296-297
    float t = 0.0f;
    if (len_sq > 0) t = (px * dx + py * dy) / len_sq;                // :300-301
    t = mathf_max(0.0f, mathf_min(1.0f, t));                         // :303
    const float cx = p0x + t * dx, cy = p0y + t * dy;                // :305-306
    const float ddx = ((float)x + 0.5f) - cx, ddy = ((float)y + 0.5f) - cy;
    const float dist_sq = ddx * ddx + ddy * ddy;                     // :310
    t_out = t;
    return dist_sq <= 0.5f * 0.5f;                                   // :312-313
}

// Texture.Sample, Texture.cs:43-63 -- nearest, wrap; texels RGBA8 row-major
// Texture.Sample in two halves, so that a caller can put independent work between the texel load and its use
// (k_raster_c: the fetch is a dependent gather with a long latency)
__device__ __forceinline__ float4 texture_unpack(uint32_t p) {              // Texture.cs:55-62: byte * (1f / 255f)
    const float inv255 = 1.0f / 255.0f;
    float4 o;
    o.x = (float)(p & 0xffu) * inv255;
    o.y = (float)((p >> 8) & 0xffu) * inv255;
    o.z = (float)((p >> 16) & 0xffu) * inv255;
    o.w = (float)(p >> 24) * inv255;
    return o;
}
// wf, hf = (float)Width, (float)Height as C# converts them in `u * Width` (exact for sizes below 2^24; passed in so that the raster
// kernel gets them from the draw's constants instead of converting per fragment).  32-bit texel index: swr_texture_create refuses
// textures of 2^30 texels or more, so index * 4 fits the unsigned 32-bit offset of a global load with a scalar base.
__device__ __forceinline__ uint32_t texture_nearest_index(int w, int h, float wf, float hf, float tu, float tv) {      // Texture.cs:43-54
    float u = tu - (float)f2i(tu);
    float v = tv - (float)f2i(tv);
    u += (u < 0) ? 1.0f : 0.0f;
    v += (v < 0) ? 1.0f : 0.0f;
    int x = f2i(u * wf);
    int y = f2i(v * hf);
    // C# '%' truncates toward zero, then `if (x < 0) x += Width`; the common case 0 <= x < w needs neither
    if ((uint32_t)x >= (uint32_t)w) { x = (x == w) ? 0 : x % w; if (x < 0) x += w; }
    if ((uint32_t)y >= (uint32_t)h) { y = (y == h) ? 0 : y % h; if (y < 0) y += h; }
    return (uint32_t)y * (uint32_t)w + (uint32_t)x;
}
__device__ __forceinline__ uint32_t texture_nearest_index(int w, int h, float tu, float tv) {
    return texture_nearest_index(w, h, (float)w, (float)h, tu, tv);
}
__device__ __forceinline__ float4 texture_sample(const uint8_t* __restrict__ tex, int w, int h, float tu, float tv) {
    return texture_unpack(reinterpret_cast<const uint32_t*>(tex)[texture_nearest_index(w, h, tu, tv)]);
}

// BUILD-DEFINED bilinear filter with wrap (row N4; no reference semantics -- the formula is stated in
// oracle/swr_oracle.c:oswr_texture_sample_bilinear and reproduced operation for operation)
// BLOCKED: the texels are stored block-linear -- 4 x 4-texel blocks of 64 B, blocks row-major (swr_texture_set_filter makes that copy
// for textures whose sides are multiples of 4; k_block_texture) -- so the 2 x 2 taps of a fragment share ONE 64-byte line 9 times out
// of 16 instead of never (row-major: two rows, one line each, W * 4 bytes apart).  The result is layout-independent by construction:
// only the address of texel (x, y) changes.
template <bool BLOCKED>
__device__ __forceinline__ uint32_t bilinear_texel_offset(int w, int x, int y) {
    if (BLOCKED) return ((((uint32_t)y >> 2) * ((uint32_t)w >> 2) + ((uint32_t)x >> 2)) << 4) + (((uint32_t)y & 3u) << 2) + ((uint32_t)x & 3u);
    return (uint32_t)y * (uint32_t)w + (uint32_t)x;
}
template <bool BLOCKED>
__device__ __forceinline__ float4 texture_sample_bilinear(const uint8_t* __restrict__ tex, int w, int h, float tu, float tv) {
    const float x = tu * (float)w - 0.5f, y = tv * (float)h - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    int ix0 = f2i(x0) % w; if (ix0 < 0) ix0 += w;
    int iy0 = f2i(y0) % h; if (iy0 < 0) iy0 += h;
    const int ix1 = ix0 + 1 == w ? 0 : ix0 + 1, iy1 = iy0 + 1 == h ? 0 : iy0 + 1;
    // (texel index < 2^30: swr_texture_create refuses larger textures)
    const uint32_t* __restrict__ t32 = reinterpret_cast<const uint32_t*>(tex);
    const uint32_t p00 = t32[bilinear_texel_offset<BLOCKED>(w, ix0, iy0)];
    const uint32_t p10 = t32[bilinear_texel_offset<BLOCKED>(w, ix1, iy0)];
    const uint32_t p01 = t32[bilinear_texel_offset<BLOCKED>(w, ix0, iy1)];
    const uint32_t p11 = t32[bilinear_texel_offset<BLOCKED>(w, ix1, iy1)];
    const float inv255 = 1.0f / 255.0f;
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float c00 = (float)((p00 >> (8 * c)) & 0xffu) * inv255, c10 = (float)((p10 >> (8 * c)) & 0xffu) * inv255;
        const float c01 = (float)((p01 >> (8 * c)) & 0xffu) * inv255, c11 = (float)((p11 >> (8 * c)) & 0xffu) * inv255;
        const float top = c00 * (1.0f - fx) + c10 * fx;
        const float bot = c01 * (1.0f - fx) + c11 * fx;
        o[c] = top * (1.0f - fy) + bot * fy;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}
// w_signed < 0 (bilinear only): `tex` is the block-linear copy of a texture of width -w_signed
__device__ __forceinline__ float4 texture_fetch(const uint8_t* __restrict__ tex, int w_signed, int h_signed, float tu, float tv) {
    if (h_signed >= 0) return texture_sample(tex, w_signed, h_signed, tu, tv);
    return w_signed < 0 ? texture_sample_bilinear<true>(tex, -w_signed, -h_signed, tu, tv) : texture_sample_bilinear<false>(tex, w_signed, -h_signed, tu, tv);
}

}  // namespace swr
