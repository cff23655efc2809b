// swr_geometry.hip.h -- vertex stage and triangle setup (near clip, viewport, cull, bbox) for gfx950.
//
// Reference path restated here (file:line under the C# repo):
//   Renderer.VertexShader              Renderer.cs:830-846
//   RenderMesh per-triangle body       Rasterizer.cs:200-229
//   ClipTriangleAgainstNearPlane       Rasterizer.cs:95-160   (+ Shaders.Lerp, Shaders.cs:50-95)
//   DrawTriangle                       Rasterizer.cs:342-399
//   RasterizeTriangle prologue         Rasterizer.cs:411-452
//
// The reference runs the vertex shader three times per triangle (six counting the dead
// avgDepth pre-pass, Rasterizer.cs:184-197); the shader is a pure function of the vertex,
// so running it once per unique vertex gives bit-identical varyings.
#pragma once
#include "swr_device.h"

namespace swr {

// one SWR_GEOM_BLOCK-thread block handles up to that many consecutive vertices / triangles of ONE draw,
// so the draw's matrices are wave-uniform (scalar loads)
struct BlockMap { uint32_t draw; uint32_t first; };

__global__ __launch_bounds__(SWR_GEOM_BLOCK) void k_vertex(const DrawParams* __restrict__ draws,
                                                const BlockMap* __restrict__ blocks,
                                                VOut* __restrict__ vout, const uint32_t* __restrict__ visible,
                                                float* __restrict__ fog_r1_of_draw0 /* &draws[0].fog_r1: it and fog_den are written, never read here */,
                                                float4* __restrict__ vnorm /* VertexOutput.Normal per VOut entry, or null: only batches
                                                                              with a DEBUG_VARYINGS draw carry it (see VOut) */,
                                                uint32_t* __restrict__ zero_words, uint32_t n_zero /* binning's per-tile counters: cleared
                                                                              here, not by a launch of their own */,
                                                uint32_t* __restrict__ zero_hist /* and the tile order's histogram + cursors */) {
    SWR_FRONT_ENTER();
    for (uint32_t i = blockIdx.x * (uint32_t)SWR_GEOM_BLOCK + threadIdx.x; i < n_zero; i += gridDim.x * (uint32_t)SWR_GEOM_BLOCK) zero_words[i] = 0u;
    if (blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < 512u; i += (uint32_t)SWR_GEOM_BLOCK) zero_hist[i] = 0u;
    // (2 x SWR_ORDER_BUCKETS = 512 words, swr_binning.hip.h)
    const BlockMap bm = blocks[blockIdx.x];
    if (visible && !visible[bm.draw]) return;          // RenderMesh was not called for this mesh (frustum culled)
    const DrawParams* __restrict__ dp = draws + bm.draw;
    const uint32_t local = bm.first + threadIdx.x;
    // A lane's 64-byte record leaves as four 16-byte stores, 64 bytes apart from its neighbour's: 256 partial line writes per wave.
    // The wave's 64 records go through LDS instead and leave as four stores of 1 KB each (consecutive lanes, consecutive 16 bytes).
    // (and the other way round for the 48-byte input vertices: three loads of 1 KB each per wave, handed out through LDS)
    __shared__ float4 s_out[SWR_GEOM_BLOCK / 64][256];
    static_assert(sizeof(s_out) <= SWR_FRONT_MAX_LDS, "k_vertex must fit beside the raster kernel (swr_device.h)");
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0, o2 = o0, o3 = o0;
    const bool live = local < dp->n_verts;
    const uint32_t wv_ = threadIdx.x >> 6, lane_ = threadIdx.x & 63u;
    const uint32_t first_ = bm.first + wv_ * 64u;                // first vertex of the wave
    const uint32_t n_live_ = dp->n_verts > first_ ? min(dp->n_verts - first_, 64u) : 0u;
    {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(dp->verts + first_);
        float4* sw = &s_out[wv_][0];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t e = (uint32_t)k * 64u + lane_;
            if (e < 3u * n_live_) sw[e] = src[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (live) {
    if (local == 0u) {      // once per draw: the refined reciprocal of the fog range (see div_core in swr_device.h)
        const float den = dp->u.fog_end - dp->u.fog_start;
        fog_r1_of_draw0[(size_t)bm.draw * (sizeof(DrawParams) / sizeof(float))] = div_operand_safe(den) ? rcp_refined(den) : 0.0f;
        fog_r1_of_draw0[(size_t)bm.draw * (sizeof(DrawParams) / sizeof(float)) + 3] = den;      // DrawParams::fog_den
    }

    const float4 q0 = s_out[wv_][3 * lane_], q1 = s_out[wv_][3 * lane_ + 1], q2 = s_out[wv_][3 * lane_ + 2];   // pos.xyz uv.x | uv.y normal.xyz | color

    float p[4] = { q0.x, q0.y, q0.z, 1.0f };
    float world[4], viewp[4], clip[4];
    const bool fma_t = (dp->nm_flags & SWR_NM_TRANSFORM_FMA) != 0u, fma_tn = (dp->nm_flags & SWR_NM_TRANSFORM_NORMAL_FMA) != 0u;
    vec4_transform(p, dp->model, world, fma_t);        // Renderer.cs:832
    vec4_transform(world, dp->view, viewp, fma_t);     // :833
    vec4_transform(viewp, dp->proj, clip, fma_t);      // :834
    float n[3] = { q1.y, q1.z, q1.w }, tn[3];
    vec3_transform_normal(n, dp->model, tn, fma_tn);   // :835
    float len = sqrtf(dot3(tn[0], tn[1], tn[2], tn[0], tn[1], tn[2]));   // Vector3.Normalize = v / Length()

    o0 = make_float4(clip[0], clip[1], clip[2], clip[3]);
    o1 = q2;
    o2 = make_float4(q0.w, q1.x, tn[0] / len, tn[1] / len);
    o3 = make_float4(tn[2] / len, world[0], world[1], world[2]);
    if (vnorm) vnorm[dp->vert_base + local] = make_float4(n[0], n[1], n[2], 0.0f);      // Normal = input.Normal, Renderer.cs:842
    }
    {
        const uint32_t wv = wv_, lane = lane_;
        float4* sw = &s_out[wv][0];                       // the wave's 64 records: record l = entries 4 l .. 4 l + 3
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // (the input vertices above have been read by every lane)
        __builtin_amdgcn_wave_barrier();
        sw[4 * lane + 0] = o0; sw[4 * lane + 1] = o1; sw[4 * lane + 2] = o2; sw[4 * lane + 3] = o3;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // one wave: its own LDS accesses complete in order
        __builtin_amdgcn_wave_barrier();
        const uint32_t first = first_, n_live = n_live_;
        float4* dst = reinterpret_cast<float4*>(vout + dp->vert_base + first);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = (uint32_t)k * 64u + lane;
            if ((e >> 2) < n_live) dst[e] = sw[e];
        }
    }
}

// a vertex moving through clip + setup: the stored varyings plus the Interpolate flag
struct SVert { VOut v; bool interp; float nrm[3]; /* VertexOutput.Normal: carried only when the batch stores it (vnorm) */ };

__device__ __forceinline__ void svert_lerp(const SVert& a, const SVert& b, float t, SVert& r) {
    // Shaders.Lerp(a, b, t, interpolate: true), Shaders.cs:50-95 (the clipper always passes true)
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v.clip[i] = nm_lerp(a.v.clip[i], b.v.clip[i], t);
#pragma unroll
    for (int i = 0; i < 2; ++i) r.v.uv[i] = nm_lerp(a.v.uv[i], b.v.uv[i], t);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v.color[i] = nm_lerp(a.v.color[i], b.v.color[i], t);
#pragma unroll
    for (int i = 0; i < 3; ++i) r.v.wn[i] = nm_lerp(a.v.wn[i], b.v.wn[i], t);      // no renormalisation, :72-73
#pragma unroll
    for (int i = 0; i < 3; ++i) r.v.wpos[i] = nm_lerp(a.v.wpos[i], b.v.wpos[i], t);
#pragma unroll
    for (int i = 0; i < 3; ++i) r.nrm[i] = nm_lerp(a.nrm[i], b.nrm[i], t);           // Vector3.Lerp(a.Normal, b.Normal, t), Shaders.cs:55
    r.interp = true;
}

__device__ __forceinline__ void store_vout(VOut* __restrict__ dst, const VOut& v) {
    float4* o = reinterpret_cast<float4*>(dst);
    o[0] = make_float4(v.clip[0], v.clip[1], v.clip[2], v.clip[3]);
    o[1] = make_float4(v.color[0], v.color[1], v.color[2], v.color[3]);
    o[2] = make_float4(v.uv[0], v.uv[1], v.wn[0], v.wn[1]);
    o[3] = make_float4(v.wn[2], v.wpos[0], v.wpos[1], v.wpos[2]);
}
__device__ __forceinline__ void load_vout(const VOut* __restrict__ src, VOut& v) {
    const float4* s = reinterpret_cast<const float4*>(src);
    float4 a = s[0], b = s[1], c = s[2], d = s[3];
    v.clip[0] = a.x; v.clip[1] = a.y; v.clip[2] = a.z; v.clip[3] = a.w;
    v.color[0] = b.x; v.color[1] = b.y; v.color[2] = b.z; v.color[3] = b.w;
    v.uv[0] = c.x; v.uv[1] = c.y; v.wn[0] = c.z; v.wn[1] = c.w;
    v.wn[2] = d.x; v.wpos[0] = d.y; v.wpos[1] = d.z; v.wpos[2] = d.w;
}

// DrawTriangle + RasterizeTriangle prologue for one (possibly clipped) triangle.
// v0,v1,v2 in submission order; r0,r1,r2 their VOut indices.  Fills rec[0] / tb[0] and returns 1 when the
// triangle reaches the tile loop.  In DebugMode.Wireframe (Rasterizer.cs:419-425) it emits instead up to three
// DrawLine records rec[0..2] (edges s0-s1, s1-s2, s2-s0, each with depths[0..1] and outputs[0..1] of the
// TRIANGLE, as the reference passes them) and returns 0 (the oracle's triangles_setup counts filled triangles only).
__device__ __forceinline__ int setup_triangle(const FrameParams& fp, int cull, uint32_t draw,
                                              const SVert& v0, const SVert& v1, const SVert& v2,
                                              uint32_t r0, uint32_t r1, uint32_t r2,
                                              TriRec* __restrict__ rec, unsigned long long* __restrict__ tb, bool wireframe,
                                              float4* __restrict__ rec_regs = nullptr /* filled mode: the record goes here instead of to
                                                                                         memory (k_setup stores a wave's records together) */) {
    const int rw = fp.width, rh = fp.height;
    // outputs = { v2, v1, v0 }  (Rasterizer.cs:367)
    const SVert* o[3] = { &v2, &v1, &v0 };
    float sx[3], sy[3], dz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float invW = 1.0f / o[i]->v.clip[3];                        // :371
        float nx = o[i]->v.clip[0] * invW, ny = o[i]->v.clip[1] * invW, nz = o[i]->v.clip[2] * invW;
        if (is_nan_or_inf(nx) || is_nan_or_inf(ny) || is_nan_or_inf(nz)) return 0;       // :378-380
        sx[i] = (nx * 0.5f + 0.5f) * (float)rw;                     // :383-386
        sy[i] = (1.0f - (ny * 0.5f + 0.5f)) * (float)rh;
        dz[i] = (nz + 1.0f) * 0.5f;                                 // :388
    }
    if (v0.v.clip[3] == 0 || v1.v.clip[3] == 0 || v2.v.clip[3] == 0) return 0;           // :393
    float area = edge_function(sx[0], sy[0], sx[1], sy[1], sx[2], sy[2]);                // :396, :411
    if (area == 0) return 0;
    bool front = area < 0;                                                               // :414
    if ((cull == SWR_CULL_BACK && !front) || (cull == SWR_CULL_FRONT && front)) return 0;
    const uint32_t flags = draw | (o[0]->interp ? SWR_FLAG_INTERP : 0u);

    if (wireframe) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int i0 = e, i1 = (e + 1) % 3;
            const float p0x = sx[i0], p0y = sy[i0], p1x = sx[i1], p1y = sy[i1];
            // DrawLine bbox, Rasterizer.cs:242-247: truncating casts of the clamped extremes
            const int minX = f2i(mathf_max(mathf_min(p0x, p1x), 0.0f));
            const int maxX = f2i(mathf_min(mathf_max(p0x, p1x), (float)(rw - 1)));
            const int minY = f2i(mathf_max(mathf_min(p0y, p1y), 0.0f));
            const int maxY = f2i(mathf_min(mathf_max(p0y, p1y), (float)(rh - 1)));
            if (minX > maxX || minY > maxY) continue;
            float4* out = reinterpret_cast<float4*>(rec + e);
            out[0] = make_float4(p0x, p1x, 0.0f, p0y);
            out[1] = make_float4(p1y, 0.0f, dz[0], dz[1]);
            out[2] = make_float4(dz[2], 0.0f, __uint_as_float(r2), __uint_as_float(r1));    // outputs[0], outputs[1]
            out[3] = make_float4(__uint_as_float(r2),                                        // third weight is 0 on outputs[0]
                                 __uint_as_float((uint32_t)minX | ((uint32_t)maxX << 16)),
                                 __uint_as_float((uint32_t)minY | ((uint32_t)maxY << 16)),
                                 __uint_as_float(flags | SWR_FLAG_LINE));
            tb[e] = (unsigned long long)(minX / SWR_TILE) | ((unsigned long long)(maxX / SWR_TILE) << 16) |
                    ((unsigned long long)(minY / SWR_TILE) << 32) | ((unsigned long long)(maxY / SWR_TILE) << 48);
        }
        return 0;
    }

    float inv_area = 1.0f / area;                                                        // :427
    float minXf = mathf_min(mathf_min(sx[0], sx[1]), sx[2]);                             // :432-435
    float maxXf = mathf_max(mathf_max(sx[0], sx[1]), sx[2]);
    float minYf = mathf_min(mathf_min(sy[0], sy[1]), sy[2]);
    float maxYf = mathf_max(mathf_max(sy[0], sy[1]), sy[2]);
    int minX = max(f2i(floorf(minXf)), 0);                                               // :437-440
    int maxX = min(f2i(ceilf(maxXf)), rw - 1);
    int minY = max(f2i(floorf(minYf)), 0);
    int maxY = min(f2i(ceilf(maxYf)), rh - 1);
    if (minX > maxX || minY > maxY) return 0;                                            // :442

    float4* out = rec_regs ? rec_regs : reinterpret_cast<float4*>(rec);
    {
    out[0] = make_float4(sx[0], sx[1], sx[2], sy[0]);
    out[1] = make_float4(sy[1], sy[2], dz[0], dz[1]);
    out[2] = make_float4(dz[2], inv_area, __uint_as_float(r2), __uint_as_float(r1));    // vref = outputs order
    out[3] = make_float4(__uint_as_float(r0),
                         __uint_as_float((uint32_t)minX | ((uint32_t)maxX << 16)),
                         __uint_as_float((uint32_t)minY | ((uint32_t)maxY << 16)),
                         __uint_as_float(flags));
    }
    // tile bbox (Rasterizer.cs:449-452), 16 bits each: tminx | tmaxx<<16 | tminy<<32 | tmaxy<<48
    tb[0] = (unsigned long long)(minX / SWR_TILE) | ((unsigned long long)(maxX / SWR_TILE) << 16) |
            ((unsigned long long)(minY / SWR_TILE) << 32) | ((unsigned long long)(maxY / SWR_TILE) << 48);
    return 1;
}

// One thread per submitted triangle.  Filled mode: slot 2*t is the triangle itself (or the first fan triangle of
// its clipped polygon), slot 2*t+1 the second fan triangle.  Wireframe: six slots per triangle, three DrawLine
// edges per fan triangle, in the reference's call order.  Slots keep submission order, which the per-tile lists preserve.
#define SWR_FRAG_DRAW(dp, bm) ((dp)->frag_draw)
__global__ __launch_bounds__(SWR_GEOM_BLOCK) void k_setup(const DrawParams* __restrict__ draws,
                                               const BlockMap* __restrict__ blocks,
                                               const VOut* __restrict__ vout_ro,
                                               VOut* __restrict__ clip_pool,     // 4 VOut per triangle, indexed by global triangle
                                               uint32_t clip_pool_base,          // VOut index of clip_pool[0]
                                               TriRec* __restrict__ recs,
                                               unsigned long long* __restrict__ slot_tb,
                                               FrameParams fp,
                                               Counters* __restrict__ counters /* 64 replicas */,
                                               const Ctrl* __restrict__ ctrl, uint32_t seq, int count_stats, int wireframe,
                                               const uint32_t* __restrict__ visible,
                                               float4* __restrict__ vnorm /* see k_vertex; the clipper's vertices get theirs here */) {
    SWR_FRONT_ENTER();
    const BlockMap bm = blocks[blockIdx.x];
    const DrawParams* __restrict__ dp = draws + bm.draw;
    const uint32_t local = bm.first + threadIdx.x;
    const bool in_range = local < dp->n_tris;
    const bool drawn = !visible || visible[bm.draw] != 0u;     // frustum-culled draws: slots invalid, nothing counted
    const bool active = in_range && drawn;
    unsigned n_setup = 0, n_clipped = 0;
    // the record of an unclipped filled triangle (nearly all): kept in registers and stored by the wave together, see the end
    float4 rec_q[4] = { make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f) };
    bool rec_valid = false;

    if (in_range && !drawn) {
        const uint32_t per_fan = wireframe ? 3u : 1u;
        const uint32_t slot = 2u * per_fan * (dp->tri_base + local);
        for (uint32_t k = 0; k < 2u * per_fan; ++k) slot_tb[slot + k] = SWR_TB_INVALID;
    }
    if (active) {
        const uint32_t gt = dp->tri_base + local;
        const uint32_t per_fan = wireframe ? 3u : 1u;
        const uint32_t slot = 2u * per_fan * gt;
        unsigned long long tbs[6] = { SWR_TB_INVALID, SWR_TB_INVALID, SWR_TB_INVALID, SWR_TB_INVALID, SWR_TB_INVALID, SWR_TB_INVALID };

        const uint16_t* __restrict__ ip = dp->idx + 3u * local;
        const uint32_t r0 = dp->vert_base + ip[0], r1 = dp->vert_base + ip[1], r2 = dp->vert_base + ip[2];
        SVert v[3];
        // only the clip positions are needed unless the triangle has to be clipped (then the varyings are fetched below)
        {
            const float4 c0 = reinterpret_cast<const float4*>(vout_ro + r0)[0], c1 = reinterpret_cast<const float4*>(vout_ro + r1)[0],
                         c2 = reinterpret_cast<const float4*>(vout_ro + r2)[0];
            v[0].v.clip[0] = c0.x; v[0].v.clip[1] = c0.y; v[0].v.clip[2] = c0.z; v[0].v.clip[3] = c0.w;
            v[1].v.clip[0] = c1.x; v[1].v.clip[1] = c1.y; v[1].v.clip[2] = c1.z; v[1].v.clip[3] = c1.w;
            v[2].v.clip[0] = c2.x; v[2].v.clip[1] = c2.y; v[2].v.clip[2] = c2.z; v[2].v.clip[3] = c2.w;
        }
        const bool interp = dp->program != SWR_PROG_FLAT_COLOR;
        v[0].interp = v[1].interp = v[2].interp = interp;

        const bool b0 = v[0].v.clip[3] <= 0, b1 = v[1].v.clip[3] <= 0, b2 = v[2].v.clip[3] <= 0;   // :208-210
        if (!(b0 && b1 && b2)) {                                                                   // :212
            if (b0 || b1 || b2) {                                                                  // :217
                n_clipped = 1;
                load_vout(vout_ro + r0, v[0].v);
                load_vout(vout_ro + r1, v[1].v);
                load_vout(vout_ro + r2, v[2].v);
                if (vnorm) {
                    const float4 n0 = vnorm[r0], n1 = vnorm[r1], n2 = vnorm[r2];
                    v[0].nrm[0] = n0.x; v[0].nrm[1] = n0.y; v[0].nrm[2] = n0.z;
                    v[1].nrm[0] = n1.x; v[1].nrm[1] = n1.y; v[1].nrm[2] = n1.z;
                    v[2].nrm[0] = n2.x; v[2].nrm[1] = n2.y; v[2].nrm[2] = n2.z;
                } else {
                    for (int i = 0; i < 3; ++i) v[i].nrm[0] = v[i].nrm[1] = v[i].nrm[2] = 0.0f;
                }
                // ClipTriangleAgainstNearPlane, Rasterizer.cs:95-160
                SVert poly[4];
                int n = 0;
                const float nearc = fp.near_clip;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const SVert& cur = v[i];
                    const SVert& nxt = v[(i + 1) % 3];
                    bool cur_in = cur.v.clip[2] >= nearc * cur.v.clip[3];                          // :112-113
                    bool nxt_in = nxt.v.clip[2] >= nearc * nxt.v.clip[3];
                    if (cur_in) { poly[n & 3] = cur; ++n; }
                    if (cur_in != nxt_in) {
                        float z0 = cur.v.clip[2], w0 = cur.v.clip[3], z1 = nxt.v.clip[2], w1 = nxt.v.clip[3];
                        float denom = (z1 - z0) - nearc * (w1 - w0);                               // :131
                        float t;
                        if (fabsf(denom) < SWR_EPSILON) {
                            t = 0.5f;
                        } else {
                            t = (z0 - nearc * w0) / (nearc * (w1 - w0) - (z1 - z0));               // :138
                            t = math_clamp(t, 0.0f, 1.0f);
                        }
                        svert_lerp(cur, nxt, t, poly[n & 3]); ++n;                                 // :142
                    }
                }
                if (n >= 3) {                                                                      // :148
                    VOut* pool = clip_pool + 4ull * gt;
                    const uint32_t pbase = clip_pool_base + 4u * gt;
                    for (int k = 0; k < n; ++k) store_vout(pool + k, poly[k].v);
                    if (vnorm) for (int k = 0; k < n; ++k) vnorm[pbase + (uint32_t)k] = make_float4(poly[k].nrm[0], poly[k].nrm[1], poly[k].nrm[2], 0.0f);
                    // fan (0, k, k+1), Rasterizer.cs:154-157
                    n_setup += setup_triangle(fp, dp->cull, SWR_FRAG_DRAW(dp, bm), poly[0], poly[1], poly[2],
                                              pbase, pbase + 1, pbase + 2, recs + slot, &tbs[0], wireframe != 0);
                    if (n == 4) n_setup += setup_triangle(fp, dp->cull, SWR_FRAG_DRAW(dp, bm), poly[0], poly[2], poly[3],
                                                          pbase, pbase + 2, pbase + 3, recs + slot + per_fan, &tbs[3], wireframe != 0);
                }
            } else {
                if (!wireframe) {
                    rec_valid = setup_triangle(fp, dp->cull, SWR_FRAG_DRAW(dp, bm), v[0], v[1], v[2], r0, r1, r2, recs + slot, &tbs[0], false, rec_q) != 0;
                    n_setup += rec_valid ? 1u : 0u;
                } else
                n_setup += setup_triangle(fp, dp->cull, SWR_FRAG_DRAW(dp, bm), v[0], v[1], v[2], r0, r1, r2, recs + slot, &tbs[0], wireframe != 0);
            }
        }
        if (wireframe) {
#pragma unroll
            for (int k = 0; k < 6; ++k) slot_tb[slot + k] = tbs[k];
        } else {
            *reinterpret_cast<ulonglong2*>(slot_tb + slot) = make_ulonglong2(tbs[0], tbs[3]);      // slot is even: one 16-byte store
        }
    }

    // A lane's 64-byte TriRec would leave as four 16-byte stores, 128 bytes apart from its neighbour's (256 partial line writes per
    // wave: 16 of the kernel's 37 us).  The wave's records go through LDS and leave as four stores of sixteen whole records each.
    {
        __shared__ float4 s_rec[SWR_GEOM_BLOCK / 64][256];
        static_assert(sizeof(s_rec) + 16 <= SWR_FRONT_MAX_LDS, "k_setup must fit beside the raster kernel (swr_device.h)");
        const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        const unsigned long long vmask = __ballot(rec_valid);
        if (vmask) {                                                   // wave-uniform
            float4* sw = &s_rec[wv][0];
            sw[4 * lane + 0] = rec_q[0]; sw[4 * lane + 1] = rec_q[1]; sw[4 * lane + 2] = rec_q[2]; sw[4 * lane + 3] = rec_q[3];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // one wave: its own LDS accesses complete in order
            __builtin_amdgcn_wave_barrier();
            // lane j's triangle is dp->tri_base + bm.first + 64 wv + j, its record slot twice that (filled mode: two slots per triangle)
            float4* dst = reinterpret_cast<float4*>(recs + 2u * (size_t)(dp->tri_base + bm.first + wv * 64u));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = (uint32_t)k * 64u + lane, j = e >> 2;
                if ((vmask >> j) & 1ull) dst[8u * j + (e & 3u)] = sw[e];
            }
        }
    }
    // block-level counter reduction, then one atomic per counter per block into a replica
    __shared__ unsigned s_cnt[3];
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long act_mask = __ballot(active);
    unsigned long long set_mask1 = __ballot(n_setup >= 1), set_mask2 = __ballot(n_setup >= 2);
    unsigned long long clip_mask = __ballot(n_clipped != 0);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_cnt[0], (unsigned)__popcll(act_mask));
        atomicAdd(&s_cnt[1], (unsigned)(__popcll(set_mask1) + __popcll(set_mask2)));
        atomicAdd(&s_cnt[2], (unsigned)__popcll(clip_mask));
    }
    __syncthreads();
    if (threadIdx.x == 0 && count_stats && !batch_poisoned(ctrl, seq)) {
        Counters* c = counters + (blockIdx.x & 63);
        atomicAdd(&c->triangles_in, (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&c->triangles_setup, (unsigned long long)s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&c->triangles_clipped, (unsigned long long)s_cnt[2]);
    }
}

}  // namespace swr
