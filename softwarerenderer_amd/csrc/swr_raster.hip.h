// swr_raster.hip.h -- pieces shared by the raster kernels (arguments, Interpolate + fragment programs) and the small
// utility kernels.  The raster kernels themselves are k_cover / k_raster_c in swr_raster_c.hip.h.
//
// Reference code restated here (file:line under the C# repo):
//   Interpolate / InterpolateData Rasterizer.cs:566-707   (shade_fragment)
//   Renderer.FragmentShader       Renderer.cs:848-860     (fs_dust2; Texture.Sample, Blend, depth functions: swr_device.h)
//   MainWindow clears / flatten   MainWindow.cs:234-240,400-436
#pragma once
#include "swr_device.h"

namespace swr {

struct RasterArgs {
    FrameParams fp;
    const TriRec* __restrict__ recs;
    const VOut* __restrict__ vout;
    const DrawParams* __restrict__ draws;
    const uint32_t* __restrict__ tile_start;
    const uint32_t* __restrict__ tile_count;
    const float4* __restrict__ vnorm;          // VertexOutput.Normal per VOut entry (DEBUG_VARYINGS batches only, else null)
    uint32_t vout_bytes;                       // size of the VOut array (< 4 GiB: k_raster_c addresses it with 32-bit byte offsets)
    const uint4* __restrict__ pair_refs;       // per pair {slot, vertex refs of outputs[0..2]} (k_cover)
    const uint4* __restrict__ tile_order;      // per dispatch position, heaviest first: {band-local tile, first list entry, pairs, -} (tile_place_block)
    uint32_t* __restrict__ tile_work;          // out: fragments tested per tile in this flush = the next flush's scheduling weight
    uint32_t n_tiles;
    float4* __restrict__ color;
    float* __restrict__ depth;
    uint32_t* __restrict__ tile_stats;     // 3 x u32 per tile: tested, shaded, written (accumulated)
    float clear_rgba[4];
    int clear_color_on, clear_depth_on;
    int depth_only_grows;                  // every draw of the batch tests Less or LessEqual (enables hi-Z in the generic kernels)
    unsigned long long* dbg;               // SWR_DEBUG_COUNTERS builds only: 8 accumulators
    const Ctrl* __restrict__ ctrl;         // poison guard (see Ctrl, batch_poisoned)
    uint32_t seq;                          // sequence number of the batch
};

// fragment inputs a built-in program may read
struct Frag {
    float clip_z;
    float4 color;
    float u, v;
    uint32_t texel;        // nearest filter: the RGBA8 texel, loaded as early as u,v are known (see shade_fragment)
    bool texel_loaded;
    float wn[3];
    float wpos[3];
};

// The per-draw constants a fragment of the reference's programs reads.  k_raster_c keeps one copy in SGPRs and reloads it
// only when the chunk's draw changes: read through `dp->` in the fragment code they were five dependent scalar-load round
// trips per 64-fragment chunk (the compiler places each s_load next to its use, behind an s_waitcnt lgkmcnt(0)).
struct DrawConsts {
    const uint8_t* tex;
    int tex_w, tex_h;
    float tex_wf, tex_hf;
    float light_direction[3], light_color[3], fog_color[3];
    float fog_end, fog_den, fog_r1;
    int program, blend, depth_test;
};
__device__ __forceinline__ DrawConsts load_draw_consts(const DrawParams* __restrict__ dp_generic) {
    // constant address space: nothing writes DrawParams while a raster kernel runs (k_vertex's fog_r1 / fog_den come from an
    // earlier launch), and with a wave-uniform address this makes every load below an s_load into SGPRs -- as plain global
    // loads the compiler picks VGPR loads here and then waits for them (and for the texel in flight) in the fragment code
    typedef const DrawParams __attribute__((address_space(4)))* const_ptr;
    const const_ptr dp = (const_ptr)(uintptr_t)dp_generic;
    DrawConsts c;
    c.tex = (const uint8_t*)dp->tex; c.tex_w = dp->tex_w; c.tex_h = dp->tex_h; c.tex_wf = dp->tex_wf; c.tex_hf = dp->tex_hf;
#pragma unroll
    for (int i = 0; i < 3; ++i) { c.light_direction[i] = dp->u.light_direction[i]; c.light_color[i] = dp->u.light_color[i]; c.fog_color[i] = dp->u.fog_color[i]; }
    c.fog_end = dp->u.fog_end; c.fog_den = dp->fog_den; c.fog_r1 = dp->fog_r1;
    c.program = dp->program; c.blend = dp->blend; c.depth_test = dp->depth_test;
    return c;
}

// Renderer.FragmentShader, Renderer.cs:848-860
__device__ __forceinline__ float4 fs_dust2(const DrawConsts& u, const Frag& f) {
    // everything that does not need the texel first (its load is in flight, see shade_fragment)
    float diffuse = mathf_max(0.25f, dot3(f.wn[0], f.wn[1], f.wn[2],
                                          -u.light_direction[0], -u.light_direction[1], -u.light_direction[2]));
    // (FogEnd - depth) / (FogEnd - FogStart), Renderer.cs:855: the denominator is per draw, its refined reciprocal was
    // computed once by k_vertex (dp->fog_r1, 0 = out of the safe range) -- see div_core in swr_device.h
    const float fog_num = u.fog_end - f.clip_z, fog_den = u.fog_den;         // (= u.fog_end - u.fog_start, k_vertex)
    float fog_q = div_core(fog_num, fog_den, u.fog_r1);
    if (!(u.fog_r1 != 0.0f && div_operand_safe(fog_num))) fog_q = fog_num / fog_den;
    float fog = math_clamp(fog_q, 0.0f, 1.0f);
    fog = (fog * fog) * (3.0f - 2.0f * fog);
    float s = 0.1f + 0.9f * diffuse;
    float4 tc = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (f.texel_loaded) tc = texture_unpack(f.texel);
    else if (u.tex) tc = texture_fetch(u.tex, u.tex_w, u.tex_h, f.u, f.v);
    float4 base = make_float4(f.color.x * tc.x, f.color.y * tc.y, f.color.z * tc.z, f.color.w * tc.w);
    float4 o;
    o.x = nm_lerp(u.fog_color[0], (base.x * s) * u.light_color[0], fog);
    o.y = nm_lerp(u.fog_color[1], (base.y * s) * u.light_color[1], fog);
    o.z = nm_lerp(u.fog_color[2], (base.z * s) * u.light_color[2], fog);
    o.w = base.w;
    return o;
}

// PHONG_4POINT: build-defined, no reference semantics (see oracle/swr_oracle.c fs_phong4 for the formula)
__device__ __forceinline__ float4 fs_phong4(const DrawParams* __restrict__ dp, const DrawConsts& dc, const Frag& f) {
    const swr_uniforms& u = dp->u;
    float4 tc = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (f.texel_loaded) tc = texture_unpack(f.texel);
    else if (dc.tex) tc = texture_fetch(dc.tex, dc.tex_w, dc.tex_h, f.u, f.v);
    float base[4] = { f.color.x * tc.x, f.color.y * tc.y, f.color.z * tc.z, f.color.w * tc.w };
    // v / |v|: three divisions by one denominator = one refined reciprocal + three division cores (div_core, swr_device.h: the
    // IEEE quotient bit for bit inside its operand range; anything else -- a zero component, a degenerate vector -- takes `/`),
    // and the square roots through sqrt_core inside its range
    auto sqrt_exact = [](float x) { return div_operand_safe(x) ? sqrt_core(x) : sqrtf(x); };
    auto normalize3 = [](const float v[3], float len, float out[3]) {
        const float r1 = rcp_refined(len);
        out[0] = div_core(v[0], len, r1); out[1] = div_core(v[1], len, r1); out[2] = div_core(v[2], len, r1);
        if (!(div_operand_safe(len) && div_operands_safe3(v[0], v[1], v[2]))) { out[0] = v[0] / len; out[1] = v[1] / len; out[2] = v[2] / len; }
    };
    float Vd[3] = { u.camera_position[0] - f.wpos[0], u.camera_position[1] - f.wpos[1], u.camera_position[2] - f.wpos[2] };
    float vl = sqrt_exact(dot3(Vd[0], Vd[1], Vd[2], Vd[0], Vd[1], Vd[2]));
    float V[3];
    normalize3(Vd, vl, V);
    float acc[3] = { 0.1f * base[0], 0.1f * base[1], 0.1f * base[2] };
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const swr_point_light& L = u.lights[l];
        float Ld[3] = { L.position[0] - f.wpos[0], L.position[1] - f.wpos[1], L.position[2] - f.wpos[2] };
        float dist = sqrt_exact(dot3(Ld[0], Ld[1], Ld[2], Ld[0], Ld[1], Ld[2]));
        float Ln[3];
        normalize3(Ld, dist, Ln);
        float ndotl = mathf_max(0.0f, dot3(f.wn[0], f.wn[1], f.wn[2], Ln[0], Ln[1], Ln[2]));
        float att = math_clamp(1.0f - dist / L.range, 0.0f, 1.0f);
        att = att * att;
        float Hd[3] = { Ln[0] + V[0], Ln[1] + V[1], Ln[2] + V[2] };
        float hl = sqrt_exact(dot3(Hd[0], Hd[1], Hd[2], Hd[0], Hd[1], Hd[2]));
        float H[3];
        normalize3(Hd, hl, H);
        float sp = mathf_max(0.0f, dot3(f.wn[0], f.wn[1], f.wn[2], H[0], H[1], H[2]));
        sp = sp * sp; sp = sp * sp; sp = sp * sp; sp = sp * sp;
        float k = L.intensity * att;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float term = (base[i] * ndotl + sp) * (L.color[i] * k);
            acc[i] = acc[i] + term;
        }
    }
    return make_float4(acc[0], acc[1], acc[2], base[3]);
}

// The varyings of one triangle's three outputs, as the fragment path reads them (VOut as four float4:
// [0] clip  [1] color  [2] uv.xy, wn.xy  [3] wn.z, wpos.xyz).  k_raster_c stages them per pair in LDS.
struct TriVaryings {
    float a_cz, b_cz, c_cz;                    // clip.z of each output (the fog depth; clip.xy are never read, clip.w is a_w ...)
    float4 a_col, b_col, c_col;
    float4 a_uvn, b_uvn, c_uvn;
    float a_wnz, b_wnz, c_wnz;
    float a_r1, b_r1, c_r1;                    // rcp_refined(clip.w) of each output (staged once per pair)
    float a_w, b_w, c_w;                       // clip.w of each output (= a_clip.w ...; staged, so that the divisions do not wait for the rows)
    bool fastdiv;                              // the three clip.w are in div_operand_safe()'s range
    float a_wpos[3], b_wpos[3], c_wpos[3];     // PHONG only
};

// Rasterizer.Interpolate for the varyings `program` reads, then the fragment program.
// A,B,C = outputs[0..2]; w0f..w2f = edge values * invArea.
// PHONG = false compiles the build-defined 4-light program out (batches without such a draw): fewer live registers
template <bool PHONG = true>
__device__ __forceinline__ float4 shade_fragment(const DrawParams* __restrict__ dp, const DrawConsts& dc, int program, bool interp,
                                                 const TriVaryings& V, float w0f, float w1f, float w2f) {
    const bool simple = program == SWR_PROG_FLAT_COLOR || program == SWR_PROG_GOURAUD;
    const float4 a_col = V.a_col, b_col = V.b_col, c_col = V.c_col;
    const float4 a_uvn = V.a_uvn, b_uvn = V.b_uvn, c_uvn = V.c_uvn;
    if (simple && !interp) return a_col;                                                             // :622-627

    // :576-578, true divisions: the denominators are per pair, so their refined reciprocals are staged and each quotient
    // is the division's own mul + 4 fma core (div_core, swr_device.h); operands outside its range take the full sequence
    float ra = div_core(w0f, V.a_w, V.a_r1);
    float rb = div_core(w1f, V.b_w, V.b_r1);
    float rc = div_core(w2f, V.c_w, V.c_r1);
    const bool fast = V.fastdiv && div_operands_safe3_arith(w0f, w1f, w2f);
    float inv_sum = (ra + rb) + rc;         // :579
    // :582.  On the fast path |ra|, |rb|, |rc| <= 2^40 / 2^-40, so |inv_sum| < 2^82: inside recip_core's range unless it
    // is tiny (cancellation), zero or NaN
    float w = recip_core(inv_sum);
    if (!(fast && __builtin_fabsf(inv_sum) >= 0x1p-40f)) {
        // one cold block for everything outside the cores' ranges (a single division on its own would be if-converted
        // into a select, i.e. computed by every fragment)
        if (!fast) {
            ra = w0f / V.a_w;
            rb = w1f / V.b_w;
            rc = w2f / V.c_w;
        }
        inv_sum = (ra + rb) + rc;
        w = 1.0f / inv_sum;
    }
#define SWR_PERSP(a_, b_, c_) ((((a_) * ra + (b_) * rb) + (c_) * rc) * w)
    Frag f;
    if (simple) {
        return make_float4(SWR_PERSP(a_col.x, b_col.x, c_col.x), SWR_PERSP(a_col.y, b_col.y, c_col.y),
                           SWR_PERSP(a_col.z, b_col.z, c_col.z), SWR_PERSP(a_col.w, b_col.w, c_col.w));     // interp (see above)
    }
    f.u = SWR_PERSP(a_uvn.x, b_uvn.x, c_uvn.x);
    f.v = SWR_PERSP(a_uvn.y, b_uvn.y, c_uvn.y);
    // The texel fetch is a dependent gather with a long latency: issue it as soon as u,v exist; the rest of the
    // interpolation, the normal and the fog term run while it is in flight.  (Bilinear textures fetch in the program.)
    f.texel = 0u; f.texel_loaded = false;
    if (dc.tex && dc.tex_h > 0) {
        // global (not generic) address space + 32-bit index: one global_load with the draw's texture pointer as scalar base
        typedef const uint32_t __attribute__((address_space(1)))* global_u32_ptr;
#ifdef SWR_ABL_TEXFIXED      // tools/ablate.py latency probe: same instructions, every lane fetches one of 64 neighbouring texels (always cache hits)
        f.texel = ((global_u32_ptr)(uintptr_t)dc.tex)[texture_nearest_index(dc.tex_w, dc.tex_h, dc.tex_wf, dc.tex_hf, f.u, f.v) & 63u];
#else
        f.texel = ((global_u32_ptr)(uintptr_t)dc.tex)[texture_nearest_index(dc.tex_w, dc.tex_h, dc.tex_wf, dc.tex_hf, f.u, f.v)];
#endif
        f.texel_loaded = true;
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the load above everything that does not feed its address
    if (interp) {
        f.color = make_float4(SWR_PERSP(a_col.x, b_col.x, c_col.x), SWR_PERSP(a_col.y, b_col.y, c_col.y),
                              SWR_PERSP(a_col.z, b_col.z, c_col.z), SWR_PERSP(a_col.w, b_col.w, c_col.w));
    } else {
        f.color = a_col;
    }
    f.clip_z = SWR_PERSP(V.a_cz, V.b_cz, V.c_cz);
#undef SWR_PERSP
    if (interp) {
        float wa = ra * w, wb = rb * w, wc = rc * w;      // :583-585
        // InterpolateData, Vector3 key: weighted sum with the NORMALISED weights, then renormalise (:680-688)
        float n0 = (a_uvn.z * wa + b_uvn.z * wb) + c_uvn.z * wc;
        float n1 = (a_uvn.w * wa + b_uvn.w * wb) + c_uvn.w * wc;
        float n2 = (V.a_wnz * wa + V.b_wnz * wb) + V.c_wnz * wc;
        float len_sq = dot3(n0, n1, n2, n0, n1, n2);
        if (len_sq > 1e-6f) {
            // 1 / MathF.Sqrt(lenSq): both correctly rounded; for lenSq in (1e-6, 1e12] the square root needs no scaling and
            // lies in [1e-3, 1e6], where the reciprocal is recip_core
            float s;
            if (len_sq <= 1.0e12f) s = recip_core(sqrt_core(len_sq));
            else s = 1.0f / sqrtf(len_sq);
            n0 = n0 * s; n1 = n1 * s; n2 = n2 * s;
        }
        f.wn[0] = n0; f.wn[1] = n1; f.wn[2] = n2;
        // Vector4 key: weighted sum only (:690-693); only PHONG_4POINT reads it
        if (PHONG) {
#pragma unroll
            for (int i = 0; i < 3; ++i) f.wpos[i] = (V.a_wpos[i] * wa + V.b_wpos[i] * wb) + V.c_wpos[i] * wc;
        }
    } else {
        f.wn[0] = a_uvn.z; f.wn[1] = a_uvn.w; f.wn[2] = V.a_wnz;
        if (PHONG) { f.wpos[0] = V.a_wpos[0]; f.wpos[1] = V.a_wpos[1]; f.wpos[2] = V.a_wpos[2]; }
    }
    if (PHONG && program == SWR_PROG_PHONG_4POINT) return fs_phong4(dp, dc, f);
    return fs_dust2(dc, f);
}

// ---- Interpolate + Renderer.FragmentShader as ONE straight-line block: speculate, then verify ----------------------------------
// shade_fragment() guards every shortcut where it is taken (division cores only for operands in range, the normal's renormalisation
// only for a length in range, the texture wrap only where the index leaves the texture, ...): eight divergent regions, i.e. eight
// times compare + exec save / restore + a skip branch that is TAKEN in the common case, and a scheduler that cannot move anything
// across them.  For the reference's own frame -- DUST2 program, nearest texture bound, fog range and light direction finite (checked
// per draw, on the scalar side) -- this version computes the common case unconditionally and only COLLECTS the conditions: `safe`
// says whether every shortcut it took was legal for this lane.  The caller ballots `safe` over the chunk; if any shaded lane is
// unsafe the whole chunk is shaded again by shade_fragment (exact, rare: degenerate weights, a texture coordinate exactly on the
// wrap seam, a collapsed normal).  A result is used only after it has been verified, so the output is the reference's bit for bit;
// and in the verified case NaNs are excluded, which makes Math.Clamp a v_med3 and MathF.Max a v_max.
// (the uniforms stay in SGPRs: copies in vector registers -- v_mul with an SGPR operand issues at the slow rate in isolation -- were
//  measured and bought nothing, profiles/r03_raster_experiments.md)
__device__ __forceinline__ float4 shade_dust2_fast(const DrawConsts& u, const TriVaryings& V, float w0f, float w1f, float w2f, bool& safe) {
    const float ra = div_core(w0f, V.a_w, V.a_r1), rb = div_core(w1f, V.b_w, V.b_r1), rc = div_core(w2f, V.c_w, V.c_r1);     // :576-578
    const float inv_sum = (ra + rb) + rc;                                                                                      // :579
    const float w = recip_core(inv_sum);                                                                                       // :582
    // (conditions are combined with `&`: a short-circuit `&&` on lane values is a divergent region again)
    bool ok = V.fastdiv & div_operands_safe3_arith(w0f, w1f, w2f) & (__builtin_fabsf(inv_sum) >= 0x1p-40f);
#define SWR_PERSP(a_, b_, c_) ((((a_) * ra + (b_) * rb) + (c_) * rc) * w)
    const float tu = SWR_PERSP(V.a_uvn.x, V.b_uvn.x, V.c_uvn.x), tv = SWR_PERSP(V.a_uvn.y, V.b_uvn.y, V.c_uvn.y);
    // Texture.Sample, Texture.cs:43-54; the wrap fix-ups are the verifier's business: an index outside the texture marks the lane unsafe
    float fu = tu - (float)f2i(tu), fv = tv - (float)f2i(tv);
    fu += (fu < 0) ? 1.0f : 0.0f;
    fv += (fv < 0) ? 1.0f : 0.0f;
    const int tx = f2i(fu * u.tex_wf), ty = f2i(fv * u.tex_hf);
    ok = ok & ((uint32_t)tx < (uint32_t)u.tex_w) & ((uint32_t)ty < (uint32_t)u.tex_h);
    // (an unsafe lane still loads: the index is clamped into the texture, its texel is never used)
#ifdef SWR_ABL_TEXFIXED      // tools/ablate.py latency probe (wrong image by design): every lane fetches one of 64 neighbouring texels
    const uint32_t ti = min((uint32_t)ty * (uint32_t)u.tex_w + (uint32_t)tx, (uint32_t)(u.tex_w * u.tex_h - 1)) & 63u;
#else
    const uint32_t ti = min((uint32_t)ty * (uint32_t)u.tex_w + (uint32_t)tx, (uint32_t)(u.tex_w * u.tex_h - 1));
#endif
    typedef const uint32_t __attribute__((address_space(1)))* global_u32_ptr;
    const uint32_t texel = ((global_u32_ptr)(uintptr_t)u.tex)[ti];
    __builtin_amdgcn_sched_barrier(0);      // keep the load above everything that does not feed its address
    const float cr = SWR_PERSP(V.a_col.x, V.b_col.x, V.c_col.x), cg = SWR_PERSP(V.a_col.y, V.b_col.y, V.c_col.y),
                cb = SWR_PERSP(V.a_col.z, V.b_col.z, V.c_col.z), ca = SWR_PERSP(V.a_col.w, V.b_col.w, V.c_col.w);
    const float clip_z = SWR_PERSP(V.a_cz, V.b_cz, V.c_cz);
#undef SWR_PERSP
    const float wa = ra * w, wb = rb * w, wc = rc * w;                                                                         // :583-585
    float n0 = (V.a_uvn.z * wa + V.b_uvn.z * wb) + V.c_uvn.z * wc;                                                             // :680-688
    float n1 = (V.a_uvn.w * wa + V.b_uvn.w * wb) + V.c_uvn.w * wc;
    float n2 = (V.a_wnz * wa + V.b_wnz * wb) + V.c_wnz * wc;
    const float len_sq = dot3(n0, n1, n2, n0, n1, n2);
    ok = ok & (len_sq > 1e-6f) & (len_sq <= 1.0e12f);
    const float sc = recip_core(sqrt_core(len_sq));
    n0 = n0 * sc; n1 = n1 * sc; n2 = n2 * sc;
    // Renderer.FragmentShader, Renderer.cs:848-860.  The normal is finite here (verified length) and the light direction is finite
    // (per-draw check), so the dot product is no NaN and MathF.Max is the plain maximum
    const float diffuse = __builtin_fmaxf(0.25f, dot3(n0, n1, n2, -u.light_direction[0], -u.light_direction[1], -u.light_direction[2]));
    const float fog_num = u.fog_end - clip_z;
    ok = ok & div_operand_safe(fog_num);
    const float fog_q = div_core(fog_num, u.fog_den, u.fog_r1);
    float fog = __builtin_amdgcn_fmed3f(fog_q, 0.0f, 1.0f);        // Math.Clamp of a verified-finite quotient
    fog = (fog * fog) * (3.0f - 2.0f * fog);
    const float sl = 0.1f + 0.9f * diffuse;
    const float4 tc = texture_unpack(texel);
    const float br = cr * tc.x, bg = cg * tc.y, bb = cb * tc.z, ba = ca * tc.w;
    float4 o;
    o.x = nm_lerp(u.fog_color[0], (br * sl) * u.light_color[0], fog);
    o.y = nm_lerp(u.fog_color[1], (bg * sl) * u.light_color[1], fog);
    o.z = nm_lerp(u.fog_color[2], (bb * sl) * u.light_color[2], fog);
    o.w = ba;
    safe = ok;
    return o;
}
// the per-draw half of the verification (wave-uniform, evaluated when the chunk's draw changes)
__device__ __forceinline__ bool dust2_fast_applies(const DrawConsts& u) {
    const float lx = u.light_direction[0], ly = u.light_direction[1], lz = u.light_direction[2];
    const bool finite_l = (lx - lx) == 0.0f && (ly - ly) == 0.0f && (lz - lz) == 0.0f;       // no NaN, no infinity
    return u.tex != nullptr && u.tex_h > 0 && u.tex_w > 0 && u.fog_r1 != 0.0f && finite_l;
}

// ---- the same for the build-defined 4-light program (cfg4): Interpolate + fs_phong4 as one straight-line block -------------------------
// fs_phong4 guards nine square roots / normalisations per fragment (view vector, and per light: distance, direction, half vector)
// plus Interpolate's three regions, and reads its 35 uniform words through a generic pointer (vector loads).  Here every core runs
// unconditionally, the conditions under which each core IS the exact operation are ANDed into `safe` (all operands of a division
// core or square-root core in [2^-40, 2^40], the interpolated normal's length in range, the texel index inside the texture), the
// light's `dist / range` becomes the division core on a reciprocal refined once per light, and the uniforms come through the
// constant address space (scalar loads).  In the verified case every operand is finite, so MathF.Max(0, x) is `x > 0 ? x : 0`
// (-0 -> +0 like mathf_max) and Math.Clamp is a median.  A chunk with an unsafe lane is shaded again by shade_fragment.
__device__ __forceinline__ float4 shade_phong4_fast(const DrawParams* __restrict__ dp_generic, const DrawConsts& u, const TriVaryings& V,
                                                   float w0f, float w1f, float w2f, bool& safe) {
    typedef const DrawParams __attribute__((address_space(4)))* const_ptr;
    const const_ptr dp = (const_ptr)(uintptr_t)dp_generic;
    const float ra = div_core(w0f, V.a_w, V.a_r1), rb = div_core(w1f, V.b_w, V.b_r1), rc = div_core(w2f, V.c_w, V.c_r1);     // :576-578
    const float inv_sum = (ra + rb) + rc;                                                                                      // :579
    const float w = recip_core(inv_sum);                                                                                       // :582
    bool ok = V.fastdiv & div_operands_safe3_arith(w0f, w1f, w2f) & (__builtin_fabsf(inv_sum) >= 0x1p-40f);
#define SWR_PERSP(a_, b_, c_) ((((a_) * ra + (b_) * rb) + (c_) * rc) * w)
    const float tu = SWR_PERSP(V.a_uvn.x, V.b_uvn.x, V.c_uvn.x), tv = SWR_PERSP(V.a_uvn.y, V.b_uvn.y, V.c_uvn.y);
    float fu = tu - (float)f2i(tu), fv = tv - (float)f2i(tv);                                                                  // Texture.cs:43-54
    fu += (fu < 0) ? 1.0f : 0.0f;
    fv += (fv < 0) ? 1.0f : 0.0f;
    const int tx = f2i(fu * u.tex_wf), ty = f2i(fv * u.tex_hf);
    ok = ok & ((uint32_t)tx < (uint32_t)u.tex_w) & ((uint32_t)ty < (uint32_t)u.tex_h);
    const uint32_t ti = min((uint32_t)ty * (uint32_t)u.tex_w + (uint32_t)tx, (uint32_t)(u.tex_w * u.tex_h - 1));
    typedef const uint32_t __attribute__((address_space(1)))* global_u32_ptr;
    const uint32_t texel = ((global_u32_ptr)(uintptr_t)u.tex)[ti];
    __builtin_amdgcn_sched_barrier(0);      // keep the load above everything that does not feed its address
    const float cr = SWR_PERSP(V.a_col.x, V.b_col.x, V.c_col.x), cg = SWR_PERSP(V.a_col.y, V.b_col.y, V.c_col.y),
                cb = SWR_PERSP(V.a_col.z, V.b_col.z, V.c_col.z), ca = SWR_PERSP(V.a_col.w, V.b_col.w, V.c_col.w);
#undef SWR_PERSP
    const float wa = ra * w, wb = rb * w, wc = rc * w;                                                                         // :583-585
    float n0 = (V.a_uvn.z * wa + V.b_uvn.z * wb) + V.c_uvn.z * wc;                                                             // :680-688
    float n1 = (V.a_uvn.w * wa + V.b_uvn.w * wb) + V.c_uvn.w * wc;
    float n2 = (V.a_wnz * wa + V.b_wnz * wb) + V.c_wnz * wc;
    const float len_sq = dot3(n0, n1, n2, n0, n1, n2);
    ok = ok & (len_sq > 1e-6f) & (len_sq <= 1.0e12f);
    const float sc = recip_core(sqrt_core(len_sq));
    n0 = n0 * sc; n1 = n1 * sc; n2 = n2 * sc;
    float wp[3];                                                                                                               // :690-693
#pragma unroll
    for (int i = 0; i < 3; ++i) wp[i] = (V.a_wpos[i] * wa + V.b_wpos[i] * wb) + V.c_wpos[i] * wc;
    // fs_phong4 (formula: oracle/swr_oracle.c)
    auto max0 = [](float x) { return x > 0.0f ? x : 0.0f; };
    auto unit = [&ok](const float v[3], float out[3]) -> float {          // out = v / |v|, returns |v|; collects the cores' conditions
        const float ll = dot3(v[0], v[1], v[2], v[0], v[1], v[2]);
        const float len = sqrt_core(ll);
        const float r1 = rcp_refined(len);
        out[0] = div_core(v[0], len, r1); out[1] = div_core(v[1], len, r1); out[2] = div_core(v[2], len, r1);
        // (ll in [2^-40, 2^40] puts len = sqrt(ll) in [2^-20, 2^20]: the division core's range for the denominator needs no test of its own)
        ok = (bool)((int)ok & (int)div_operand_safe(ll) & (int)div_operands_safe3(v[0], v[1], v[2]));
        return len;
    };
    const float4 tc = texture_unpack(texel);
    const float base[4] = { cr * tc.x, cg * tc.y, cb * tc.z, ca * tc.w };
    const float Vd[3] = { dp->u.camera_position[0] - wp[0], dp->u.camera_position[1] - wp[1], dp->u.camera_position[2] - wp[2] };
    float Vn[3];
    (void)unit(Vd, Vn);
    float acc[3] = { 0.1f * base[0], 0.1f * base[1], 0.1f * base[2] };
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const float lpx = dp->u.lights[l].position[0], lpy = dp->u.lights[l].position[1], lpz = dp->u.lights[l].position[2];
        const float range = dp->u.lights[l].range, inten = dp->u.lights[l].intensity;
        const float lc[3] = { dp->u.lights[l].color[0], dp->u.lights[l].color[1], dp->u.lights[l].color[2] };
        const float Ld[3] = { lpx - wp[0], lpy - wp[1], lpz - wp[2] };
        float Ln[3];
        const float dist = unit(Ld, Ln);
        const float ndotl = max0(dot3(n0, n1, n2, Ln[0], Ln[1], Ln[2]));
        ok = ok & div_operand_safe(range);                                    // (dist: checked by unit())
        const float q = div_core(dist, range, rcp_refined(range));
        float att = __builtin_amdgcn_fmed3f(1.0f - q, 0.0f, 1.0f);
        att = att * att;
        const float Hd[3] = { Ln[0] + Vn[0], Ln[1] + Vn[1], Ln[2] + Vn[2] };
        float H[3];
        (void)unit(Hd, H);
        float sp = max0(dot3(n0, n1, n2, H[0], H[1], H[2]));
        sp = sp * sp; sp = sp * sp; sp = sp * sp; sp = sp * sp;
        const float k = inten * att;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float term = (base[i] * ndotl + sp) * (lc[i] * k);
            acc[i] = acc[i] + term;
        }
    }
    safe = ok;
    return make_float4(acc[0], acc[1], acc[2], base[3]);
}
__device__ __forceinline__ bool phong4_fast_applies(const DrawConsts& u) { return u.tex != nullptr && u.tex_h > 0 && u.tex_w > 0; }

// SWR_PROG_DEBUG_VARYINGS: Rasterizer.Interpolate for the varyings no other built-in program reads -- Normal (Rasterizer.cs:610-613),
// ScreenCoords (:390, :598-601), Barycentric (:583-585, :638) -- and the build-defined program that returns them.  Plain IEEE
// divisions (a debug program: no division cores).  na/nb/nc = the three outputs' Normal, s?x/s?y = their screen positions (TriRec).
__device__ __forceinline__ float4 shade_debug_varyings(float w0f, float w1f, float w2f, float wa_clip, float wb_clip, float wc_clip,
                                                       float4 na, float4 nb, float4 nc, const float sx[3], const float sy[3],
                                                       float inv_width, float inv_height) {
    const float ra = w0f / wa_clip, rb = w1f / wb_clip, rc = w2f / wc_clip;          // :576-578
    const float inv_sum = (ra + rb) + rc;                                             // :579
    const float w = 1.0f / inv_sum;                                                   // :582
    const float wa = ra * w, wb = rb * w;                                             // :583-584
#define SWR_PERSP_D(a_, b_, c_) ((((a_) * ra + (b_) * rb) + (c_) * rc) * w)
    // outputs[i].ScreenCoords = (screenCoords[i].X * invWidth, screenCoords[i].Y * invHeight), :390
    const float scx = SWR_PERSP_D(sx[0] * inv_width, sx[1] * inv_width, sx[2] * inv_width);
    const float scy = SWR_PERSP_D(sy[0] * inv_height, sy[1] * inv_height, sy[2] * inv_height);
    const float nx = SWR_PERSP_D(na.x, nb.x, nc.x), ny = SWR_PERSP_D(na.y, nb.y, nc.y), nz = SWR_PERSP_D(na.z, nb.z, nc.z);
#undef SWR_PERSP_D
    return make_float4(scx + nx, scy + ny, wa + nz, wb + 0.5f);
}

// ---- small utility kernels -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_clear(float4* __restrict__ color, float* __restrict__ depth, size_t n,
                                               float4 rgba, int do_color, int do_depth, const Ctrl* __restrict__ ctrl, uint32_t seq) {
    if (batch_poisoned(ctrl, seq)) return;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        if (do_color) color[i] = rgba;
        if (do_depth) depth[i] = SWR_FLOAT_MINVALUE;
    }
}

// MainWindow.OnRender's Vector4 -> Vector3 flatten (MainWindow.cs:234-240), on the device: RGB float, 12 B per pixel
__global__ __launch_bounds__(256) void k_flatten_rgb(const float4* __restrict__ color, float* __restrict__ rgb, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const float4 c = color[i];
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}

__global__ __launch_bounds__(256) void k_texture_sample(const uint8_t* __restrict__ tex, int w, int h,
                                                        const float2* __restrict__ uv, int n, float4* __restrict__ out) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = texture_sample(tex, w, h, uv[i].x, uv[i].y);
}

// row-major RGBA8 -> block-linear (4 x 4-texel blocks of 64 B, see bilinear_texel_offset); w and h are multiples of 4
__global__ __launch_bounds__(256) void k_block_texture(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int w, int h) {
    const size_t n = (size_t)w * (size_t)h;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const int x = (int)(i % (size_t)w), y = (int)(i / (size_t)w);
        dst[bilinear_texel_offset<true>(w, x, y)] = src[i];
    }
}

// sums the per-tile fragment counters; one block (accumulate: adds to out3 instead of overwriting it)
__global__ __launch_bounds__(1024) void k_reduce_tile_stats(const uint32_t* __restrict__ ts, uint32_t n_tiles,
                                                            unsigned long long* __restrict__ out3, int accumulate) {
    __shared__ unsigned long long s[3][16];
    unsigned long long t[3] = { 0, 0, 0 };
    for (uint32_t i = threadIdx.x; i < n_tiles; i += 1024u) { t[0] += ts[3 * i]; t[1] += ts[3 * i + 1]; t[2] += ts[3 * i + 2]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        unsigned lo = (unsigned)t[c], hi = (unsigned)(t[c] >> 32);
        for (int off = 32; off > 0; off >>= 1) {
            unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)hi, off) << 32) | (unsigned)__shfl_xor((int)lo, off);
            t[c] += o; lo = (unsigned)t[c]; hi = (unsigned)(t[c] >> 32);
        }
        if ((threadIdx.x & 63) == 0) s[c][threadIdx.x >> 6] = t[c];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long sum = 0;
        for (int wv = 0; wv < 16; ++wv) sum += s[threadIdx.x][wv];
        out3[threadIdx.x] = accumulate ? out3[threadIdx.x] + sum : sum;
    }
}

// ---- self-test of the exact division / sqrt cores (swr_device.h) against the compiler's own `/` and sqrtf ----
// mode 0: random operands over the whole safe range; 1: quotients next to a rounding midpoint (n = d * (q + half ulp),
// the hard cases of a final correction step); 2: exponents and significands at the guard's boundaries; 3: operands of
// the shapes the raster kernel divides (edge weights in [-1, 0] by clip.w, fog ranges).  Sqrt: random, and x next to
// s*s for random s (rounding boundaries of the square root).  out: [0] divisions tested, [1] division mismatches,
// [2] sqrt tested, [3] sqrt mismatches, [4..7] first mismatching {n, d, got, want} bit patterns.
__device__ __forceinline__ unsigned long long selftest_mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float selftest_operand(unsigned long long r, int emin_field, int emax_field) {
    const uint32_t mant = (uint32_t)r & 0x7fffffu, sign = (uint32_t)(r >> 23) & 1u;
    const uint32_t e = (uint32_t)emin_field + (uint32_t)((r >> 24) % (unsigned long long)(emax_field - emin_field + 1));
    return __uint_as_float((sign << 31) | (e << 23) | mant);
}
__global__ __launch_bounds__(256) void k_selftest_division(unsigned long long seed, int iters, unsigned long long* __restrict__ out) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    unsigned div_bad = 0, sqrt_bad = 0, div_n = 0, sqrt_n = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned long long r0 = selftest_mix(seed ^ (gid * 0x100000001B3ull + (unsigned long long)it));
        const unsigned long long r1 = selftest_mix(r0), r2 = selftest_mix(r1);
        const int mode = (int)(r2 >> 61) & 3;
        float n, d;
        if (mode == 0) {
            n = selftest_operand(r0, 87, 166); d = selftest_operand(r1, 87, 166);
        } else if (mode == 1) {
            d = selftest_operand(r1, 120, 134);
            const double q = 1.0 + (double)((uint32_t)r0 & 0x7fffffu) * (1.0 / 8388608.0) + (1.0 / 16777216.0);   // midpoint of two floats in [1, 2)
            const int sh = (int)((r0 >> 32) % 41ull) - 20;
            n = (float)(ldexp(q, sh) * (double)d);                       // nearest float to d * midpoint
            if ((r0 >> 60) & 1ull) n = __uint_as_float(__float_as_uint(n) + (((r0 >> 61) & 1ull) ? 1u : 0xffffffffu));   // and its neighbours
        } else if (mode == 2) {
            const uint32_t efs[6] = { 87u, 88u, 127u, 165u, 166u, 126u };
            const uint32_t mts[6] = { 0u, 1u, 0x7fffffu, 0x7ffffeu, 0x400000u, (uint32_t)r2 & 0x7fffffu };
            n = __uint_as_float(((uint32_t)(r0 >> 40) & 1u) << 31 | efs[(r0 >> 8) % 6ull] << 23 | mts[(r0 >> 16) % 6ull]);
            d = __uint_as_float(((uint32_t)(r1 >> 40) & 1u) << 31 | efs[(r1 >> 8) % 6ull] << 23 | mts[(r1 >> 16) % 6ull]);
            if (((r2 >> 8) & 15ull) == 0ull) n = 1099511627776.0f;      // 2^40 itself, the top of the range
            if (((r2 >> 12) & 15ull) == 0ull) d = -1099511627776.0f;
        } else {
            n = -(float)((uint32_t)r0 & 0xffffffu) * (1.0f / 16777216.0f) - 1.0e-7f;                       // an edge weight
            d = 0.05f + (float)((uint32_t)r1 & 0xfffffu) * (2000.0f / 1048576.0f);                          // a clip.w
            if ((r2 >> 20) & 1ull) { n = 25.0f - (float)((uint32_t)r0 & 0xffffffu) * (60.0f / 16777216.0f); d = 24.0f; }   // fog
        }
        if (div_operand_safe(n) && div_operand_safe(d)) {
            const float got = div_core(n, d, rcp_refined(d));
            const float want = n / d;
            ++div_n;
            if (__float_as_uint(got) != __float_as_uint(want)) {
                if (atomicAdd(&out[1], 1ull) == 0ull) {
                    out[4] = __float_as_uint(n); out[5] = __float_as_uint(d); out[6] = __float_as_uint(got); out[7] = __float_as_uint(want);
                }
                ++div_bad;
            }
        }
        // square roots: a random operand, or the neighbourhood of a perfect square's rounding boundary
        float x = fabsf(selftest_operand(r2, 87, 166));
        if ((r1 >> 50) & 1ull) {
            const float sroot = fabsf(selftest_operand(r0, 108, 146));
            const float sq = sroot * sroot;
            x = __uint_as_float(__float_as_uint(sq) + (uint32_t)((r1 >> 52) % 5ull) - 2u);
        }
        if (div_operand_safe(x)) {
            const float got = sqrt_core(x), want = sqrtf(x);
            ++sqrt_n;
            if (__float_as_uint(got) != __float_as_uint(want)) { atomicAdd(&out[3], 1ull); ++sqrt_bad; }
        }
    }
    (void)div_bad; (void)sqrt_bad;
    atomicAdd(&out[0], (unsigned long long)div_n);
    atomicAdd(&out[2], (unsigned long long)sqrt_n);
}

// recip_core against the compiler's 1.0f / d for EVERY float d with |d| in [2^-40, 2^83], both signs (2.06e9 operands, a few
// milliseconds): counted into the same out[0] / out[1] / out[4..7] as the divisions above (n = 1)
__global__ __launch_bounds__(256) void k_selftest_recip(unsigned long long* __restrict__ out) {
    const uint32_t span = SWR_RCP_HI_BITS - SWR_DIV_LO_BITS;
    unsigned n_ok = 0, bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i <= (unsigned long long)span; i += (unsigned long long)gridDim.x * 256ull) {
#pragma unroll
        for (uint32_t sign = 0u; sign < 2u; ++sign) {
            const float d = __uint_as_float((SWR_DIV_LO_BITS + (uint32_t)i) | (sign << 31));
            const float got = recip_core(d), want = 1.0f / d;
            ++n_ok;
            if (__float_as_uint(got) != __float_as_uint(want)) {
                if (atomicAdd(&out[1], 1ull) == 0ull) {
                    out[4] = 0x3f800000ull; out[5] = __float_as_uint(d); out[6] = __float_as_uint(got); out[7] = __float_as_uint(want);
                }
                ++bad;
            }
        }
    }
    (void)bad;
    atomicAdd(&out[0], (unsigned long long)n_ok);
}

// Rasterizer.Interpolate (public API, Rasterizer.cs:566-640), batched: one thread per weight triple.
// verts: 3 records of 20 floats {clip4,color4,uv2,normal3,screen2,worldNormal3,pad2}
// out:   n records of 24 floats {clip4,color4,uv2,normal3,screen2,worldNormal3,bary3,pad3}
__global__ __launch_bounds__(256) void k_interpolate(const float* __restrict__ verts, const float* __restrict__ wts,
                                                     int n, int interpolate, float* __restrict__ out) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* A = verts; const float* B = verts + 20; const float* C = verts + 40;
    float w0 = wts[3 * i], w1 = wts[3 * i + 1], w2 = wts[3 * i + 2];
    float ra = w0 / A[3], rb = w1 / B[3], rc = w2 / C[3];
    float inv_sum = (ra + rb) + rc;
    float w = 1.0f / inv_sum;
    float wa = ra * w, wb = rb * w, wc = rc * w;
    float* o = out + 24 * (size_t)i;
#define SWR_PERSP_I(k_) ((((A[k_]) * ra + (B[k_]) * rb) + (C[k_]) * rc) * w)
    for (int k = 0; k < 4; ++k) o[k] = SWR_PERSP_I(k);                 // ClipPosition
    for (int k = 8; k < 10; ++k) o[k] = SWR_PERSP_I(k);                // TexCoord
    for (int k = 13; k < 15; ++k) o[k] = SWR_PERSP_I(k);               // ScreenCoords
    if (interpolate) {
        for (int k = 4; k < 8; ++k) o[k] = SWR_PERSP_I(k);             // Color
        for (int k = 10; k < 13; ++k) o[k] = SWR_PERSP_I(k);           // Normal
        float v[3];
        for (int k = 0; k < 3; ++k) v[k] = (A[15 + k] * wa + B[15 + k] * wb) + C[15 + k] * wc;
        float len_sq = dot3(v[0], v[1], v[2], v[0], v[1], v[2]);
        if (len_sq > 1e-6f) { float s = 1.0f / sqrtf(len_sq); v[0] = v[0] * s; v[1] = v[1] * s; v[2] = v[2] * s; }
        o[15] = v[0]; o[16] = v[1]; o[17] = v[2];
    } else {
        for (int k = 4; k < 8; ++k) o[k] = A[k];
        for (int k = 10; k < 13; ++k) o[k] = A[k];
        for (int k = 15; k < 18; ++k) o[k] = A[k];
    }
#undef SWR_PERSP_I
    o[18] = wa; o[19] = wb; o[20] = wc;
    o[21] = o[22] = o[23] = 0.0f;
}

}  // namespace swr
