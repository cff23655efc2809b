// swr_raster_q.hip.h -- k_raster_q: per-tile rasteriser with an LDS fragment queue (the fast path).
//
// Same contract and arithmetic as k_raster (swr_raster.hip.h; reference Rasterizer.cs:462-538), but the
// expensive part -- Interpolate + fragment program + blend, ~300 VALU per fragment -- no longer runs once
// per (triangle, pixel-slot) at a few % lane utilisation.  Coverage and the depth test run per triangle
// (cheap, exact edge chains); fragments that pass are COMPACTED with wavefront ballot + mbcnt prefix into a
// 64-entry queue in LDS, across triangles, and shaded 64 at a time.
//
// Order semantics are kept exact:
//   * a pixel has at most ONE queued fragment: before a triangle's fragment is depth-tested against a pixel
//     that still has a queued fragment ("pending"), the queue is flushed, so every depth test sees the Z the
//     serial schedule would see, and blending sees the colour it would see;
//   * Z is written at flush time only if the shaded alpha is > 0 (Rasterizer.cs:511-518), exactly like the
//     reference -- nothing is speculated;
//   * a queue holds fragments of one draw (uniform state / texture / program are then wave-uniform);
//   * BlendMode.None needs the row early-out (Rasterizer.cs:520-523), which depends on the alpha of pixels to
//     the left: batches containing such a draw use k_raster (immediate shading) instead.
//
// Tile colour (float4) and Z (float) live in LDS for the whole list: 5 KiB per wave, read from HBM at most
// once (not at all when the clear is fused) and written back once with fully coalesced 1-KiB wave stores.
#pragma once
#include "swr_device.h"
#include "swr_raster.hip.h"

namespace swr {

struct __attribute__((aligned(16))) WaveLds {
    float4 col[256];        // pixel p = (y - y0) * 16 + (x - x0)
    float z[256];
    float4 qw[64];          // queued fragment: w0f, w1f, w2f, depth
    uint4 tbl[64];          // triangle table of the queue: vref0, vref1, vref2, draw_flags
    uint32_t qi[64];        // pixel | table index << 8
};

__global__ __launch_bounds__(256) void k_raster_q(RasterArgs a) {
    if (a.ctrl->poison) return;
    __shared__ WaveLds s_w[4];

    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const uint32_t q = nb >> 3, r = nb & 7u, xcd = b & 7u, kk = b >> 3;
    const uint32_t blk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + kk;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bx = (int)(blk % (uint32_t)a.blocks_x), by = (int)(blk / (uint32_t)a.blocks_x);
    const int tx = bx * 2 + (wave & 1);
    const int ty_local = by * 2 + (wave >> 1);
    const int ty = a.fp.band_ty0 + ty_local;
    if (tx >= a.fp.tiles_x || ty >= a.fp.band_ty1) return;
    const uint32_t tile = (uint32_t)(ty_local * a.fp.tiles_x + tx);
    const uint32_t n = a.tile_count[tile];
    if (n == 0 && !a.clear_color_on && !a.clear_depth_on) return;
    const uint32_t start = a.tile_start[tile];
    WaveLds& L = s_w[wave];

    const int W = a.fp.width, H = a.fp.height;
    const int x0 = tx * SWR_TILE, y0 = ty * SWR_TILE;
    const int tile_end_x = min(x0 + SWR_TILE - 1, W - 1), tile_end_y = min(y0 + SWR_TILE - 1, H - 1);

    // ---- tile init: clear fused, or one coalesced read of the framebuffer ----
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int p = rr * 64 + lane;
        const int gx = x0 + (p & 15), gy = y0 + (p >> 4);
        const bool inb = gx < W && gy < H;
        const size_t gi = (size_t)(gy - a.fp.band_y0) * (size_t)W + (size_t)gx;
        float4 c;
        if (a.clear_color_on) c = make_float4(a.clear_rgba[0], a.clear_rgba[1], a.clear_rgba[2], a.clear_rgba[3]);
        else c = inb ? a.color[gi] : make_float4(0.f, 0.f, 0.f, 0.f);
        float zz;
        if (a.clear_depth_on) zz = SWR_FLOAT_MINVALUE;
        else zz = inb ? a.depth[gi] : SWR_FLOAT_MINVALUE;
        L.col[p] = c;
        L.z[p] = zz;
    }

    // owner mapping for coverage / depth: lane = (row, quad) holds pixels p = lane*4 + s, s = 0..3
    const int row = lane >> 2, quad = lane & 3;
    const int py = y0 + row;
    const int pxb = x0 + quad * 4;
    float4 zreg = *reinterpret_cast<const float4*>(&L.z[lane * 4]);
    unsigned pending = 0;            // bit s: this lane's pixel s has a queued fragment
    int qlen = 0, tbl_len = 0;       // wave-uniform
    uint32_t cur_draw = 0xffffffffu; // draw of the queued fragments
    uint32_t tri_draw = 0xffffffffu; // draw of the current triangle (its depth test is cached below)
    int tri_depth_test = SWR_DEPTH_LESSEQUAL;
    unsigned n_tested = 0, n_shaded = 0, n_written = 0;

    // shades and applies every queued fragment (<= 64, one per lane), then re-syncs the owners' Z registers
#define SWR_FLUSH()                                                                                              \
    do {                                                                                                         \
        const DrawParams* __restrict__ cdp = a.draws + cur_draw;                                                 \
        const int f_program = cdp->program, f_blend = cdp->blend, f_dt = cdp->depth_test;                        \
        if (lane < qlen) {                                                                                       \
            const float4 wv = L.qw[lane];                                                                        \
            const uint32_t qi_ = L.qi[lane];                                                                     \
            const uint4 te = L.tbl[qi_ >> 8];                                                                    \
            const int pix = (int)(qi_ & 255u);                                                                   \
            const float4 src = shade_fragment<true>(cdp, f_program, (te.w >> 31) != 0u, a.vout + te.x, a.vout + te.y,  \
                                              a.vout + te.z, wv.x, wv.y, wv.z);                                  \
            ++n_shaded;                                                                                          \
            if (src.w > 0.0f) {                                              /* Rasterizer.cs:511 */             \
                const float4 dst = L.col[pix];                                                                   \
                L.col[pix] = blend(src, dst, f_blend);                       /* :513-515 */                      \
                if (f_dt != SWR_DEPTH_DISABLED) L.z[pix] = wv.w;             /* :517-518 */                      \
                ++n_written;                                                                                     \
            }                                                                                                    \
        }                                                                                                        \
        zreg = *reinterpret_cast<const float4*>(&L.z[lane * 4]);                                                 \
        pending = 0; qlen = 0; tbl_len = 0; tri_tbl = -1;                                                        \
    } while (0)

    int tri_tbl = -1;
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t my_slot = (base + (uint32_t)lane < n) ? a.tile_list[start + base + (uint32_t)lane] : 0u;
        const int cnt = (int)min(64u, n - base);
        for (int k = 0; k < cnt; ++k) {
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)my_slot, k);
            const TriRec* __restrict__ rp = a.recs + slot;
            const float s0x = rp->sx[0], s1x = rp->sx[1], s2x = rp->sx[2];
            const float s0y = rp->sy[0], s1y = rp->sy[1], s2y = rp->sy[2];
            const uint32_t bbx = rp->bbox_x, bby = rp->bbox_y;
            const int minX = (int)(bbx & 0xffffu), maxX = (int)(bbx >> 16);
            const int minY = (int)(bby & 0xffffu), maxY = (int)(bby >> 16);
            const int startX = max(minX, x0), endX = min(maxX, tile_end_x);        // Rasterizer.cs:471-474
            const int startY = max(minY, y0), endY = min(maxY, tile_end_y);
            if (startX > endX || startY > endY) continue;                          // :476

            const float a01 = s0y - s1y, b01 = s1x - s0x;                          // :445-447
            const float a12 = s1y - s2y, b12 = s2x - s1x;
            const float a20 = s2y - s0y, b20 = s0x - s2x;
            const float fsx = (float)startX, fsy = (float)startY;
            float w0 = a12 * (fsx - s1x) + b12 * (fsy - s1y);                      // :481-483
            float w1 = a20 * (fsx - s2x) + b20 * (fsy - s2y);
            float w2 = a01 * (fsx - s0x) + b01 * (fsy - s0y);
            const int nrow = py - startY, nrow_max = endY - startY;
            for (int i = 0; i < nrow_max; ++i) {                                   // :532-534
                if (i < nrow) { w0 += b12; w1 += b20; w2 += b01; }
            }
            const int npre = pxb - startX, npre_max = min(12, endX - startX);
            for (int i = 0; i < npre_max; ++i) {                                   // :527-529
                if (i < npre) { w0 += a12; w1 += a20; w2 += a01; }
            }
            const bool rowok = py >= startY && py <= endY;

            const float d0 = rp->depth[0], d1 = rp->depth[1], d2 = rp->depth[2];
            const float inv_area = rp->inv_area;
            const uint32_t dflags = rp->draw_flags;
            const uint32_t vr0 = rp->vref[0], vr1 = rp->vref[1], vr2 = rp->vref[2];
            const uint32_t draw = dflags & SWR_DRAW_MASK;
            if (draw != tri_draw) { tri_draw = draw; tri_depth_test = a.draws[draw].depth_test; }
            tri_tbl = -1;

            for (int s = 0; s < 4; ++s) {
                const int px = pxb + s;
                const bool valid = rowok && px >= startX && px <= endX;
                bool inside = valid && ((w0 >= 0 && w1 >= 0 && w2 >= 0) || (w0 <= 0 && w1 <= 0 && w2 <= 0));   // :493-494
                if (__any(inside)) {
                    n_tested += inside ? 1u : 0u;
                    const float w0f = w0 * inv_area, w1f = w1 * inv_area, w2f = w2 * inv_area;              // :498-500
                    const float d = (d0 * w0f + d1 * w1f) + d2 * w2f;                                          // :502
                    bool pass = false;
                    unsigned long long m = 0;
                    int c = 0;
                    for (int attempt = 0; attempt < 2; ++attempt) {
                        const bool conflict = __any(inside && ((pending >> s) & 1u));
                        const float zs = s == 0 ? zreg.x : (s == 1 ? zreg.y : (s == 2 ? zreg.z : zreg.w));
                        pass = inside && depth_func(tri_depth_test, d, zs);                                    // :505
                        m = __ballot(pass);
                        c = __popcll(m);
                        const bool need = qlen > 0 && (conflict || (c > 0 && (qlen + c > 64 || tri_draw != cur_draw)));
                        if (!need) break;
                        SWR_FLUSH();
                        if (!conflict) break;      // Z of these pixels did not change: the test above still holds
                    }
                    if (c) {
                        cur_draw = tri_draw;
                        if (tri_tbl < 0) {
                            tri_tbl = tbl_len++;
                            if (lane == 0) L.tbl[tri_tbl] = make_uint4(vr0, vr1, vr2, dflags);
                        }
                        if (pass) {
                            const int pos = qlen + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                            L.qw[pos] = make_float4(w0f, w1f, w2f, d);
                            L.qi[pos] = (uint32_t)(lane * 4 + s) | ((uint32_t)tri_tbl << 8);
                            pending |= 1u << s;
                        }
                        qlen += c;
                    }
                }
                // x + (-0.0f) == x for every x: lanes left of startX keep the row value
                const bool step = px >= startX;
                w0 += step ? a12 : -0.0f; w1 += step ? a20 : -0.0f; w2 += step ? a01 : -0.0f;
            }
        }
    }
    if (qlen > 0) SWR_FLUSH();
#undef SWR_FLUSH

    // ---- write back: each wave store covers 4 rows x 256 B (colour) / 4 rows x 64 B (Z) ----
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int p = rr * 64 + lane;
        const int gx = x0 + (p & 15), gy = y0 + (p >> 4);
        if (gx < W && gy < H) {
            const size_t gi = (size_t)(gy - a.fp.band_y0) * (size_t)W + (size_t)gx;
            a.color[gi] = L.col[p];
            a.depth[gi] = L.z[p];
        }
    }

#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_tested += (unsigned)__shfl_xor((int)n_tested, off);
        n_shaded += (unsigned)__shfl_xor((int)n_shaded, off);
        n_written += (unsigned)__shfl_xor((int)n_written, off);
    }
    if (lane == 0 && n > 0) {
        uint32_t* ts = a.tile_stats + 3u * tile;
        ts[0] += n_tested; ts[1] += n_shaded; ts[2] += n_written;
    }
}

}  // namespace swr
