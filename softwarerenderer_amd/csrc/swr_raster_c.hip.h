// swr_raster_c.hip.h -- k_cover + k_raster_c: coverage masks per (triangle, tile) pair, then a per-tile
// fragment-stream kernel: the raster path (reference Rasterizer.cs:462-538 and, in wireframe mode, :232-340).
//
// k_cover   -- ONE LANE PER (triangle, tile) PAIR over the whole frame (pairs are the entries of the sorted
//   tile lists, so waves are fully packed regardless of how few triangles a tile has).  Each lane walks the
//   pixels of bbox /\ tile in exactly the reference's order with the reference's incremental float32 edge
//   stepping (Rasterizer.cs:481-534) -- bit-exact by construction -- and records coverage (:493-494) as a
//   256-bit mask (16 rows x 16 bits) plus its popcount.
// k_raster_c -- one wave per 16x16 tile.  The tile's fragment stream is the concatenation, in submission
//   order, of the set bits of its pairs' masks (row-major inside a pair = the reference's pixel order).  64
//   consecutive fragments are taken at a time, one per lane; the chunk is cut at the first fragment whose pixel
//   already occurs earlier in the chunk (ds_min owner election) or whose draw differs, so inside a chunk every
//   pixel is touched once and state is uniform: depth test, Interpolate, fragment program, blend and the
//   colour/Z update (tile-resident in LDS) are order-independent inside a chunk, and chunks run in stream order
//   = the serial schedule of the reference (>= ties, blending, alpha-gated Z writes all exact).  Each fragment
//   lane recomputes its own edge values by replaying its (y-startY)+(x-startX) add chain at full lane utilisation.
#pragma once
#include "swr_device.h"
#include "swr_raster.hip.h"

namespace swr {

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

struct CoverArgs {
    const TriRec* __restrict__ recs;
    const uint32_t* __restrict__ tile_list;   // sorted: slot id per pair
    const uint32_t* __restrict__ pair_tile;   // band-local tile index per pair
    uint4* __restrict__ masks;                // 2 x uint4 per pair: row r -> bits (r & 1) * 16 .. of word r >> 1
    uint16_t* __restrict__ counts;            // popcount of the mask
    const unsigned long long* __restrict__ n_pairs;   // device-resident pair total of this batch
    const Ctrl* __restrict__ ctrl;
    FrameParams fp;
};

// LINES: the batch is DebugMode.Wireframe (DrawLine records); compiled out of the filled-triangle instantiation
template <bool LINES>
__global__ __launch_bounds__(256) void k_cover(CoverArgs a) {
    __shared__ uint16_t s_rows[256][18];      // 16 row masks per lane (+2 pad: 9 dwords per lane, conflict-free)
    if (a.ctrl->poison) return;
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n_pairs = (uint32_t)*a.n_pairs;
    uint16_t* mrow16 = s_rows[threadIdx.x];
#pragma unroll
    for (int i = 0; i < 16; ++i) mrow16[i] = 0;
    int cnt = 0;
    if (p < n_pairs) {
        const uint32_t slot = a.tile_list[p];
        const uint32_t tile = a.pair_tile[p];
        const int tx = (int)(tile % (uint32_t)a.fp.tiles_x), ty = a.fp.band_ty0 + (int)(tile / (uint32_t)a.fp.tiles_x);
        const int x0 = tx * SWR_TILE, y0 = ty * SWR_TILE;
        const int tile_end_x = min(x0 + SWR_TILE - 1, a.fp.width - 1), tile_end_y = min(y0 + SWR_TILE - 1, a.fp.height - 1);
        const float4* __restrict__ rq = reinterpret_cast<const float4*>(a.recs + slot);
        const float4 r0 = rq[0], r1 = rq[1], r3 = rq[3];
        const float s0x = r0.x, s1x = r0.y, s2x = r0.z, s0y = r0.w, s1y = r1.x, s2y = r1.y;
        const uint32_t bbx = __float_as_uint(r3.y), bby = __float_as_uint(r3.z);
        const int startX = max((int)(bbx & 0xffffu), x0), endX = min((int)(bbx >> 16), tile_end_x);     // Rasterizer.cs:471-474
        const int startY = max((int)(bby & 0xffffu), y0), endY = min((int)(bby >> 16), tile_end_y);
        const bool is_line = LINES && (__float_as_uint(r3.w) & SWR_FLAG_LINE) != 0u;
        if (is_line && startX <= endX && startY <= endY) {
            // DrawLine, Rasterizer.cs:292-313: every pixel of bbox /\ tile, centre within 0.5 px of the segment
            for (int y = startY; y <= endY; ++y) {
                uint32_t rowbits = 0;
                for (int x = startX; x <= endX; ++x) {
                    float t;
                    if (line_test(s0x, s0y, s1x, s1y, x, y, t)) rowbits |= 1u << (x - x0);
                }
                mrow16[y - y0] = (uint16_t)rowbits;
            }
        } else if (startX <= endX && startY <= endY) {                                                    // :476
            const float a01 = s0y - s1y, b01 = s1x - s0x;                                                 // :445-447
            const float a12 = s1y - s2y, b12 = s2x - s1x;
            const float a20 = s2y - s0y, b20 = s0x - s2x;
            const float fsx = (float)startX, fsy = (float)startY;
            float w0r = a12 * (fsx - s1x) + b12 * (fsy - s1y);                                            // :481-483
            float w1r = a20 * (fsx - s2x) + b20 * (fsy - s2y);
            float w2r = a01 * (fsx - s0x) + b01 * (fsy - s0y);
            // Fast path when nothing in the chain can be NaN/Inf (all nine inputs finite and < 1e30: at most 30 adds
            // cannot overflow): for finite values "all >= 0" <=> min3 >= 0 and "all <= 0" <=> max3 <= 0 (signed zeros
            // satisfy both, exactly like the reference's comparisons), and the loops are plain row / column nests.
            // Anything else (overflowed screen coordinates) takes the literal six-comparison walk below.
            const float lim = 1.0e30f;
            const bool fast = fabsf(a12) < lim && fabsf(a20) < lim && fabsf(a01) < lim && fabsf(b12) < lim && fabsf(b20) < lim &&
                              fabsf(b01) < lim && fabsf(w0r) < lim && fabsf(w1r) < lim && fabsf(w2r) < lim;
            if (fast) {
                const uint32_t bit0 = 1u << (startX - x0);
                for (int y = startY; y <= endY; ++y) {
                    float w0 = w0r, w1 = w1r, w2 = w2r;
                    uint32_t rowbits = 0, bit = bit0;
                    for (int x = startX; x <= endX; ++x) {
                        const bool inside = fminf(fminf(w0, w1), w2) >= 0.0f || fmaxf(fmaxf(w0, w1), w2) <= 0.0f;   // :493-494
                        rowbits |= inside ? bit : 0u;
                        bit <<= 1;
                        w0 += a12; w1 += a20; w2 += a01;                                                  // :527-529
                    }
                    mrow16[y - y0] = (uint16_t)rowbits;
                    w0r += b12; w1r += b20; w2r += b01;                                                   // :532-534
                }
            } else {
                float w0 = w0r, w1 = w1r, w2 = w2r;
                const int area = (endX - startX + 1) * (endY - startY + 1);
                int x = startX, y = startY;
                uint32_t rowbits = 0;
                for (int it = 0; it < area; ++it) {
                    const bool inside = (w0 >= 0 && w1 >= 0 && w2 >= 0) || (w0 <= 0 && w1 <= 0 && w2 <= 0);  // :493-494
                    rowbits |= inside ? (1u << (x - x0)) : 0u;
                    if (x == endX) {                                                                          // :532-534, :487-489
                        mrow16[y - y0] = (uint16_t)rowbits;
                        rowbits = 0;
                        w0r += b12; w1r += b20; w2r += b01;
                        w0 = w0r; w1 = w1r; w2 = w2r;
                        x = startX; ++y;
                    } else {                                                                                  // :527-529
                        w0 += a12; w1 += a20; w2 += a01;
                        ++x;
                    }
                }
            }
        }
        uint32_t mw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {          // word i = rows 2i (low half) and 2i+1 (high half)
            mw[i] = (uint32_t)mrow16[2 * i] | ((uint32_t)mrow16[2 * i + 1] << 16);
            cnt += __popc(mw[i]);
        }
        a.masks[2 * (size_t)p] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
        a.masks[2 * (size_t)p + 1] = make_uint4(mw[4], mw[5], mw[6], mw[7]);
        a.counts[p] = (uint16_t)cnt;
    }
}

// pairs per batch: 64 fills the wave's lanes in the batch set-up; 32 halves the LDS staging so a fifth wave fits per SIMD
#ifndef SWR_BATCH
#define SWR_BATCH 32
#endif
#ifndef SWR_RASTER_MINWAVES
#define SWR_RASTER_MINWAVES 5
#endif
struct __attribute__((aligned(16))) WaveLdsC {
    float4 col[256];                 // pixel p = (y - y0) * 16 + (x - x0)
    float z[256];
    uint32_t owner[256];             // chunk duplicate election (0xffffffff when idle)
    uint32_t mask[SWR_BATCH][8];     // batch: coverage masks
    uint32_t slot[SWR_BATCH];        // batch: triangle slot ids
    uint32_t pre[SWR_BATCH + 4];     // batch: exclusive prefix of covered counts, pre[SWR_BATCH] = total
};

// index (0..31) of the k-th (0-based) set bit of w; requires k < popc(w)
__device__ __forceinline__ int kth_set_bit32(uint32_t w, int k) {
    int base = 0;
#pragma unroll
    for (int width = 16; width >= 1; width >>= 1) {
        const uint32_t lowmask = (width == 32) ? 0xffffffffu : ((1u << width) - 1u);
        const int c = __popc((w >> base) & lowmask);
        const bool up = k >= c;
        k -= up ? c : 0;
        base += up ? width : 0;
    }
    return base;
}

// SWR_RASTER_WPB waves (tiles) per workgroup: 1 lets the dispatcher backfill a finished tile's slot at once
// (tile list lengths vary a lot), 4 shares one workgroup launch between a 2x2 tile quad.
#ifndef SWR_RASTER_WPB
#define SWR_RASTER_WPB 1
#endif
// PROG / BLEND / DT >= 0: every draw of the batch has that program / blend mode / depth test (compile-time state:
// the switches fold away); -1 = read them from the draw at run time.
// EARLYOUT: some draw of the batch uses BlendMode.None, whose row early-out (Rasterizer.cs:520-523) is applied per chunk.
template <bool LINES, bool PHONG, int PROG = -1, int BLEND = -1, int DT = -1, bool EARLYOUT = false>
__global__ __launch_bounds__(64 * SWR_RASTER_WPB, SWR_RASTER_MINWAVES) void k_raster_c(RasterArgs a, const uint4* __restrict__ masks,
                                                                  const uint16_t* __restrict__ counts) {
    __shared__ WaveLdsC s_w[SWR_RASTER_WPB];
    if (a.ctrl->poison) return;

    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const uint32_t q = nb >> 3, r = nb & 7u, xcd = b & 7u, kk = b >> 3;
    const uint32_t blk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + kk;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#if SWR_RASTER_WPB == 4
    const int bx = (int)(blk % (uint32_t)a.blocks_x), by = (int)(blk / (uint32_t)a.blocks_x);
    const int tx = bx * 2 + (wave & 1);
    const int ty_local = by * 2 + (wave >> 1);
#else
    // one tile per workgroup; walk tiles in 2x2 quads so that neighbours (shared triangles) run close in time
    const uint32_t quad = blk >> 2, sub = blk & 3u;
    const int bx = (int)(quad % (uint32_t)a.blocks_x), by = (int)(quad / (uint32_t)a.blocks_x);
    const int tx = bx * 2 + (int)(sub & 1u);
    const int ty_local = by * 2 + (int)(sub >> 1);
#endif
    const int ty = a.fp.band_ty0 + ty_local;
    if (tx >= a.fp.tiles_x || ty >= a.fp.band_ty1) return;
    const uint32_t tile = (uint32_t)(ty_local * a.fp.tiles_x + tx);
    const uint32_t n = a.tile_count[tile];
    if (n == 0 && !a.clear_color_on && !a.clear_depth_on) return;
    const uint32_t start = a.tile_start[tile];
    WaveLdsC& L = s_w[wave];

    const int W = a.fp.width, H = a.fp.height;
    const int x0 = tx * SWR_TILE, y0 = ty * SWR_TILE;

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
        L.owner[p] = 0xffffffffu;
    }
    unsigned n_tested = 0, n_shaded = 0, n_written = 0;
    uint32_t carry_key = 0xffffffffu;      // EARLYOUT: (pair, row) of the previous chunk's last fragment ...
    bool carry_dead = false;               // ... and whether that row segment has already hit its `break`
#ifdef SWR_DEBUG_COUNTERS
    unsigned dbg_batches = 0, dbg_chunks = 0, dbg_chunk_lanes = 0, dbg_chain = 0;
#endif

    for (uint32_t base = 0; base < n; base += (uint32_t)SWR_BATCH) {
        // ---- batch: the next (up to) 64 pairs of this tile; masks and counts staged in LDS ----
        const bool have = lane < SWR_BATCH && base + (uint32_t)lane < n;
        const uint32_t pidx = start + base + (uint32_t)lane;
        const uint32_t slot = have ? a.tile_list[pidx] : 0u;
        const int cnt = have ? (int)counts[pidx] : 0;
        uint4 m0 = make_uint4(0u, 0u, 0u, 0u), m1 = m0;
        if (cnt > 0) { m0 = masks[2 * (size_t)pidx]; m1 = masks[2 * (size_t)pidx + 1]; }
        const int cincl = wave_incl_scan(cnt, lane);
        const int total = __shfl(cincl, 63);
        if (total == 0) continue;
        if (lane < SWR_BATCH) {
            *reinterpret_cast<uint4*>(&L.mask[lane][0]) = m0;
            *reinterpret_cast<uint4*>(&L.mask[lane][4]) = m1;
            L.slot[lane] = slot;
            L.pre[lane] = (uint32_t)(cincl - cnt);
        }
        if (lane == 0) L.pre[SWR_BATCH] = (uint32_t)total;
        n_tested += (unsigned)cnt;
#ifdef SWR_DEBUG_COUNTERS
        ++dbg_batches;
#endif

        // ---- fragment stream of the batch, 64 at a time ----
        for (int pos = 0; pos < total;) {
            const int g = pos + lane;
            const bool valid = g < total;
            // pair of fragment g: largest t with pre[t] <= g  (pre is non-decreasing, pre[0] = 0)
            int lo = 0, hi = SWR_BATCH;
#pragma unroll
            for (int it = 0; it < (SWR_BATCH == 64 ? 6 : 5); ++it) {
                const int mid = (lo + hi) >> 1;
                const bool le = (int)L.pre[mid] <= g;
                lo = le ? mid : lo;
                hi = le ? hi : mid;
            }
#ifdef SWR_ABLATE_SEARCH
            const int t = (lane >> 4) & 3; asm volatile("" :: "v"(lo));
#else
            const int t = lo;
#endif
            int k = valid ? g - (int)L.pre[t] : 0;
            const uint32_t fslot = L.slot[t];
            const TriRec* __restrict__ rp = a.recs + fslot;
            const float4* __restrict__ fq = reinterpret_cast<const float4*>(rp);
            const float4 f0 = fq[0], f1 = fq[1], f2 = fq[2], f3 = fq[3];
            // k-th covered pixel of pair t in row-major order
            const uint4 ma = *reinterpret_cast<const uint4*>(&L.mask[t][0]);
            const uint4 mb = *reinterpret_cast<const uint4*>(&L.mask[t][4]);
            int pix = 0;
            {
                // k-th set bit of the 256-bit mask: 3-level selection over the 8 words, then inside the word
                const int c0 = __popc(ma.x), c1 = __popc(ma.y), c2 = __popc(ma.z), c3 = __popc(ma.w);
                const int c4 = __popc(mb.x), c5 = __popc(mb.y), c6 = __popc(mb.z);
                const int s01 = c0 + c1, s23 = c2 + c3, s45 = c4 + c5, s0123 = s01 + s23;
                const bool up1 = k >= s0123;
                k -= up1 ? s0123 : 0;
                const uint32_t a0 = up1 ? mb.x : ma.x, a1 = up1 ? mb.y : ma.y, a2 = up1 ? mb.z : ma.z, a3 = up1 ? mb.w : ma.w;
                const int sa = up1 ? s45 : s01, ca0 = up1 ? c4 : c0, ca2 = up1 ? c6 : c2;
                const bool up2 = k >= sa;
                k -= up2 ? sa : 0;
                const uint32_t b0 = up2 ? a2 : a0, b1 = up2 ? a3 : a1;
                const int cb0 = up2 ? ca2 : ca0;
                const bool up3 = k >= cb0;
                k -= up3 ? cb0 : 0;
                const uint32_t wsel = up3 ? b1 : b0;
                const int wi = (up1 ? 4 : 0) + (up2 ? 2 : 0) + (up3 ? 1 : 0);
                const bool okk = valid && k < __popc(wsel);          // always true for a valid fragment
                pix = wi * 32 + kth_set_bit32(wsel, okk ? k : 0);
            }
#ifdef SWR_ABLATE_KTH
            asm volatile("" :: "v"(pix)); pix = (g * 7) & 255;
#endif
            // duplicate election: the lowest lane touching a pixel owns it; any other lane on that pixel must wait
            if (valid) atomicMin(&L.owner[pix], (uint32_t)lane);
            const uint32_t dflags = __float_as_uint(f3.w);
            const uint32_t draw = dflags & SWR_DRAW_MASK;
            const uint32_t draw0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)draw);
            const bool dup = valid && L.owner[pix] != (uint32_t)lane;
            const unsigned long long stop = __ballot(!valid || dup || draw != draw0);
            const int cut = stop ? (__ffsll((long long)stop) - 1) : 64;        // >= 1: lane 0 is valid, never dup, own draw
            if (valid) L.owner[pix] = 0xffffffffu;
            const bool act = lane < cut;
#ifdef SWR_DEBUG_COUNTERS
            ++dbg_chunks; dbg_chunk_lanes += (unsigned)cut;
#endif

            const DrawParams* __restrict__ cdp = a.draws + draw0;
            const int f_program = PROG >= 0 ? PROG : cdp->program, f_blend = BLEND >= 0 ? BLEND : cdp->blend, f_dt = DT >= 0 ? DT : cdp->depth_test;
            bool e_pass = false, e_alpha = false;          // EARLYOUT: results held until the row kills are known
            float e_d = 0.0f;
            float4 e_src = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act) {
                const float t0x = f0.x, t1x = f0.y, t2x = f0.z, t0y = f0.w, t1y = f1.x, t2y = f1.y;
                const float d0 = f1.z, d1 = f1.w, d2 = f2.x, inv_area = f2.y;
                const uint32_t fbx = __float_as_uint(f3.y), fby = __float_as_uint(f3.z);
                const int fsX = max((int)(fbx & 0xffffu), x0), fsY = max((int)(fby & 0xffffu), y0);
                const int px = x0 + (pix & 15), py = y0 + (pix >> 4);
                const bool is_line = LINES && (dflags & SWR_FLAG_LINE) != 0u;
                float w0f, w1f, w2f, d;
                if (is_line) {
                    // DrawLine fragment, Rasterizer.cs:299-322: weights (1-t, t, 0) on outputs[0], outputs[1], outputs[0]
                    float t;
                    (void)line_test(t0x, t0y, t1x, t1y, px, py, t);
                    w0f = 1.0f - t; w1f = t; w2f = 0.0f;
                    d = 1.0f / (d0 * (1.0f - t) + d1 * t);                                                // :315
                } else {
                    const float a01 = t0y - t1y, b01 = t1x - t0x;
                    const float a12 = t1y - t2y, b12 = t2x - t1x;
                    const float a20 = t2y - t0y, b20 = t0x - t2x;
                    const float fsx = (float)fsX, fsy = (float)fsY;
                    float w0 = a12 * (fsx - t1x) + b12 * (fsy - t1y);                                     // :481-483
                    float w1 = a20 * (fsx - t2x) + b20 * (fsy - t2y);
                    float w2 = a01 * (fsx - t0x) + b01 * (fsy - t0y);
                    const int nrow = py - fsY, ncol = px - fsX;
                    for (int i = 0; i < nrow; ++i) { w0 += b12; w1 += b20; w2 += b01; }                   // :532-534
                    for (int i = 0; i < ncol; ++i) { w0 += a12; w1 += a20; w2 += a01; }                   // :527-529
                    w0f = w0 * inv_area; w1f = w1 * inv_area; w2f = w2 * inv_area;                        // :498-500
                    d = (d0 * w0f + d1 * w1f) + d2 * w2f;                                                 // :502
                }
                if (!EARLYOUT) {
                    if (depth_func(f_dt, d, L.z[pix])) {                                                   // :505 / :318
                        ++n_shaded;
                        const float4 src = shade_fragment<true, PHONG>(cdp, f_program, (dflags & SWR_FLAG_INTERP) != 0u,
                                                                a.vout + __float_as_uint(f2.z), a.vout + __float_as_uint(f2.w),
                                                                a.vout + __float_as_uint(f3.x), w0f, w1f, w2f);     // :507-509 / :321-323
                        // triangles: W > 0 (:511); lines: W != 0 (:325)
                        if (is_line ? (src.w != 0.0f) : (src.w > 0.0f)) {
                            const float4 dst = L.col[pix];
                            L.col[pix] = blend(src, dst, f_blend);                                         // :513-515
                            if (f_dt != SWR_DEPTH_DISABLED) L.z[pix] = d;                                  // :517-518
                            ++n_written;
                        }
                    }
                } else {
                    e_d = d;
                    e_pass = depth_func(f_dt, d, L.z[pix]);
                    if (e_pass) {
                        e_src = shade_fragment<true, PHONG>(cdp, f_program, (dflags & SWR_FLAG_INTERP) != 0u,
                                                            a.vout + __float_as_uint(f2.z), a.vout + __float_as_uint(f2.w),
                                                            a.vout + __float_as_uint(f3.x), w0f, w1f, w2f);
                        e_alpha = is_line ? (e_src.w != 0.0f) : (e_src.w > 0.0f);
                    }
                }
            }
            if (EARLYOUT) {
                // canEarlyOut (Rasterizer.cs:430,520-523): under BlendMode.None the first fragment of a row (within this
                // triangle and tile) that passes depth but fails alpha ends the row -- nothing to its right is visited.
                // Fragments of one (pair, row) are consecutive in the stream, so inside a chunk that is a segmented
                // "any failure to my left" over lanes; a segment cut by the chunk boundary carries its state over.
                bool killed = false;
                const bool none = f_blend == SWR_BLEND_NONE;                       // uniform: a chunk holds one draw
                const uint32_t key = ((base + (uint32_t)t) << 4) | (uint32_t)(pix >> 4);
                if (none) {
                    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
                    const bool head = act && (lane == 0 || key != prev);
                    const unsigned long long H = __ballot(head);
                    const unsigned long long F = __ballot(act && e_pass && !e_alpha);
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const unsigned long long hm = H & (below | (1ull << lane));
                    const int headlane = hm ? 63 - __clzll((long long)hm) : 0;
                    const unsigned long long seg_below = below & ~((1ull << headlane) - 1ull);
                    killed = act && (F & seg_below) != 0ull;
                    if (act && headlane == 0 && carry_dead && key == carry_key) killed = true;
                    const bool dead_after = killed || (act && e_pass && !e_alpha);
                    carry_key = (uint32_t)__shfl((int)key, cut - 1);
                    carry_dead = __shfl((int)dead_after, cut - 1) != 0;
                } else {
                    carry_dead = false;
                }
                if (act) {
                    if (killed) {
                        --n_tested;                                                // never visited by the reference
                    } else if (e_pass) {
                        ++n_shaded;
                        if (e_alpha) {
                            const float4 dst = L.col[pix];
                            L.col[pix] = blend(e_src, dst, f_blend);
                            if (f_dt != SWR_DEPTH_DISABLED) L.z[pix] = e_d;
                            ++n_written;
                        }
                    }
                }
            }
            pos += cut;
        }
    }

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
#ifdef SWR_DEBUG_COUNTERS
    if (lane == 0 && a.dbg) {
        atomicAdd(&a.dbg[0], (unsigned long long)dbg_batches); atomicAdd(&a.dbg[1], (unsigned long long)dbg_chunks);
        atomicAdd(&a.dbg[4], (unsigned long long)dbg_chunk_lanes); atomicAdd(&a.dbg[3], (unsigned long long)dbg_chain);
    }
#endif
}

}  // namespace swr
