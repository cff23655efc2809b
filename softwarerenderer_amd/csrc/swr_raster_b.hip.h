// swr_raster_b.hip.h -- k_raster_b: two-phase per-tile rasteriser (the fast path).
//
// Same contract and arithmetic as k_raster (swr_raster.hip.h; reference Rasterizer.cs:462-538).  One wave owns
// one 16x16 tile and walks the tile's triangle list in batches of up to 64 triangles:
//
//   phase 1 -- ONE LANE PER (triangle, tile) PAIR.  Each lane loads its triangle's setup record (64 loads in
//     flight instead of one scalar load chain per triangle) and walks the pixels of bbox /\ tile in exactly the
//     reference's order with the reference's incremental float32 edge stepping (Rasterizer.cs:481-534), so the
//     edge values are bit-exact by construction.  Covered pixels (:493-494) are appended, as one byte each, to
//     the lane's own region of an LDS byte pool: the concatenation over lanes is the tile's fragment stream in
//     submission order.
//   phase 2 -- ONE LANE PER FRAGMENT.  64 consecutive fragments of the stream are taken at a time; the chunk is
//     cut at the first fragment whose pixel already occurs earlier in the chunk (ds_min owner election) or whose
//     draw differs from the first one's, so inside a chunk every pixel is touched once and state is uniform:
//     depth test, Interpolate, fragment program, blend and the colour/Z update (tile-resident in LDS) are then
//     order-independent inside the chunk and the chunks run in stream order = the serial schedule of the
//     reference (ties under >=, blending, alpha-gated Z writes all exact).  Each fragment lane recomputes its
//     own edge values by replaying its (y-startY)+(x-startX) chain -- at full lane utilisation.
//
// BlendMode.None (row early-out, :520-523) is not handled here: such batches use k_raster.
#pragma once
#include "swr_device.h"
#include "swr_raster.hip.h"

namespace swr {

#define SWR_REGION_CAP 3072

struct __attribute__((aligned(16))) WaveLdsB {
    float4 col[256];                 // pixel p = (y - y0) * 16 + (x - x0)
    float z[256];
    uint32_t owner[256];             // chunk duplicate election (0xffffffff when idle)
    uint32_t slot[64];               // batch: triangle slot ids
    uint32_t pre[65];                // batch: exclusive prefix of covered counts, pre[B] = total
    uint32_t rs[64];                 // batch: region start of each triangle
    uint32_t pad[63];
    uint8_t region[SWR_REGION_CAP];  // covered pixel indices
};

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_raster_b(RasterArgs a) {
    if (a.ctrl->poison) return;
    __shared__ WaveLdsB s_w[4];

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
    WaveLdsB& L = s_w[wave];

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
        L.owner[p] = 0xffffffffu;
    }
    unsigned n_tested = 0, n_shaded = 0, n_written = 0;
#ifdef SWR_DEBUG_COUNTERS
    unsigned dbg_batches = 0, dbg_chunks = 0, dbg_p1_iters = 0, dbg_chain_iters = 0, dbg_chunk_lanes = 0, dbg_shade_chunks = 0, dbg_tris = 0;
#endif

    for (uint32_t base = 0; base < n;) {
        // ================= batch formation: as many triangles as fit the byte pool (>= 1) =================
        const bool have = base + (uint32_t)lane < n;
        const uint32_t slot = have ? a.tile_list[start + base + (uint32_t)lane] : 0u;
        const float4* __restrict__ rq = reinterpret_cast<const float4*>(a.recs + slot);
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
        uint32_t bbx = 0, bby = 0;
        if (have) {
            r0 = rq[0]; r1 = rq[1];
            const float4 r3 = rq[3];
            bbx = __float_as_uint(r3.y); bby = __float_as_uint(r3.z);
        }
        const float s0x = r0.x, s1x = r0.y, s2x = r0.z, s0y = r0.w, s1y = r1.x, s2y = r1.y;
        const int startX = max((int)(bbx & 0xffffu), x0), endX = min((int)(bbx >> 16), tile_end_x);     // Rasterizer.cs:471-474
        const int startY = max((int)(bby & 0xffffu), y0), endY = min((int)(bby >> 16), tile_end_y);
        const bool nonempty = have && startX <= endX && startY <= endY;                                   // :476
        const int area = nonempty ? (endX - startX + 1) * (endY - startY + 1) : 0;
        const int incl = wave_incl_scan(area, lane);
        const unsigned long long fit = __ballot(have && incl <= SWR_REGION_CAP);
        const int B = __popcll(fit);                      // prefix-monotone, so the fitting lanes are 0..B-1; B >= 1
        const bool mine = lane < B;
        const int rs = incl - area;

        // ================= phase 1: lane per (triangle, tile) pair =================
        int cnt = 0;
        if (mine && area > 0) {
            const float a01 = s0y - s1y, b01 = s1x - s0x;                                                 // :445-447
            const float a12 = s1y - s2y, b12 = s2x - s1x;
            const float a20 = s2y - s0y, b20 = s0x - s2x;
            const float fsx = (float)startX, fsy = (float)startY;
            float w0r = a12 * (fsx - s1x) + b12 * (fsy - s1y);                                            // :481-483
            float w1r = a20 * (fsx - s2x) + b20 * (fsy - s2y);
            float w2r = a01 * (fsx - s0x) + b01 * (fsy - s0y);
            float w0 = w0r, w1 = w1r, w2 = w2r;
            int x = startX, y = startY;
            int cur = rs;
            for (int it = 0; it < area; ++it) {
                const bool inside = (w0 >= 0 && w1 >= 0 && w2 >= 0) || (w0 <= 0 && w1 <= 0 && w2 <= 0);  // :493-494
                if (inside) { L.region[cur] = (uint8_t)(((y - y0) << 4) | (x - x0)); ++cur; }
                if (x == endX) {                                                                          // :532-534, :487-489
                    w0r += b12; w1r += b20; w2r += b01;
                    w0 = w0r; w1 = w1r; w2 = w2r;
                    x = startX; ++y;
                } else {                                                                                  // :527-529
                    w0 += a12; w1 += a20; w2 += a01;
                    ++x;
                }
            }
            cnt = cur - rs;
        }
#ifdef SWR_DEBUG_COUNTERS
        { int mx = mine ? area : 0; for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off)); dbg_p1_iters += (unsigned)mx; ++dbg_batches; dbg_tris += (unsigned)B; }
#endif
        const int cincl = wave_incl_scan(cnt, lane);
        const int total = __shfl(cincl, 63);
        L.slot[lane] = slot;
        L.pre[lane] = (uint32_t)(cincl - cnt);
        L.rs[lane] = (uint32_t)rs;
        if (lane == 0) L.pre[64] = (uint32_t)total;
        n_tested += (unsigned)cnt;

        // ================= phase 2: lane per fragment, chunks in stream order =================
#ifdef SWR_ABLATE_PHASE2
        asm volatile("" :: "v"(total));
        for (int pos = 0; pos < 0;) {
#else
        for (int pos = 0; pos < total;) {
#endif
            const int g = pos + lane;
            const bool valid = g < total;
            // triangle of fragment g: largest t with pre[t] <= g  (pre is non-decreasing, pre[0] = 0)
            int lo = 0, hi = 64;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi) >> 1;
                const bool le = (int)L.pre[mid] <= g;
                lo = le ? mid : lo;
                hi = le ? hi : mid;
            }
            const int t = lo;
            const int pix = valid ? (int)L.region[L.rs[t] + (uint32_t)(g - (int)L.pre[t])] : 0;
            const uint32_t fslot = L.slot[t];
            // duplicate election: the lowest lane touching a pixel owns it; any other lane on that pixel must wait
            if (valid) atomicMin(&L.owner[pix], (uint32_t)lane);
            const TriRec* __restrict__ rp = a.recs + fslot;
            const float4* __restrict__ fq = reinterpret_cast<const float4*>(rp);
            const float4 f0 = fq[0], f1 = fq[1], f2 = fq[2], f3 = fq[3];
            const bool dup = valid && L.owner[pix] != (uint32_t)lane;
            const uint32_t dflags = __float_as_uint(f3.w);
            const uint32_t draw = dflags & SWR_DRAW_MASK;
            const uint32_t draw0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)draw);
            const unsigned long long stop = __ballot(!valid || dup || draw != draw0);
            const int cut = stop ? (__ffsll((long long)stop) - 1) : 64;        // >= 1: lane 0 is valid, never dup, own draw
            if (valid) L.owner[pix] = 0xffffffffu;
            const bool act = lane < cut;
#ifdef SWR_DEBUG_COUNTERS
            ++dbg_chunks; dbg_chunk_lanes += (unsigned)cut;
#endif

            const DrawParams* __restrict__ cdp = a.draws + draw0;
            const int f_program = cdp->program, f_blend = cdp->blend, f_dt = cdp->depth_test;
            if (act) {
                const float t0x = f0.x, t1x = f0.y, t2x = f0.z, t0y = f0.w, t1y = f1.x, t2y = f1.y;
                const float d0 = f1.z, d1 = f1.w, d2 = f2.x, inv_area = f2.y;
                const uint32_t fbx = __float_as_uint(f3.y), fby = __float_as_uint(f3.z);
                const int fsX = max((int)(fbx & 0xffffu), x0), fsY = max((int)(fby & 0xffffu), y0);
                const int px = x0 + (pix & 15), py = y0 + (pix >> 4);
                const float a01 = t0y - t1y, b01 = t1x - t0x;
                const float a12 = t1y - t2y, b12 = t2x - t1x;
                const float a20 = t2y - t0y, b20 = t0x - t2x;
                const float fsx = (float)fsX, fsy = (float)fsY;
                float w0 = a12 * (fsx - t1x) + b12 * (fsy - t1y);
                float w1 = a20 * (fsx - t2x) + b20 * (fsy - t2y);
                float w2 = a01 * (fsx - t0x) + b01 * (fsy - t0y);
                const int nrow = py - fsY, ncol = px - fsX;
#ifdef SWR_DEBUG_COUNTERS
                { int mx = nrow + ncol; for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off)); if (lane == 0) dbg_chain_iters += (unsigned)mx; }
#endif
                for (int i = 0; i < nrow; ++i) { w0 += b12; w1 += b20; w2 += b01; }
                for (int i = 0; i < ncol; ++i) { w0 += a12; w1 += a20; w2 += a01; }
                const float w0f = w0 * inv_area, w1f = w1 * inv_area, w2f = w2 * inv_area;               // :498-500
                const float d = (d0 * w0f + d1 * w1f) + d2 * w2f;                                         // :502
                if (depth_func(f_dt, d, L.z[pix])) {                                                       // :505
                    ++n_shaded;
#ifdef SWR_ABLATE_SHADE
                    const float4 src = make_float4(w0f, w1f, w2f, 1.0f);
#else
                    const float4 src = shade_fragment<true>(cdp, f_program, (dflags >> 31) != 0u,
                                                      a.vout + __float_as_uint(f2.z), a.vout + __float_as_uint(f2.w),
                                                      a.vout + __float_as_uint(f3.x), w0f, w1f, w2f);     // :507-509
#endif
                    if (src.w > 0.0f) {                                                                    // :511
                        const float4 dst = L.col[pix];
                        L.col[pix] = blend(src, dst, f_blend);                                             // :513-515
                        if (f_dt != SWR_DEPTH_DISABLED) L.z[pix] = d;                                      // :517-518
                        ++n_written;
                    }
                }
            }
            pos += cut;
        }
        base += (uint32_t)B;
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
        atomicAdd(&a.dbg[2], (unsigned long long)dbg_p1_iters); atomicAdd(&a.dbg[3], (unsigned long long)dbg_chain_iters);
        atomicAdd(&a.dbg[4], (unsigned long long)dbg_chunk_lanes); atomicAdd(&a.dbg[5], (unsigned long long)dbg_tris);
    }
#endif
}

}  // namespace swr
