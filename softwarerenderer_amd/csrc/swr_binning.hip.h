// swr_binning.hip.h -- (triangle, 16x16 tile) pair binning that preserves submission order.
//
// The reference visits, for every triangle, every 16x16 tile of its clamped pixel bbox
// (Rasterizer.cs:449-476) and the per-pixel result depends on the ORDER triangles reach a
// pixel (depth ties under `>=`, blending).  The oracle of record is the serial order, so each
// tile's list must hold its triangles in ascending slot order:
//   k_bin<COUNT>  : tile_count[tile] += 1 per pair               (integer atomics, order-free)
//   k_scan        : exclusive scan -> tile_start, total pairs (k_scan_sums + k_scan_apply)
//   k_bin<FILL>   : list[tile_start + atomic cursor] = slot       (arbitrary order inside a tile)
//   k_sort_tiles  : one wave per tile sorts its segment ascending (restores submission order)
// plus the raster kernel's dispatch order (tiles by descending work, a counting sort) without launches of its own: the weight of a
// tile is its pair count of THIS flush + the fragments the raster kernel counted in it in the PREVIOUS flush (any permutation is
// correct; a frame resembles the one before), so k_scan_apply -- which holds the counts -- builds the histogram and the last blocks
// of k_bin<FILL>'s grid place the tiles (tile_place_block).
#pragma once
#include "swr_device.h"

#define SWR_ORDER_BUCKETS 256      // heaviest-first tile order: counting-sort buckets (see tile_place_block); k_vertex clears 2 x 256 words

namespace swr {

struct BinArgs {
    const unsigned long long* __restrict__ slot_tb;
    const TriRec* __restrict__ recs;
    uint32_t slot_lo, slot_hi;           // slots [lo, hi) are binned in this round
    uint32_t spt;                        // slots per thread = slots per triangle (2 filled, 6 wireframe): the odd
                                         // fan slots are almost always empty, so a thread walks its triangle's slots
    int tiles_x;
    int band_ty0, band_ty1;              // tile rows binned: the contiguous band, or [0, tiles_y) with `band` deciding row by row
    BandMap band;                        // tile-row ownership (contiguous band or interleaved stripes)
    int width, height;
    uint32_t* __restrict__ tile_count;   // COUNT: incremented; FILL: used as cursor (zeroed again before)
    const uint32_t* __restrict__ tile_start;
    uint32_t* __restrict__ tile_list;
    uint32_t list_capacity;
    Counters* __restrict__ counters;
    Ctrl* __restrict__ ctrl;
    uint32_t seq;                        // sequence number of the batch (what a list overflow reports in Ctrl::first_bad)
    uint32_t replayable;                 // 1: optimistic flush -- a list overflow poisons the batch (see bin_overflow)
    const unsigned long long* __restrict__ total;
    uint8_t* __restrict__ want;          // per slot: which of its (<= 8) tiles passed pair_may_cover -- written by COUNT, read by FILL
    uint32_t tpw;                        // triangles per wave: 64, or fewer for small batches (a wave works through its big
                                         // triangles one after the other: with few triangles more, emptier waves finish sooner)
    // FILL only: blocks [bin_blocks, gridDim.x) place the tiles in the raster kernel's dispatch order (tile_place_block)
    uint32_t bin_blocks;
    int order_tiles_y;                   // tile rows of the band
    const uint8_t* __restrict__ tile_bucket;    // k_scan_apply: order_bucket of every tile
    const uint32_t* __restrict__ order_hist;    // tiles per bucket
    uint32_t* __restrict__ order_cursor;        // per bucket: tiles placed so far
    uint4* __restrict__ tile_order;             // per dispatch position: {tile, first list entry, pairs, -} (tile_place_block)
};

// Can triangle (sx, sy, pixel bbox) cover ANY pixel of tile (tx, ty)?  Conservative: returns false only when
// provably no pixel of bbox /\ tile can pass the reference's coverage test (all three incrementally stepped
// float32 edge values >= 0, or all three <= 0; Rasterizer.cs:481-494).  The reference visits such tiles and
// finds nothing, so dropping the pair changes no pixel and no counter.
//
// Proof sketch.  Let R be the pixel rectangle bbox /\ tile and, for edge k with float coefficients (a, b) and
// reference vertex (rx, ry), E(x,y) = a*(x-rx) + b*(y-ry) in real arithmetic; u = 2^-24.  Every value the
// reference's float chain takes is fl-arithmetic on points of R: the start value costs <= 5 roundings of
// quantities bounded by M = |a|*max|x-rx| + |b|*max|y-ry| over R, and each of the <= 30 chain adds rounds a value
// of magnitude <= M(1+tiny); so |W - E| <= 35uM(1+tiny) at every pixel of R.  E is linear, so its extrema over R
// are at corners; evaluating them in float32 as below costs <= 3 roundings per term, |Efl - E| <= 6uM, and the
// float M' satisfies M' >= M(1-4u).  With delta = 64uM':
//   Efl_max < -delta  =>  W <= E_max + 35uM <= Efl_max + 41uM < -64uM(1-4u) + 41uM < 0 on all of R  (kills "all >= 0")
//   Efl_min >  delta  =>  W > 0 on all of R                                                          (kills "all <= 0")
// The analysis needs every product and sum finite: screen coordinates below 1e15 in magnitude (tested once per triangle, not
// per tile and edge as in round 2) bound every term by 1e31; anything else (NaN / Inf included) means "keep".
__device__ __forceinline__ bool pair_may_cover(const float sx[3], const float sy[3], int minX, int maxX, int minY, int maxY,
                                               int tx, int ty, int width, int height, bool is_line) {
    if (is_line) return true;         // DrawLine edges: keep every tile of the line's bbox (the test below is for triangles)
    const int x0 = tx * SWR_TILE, y0 = ty * SWR_TILE;
    const int startX = max(minX, x0), endX = min(maxX, min(x0 + SWR_TILE - 1, width - 1));
    const int startY = max(minY, y0), endY = min(maxY, min(y0 + SWR_TILE - 1, height - 1));
    if (startX > endX || startY > endY) return false;                 // Rasterizer.cs:476: nothing visited
    // edge k: coefficients exactly as RasterizeTriangle forms them (Rasterizer.cs:445-447), reference vertex :481-483
    const float ea[3] = { sy[1] - sy[2], sy[2] - sy[0], sy[0] - sy[1] };   // a12, a20, a01
    const float eb[3] = { sx[2] - sx[1], sx[0] - sx[2], sx[1] - sx[0] };   // b12, b20, b01
    const float rx[3] = { sx[1], sx[2], sx[0] };
    const float ry[3] = { sy[1], sy[2], sy[0] };
    const float fxs = (float)startX, fxe = (float)endX, fys = (float)startY, fye = (float)endY;
    const float lim = 1.0e15f;
    const bool tame = fabsf(sx[0]) < lim && fabsf(sx[1]) < lim && fabsf(sx[2]) < lim && fabsf(sy[0]) < lim && fabsf(sy[1]) < lim && fabsf(sy[2]) < lim;
    if (!tame) return true;
    bool any_neg = false, any_pos = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = ea[k], b = eb[k];
        const float dxs = fxs - rx[k], dxe = fxe - rx[k], dys = fys - ry[k], dye = fye - ry[k];
        const float ax_s = a * dxs, ax_e = a * dxe, by_s = b * dys, by_e = b * dye;
        const float emax = fmaxf(ax_s, ax_e) + fmaxf(by_s, by_e);
        const float emin = fminf(ax_s, ax_e) + fminf(by_s, by_e);
        const float m = fabsf(a) * fmaxf(fabsf(dxs), fabsf(dxe)) + fabsf(b) * fmaxf(fabsf(dys), fabsf(dye));
        const float delta = m * (64.0f / 16777216.0f);
        any_neg = any_neg || emax < -delta;
        any_pos = any_pos || emin > delta;
    }
    return !(any_neg && any_pos);
}

// A FILL position beyond the list capacity.  k_scan_apply already refuses batches whose COUNT total does not fit, so this
// means COUNT and FILL disagreed (a bug) or a debug capacity (SWR_DEBUG_FILL_CAPACITY) -- either way the pair must not be
// dropped silently: the sticky flag goes up, and in an optimistic flush the batch poisons itself BEFORE any of its later
// kernels (sort, cover, raster all return on poison) touches the framebuffer, so the host replays it exactly.
__device__ __forceinline__ void bin_overflow(const BinArgs& a) {
    a.counters->overflow = 1u;
    if (a.ctrl->host_flag) __hip_atomic_store(a.ctrl->host_flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.replayable) {
        atomicMin(&a.ctrl->first_bad, a.seq);
        atomicMax(&a.ctrl->need, 2ull * (unsigned long long)a.list_capacity + 4096ull);
        __hip_atomic_store(&a.ctrl->poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.ctrl->host_flag) __hip_atomic_store(a.ctrl->host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- heaviest-first tile order for the raster kernel ---------------------------------------------
// One wave rasterises one tile and tiles differ a lot in work, so the launch ends with a few long tiles running on
// an otherwise idle chip.  Workgroups are dispatched in index order as slots free up: handing out the tiles by
// descending work makes that dispatch a longest-first schedule.  Weight of a tile = the fragments k_raster_c counted in it
// in the previous flush (RasterArgs::tile_work) + a per-pair cost on this flush's pair count: both are known when the scan
// runs, so the order costs no launch (round 2: k_cover summed this flush's fragments per tile, then k_tile_hist and
// k_tile_place ran between k_cover and the raster kernel: 11 us of launches + k_cover's tail).  With pair counts alone cfg2's
// raster kernel is 10 % slower, cfg3's the same (profiles/r03_raster_experiments.md, section 9).
// Counting sort over 8 buckets per octave; the order inside a bucket is arbitrary (tiles are independent).
__device__ __forceinline__ uint32_t order_bucket(uint32_t w) {
    if (w == 0u) return 0u;
    const int e = 31 - __clz((int)w);
    const uint32_t m = e >= 3 ? ((w >> (e - 3)) & 7u) : ((w << (3 - e)) & 7u);
    return min((uint32_t)(e * 8) + m + 1u, (uint32_t)SWR_ORDER_BUCKETS - 1u);
}
__device__ __forceinline__ uint32_t tile_weight(uint32_t frags, uint32_t pairs) { return frags + 16u * pairs; }

// Thread i of an ordering block -> tile: a block takes a 16x16-tile region, a wave an 8x8 quarter of it, so that
// tiles which end up next to each other in the order (same block, same bucket, consecutive LDS ranks) are neighbours
// on screen: the raster kernel hands runs of 64 consecutive entries to one XCD, whose L2 then serves the triangle
// records and vertices that neighbouring tiles share.  Returns 0xffffffff outside the band.
__device__ __forceinline__ uint32_t order_tile_of_thread(uint32_t block, int tiles_x, int tiles_y) {
    const int sbx = (tiles_x + 15) >> 4;
    const int bx = (int)(block % (uint32_t)sbx), by = (int)(block / (uint32_t)sbx);
    const int t = threadIdx.x, q = t >> 6, w = t & 63;
    const int tx = bx * 16 + (q & 1) * 8 + (w & 7), ty = by * 16 + (q >> 1) * 8 + (w >> 3);
    return (tx < tiles_x && ty < tiles_y) ? (uint32_t)(ty * tiles_x + tx) : 0xffffffffu;
}
__host__ __device__ inline uint32_t order_blocks(int tiles_x, int tiles_y) { return (uint32_t)(((tiles_x + 15) / 16) * ((tiles_y + 15) / 16)); }

// One 256-thread block of the placement (the blocks behind k_bin<FILL>'s own): every tile of its 16x16 region gets its position in
// the order = tiles in heavier buckets + tiles of its bucket placed so far.  s_suf / s_cnt: SWR_ORDER_BUCKETS words of LDS each.
__device__ __forceinline__ void tile_place_block(const BinArgs& a, uint32_t block, uint32_t* s_suf, uint32_t* s_cnt) {
    const uint32_t t = threadIdx.x;
    s_suf[t] = a.order_hist[t];
    s_cnt[t] = 0u;
    __syncthreads();
    for (uint32_t off = 1; off < (uint32_t)SWR_ORDER_BUCKETS; off <<= 1) {       // inclusive suffix sum: tiles in this and heavier buckets
        const uint32_t v = t + off < (uint32_t)SWR_ORDER_BUCKETS ? s_suf[t + off] : 0u;
        __syncthreads();
        s_suf[t] += v;
        __syncthreads();
    }
    const uint32_t i = order_tile_of_thread(block, a.tiles_x, a.order_tiles_y);
    uint32_t b = 0, rank = 0;
    if (i != 0xffffffffu) {
        b = a.tile_bucket[i];
        rank = atomicAdd(&s_cnt[b], 1u);
    }
    __syncthreads();
    // one global reservation per (block, bucket): same-address atomics with a return value are slow
    const uint32_t mine = s_cnt[t];
    __syncthreads();
    if (mine) s_cnt[t] = (s_suf[t] - a.order_hist[t]) + atomicAdd(&a.order_cursor[t], mine);
    __syncthreads();
    // The record carries what the tile's sort wave and raster wave would otherwise fetch with two more dependent loads (order -> tile ->
    // start / count -> list): its list start, and its pair count as the difference of two starts -- the scan is final here, the counts
    // are not (this block runs in k_bin<FILL>'s grid, whose cursors they are).
    if (i != 0xffffffffu) {
        const uint32_t n_tiles = (uint32_t)(a.tiles_x * a.order_tiles_y);
        const uint32_t st = a.tile_start[i];
        const unsigned long long nxt = i + 1u < n_tiles ? (unsigned long long)a.tile_start[i + 1u] : *a.total;
        a.tile_order[s_cnt[b] + rank] = make_uint4(i, st, (uint32_t)(nxt - st), 0u);
    }
}

// what binning needs to know about one primitive slot
struct SlotData {
    int tminx, tminy, nx, ny;            // tile bbox clamped to the band (nx = ny = 0: nothing to bin)
    float sx[3], sy[3];
    int minX, maxX, minY, maxY;          // pixel bbox
    bool is_line;
};
__device__ __forceinline__ SlotData slot_none() {
    SlotData s;
    s.tminx = s.tminy = s.nx = s.ny = 0;
    s.sx[0] = s.sx[1] = s.sx[2] = s.sy[0] = s.sy[1] = s.sy[2] = 0.f;
    s.minX = s.minY = 0; s.maxX = s.maxY = -1; s.is_line = false;
    return s;
}
// tile bbox word (k_setup) -> clamped to the band; false when the slot has nothing in this band
__device__ __forceinline__ bool slot_tiles(const BinArgs& a, unsigned long long tb, SlotData& s) {
    if (tb == SWR_TB_INVALID) return false;
    s.tminx = (int)(tb & 0xffff);
    const int tmaxx = (int)((tb >> 16) & 0xffff);
    s.tminy = max((int)((tb >> 32) & 0xffff), a.band_ty0);
    const int tmaxy = min((int)((tb >> 48) & 0xffff), a.band_ty1 - 1);
    s.nx = tmaxx - s.tminx + 1;
    s.ny = tmaxy - s.tminy + 1;
    if (s.ny <= 0) { s.nx = 0; s.ny = 0; return false; }
    return true;
}
// with_rec = false: only the tile bbox (FILL of a small triangle replays COUNT's decisions from a.want)
// tb = a.slot_tb[slot] (SWR_TB_INVALID for a slot outside the round): the caller fetches it one slot ahead
__device__ __forceinline__ SlotData slot_load(const BinArgs& a, uint32_t slot, unsigned long long tb, bool with_rec_small) {
    SlotData s = slot_none();
    if (slot_tiles(a, tb, s) && (with_rec_small || s.nx * s.ny > 8)) {
        const float4* __restrict__ rq = reinterpret_cast<const float4*>(a.recs + slot);
        const float4 r0 = rq[0], r1 = rq[1], r3 = rq[3];
        s.sx[0] = r0.x; s.sx[1] = r0.y; s.sx[2] = r0.z; s.sy[0] = r0.w; s.sy[1] = r1.x; s.sy[2] = r1.y;
        const uint32_t bbx = __float_as_uint(r3.y), bby = __float_as_uint(r3.z);
        s.minX = (int)(bbx & 0xffffu); s.maxX = (int)(bbx >> 16); s.minY = (int)(bby & 0xffffu); s.maxY = (int)(bby >> 16);
        s.is_line = (__float_as_uint(r3.w) & SWR_FLAG_LINE) != 0u;
    }
    return s;
}

// Triangles spanning more than 8 tiles are spread over the wave, every lane a different tile (every lane must call).
// Those of <= 64 tiles (one round of lanes) go four at a time: all four atomics are issued before any returned base is
// used, so FILL pays one atomic round trip per four triangles instead of one each; larger ones take the plain loop.
template <bool FILL>
__device__ __forceinline__ void bin_big(const BinArgs& a, const SlotData& sd, uint32_t slot, bool big) {
    const int lane = threadIdx.x & 63;
    const int nt = sd.nx * sd.ny;
    unsigned long long med = __ballot(big && nt <= 64);
    while (med) {
        bool want[4];
        uint32_t tile[4], sslot[4], base[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            want[j] = false; tile[j] = 0u; sslot[j] = 0u; base[j] = 0u;
            if (med) {                                             // wave-uniform
                const int src = __ffsll((long long)med) - 1;
                med &= med - 1;
                const int s_tminx = __shfl(sd.tminx, src), s_tminy = __shfl(sd.tminy, src);
                const int s_nx = __shfl(sd.nx, src), s_nt = __shfl(nt, src);
                sslot[j] = (uint32_t)__shfl((int)slot, src);
                float bsx[3], bsy[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { bsx[k] = __shfl(sd.sx[k], src); bsy[k] = __shfl(sd.sy[k], src); }
                const int bminX = __shfl(sd.minX, src), bmaxX = __shfl(sd.maxX, src), bminY = __shfl(sd.minY, src), bmaxY = __shfl(sd.maxY, src);
                const bool b_line = __shfl((int)sd.is_line, src) != 0;
                if (lane < s_nt) {
                    const int ty = s_tminy + lane / s_nx, tx = s_tminx + lane % s_nx;
                    want[j] = band_local_row(a.band, ty) >= 0 && pair_may_cover(bsx, bsy, bminX, bmaxX, bminY, bmaxY, tx, ty, a.width, a.height, b_line);
                    tile[j] = (uint32_t)(band_local_row(a.band, ty) * a.tiles_x + tx);
                }
                if (want[j]) {
                    if (FILL) base[j] = atomicAdd(&a.tile_count[tile[j]], 1u);
                    else atomicAdd(&a.tile_count[tile[j]], 1u);
                }
            }
        }
        if (FILL) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (want[j]) {
                    const uint32_t at = a.tile_start[tile[j]] + base[j];
                    if (at < a.list_capacity) a.tile_list[at] = sslot[j];
                    else bin_overflow(a);
                }
            }
        }
    }
    unsigned long long m = __ballot(big && nt > 64);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int s_tminx = __shfl(sd.tminx, src), s_tminy = __shfl(sd.tminy, src);
        const int s_nx = __shfl(sd.nx, src), s_nt = __shfl(nt, src);
        const uint32_t s_slot = (uint32_t)__shfl((int)slot, src);
        float bsx[3], bsy[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { bsx[k] = __shfl(sd.sx[k], src); bsy[k] = __shfl(sd.sy[k], src); }
        const int bminX = __shfl(sd.minX, src), bmaxX = __shfl(sd.maxX, src), bminY = __shfl(sd.minY, src), bmaxY = __shfl(sd.maxY, src);
        const bool b_line = __shfl((int)sd.is_line, src) != 0;
        for (int i0 = 0; i0 < s_nt; i0 += 64) {
            const int i = i0 + lane;
            bool want = i < s_nt;
            int ty = 0, tx = 0;
            if (want) {
                ty = s_tminy + i / s_nx; tx = s_tminx + i % s_nx;
                want = band_local_row(a.band, ty) >= 0 && pair_may_cover(bsx, bsy, bminX, bmaxX, bminY, bmaxY, tx, ty, a.width, a.height, b_line);
            }
            if (want) {
                const uint32_t tile = (uint32_t)(band_local_row(a.band, ty) * a.tiles_x + tx);
                if (FILL) {
                    const uint32_t at = a.tile_start[tile] + atomicAdd(&a.tile_count[tile], 1u);
                    if (at < a.list_capacity) a.tile_list[at] = s_slot;
                    else bin_overflow(a);
                } else {
                    atomicAdd(&a.tile_count[tile], 1u);
                }
            }
        }
    }
}

// Thread per submitted triangle, walking its `spt` slots (the odd fan slots are almost always empty).
// Triangles of <= 8 tiles (nearly all): the block's (tile, slot) pairs are first combined in an LDS hash table
// keyed by tile -- neighbouring triangles share tiles, so 256 triangles touch few distinct ones -- and only one
// global atomic per distinct tile leaves the block; FILL gets each pair's rank from the LDS add and the tile's base
// from that one global atomic.  The order inside a tile's list is irrelevant here (k_sort_tiles restores it).
#define SWR_BIN_TABLE_LOG2 8
#define SWR_BIN_TABLE (1 << SWR_BIN_TABLE_LOG2)     // a block rarely touches more than a few hundred distinct tiles; probing is bounded and
                                                  // a pair that finds no slot goes to the global counter directly
// one slot per thread of a 256-thread block (tb = its tile bbox word, SWR_TB_INVALID: nothing); every thread of the block must call
template <bool FILL>
__device__ __forceinline__ void bin_block_slots(const BinArgs& a, uint32_t slot, unsigned long long tb, uint32_t want_in, uint32_t* s_key, uint32_t* s_val) {
    const SlotData sd = slot_load(a, slot, tb, !FILL);
    const int nt = sd.nx * sd.ny;
    const bool big = nt > 8;
    const int nt_small = big ? 0 : nt;
    if (!__syncthreads_or(nt != 0)) return;                    // block-uniform (also orders the table's reuse)
    for (int e = threadIdx.x; e < SWR_BIN_TABLE; e += 256) { s_key[e] = 0u; s_val[e] = 0u; }
    __syncthreads();
    // phase 1: every wanted (tile, slot) pair into the table; packed[i] = entry | rank << 12 | wanted << 31
    uint32_t packed[8];
    int wtx = sd.tminx, wty = sd.tminy;                         // row-major walk of the tile bbox without integer division
    uint32_t wmask = (FILL && nt_small) ? want_in : 0u;            // FILL: COUNT's decisions (a.want[slot], fetched with tb)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        packed[i] = 0u;
        if (i < nt_small) {
            bool want;
            if (FILL) want = ((wmask >> i) & 1u) != 0u;
            else {
                want = band_local_row(a.band, wty) >= 0 && pair_may_cover(sd.sx, sd.sy, sd.minX, sd.maxX, sd.minY, sd.maxY, wtx, wty, a.width, a.height, sd.is_line);
                wmask |= want ? (1u << i) : 0u;
            }
            if (want) {
                const uint32_t tile = (uint32_t)(band_local_row(a.band, wty) * a.tiles_x + wtx);
                uint32_t e = (tile * 0x9E3779B1u) >> (32 - SWR_BIN_TABLE_LOG2);
                bool placed = false;
                for (int probe = 0; probe < 16; ++probe) {
                    const uint32_t prev = atomicCAS(&s_key[e], 0u, tile + 1u);
                    if (prev == 0u || prev == tile + 1u) { placed = true; break; }
                    e = (e + 1u) & (SWR_BIN_TABLE - 1u);
                }
                if (placed) {
                    const uint32_t rank = atomicAdd(&s_val[e], 1u);       // < 2048
                    packed[i] = e | (rank << 12) | 0x80000000u;
                } else if (FILL) {                                      // crowded table (many distinct tiles): go direct
                    const uint32_t at = a.tile_start[tile] + atomicAdd(&a.tile_count[tile], 1u);
                    if (at < a.list_capacity) a.tile_list[at] = slot;
                    else bin_overflow(a);
                } else {
                    atomicAdd(&a.tile_count[tile], 1u);
                }
            }
            ++wtx;
            if (wtx >= sd.tminx + sd.nx) { wtx = sd.tminx; ++wty; }
        }
    }
    if (!FILL && nt_small) a.want[slot] = (uint8_t)wmask;
    __syncthreads();
    // phase 2: one global atomic per distinct tile of the block
    for (int e = threadIdx.x; e < SWR_BIN_TABLE; e += 256) {
        const uint32_t key = s_key[e];
        if (key) {
            const uint32_t tile = key - 1u, n = s_val[e];
            if (FILL) s_val[e] = a.tile_start[tile] + atomicAdd(&a.tile_count[tile], n);
            else atomicAdd(&a.tile_count[tile], n);
        }
    }
    if (FILL) {
        __syncthreads();
        // phase 3: list position = the tile's base for this block + the pair's rank in the block
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (packed[i] & 0x80000000u) {
                const uint32_t at = s_val[packed[i] & 0xfffu] + ((packed[i] >> 12) & 0x7ffffu);
                if (at < a.list_capacity) a.tile_list[at] = slot;
                else bin_overflow(a);
            }
        }
    }
    bin_big<FILL>(a, sd, slot, big);
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_bin(BinArgs a) {
    SWR_FRONT_ENTER();
    __shared__ uint32_t s_key[SWR_BIN_TABLE];      // tile + 1; 0 = empty
    __shared__ uint32_t s_val[SWR_BIN_TABLE];      // pairs of this block in the tile; FILL: then their first list position
    if (FILL && batch_poisoned(a.ctrl, a.seq)) return;
    if (FILL && blockIdx.x >= a.bin_blocks) {                   // the grid's last blocks: the raster kernel's tile order
        static_assert(SWR_BIN_TABLE >= SWR_ORDER_BUCKETS, "tile_place_block borrows the hash table's LDS");
        tile_place_block(a, blockIdx.x - a.bin_blocks, s_key, s_val);
        return;
    }
    const uint32_t wave_id = (blockIdx.x * 256u + threadIdx.x) >> 6, lane_id = threadIdx.x & 63u;
    const bool has_tri = lane_id < a.tpw;
    const uint32_t first = a.slot_lo + (wave_id * a.tpw + lane_id) * a.spt;
    // the tile bbox word of slot si + 1 is on its way while slot si is binned (a block's life is a chain of memory round trips and
    // barriers: this takes one round trip out of every slot but the first)
    auto in_round = [&](uint32_t si) { return has_tri && si < a.spt && first + si < a.slot_hi; };
    auto tb_of = [&](uint32_t si) { return in_round(si) ? a.slot_tb[first + si] : SWR_TB_INVALID; };
    auto want_of = [&](uint32_t si) { return (FILL && in_round(si)) ? (uint32_t)a.want[first + si] : 0u; };   // (a byte of a slot COUNT never wrote is never used)
    unsigned long long tb = tb_of(0);
    uint32_t wm = want_of(0);
    for (uint32_t si = 0; si < a.spt; ++si) {
        const unsigned long long tb_next = tb_of(si + 1);
        const uint32_t wm_next = want_of(si + 1);
        bin_block_slots<FILL>(a, first + si, tb, wm, s_key, s_val);
        tb = tb_next; wm = wm_next;
    }
}

// Exclusive scan of the per-tile counts in two launches (no inter-block waiting, no dispatch-order assumption; a single launch that
// hands out tickets and exchanges the chunk sums through memory was built and measured: 0.0669 against 0.0667 ms for the bin stage --
// the exchange's dependent round trips cost what the second launch costs -- and it spins, so it was not kept):
//   k_scan_sums : block b sums its SWR_SCAN_BLOCK counts -> sums[b]
//   k_scan_apply: block b adds sums[0..b) (<= 1024 values at 8192^2, 4096 at the 2^20-tile limit) to a local scan of its counts
// Blocks of 256 threads (round 3: 1024): a 16-wave workgroup never fits beside a raster kernel (swr_device.h, SWR_FRONT_MAX_LDS).
#define SWR_SCAN_BLOCK 256
__device__ __forceinline__ unsigned long long block_sum(unsigned long long v, unsigned long long* s_part) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        v += ((unsigned long long)(unsigned)__shfl_xor((int)hi, off) << 32) | (unsigned)__shfl_xor((int)lo, off);
    }
    if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long t = 0;
    for (int w = 0; w < SWR_SCAN_BLOCK / 64; ++w) t += s_part[w];
    return t;
}

__global__ __launch_bounds__(SWR_SCAN_BLOCK) void k_scan_sums(const uint32_t* __restrict__ count, uint32_t n,
                                                    unsigned long long* __restrict__ sums) {
    SWR_FRONT_ENTER();
    __shared__ unsigned long long s_part[SWR_SCAN_BLOCK / 64];
    const uint32_t i = blockIdx.x * (uint32_t)SWR_SCAN_BLOCK + threadIdx.x;
    const unsigned long long t = block_sum(i < n ? count[i] : 0u, s_part);
    if (threadIdx.x == 0) sums[blockIdx.x] = t;
}

__global__ __launch_bounds__(SWR_SCAN_BLOCK) void k_scan_apply(uint32_t* __restrict__ count, uint32_t* __restrict__ start,
                                                     uint32_t n, const unsigned long long* __restrict__ sums,
                                                     unsigned long long* __restrict__ total_out,
                                                     unsigned long long capacity, uint32_t seq, Ctrl* __restrict__ ctrl,
                                                     Counters* __restrict__ counters, int poison_on_overflow,
                                                     const uint32_t* __restrict__ tile_work, uint32_t* __restrict__ order_hist,
                                                     uint8_t* __restrict__ tile_bucket) {
    SWR_FRONT_ENTER();
    __shared__ unsigned long long s_part[SWR_SCAN_BLOCK / 64];
    __shared__ unsigned long long s_wave[SWR_SCAN_BLOCK / 64];
    __shared__ uint32_t s_oh[SWR_ORDER_BUCKETS];
    static_assert(SWR_ORDER_BUCKETS <= SWR_SCAN_BLOCK, "one thread per bucket clears / flushes the block's histogram");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid < (uint32_t)SWR_ORDER_BUCKETS) s_oh[tid] = 0u;          // (block_sum below has the barrier)
    // offset of this block = sum of the earlier blocks' sums (gridDim.x <= 4096)
    unsigned long long mine = 0ull;
    for (uint32_t j = tid; j < blockIdx.x; j += (uint32_t)SWR_SCAN_BLOCK) mine += sums[j];
    const unsigned long long off = block_sum(mine, s_part);
    const uint32_t i = blockIdx.x * (uint32_t)SWR_SCAN_BLOCK + tid;
    const uint32_t c = i < n ? count[i] : 0u;
    // the tile's place in the raster kernel's dispatch order: bucket of its weight (see order_bucket), histogram of the buckets
    if (i < n) {
        const uint32_t ob = order_bucket(tile_weight(tile_work[i], c));
        tile_bucket[i] = (uint8_t)ob;
        atomicAdd(&s_oh[ob], 1u);
    }
    unsigned long long incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned lo = (unsigned)incl, hi = (unsigned)(incl >> 32);
        unsigned long long v = ((unsigned long long)(unsigned)__shfl_up((int)hi, o) << 32) | (unsigned)__shfl_up((int)lo, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63u) s_wave[wv] = incl;
    __syncthreads();
    if (tid < (uint32_t)SWR_ORDER_BUCKETS && s_oh[tid]) atomicAdd(&order_hist[tid], s_oh[tid]);
    unsigned long long wave_off = 0;
    for (uint32_t w = 0; w < wv; ++w) wave_off += s_wave[w];
    const unsigned long long excl = off + wave_off + incl - c;
    if (i < n) {
        start[i] = (uint32_t)min(excl, 0xffffffffull);
        count[i] = 0u;              // k_bin<FILL> uses the counts as cursors and rebuilds them: no separate clear
    }
    if (blockIdx.x == gridDim.x - 1 && tid == (uint32_t)SWR_SCAN_BLOCK - 1u) {
        const unsigned long long total = excl + c;
        *total_out = total;
        if (total > capacity) {                     // does not fit: poison this and every later batch
            if (!poison_on_overflow) return;
            atomicMin(&ctrl->first_bad, seq);
            atomicMax(&ctrl->need, total);
            __hip_atomic_store(&ctrl->poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ctrl->host_flag) __hip_atomic_store(ctrl->host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (poison_on_overflow && !batch_poisoned(ctrl, seq)) {
            atomicAdd(&counters->tile_pairs, total);        // MODE_SYNC rounds are counted by the host
        }
    }
}

// ---- per-tile ascending sort -------------------------------------------------------------
// Normalised bitonic network (first step of each merge is the "flip" i <-> block_end - i, the
// rest are half-cleaners): every comparator puts the minimum at the lower index, so padding
// the tail virtually with 0xffffffff needs no storage.
#define SWR_SORT_LDS 2048

__device__ __forceinline__ void cmpx_lds(uint32_t* s, uint32_t i, uint32_t p, uint32_t n) {
    if (p < n) {            // p > i always; p >= n means "padding" = max, nothing moves
        uint32_t a = s[i], b = s[p];
        if (a > b) { s[i] = b; s[p] = a; }
    }
}
__device__ __forceinline__ void cmpx_glb(uint32_t* g, uint32_t i, uint32_t p, uint32_t n) {
    if (p < n) {
        uint32_t a = g[i], b = g[p];
        if (a > b) { g[i] = b; g[p] = a; }
    }
}

// One wave sorts one tile at a time, SWR_SORT_TPW tiles one after the other, SWR_SORT_TPB waves per block (they never meet: no block
// barrier).  A tile's sort is ~850 cycles of one wave, and 65,536 waves that short are bound by the rate at which waves can be
// launched (about one per clock chip-wide: 3.5 waves resident per CU on average, 26 us); fewer, longer waves are not.
#define SWR_SORT_TPB 1                 // (round 3: 4 -- 32 KB of LDS per block; one wave and 8 KB fit beside the raster kernel, swr_device.h)
#define SWR_SORT_TPW 4
// what a one-wave workgroup's __syncthreads() amounts to: the wave's own LDS / global accesses complete in order
#define SWR_SORT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
// key_in: entry `lane` of the segment when n <= 64 (the kernel fetches the first 64 entries of all its tiles in one round trip)
__device__ __forceinline__ void sort_tile(uint32_t n, uint32_t start, uint32_t* __restrict__ tile_list, uint32_t tile,
                                          uint32_t* __restrict__ pair_tile, uint32_t* s_keys, uint32_t key_in) {
    if (n == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    // band-local tile index of every pair of this segment (k_cover reads it): contiguous, coalesced
    for (uint32_t i = lane; i < n; i += 64) pair_tile[start + i] = tile;
    if (n < 2) return;
    uint32_t* seg = tile_list + start;

    if (n <= 64) {
        // In registers.  Partners inside a row of 16 lanes are reached on the DPP path (no LDS round trip): quad_perm for
        // distances 1, 2 and the flip of 4, row_half_mirror / row_mirror for the flips of 8 / 16, row_ror for the xor
        // of 4 / 8; only the three steps that cross rows (flips of 32 and 64, xor 16) go through ds_bpermute.
        uint32_t key = lane < n ? key_in : 0xffffffffu;
#define SWR_CMPX(other_expr, lower_cond) { const uint32_t other = (uint32_t)(other_expr); \
                                            key = (lower_cond) ? min(key, other) : max(key, other); }
#define SWR_DPP(ctrl) __builtin_amdgcn_update_dpp(0, (int)key, ctrl, 0xf, 0xf, false)
#define SWR_XOR1  SWR_CMPX(SWR_DPP(0xB1), (lane & 1u) == 0u)                         /* quad_perm [1,0,3,2] */
#define SWR_XOR2  SWR_CMPX(SWR_DPP(0x4E), (lane & 2u) == 0u)                         /* quad_perm [2,3,0,1] */
#define SWR_XOR4  { const int up = SWR_DPP(0x124), dn = SWR_DPP(0x12C);              /* row_ror:4 = lane-4, row_ror:12 = lane+4 */ \
                    SWR_CMPX((lane & 4u) ? up : dn, (lane & 4u) == 0u) }
#define SWR_XOR8  SWR_CMPX(SWR_DPP(0x128), (lane & 8u) == 0u)                        /* row_ror:8 */
#define SWR_XOR16 SWR_CMPX(__shfl((int)key, (int)(lane ^ 16u)), (lane & 16u) == 0u)
        SWR_XOR1                                                                      // k = 2 (its flip is xor 1)
        SWR_CMPX(SWR_DPP(0x1B), (lane & 2u) == 0u) SWR_XOR1                           // k = 4: flip = quad_perm [3,2,1,0]
        SWR_CMPX(SWR_DPP(0x141), (lane & 4u) == 0u) SWR_XOR2 SWR_XOR1                 // k = 8: flip = row_half_mirror
        SWR_CMPX(SWR_DPP(0x140), (lane & 8u) == 0u) SWR_XOR4 SWR_XOR2 SWR_XOR1        // k = 16: flip = row_mirror
        SWR_CMPX(__shfl((int)key, (int)(lane ^ 31u)), (lane & 16u) == 0u) SWR_XOR8 SWR_XOR4 SWR_XOR2 SWR_XOR1              // k = 32
        SWR_CMPX(__shfl((int)key, (int)(lane ^ 63u)), (lane & 32u) == 0u) SWR_XOR16 SWR_XOR8 SWR_XOR4 SWR_XOR2 SWR_XOR1    // k = 64
#undef SWR_XOR16
#undef SWR_XOR8
#undef SWR_XOR4
#undef SWR_XOR2
#undef SWR_XOR1
#undef SWR_DPP
#undef SWR_CMPX
        if (lane < n) seg[lane] = key;
        return;
    }

    uint32_t m = 1;
    while (m < n) m <<= 1;
    const bool in_lds = n <= SWR_SORT_LDS;
    uint32_t* buf = in_lds ? s_keys : seg;
    if (in_lds) {
        for (uint32_t i = lane; i < n; i += 64) s_keys[i] = seg[i];
        SWR_SORT_WAVE_SYNC();
    }
    for (uint32_t k = 2; k <= m; k <<= 1) {
        // flip: within each block of k, i in the lower half pairs with block_base + k-1 - offset
        for (uint32_t t = lane; t < (m >> 1); t += 64) {
            uint32_t blk = t / (k >> 1), off = t % (k >> 1);
            uint32_t i = blk * k + off, p = blk * k + (k - 1 - off);
            if (i < n) { if (in_lds) cmpx_lds(buf, i, p, n); else cmpx_glb(buf, i, p, n); }
        }
        SWR_SORT_WAVE_SYNC();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (m >> 1); t += 64) {
                uint32_t i = 2 * j * (t / j) + (t % j), p = i + j;
                if (i < n) { if (in_lds) cmpx_lds(buf, i, p, n); else cmpx_glb(buf, i, p, n); }
            }
            SWR_SORT_WAVE_SYNC();
        }
    }
    if (in_lds) {
        for (uint32_t i = lane; i < n; i += 64) seg[i] = s_keys[i];
    }
}

__global__ __launch_bounds__(64 * SWR_SORT_TPB) void k_sort_tiles(const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ tile_count,
                                                   uint32_t* __restrict__ tile_list, uint32_t n_tiles,
                                                   uint32_t* __restrict__ pair_tile, const Ctrl* __restrict__ ctrl, uint32_t seq,
                                                   const uint4* __restrict__ tile_order /* heaviest first: the few tiles with hundreds of
                                                       pairs sort for 10+ us in one wave and must not start last */) {
    SWR_FRONT_ENTER();
    __shared__ uint32_t s_keys_all[SWR_SORT_TPB][SWR_SORT_LDS];
    static_assert(sizeof(s_keys_all) <= SWR_FRONT_MAX_LDS, "k_sort_tiles must fit beside the raster kernel (swr_device.h)");
    if (batch_poisoned(ctrl, seq)) return;
    static_assert(SWR_SORT_TPW <= 64, "one lane per tile of the wave fetches its count and start");
    const uint32_t first = (blockIdx.x * SWR_SORT_TPB + (threadIdx.x >> 6)) * SWR_SORT_TPW;
    const uint32_t lane = threadIdx.x & 63u;
    // counts and starts of the wave's tiles in one round trip (lane t: tile first + t)
    // wave w takes entries w, w + waves, w + 2 waves, ... of the order: its first tile is among the heaviest, its last among the lightest
    const uint32_t waves = gridDim.x * SWR_SORT_TPB, entry = (first / SWR_SORT_TPW) + lane * waves;
    const bool mine = lane < SWR_SORT_TPW && entry < n_tiles;
    const uint4 desc_l = mine ? tile_order[entry] : make_uint4(0u, 0u, 0u, 0u);      // {tile, start, pairs}: one load instead of three dependent ones
    const uint32_t tile_l = desc_l.x;
    const uint32_t cnt_l = desc_l.z, st_l = desc_l.y;
    uint32_t keys[SWR_SORT_TPW];           // entry `lane` of every tile's segment: all in flight before the first sort
#pragma unroll
    for (int t = 0; t < SWR_SORT_TPW; ++t) {
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, t), start = (uint32_t)__builtin_amdgcn_readlane((int)st_l, t);
        keys[t] = (lane < n && n <= 64u) ? tile_list[start + lane] : 0xffffffffu;
    }
#pragma unroll
    for (int t = 0; t < SWR_SORT_TPW; ++t) {
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, t), start = (uint32_t)__builtin_amdgcn_readlane((int)st_l, t);
        sort_tile(n, start, tile_list, (uint32_t)__builtin_amdgcn_readlane((int)tile_l, t), pair_tile, s_keys_all[threadIdx.x >> 6], keys[t]);
        SWR_SORT_WAVE_SYNC();              // the next tile reuses the wave's LDS slice
    }
}

}  // namespace swr
