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
//   is already claimed in the chunk (LDS bitmap election) or whose draw differs, so inside a chunk every
//   pixel is touched once and state is uniform: depth test, Interpolate, fragment program, blend and the
//   colour/Z update (tile-resident in LDS) are order-independent inside a chunk, and chunks run in stream order
//   = the serial schedule of the reference (>= ties, blending, alpha-gated Z writes all exact).  Each fragment
//   lane recomputes its own edge values by replaying its (y-startY)+(x-startX) add chain at full lane utilisation.
//   Pairs are taken 16 at a time: one lane per pair fetches the triangle record and the three outputs ONCE and
//   stages what fragments need in LDS (per-fragment gathers through the vector L1 were the first bound), pairs that
//   provably fail the depth test everywhere are dropped against the tile's minimum stored depth (hi-Z), and tiles
//   are dispatched heaviest first (tile_place_block in swr_binning.hip.h).
//   A wave's time per chunk is a chain of dependent LDS / memory round trips that four waves per SIMD (LDS: 10 KB per wave) do
//   not cover, so the links that can run early do: the next window's head words are read at the cut, each lane's next pair and
//   that pair's prefix counts between replay and shading, the row-start table and stored depth ahead of the election's atomic.
#pragma once
#include <type_traits>
#include "swr_device.h"
#include "swr_raster.hip.h"

namespace swr {

// Wave-synchronous LDS hand-off point: lanes write LDS above it, OTHER lanes of the same wave read below it.
//  * hardware: one wave's LDS instructions execute in issue order, so a later ds_read sees an earlier ds_write of any lane;
//  * compiler: program order between the write and the read is kept because the two accesses MAY alias (same LDS array,
//    unrelated run-time indices) -- no pass may swap a store with a later load it cannot prove disjoint.
// So the marker expands to nothing, deliberately.  Every real barrier was measured and rejected: a wavefront-scope
// release/acquire fence (also the LDS-only form), __builtin_amdgcn_wave_barrier() and __builtin_amdgcn_sched_barrier(0)
// all make LLVM treat the per-draw constants as clobberable, which turns their scalar loads (s_load through the constant
// cache, 28 in the DUST2 kernel) into vector loads (16 left) with a vmcnt(0) wait in every chunk: k_raster_c 0.46 -> 0.58 ms
// on cfg3 (gpurun_out/ab_r02j.txt).  An `asm volatile` block in the batch loop has the same kind of cost (it drains the
// window prefetch at the loop top): wave_min's DPP ladder is a plain `asm` for that reason.
// What makes this safe to ship is verification, not the memory model: (1) libswr_hip_test.so is built with SWR_WAVE_LDS_FENCE --
// a real wavefront-scope release / wave_barrier / acquire at every hand-off -- and tests/test_gpu_testlib.py requires its frames to
// equal the unfenced product's bit for bit; (2) tests/test_lds_handoff_asm.py disassembles the product build and checks that at
// each hand-off the ds_write / returning atomic precedes the dependent ds_read in the emitted code; (3) swr_build_info() names the
// compiler and source hash, and tests/test_gpu_api.py compares them with the pair the sweeps ran on (profiles/verified_build.json).
// The one COLD hand-off (k_cover's widest-box exchange, once per 256 pairs) carries the real fence in every build.
#define SWR_WAVE_LDS_FENCE_REAL() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                                       __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#ifndef SWR_WAVE_LDS_SYNC
#ifdef SWR_WAVE_LDS_FENCE
#define SWR_WAVE_LDS_SYNC() SWR_WAVE_LDS_FENCE_REAL()
#else
#define SWR_WAVE_LDS_SYNC() ((void)0)
#endif
#endif

// wave64 ballot straight from a bool: HIP's __ballot(int) first materialises the predicate as 0 / 1 in a VGPR and compares it
// with zero again (two VALU instructions per use); the builtin takes the condition mask as it is
#define SWR_BALLOT(cond) __builtin_amdgcn_ballot_w64((bool)(cond))

// Inclusive prefix sum over the wave, on the DPP path (no LDS round trips): Hillis-Steele inside each row of 16 lanes
// with row_shr, then the row totals with row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3).  All 64 lanes must be active.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31
    return v;
}
// minimum over the wave (same DPP ladder; the last lane ends up with the result).  v_min_f32 with a DPP source operand:
// a lane whose source lane does not exist is not written (no bound_ctrl), i.e. keeps its value; s_nop 1 = the two wait
// states a DPP read needs after a VALU write of the same register.  NaNs are dropped like fminf does.
__device__ __forceinline__ float wave_min(float x) {
    asm("s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1"
                 : "+v"(x));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

struct CoverArgs {
    const TriRec* __restrict__ recs;
    const uint32_t* __restrict__ tile_list;   // sorted: slot id per pair
    const uint32_t* __restrict__ pair_tile;   // band-local tile index per pair
    uint4* __restrict__ masks;                // 2 x uint4 per pair: row r -> bits (r & 1) * 16 .. of word r >> 1
    uint2* __restrict__ info;                 // {popcount of the mask, hi-Z bound: an upper bound of the pair's fragment depths
                                              //  over bbox /\ tile as float bits (+inf = no bound)}: one load in the raster kernel's window
    uint4* __restrict__ refs;                 // {slot, vertex references of outputs[0..2]}: the raster kernel's batch set-up
                                              // then needs no load that depends on another load
    const unsigned long long* __restrict__ n_pairs;   // device-resident pair total of this batch
    const Ctrl* __restrict__ ctrl;
    uint32_t seq;                             // sequence number of the batch (batch_poisoned)
    FrameParams fp;
    unsigned long long* dbg;                  // SWR_DEBUG_COVER builds only (tools/debug_counters.py)
};

// pairs (= threads) per k_cover block, and the quantisation of the shape sort's key (rows, columns)
#define SWR_COVER_BLOCK 256
#define SWR_COVER_HQ 1                  // rows exact, columns in fours: 92 instead of 101 executed pixel steps per lane (49 useful), cover -2 %
#define SWR_COVER_WQ 4
// LINES: the batch is DebugMode.Wireframe (DrawLine records); compiled out of the filled-triangle instantiation
template <bool LINES>
__global__ __launch_bounds__(SWR_COVER_BLOCK) void k_cover(CoverArgs a) {
    SWR_FRONT_ENTER();
    constexpr int CB = SWR_COVER_BLOCK;
    constexpr int NB = 1 + (16 / SWR_COVER_HQ) * (16 / SWR_COVER_WQ);
    // One LDS record of 18 halfwords per pair (9 dwords per lane: conflict-free), used three ways in turn by exactly two lanes -- the
    // pair's own thread and the lane that walks it -- so nothing else is needed (round 3 kept the list entries and the bound in arrays
    // of their own: 13,088 B; this is 10,004 B, and a block that fits in 10,240 B runs BESIDE the previous flush's raster kernel,
    // swr_device.h):  [0..3] the pair's list entries {slot, tile} as the own thread read them, in pair order (handed to the walker
    // before the sort's barriers);  then [0..15] the 16 row masks the walker produces;  [16..17] its hi-Z bound (float bits).
    // Results leave in PAIR order, not in the walk's sorted order: after a barrier thread t packs and stores pair t -- consecutive
    // lanes, consecutive addresses (in sorted order every lane's 16-byte stores went to a different line: k_cover is memory bound,
    // and scattered partial-line writes are what cost k_setup 16 of its 37 us).  All accesses are halfword accesses on purpose: one
    // type, so the compiler keeps the walker's reads of [0..3] ahead of its own clearing stores.
    __shared__ uint16_t s_rows[CB][18];
    __shared__ uint32_t s_hist[NB];           // work sort: pairs per bucket, then bucket bases
    __shared__ uint16_t s_perm[CB];           // sorted position -> thread whose pair it is
    __shared__ uint32_t s_wmax[CB / 64];      // per wave: widest bbox /\ tile among its lanes on the fast path
    static_assert(sizeof(s_rows) + sizeof(s_hist) + sizeof(s_perm) + sizeof(s_wmax) + 64 <= SWR_FRONT_MAX_LDS, "k_cover must fit beside the raster kernel");
    if (batch_poisoned(a.ctrl, a.seq)) return;
    const uint32_t n_pairs = (uint32_t)*a.n_pairs;
    // Workgroups go round-robin to the 8 XCDs, each with an L2 of its own, and a triangle's pairs sit in neighbouring tiles: the one to
    // the right a few dozen pairs away, the one below a whole tile row (thousands of pairs) away.  XCD x therefore takes the x-th
    // CONTIGUOUS eighth of the pair array (of the pairs that exist, not of the grid, which covers the list capacity): the block that
    // meets a TriRec again a tile row later runs on the same XCD while the line is still in its L2.  SWR_COVER_LINEAR_BLOCKS: A/B.
    uint32_t block;
    {
        const uint32_t nb = (n_pairs + (uint32_t)CB - 1u) / (uint32_t)CB, per_xcd = (nb + 7u) >> 3;
        const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
        if (k >= per_xcd) return;                                   // (block-uniform)
        block = xcd * per_xcd + k;
    }
    const uint32_t p_own = block * (uint32_t)CB + threadIdx.x;
    // A lane's loop length is the area of bbox /\ tile (1..256 pixels) and a wave runs as long as its longest lane, so
    // the block first sorts its 256 pairs by that area (counting sort in LDS): each wave then holds pairs of similar
    // length.  Results are written at the pair's own index, so nothing downstream sees the permutation.
    for (int i = threadIdx.x; i < NB; i += CB) s_hist[i] = 0u;
    if (threadIdx.x < CB / 64) s_wmax[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t bucket = 0, rank = 0;
    {
        int area = 0;
        if (p_own < n_pairs) {
            const uint32_t slot = a.tile_list[p_own], tile = a.pair_tile[p_own];
            s_rows[threadIdx.x][0] = (uint16_t)slot; s_rows[threadIdx.x][1] = (uint16_t)(slot >> 16);
            s_rows[threadIdx.x][2] = (uint16_t)tile; s_rows[threadIdx.x][3] = (uint16_t)(tile >> 16);
            const int tx = (int)(tile % (uint32_t)a.fp.tiles_x), ty = band_global_row(band_map(a.fp), (int)(tile / (uint32_t)a.fp.tiles_x));
            const int x0 = tx * SWR_TILE, y0 = ty * SWR_TILE;
            const float4 r3 = reinterpret_cast<const float4*>(a.recs + slot)[3];
            {   // {slot, vertex references of outputs[0..2]}: words 10, 11, 12 of the TriRec
                const float2 r2zw = reinterpret_cast<const float2*>(a.recs + slot)[5];
                a.refs[p_own] = make_uint4(slot, __float_as_uint(r2zw.x), __float_as_uint(r2zw.y), __float_as_uint(r3.x));
            }
            const uint32_t bbx = __float_as_uint(r3.y), bby = __float_as_uint(r3.z);
            const int w = min((int)(bbx >> 16), min(x0 + SWR_TILE - 1, a.fp.width - 1)) - max((int)(bbx & 0xffffu), x0) + 1;
            const int h = min((int)(bby >> 16), min(y0 + SWR_TILE - 1, a.fp.height - 1)) - max((int)(bby & 0xffffu), y0) + 1;
            area = (w > 0 && h > 0) ? w * h : 0;
            // a wave walks max-height rows x max-width columns of its lanes, so group by SHAPE, not just area: bucket = (rows, columns)
            // quantised to SWR_COVER_HQ / SWR_COVER_WQ -- 1 + 16 * 4 buckets, tallest / widest first (0 = nothing to walk)
            if (area > 0) area = 1 + ((h + SWR_COVER_HQ - 1) / SWR_COVER_HQ - 1) * (16 / SWR_COVER_WQ) + ((w + SWR_COVER_WQ - 1) / SWR_COVER_WQ - 1);
        }
        bucket = (uint32_t)area;                                 // 0..NB-1
        rank = atomicAdd(&s_hist[bucket], 1u);
    }
    __syncthreads();
    // bucket bases, longest first: base[b] = pairs in the buckets above b.  One wave, one DPP scan over the 64 non-empty-walk buckets
    // taken in descending order (lane l holds bucket NB - 1 - l), bucket 0 (nothing to walk) behind them all.  (Round 3: thread 0
    // walked the 65 buckets one LDS round trip after the other while 255 threads waited at the barrier.)
    static_assert(NB == 65, "one lane per bucket 1..64");
    if (threadIdx.x < 64) {
        const uint32_t b = (uint32_t)(NB - 1) - threadIdx.x;                  // 64 .. 1
        const uint32_t cnt_b = s_hist[b];
        const uint32_t incl = (uint32_t)wave_incl_scan((int)cnt_b, (int)threadIdx.x);
        s_hist[b] = incl - cnt_b;
        if (threadIdx.x == 63) s_hist[0] = incl;                             // every walking pair precedes the empty ones
    }
    __syncthreads();
    s_perm[s_hist[bucket] + rank] = (uint16_t)threadIdx.x;
    __syncthreads();
    const uint32_t owner = s_perm[threadIdx.x];
    const uint32_t p = block * (uint32_t)CB + owner;
    uint16_t* mrow16 = s_rows[owner];
    // (the pair's list entries, left there by its own thread before the sort's barriers; read before the rows are cleared)
    const uint32_t slot_w = (uint32_t)mrow16[0] | ((uint32_t)mrow16[1] << 16), tile_w = (uint32_t)mrow16[2] | ((uint32_t)mrow16[3] << 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) mrow16[i] = 0;
    // pack 16 row masks into the pair's 256-bit mask, count them, flag rows that are not one run; store mask and info of pair `pi`
    auto emit = [&a](const uint16_t* rows, uint32_t pi, float zb) {
        uint32_t mw[8];
        int cnt = 0;
#ifdef SWR_NO_SIMPLE_SELECT                    // test builds: every pair takes the raster kernel's general k-th-set-bit search
        uint32_t not_run = 1u;
#else
        uint32_t not_run = 0u;                 // a row whose covered pixels are not ONE run leaves a bit here
#endif
#pragma unroll
        for (int i = 0; i < 8; ++i) {          // word i = rows 2i (low half) and 2i+1 (high half)
            const uint32_t lo = rows[2 * i], hi = rows[2 * i + 1];
            not_run |= (lo & (lo + (lo & (0u - lo)))) | (hi & (hi + (hi & (0u - hi))));
            mw[i] = lo | (hi << 16);
            cnt += __popc(mw[i]);
        }
        if (cnt) {                             // (the raster kernel drops an empty pair before it looks at its mask)
            a.masks[2 * (size_t)pi] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
            a.masks[2 * (size_t)pi + 1] = make_uint4(mw[4], mw[5], mw[6], mw[7]);
        }
        // (along a row every edge value is monotone -- it is stepped by a constant -- so "all >= 0" and "all <= 0" are
        // intervals and a row is one run unless both are non-empty and apart: sliver triangles only.  The raster kernel
        // selects the k-th pixel of a run arithmetically and searches bit by bit only in pairs without this flag.)
        a.info[pi] = make_uint2((uint32_t)cnt | (not_run == 0u ? SWR_INFO_SIMPLE : 0u), __float_as_uint(zb));
    };
    if (p < n_pairs) {
        const uint32_t slot = slot_w, tile = tile_w;
        const int tx = (int)(tile % (uint32_t)a.fp.tiles_x), ty = band_global_row(band_map(a.fp), (int)(tile / (uint32_t)a.fp.tiles_x));
        const int x0 = tx * SWR_TILE, y0 = ty * SWR_TILE;
        const int tile_end_x = min(x0 + SWR_TILE - 1, a.fp.width - 1), tile_end_y = min(y0 + SWR_TILE - 1, a.fp.height - 1);
        const float4* __restrict__ rq = reinterpret_cast<const float4*>(a.recs + slot);
        const float4 r0 = rq[0], r1 = rq[1], r2 = rq[2], r3 = rq[3];      // (r3 through LDS as well: no change, measured)
        const float s0x = r0.x, s1x = r0.y, s2x = r0.z, s0y = r0.w, s1y = r1.x, s2y = r1.y;
        const uint32_t bbx = __float_as_uint(r3.y), bby = __float_as_uint(r3.z);
        const int startX = max((int)(bbx & 0xffffu), x0), endX = min((int)(bbx >> 16), tile_end_x);     // Rasterizer.cs:471-474
        const int startY = max((int)(bby & 0xffffu), y0), endY = min((int)(bby >> 16), tile_end_y);
        const bool is_line = LINES && (__float_as_uint(r3.w) & SWR_FLAG_LINE) != 0u;
        // Hierarchical-Z bound for the raster kernel (used there only under Less / LessEqual, where stored depth only
        // grows): U >= every fragment depth of this pair.  Proof (u = 2^-24, R = bbox /\ tile, M_i as in
        // pair_may_cover, S = sum |d_i invArea| M_i): every edge value of the reference's chain is within 35uM_i of the
        // exact edge function (swr_binning.hip.h), so the fragment's float depth (two products, two sums) is within
        // 39.3uS of the exact affine depth; that is maximal at a corner of R; the corners evaluated in float below are
        // within 7.3uS; margin used: 64uS.  Lines and non-finite cases get +inf (never hidden).
        float zbound = __uint_as_float(0x7f800000u);
        if (!is_line && startX <= endX && startY <= endY) {
            const float fxs = (float)startX, fxe = (float)endX, fys = (float)startY, fye = (float)endY;
            const float sx[3] = { r0.x, r0.y, r0.z }, sy[3] = { r0.w, r1.x, r1.y };
            const float dd[3] = { r1.z, r1.w, r2.x };
            const float inv_area = r2.y;
            // edge k (weight of depths[k]): a12,b12 about vertex 1; a20,b20 about vertex 2; a01,b01 about vertex 0
            const float ea[3] = { sy[1] - sy[2], sy[2] - sy[0], sy[0] - sy[1] };
            const float eb[3] = { sx[2] - sx[1], sx[0] - sx[2], sx[1] - sx[0] };
            const float rx[3] = { sx[1], sx[2], sx[0] }, ry[3] = { sy[1], sy[2], sy[0] };
            float c00 = 0.f, c10 = 0.f, c01 = 0.f, c11 = 0.f, S = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float kk = dd[k] * inv_area;
                const float dxs = fxs - rx[k], dxe = fxe - rx[k], dys = fys - ry[k], dye = fye - ry[k];
                const float xs = ea[k] * dxs, xe = ea[k] * dxe, ys = eb[k] * dys, ye = eb[k] * dye;
                c00 += kk * (xs + ys); c10 += kk * (xe + ys); c01 += kk * (xs + ye); c11 += kk * (xe + ye);
                S += fabsf(kk) * (fabsf(ea[k]) * fmaxf(fabsf(dxs), fabsf(dxe)) + fabsf(eb[k]) * fmaxf(fabsf(dys), fabsf(dye)));
            }
            const float U = fmaxf(fmaxf(c00, c10), fmaxf(c01, c11)) + S * (64.0f / 16777216.0f);
            const bool finite = S < 1.0e30f && c00 == c00 && c10 == c10 && c01 == c01 && c11 == c11;   // fmaxf drops NaNs
            if (finite) zbound = U;
        }
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
                // The column loop runs a WAVE-UNIFORM number of steps (the widest box among this wave's lanes: a wave
                // executes its longest lane anyway), so the per-pixel work is the three chain adds, the test and one
                // shift-or -- no per-lane loop bookkeeping.  A narrower lane steps past its endX; those values are
                // never looked at (colmask) and stay finite (<= 31 adds of values below 1e30).
                const int width = endX - startX + 1;
                atomicMax(&s_wmax[threadIdx.x >> 6], (uint32_t)width);
                SWR_WAVE_LDS_FENCE_REAL();        // cold (once per lane and 256-pair block): the real fence costs nothing measurable here
                const int wsteps = __builtin_amdgcn_readfirstlane((int)s_wmax[threadIdx.x >> 6]);
                const uint32_t colmask = (1u << width) - 1u;
                const int sh = startX - x0;
                for (int y = startY; y <= endY; ++y) {
                    float w0 = w0r, w1 = w1r, w2 = w2r;
                    uint32_t acc = 0;                      // pixel startX + i ends at bit wsteps - 1 - i
                    for (int i = 0; i < wsteps; ++i) {
                        // inside = all >= 0 || all <= 0 (:493-494) as lane masks (a ballot of ONE compare is the compare's own mask), then
                        // acc = 2 acc + inside as ONE instruction: an add with that mask as carry-in (the compiler's own form is
                        // v_cndmask 0/1 + v_lshl_or: two of the nine vector instructions of this loop)
                        {
                            const unsigned long long in_mask = SWR_BALLOT(fminf(fminf(w0, w1), w2) >= 0.0f) | SWR_BALLOT(fmaxf(fmaxf(w0, w1), w2) <= 0.0f);
                            unsigned long long carry_out;
                            asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(acc), "=s"(carry_out) : "v"(acc), "s"(in_mask));
                        }
                        w0 += a12; w1 += a20; w2 += a01;                                                  // :527-529
                    }
                    const uint32_t rowbits = ((__brev(acc) >> (32 - wsteps)) & colmask) << sh;
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
        mrow16[16] = (uint16_t)__float_as_uint(zbound); mrow16[17] = (uint16_t)(__float_as_uint(zbound) >> 16);
    }
    __syncthreads();
    if (p_own < n_pairs) {
        const uint16_t* own = s_rows[threadIdx.x];
        emit(own, p_own, __uint_as_float((uint32_t)own[16] | ((uint32_t)own[17] << 16)));
    }
    // (round 2 summed the counts per tile here -- the raster kernel's scheduling weight -- with a block barrier, a wave scan and an
    //  atomic per run of equal tiles; the weight now comes from the raster kernel's own count of the previous flush, swr_binning.hip.h)
}

// pairs per batch.  Everything a fragment needs from its triangle (the TriRec and the three outputs' varyings)
// is staged in LDS once per batch by one lane per pair: per-fragment gathers of that data (256 B per fragment
// through the vector L1, which is what bounded the kernel) become LDS broadcasts.
#define SWR_BATCH 16
#define SWR_BATCH_FRAGS 2048                   // fragments per batch (a pair covers <= 256 pixels, so at least 8 pairs always fit)
#define SWR_WINDOW 32                          // candidate pairs examined per batch (<= 64)
// (every size here was swept: 8 / 12 / 13 / 20 / 24 / 28-pair batches, windows of 24-64, 1536 fragments -- profiles/r02_final_ablations.txt,
//  profiles/r03_raster_experiments.md; the switches are gone, the numbers stay)
// Where the three outputs' varyings of a pair come from when a fragment is shaded (VG in the code):
//   staged in LDS (round 2's layout): nine rows per pair (twelve with the 4-light program's world position), 292 B per pair;
//   from the vertex-stage output in HBM: buffer loads with staged byte offsets -- the lanes of one pair read the same 16-byte rows,
//     a chunk touches the vertices of 3-4 pairs -- and only what the chain replay, the depth test and Interpolate's divisions need
//     stays staged: 176 B per pair, 8,336 B per wave with 16 pairs = 18 waves per CU instead of 16 (LDS is allocated in units of
//     1,280 B, tools/ubench/lds_occupancy.hip).
// Measured on cfg3 (profiles/r03_raster_experiments.md): the loads cost 7 % at equal occupancy and the two extra waves return 4-5 %,
// so the kernels WITHOUT the 4-light program keep the staged layout (VG = false); the kernels that carry it (PHONG = true: staged, they
// had to shrink their batch to 12 pairs to stay at 16 waves) take the loads (VG = true): 16 pairs per batch again and 18 waves, cfg4 -5 %.
// When the rows are requested (VG): the uv rows by the fragments that survive the chunk's cut, ahead of the chain replay (the texel
// address hangs on them), the rest by the fragments that passed the depth test (cfg3, all kernels VG: 0.4290 ms against 0.4333 for
// everything after the depth test, 0.4303 for everything ahead of the election, 0.4241 but spilling for everything after the cut).
// The DUST2 kernel shades with the straight-line speculate-then-verify shader (shade_dust2_fast, swr_raster.hip.h).
// staged float4 rows per pair (everything per-pair is computed once here instead of once per fragment):
//   0: edge values at the pair's first pixel (w0,w1,w2 of Rasterizer.cs:481-483), invArea   [lines: t0x,t1x,t0y,t1y]
//   1: depths[0..2], draw/flags word          2: column steps a12,a20,a01, first pixel (x,y inside the tile, 8 bits each)
//   3: row steps b12,b20,b01, stream position of the pair's first fragment (int bits)
//   VG:              4: byte offsets of outputs[0..2] in the VOut array, clip.w of outputs[0]
//                    5: refined reciprocals of the three clip.w (Interpolate's divisions, see div_core), clip.w of outputs[1]
//                    (clip.w of outputs[2] rides in the row-start entry)
//                    6: byte offset of the pair's TriRec (DEBUG_VARYINGS reads the three screen positions from it)
//   else 4-6 / 7-9 / 10-12: outputs[0] / [1] / [2] as {clip (x replaced by the output's wn.z, y by the refined reciprocal of clip.w),
//   color, uv + wn.xy};  PHONG adds 13-15 = {wn.z, wpos} of each
#define SWR_VROW(r, idx) L.stage[r][idx]
template <bool PHONG>
struct __attribute__((aligned(16))) WaveLdsC {
    static constexpr bool VG = PHONG;
    static constexpr int NQ = VG ? 7 : 13;
    static constexpr int BATCH = SWR_BATCH;        // pairs staged per batch
    static constexpr int WINDOW = SWR_WINDOW;      // candidate pairs examined per batch
    static constexpr int RT_ROWS = VG ? 8 : 4;     // a row-start entry every RT_ROWS rows of a pair
    static constexpr int RT_N = 16 / RT_ROWS - 1;  // entries per pair (rows RT_ROWS, 2 RT_ROWS, ...)
    float4 col[256];                 // pixel p = (y - y0) * 16 + (x - x0)
    float z[256];
    float4 stage[NQ][BATCH];         // batch (non-empty pairs only, compacted): per-pair fragment inputs
    uint32_t mask[BATCH][8];         // coverage masks
    uint32_t wpre[BATCH][4];         // exclusive prefix of the 8 word popcounts, 16-bit fields (word j -> field j)
    uint32_t head[SWR_BATCH_FRAGS / 32 + 4];   // bit (first - 1) set for every pair t >= 1 (first = its stream position): pair of fragment g = #bits below g
    uint32_t touched[32];            // chunk duplicate election: pixel p claimed <=> bit (p >> 5) of word (p & 31) -- neighbouring
                                     // pixels (the usual content of a chunk) fall into different words: no same-address atomics
    // row-start table: the pair's edge values at the start of its rows RT_ROWS (q + 1), q = 0 .. RT_N - 1, i.e. the reference's row chain
    // (Rasterizer.cs:532-534) run once per pair at staging: a fragment's row replay is then at most RT_ROWS - 1 add steps from the
    // nearest entry instead of up to 15.  Staged-varyings layout: three planes of floats (address = plane + 4 t: no multiply);
    // VG layout: one float4 per pair whose .w carries clip.w of outputs[2].
    float rowtab3[VG ? 1 : RT_N][3][VG ? 1 : BATCH];
    float4 rowtab4[VG ? RT_N : 1][VG ? BATCH : 1];
    __device__ __forceinline__ void rt_store(int q, int t, float x, float y, float z, float w) {
        if constexpr (VG) rowtab4[q][t] = make_float4(x, y, z, w);
        else { rowtab3[q][0][t] = x; rowtab3[q][1][t] = y; rowtab3[q][2][t] = z; }
    }
    __device__ __forceinline__ void rt_load(int q, int t, float& x, float& y, float& z) const {
        if constexpr (VG) { const float4 e = rowtab4[q][t]; x = e.x; y = e.y; z = e.z; }
        else { x = rowtab3[q][0][t]; y = rowtab3[q][1][t]; z = rowtab3[q][2][t]; }
    }
    __device__ __forceinline__ float rt_w(int t) const { return rowtab4[0][VG ? t : 0].w; }
};

// tools/phase_times.py (-DSWR_DEBUG_PHASES, never the product): shader-clock ticks a wave spends in each phase of k_raster_c, summed
// over every 64th wave (dispatch index: spread evenly over the heaviest-first order) into RasterArgs::dbg.  Every mark is an s_memtime + s_waitcnt lgkmcnt(0), i.e. it also waits for the wave's
// outstanding LDS operations: the marked kernel runs slower than the product and reads that wait into the phase that ends there.
//   0 tile init   1 batch: window, hi-Z, selection   2 batch: staging (loads, edge set-up, LDS writes)   3 chunk: lookup (-> pixel)
//   4 chunk: election + cut   5 chunk: chain replay + depth   6 chunk: shading, blend, next lookup   7 write-back + statistics
#ifdef SWR_DEBUG_PHASES
#define SWR_PHASE_DECL unsigned ph_t = (unsigned)__builtin_readcyclecounter(), ph_acc0 = 0u, ph_acc1 = 0u, ph_acc2 = 0u, ph_acc3 = 0u, \
                                ph_acc4 = 0u, ph_acc5 = 0u, ph_acc6 = 0u, ph_acc7 = 0u;
#define SWR_PHASE(i) do { const unsigned now_ = (unsigned)__builtin_readcyclecounter(); ph_acc##i += now_ - ph_t; ph_t = now_; } while (0)
#define SWR_PHASE_FLUSH(dbg) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63u) == 0u && (dbg)) {   /* every 64th wave: 65,536 waves on 8 addresses are a storm */ \
        atomicAdd(&(dbg)[0], (unsigned long long)ph_acc0); atomicAdd(&(dbg)[1], (unsigned long long)ph_acc1); \
        atomicAdd(&(dbg)[2], (unsigned long long)ph_acc2); atomicAdd(&(dbg)[3], (unsigned long long)ph_acc3); \
        atomicAdd(&(dbg)[4], (unsigned long long)ph_acc4); atomicAdd(&(dbg)[5], (unsigned long long)ph_acc5); \
        atomicAdd(&(dbg)[6], (unsigned long long)ph_acc6); atomicAdd(&(dbg)[7], (unsigned long long)ph_acc7); } } while (0)
#else
#define SWR_PHASE_DECL
#define SWR_PHASE(i) ((void)0)
#define SWR_PHASE_FLUSH(dbg) ((void)0)
#endif

// n steps of the reference's incremental edge chain (Rasterizer.cs:527-534) for this lane, 0 <= n <= MAXN: exactly n float additions per
// edge, in order.  History: as a per-lane loop (`for (i < n)`) this was three adds, a compare, two exec updates and a TAKEN branch per
// iteration; as predicated C++ (`if (i < n) w += s`, round 4) the compiler if-converts every step into three adds, a compare and THREE
// v_cndmask -- and a v_cndmask fed by vcc is the most expensive vector instruction this kernel issues (tools/ubench/valu_rate.hip: 23
// cycles per wave-instruction against 2.5 for the add; profiles/r04_valu_issue_model.md): ~33 of them per chunk.  Here the predicate
// lives where the hardware keeps predicates: each step narrows EXEC with one v_cmpx (the lanes with steps left only ever become fewer),
// the adds run under it, every fourth step leaves when no lane is left, and EXEC is restored at the end -- inside ONE asm statement, so
// the compiler never sees EXEC change.  Operand order of the adds as the compiler had it (step first).
#define SWR_RP_STEP(I) "v_cmpx_lt_i32_e32 vcc, " #I ", %[n]\n v_add_f32_e32 %[w0], %[s0], %[w0]\n v_add_f32_e32 %[w1], %[s1], %[w1]\n v_add_f32_e32 %[w2], %[s2], %[w2]\n"
#define SWR_RP_OUT "s_nop 4\n s_cbranch_execz .Lrp_end_%=\n"       /* (wait states between the VALU write of EXEC and the branch on EXECZ: insurance, 3 per column replay) */
template <int MAXN>
__device__ __forceinline__ void replay_chain(float& w0, float& w1, float& w2, int n, float s0, float s1, float s2) {
    static_assert(MAXN == 3 || MAXN == 7 || MAXN == 15, "row replay (RT_ROWS - 1) or column replay (15)");
    unsigned long long saved;
    if constexpr (MAXN == 3) {
        asm("s_mov_b64 %[sv], exec\n"
            SWR_RP_STEP(0) SWR_RP_STEP(1) SWR_RP_STEP(2)
            "s_mov_b64 exec, %[sv]"
            : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [sv] "=&s"(saved) : [n] "v"(n), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2) : "vcc");
    } else if constexpr (MAXN == 7) {
        asm("s_mov_b64 %[sv], exec\n"
            SWR_RP_STEP(0) SWR_RP_STEP(1) SWR_RP_STEP(2) SWR_RP_STEP(3) SWR_RP_OUT
            SWR_RP_STEP(4) SWR_RP_STEP(5) SWR_RP_STEP(6)
            ".Lrp_end_%=:\n s_mov_b64 exec, %[sv]"
            : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [sv] "=&s"(saved) : [n] "v"(n), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2) : "vcc");
    } else {
        asm("s_mov_b64 %[sv], exec\n"
            SWR_RP_STEP(0) SWR_RP_STEP(1) SWR_RP_STEP(2) SWR_RP_STEP(3) SWR_RP_OUT
            SWR_RP_STEP(4) SWR_RP_STEP(5) SWR_RP_STEP(6) SWR_RP_STEP(7) SWR_RP_OUT
            SWR_RP_STEP(8) SWR_RP_STEP(9) SWR_RP_STEP(10) SWR_RP_STEP(11) SWR_RP_OUT
            SWR_RP_STEP(12) SWR_RP_STEP(13) SWR_RP_STEP(14)
            ".Lrp_end_%=:\n s_mov_b64 exec, %[sv]"
            : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [sv] "=&s"(saved) : [n] "v"(n), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2) : "vcc");
    }
}

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

// PROG / BLEND / DT >= 0: every draw of the batch has that program / blend mode / depth test (compile-time state:
// the switches fold away); -1 = read them from the draw at run time.
// EARLYOUT: some draw of the batch uses BlendMode.None, whose row early-out (Rasterizer.cs:520-523) is applied per chunk.
// LDS is allocated in units of 1,280 B (tools/ubench/lds_occupancy.hip: 16 one-wave workgroups per CU up to 10,240 B, 18 up to 8,960 B,
// 21 up to 7,680 B, 25 up to 6,400 B, 32 up to 5,120 B)
static_assert(sizeof(WaveLdsC<false>) <= 10240 && sizeof(WaveLdsC<true>) <= 8960, "16 (18) waves per CU need <= 10,240 (8,960) B of LDS per wave");
template <bool LINES, bool PHONG, int PROG = -1, int BLEND = -1, int DT = -1, bool EARLYOUT = false>
__global__ __launch_bounds__(64, (PHONG && PROG != SWR_PROG_PHONG_4POINT) ? 5 : 4) void k_raster_c(RasterArgs a, const uint4* __restrict__ masks,
                                                                  const uint2* __restrict__ info) {
    __shared__ WaveLdsC<PHONG> s_w;
    if (batch_poisoned(a.ctrl, a.seq)) return;
    SWR_PHASE_DECL

    const int lane = threadIdx.x & 63;
    // hi-Z needs every draw of the batch to use Less / LessEqual (stored depth only grows): compile-time state, or the host's word
    const bool HIZ = !LINES && (DT == SWR_DEPTH_LESS || DT == SWR_DEPTH_LESSEQUAL || (DT < 0 && a.depth_only_grows != 0));
    // Workgroups go round-robin to the 8 XCDs, in index order as slots free up.  The order array is cut in segments of
    // 64 entries (neighbouring tiles of similar weight, see order_tile_of_thread); XCD x works through segments x, x+8, ...:
    // still heaviest-first chip-wide (granularity 512 tiles), and a segment's shared triangle data stays in one L2.
    uint4 desc;                            // {tile, first list entry, pairs}: ONE scalar load, then the window fetch (round 3: three dependent loads)
    {
        const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
        const uint32_t j = (((k >> 6) << 3) + xcd) * 64u + (k & 63u);
        if (j >= a.n_tiles) return;
        desc = a.tile_order[j];
    }
    const uint32_t tile_o = desc.x;
    const int tx = (int)(tile_o % (uint32_t)a.fp.tiles_x), ty_local = (int)(tile_o / (uint32_t)a.fp.tiles_x);
    if (tx >= a.fp.tiles_x || ty_local >= a.fp.band_tile_rows) return;
    const int ty = band_global_row(band_map(a.fp), ty_local);
    const uint32_t tile = (uint32_t)(ty_local * a.fp.tiles_x + tx);
    const uint32_t n = desc.z;              // (= tile_count[tile] after FILL: the difference of two list starts)
    if (n == 0 && !a.clear_color_on && !a.clear_depth_on) {
        if (threadIdx.x == 0) a.tile_work[tile] = 0u;
        return;
    }
    const uint32_t start = desc.y;
    WaveLdsC<PHONG>& L = s_w;
    constexpr bool VG = WaveLdsC<PHONG>::VG;
    constexpr int RT_ROWS = WaveLdsC<PHONG>::RT_ROWS;
    // the VOut array as a raw buffer: one 32-bit byte offset per vertex in a VGPR, the 16-byte row as the instruction's immediate
    const __amdgpu_buffer_rsrc_t vout_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.vout, 0, (int)a.vout_bytes, 0x00020000);
    // SWR_PROG_DEBUG_VARYINGS (its own instantiations, PROG = 4: a batch holds either only such draws or none, swr_render_mesh flushes
    // in between): Normal of the three outputs from the side array, screen positions from the TriRec
    auto shade_debug = [&L, &a](int t, float w0f, float w1f, float w2f) {
        if (!(PHONG && VG && PROG == SWR_PROG_DEBUG_VARYINGS)) return make_float4(0.f, 0.f, 0.f, 0.f);
        const __amdgpu_buffer_rsrc_t nrm_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.vnorm, 0, (int)(a.vout_bytes >> 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.recs, 0, 0x7ffffff0, 0x00020000);
        const float4 q4 = L.stage[VG ? 4 : 0][t], q5 = L.stage[VG ? 5 : 0][t], q6 = L.stage[VG ? 6 : 0][t];
        auto ld4 = [](__amdgpu_buffer_rsrc_t r, uint32_t off) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
            return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        };
        const float4 na = ld4(nrm_rsrc, __float_as_uint(q4.x) >> 2), nb = ld4(nrm_rsrc, __float_as_uint(q4.y) >> 2), nc = ld4(nrm_rsrc, __float_as_uint(q4.z) >> 2);
        const uint32_t ro = __float_as_uint(q6.x);
        const float4 r0 = ld4(rec_rsrc, ro), r1 = ld4(rec_rsrc, ro + 16u);
        const float sx[3] = { r0.x, r0.y, r0.z }, sy[3] = { r0.w, r1.x, r1.y };
        float wc_clip = 0.0f;
        if constexpr (VG) wc_clip = L.rt_w(t);
        const float inv_w = 1.0f / (float)(a.fp.width - 1), inv_h = 1.0f / (float)(a.fp.height - 1);      // Rasterizer.cs:362-363
        return shade_debug_varyings(w0f, w1f, w2f, q4.w, q5.w, wc_clip, na, nb, nc, sx, sy, inv_w, inv_h);
    };
    // part: 0 = everything, 1 = only the three uv rows (requested ahead of the chain replay, the texel address hangs
    // on them), 2 = everything but the uv rows, which `V` already holds
    auto load_varyings = [&L, vout_rsrc](int t, bool fastdiv, int part = 0, TriVaryings V = TriVaryings()) {
        if (VG) {
            const float4 q4 = L.stage[VG ? 4 : 0][t], q5 = L.stage[VG ? 5 : 0][t];
            const uint32_t oa = __float_as_uint(q4.x), ob = __float_as_uint(q4.y), oc = __float_as_uint(q4.z);
            auto row = [&](uint32_t off, int r) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(vout_rsrc, (int)off + 16 * r, 0, 0);
                return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            };
            auto word = [&](uint32_t off, int byte) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vout_rsrc, (int)off + byte, 0, 0)); };
            // in the order Interpolate consumes them: uv (the texel address), clip.z, colour, the normal's z
            if (part != 2) { V.a_uvn = row(oa, 2); V.b_uvn = row(ob, 2); V.c_uvn = row(oc, 2); }
            if (part == 1) return V;
            V.a_cz = word(oa, 8); V.b_cz = word(ob, 8); V.c_cz = word(oc, 8);
            V.a_col = row(oa, 1); V.b_col = row(ob, 1); V.c_col = row(oc, 1);
            if (PHONG) {
                const float4 a3 = row(oa, 3), b3 = row(ob, 3), c3 = row(oc, 3);
                V.a_wnz = a3.x; V.b_wnz = b3.x; V.c_wnz = c3.x;
                V.a_wpos[0] = a3.y; V.a_wpos[1] = a3.z; V.a_wpos[2] = a3.w;
                V.b_wpos[0] = b3.y; V.b_wpos[1] = b3.z; V.b_wpos[2] = b3.w;
                V.c_wpos[0] = c3.y; V.c_wpos[1] = c3.z; V.c_wpos[2] = c3.w;
            } else {
                V.a_wnz = word(oa, 48); V.b_wnz = word(ob, 48); V.c_wnz = word(oc, 48);
            }
            // the divisions' operands come from LDS (staged once per pair): they do not wait for the rows above
            V.a_r1 = q5.x; V.b_r1 = q5.y; V.c_r1 = q5.z;
            V.a_w = q4.w; V.b_w = q5.w;
            if constexpr (VG) V.c_w = L.rt_w(t);
            V.fastdiv = fastdiv;
            return V;
        }
        const float4 a_clip = SWR_VROW(VG ? 0 : 4, t), b_clip = SWR_VROW(VG ? 0 : 7, t), c_clip = SWR_VROW(VG ? 0 : 10, t);
        V.a_col = SWR_VROW(VG ? 0 : 5, t); V.a_uvn = SWR_VROW(VG ? 0 : 6, t);
        V.b_col = SWR_VROW(VG ? 0 : 8, t); V.b_uvn = SWR_VROW(VG ? 0 : 9, t);
        V.c_col = SWR_VROW(VG ? 0 : 11, t); V.c_uvn = SWR_VROW(VG ? 0 : 12, t);
        V.a_wnz = a_clip.x; V.b_wnz = b_clip.x; V.c_wnz = c_clip.x;
        V.a_r1 = a_clip.y; V.b_r1 = b_clip.y; V.c_r1 = c_clip.y;
        V.a_cz = a_clip.z; V.b_cz = b_clip.z; V.c_cz = c_clip.z;
        V.a_w = a_clip.w; V.b_w = b_clip.w; V.c_w = c_clip.w;
        V.fastdiv = fastdiv;
        if (PHONG) {
            const float4 a3 = SWR_VROW((PHONG && !VG) ? 13 : 0, t), b3 = SWR_VROW((PHONG && !VG) ? 14 : 0, t), c3 = SWR_VROW((PHONG && !VG) ? 15 : 0, t);
            V.a_wpos[0] = a3.y; V.a_wpos[1] = a3.z; V.a_wpos[2] = a3.w;
            V.b_wpos[0] = b3.y; V.b_wpos[1] = b3.z; V.b_wpos[2] = b3.w;
            V.c_wpos[0] = c3.y; V.c_wpos[1] = c3.z; V.c_wpos[2] = c3.w;
        }
        return V;
    };

    const int W = a.fp.width, H = a.fp.height;
    const int x0 = tx * SWR_TILE, y0 = ty * SWR_TILE;

    // A sliding window of WINDOW candidate pairs (lane i holds entry base + i of the tile list: references and
    // count, fetched while the previous batch is rasterised).  Each batch stages the first BATCH candidates that
    // survive (non-empty, not hidden by hi-Z) and consumes the list up to the last of them, so batches are full
    // although a third of the candidates drops out.
    uint4 ref_w = make_uint4(0u, 0u, 0u, 0u);
    int cnt_w = 0;
    float zb_w = 0.0f;                     // hi-Z bound of the entry (k_cover)
    constexpr int BATCH = WaveLdsC<PHONG>::BATCH, WINDOW = WaveLdsC<PHONG>::WINDOW;
    if (lane < WINDOW && (uint32_t)lane < n) {
        ref_w = a.pair_refs[start + (uint32_t)lane];
        const uint2 pi = info[start + (uint32_t)lane];
        cnt_w = (int)pi.x; zb_w = __uint_as_float(pi.y);          // cnt_w keeps k_cover's SWR_INFO_SIMPLE bit (the sign)
    }

    // ---- tile init: clear fused, or one coalesced read of the framebuffer ----
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int p = rr * 64 + lane;
        const int gx = x0 + (p & 15), gy = y0 + (p >> 4);
        const bool inb = gx < W && gy < H;
        const size_t gi = (size_t)(ty_local * SWR_TILE + (p >> 4)) * (size_t)W + (size_t)gx;     // the band's buffers hold its tile rows consecutively
        float4 c;
        if (a.clear_color_on) c = make_float4(a.clear_rgba[0], a.clear_rgba[1], a.clear_rgba[2], a.clear_rgba[3]);
        else c = inb ? a.color[gi] : make_float4(0.f, 0.f, 0.f, 0.f);
        float zz;
        if (a.clear_depth_on) zz = SWR_FLOAT_MINVALUE;
        else zz = inb ? a.depth[gi] : SWR_FLOAT_MINVALUE;
        L.col[p] = c;
        L.z[p] = zz;
    }
    unsigned n_tested = 0, n_shaded = 0, n_written = 0;      // per lane, summed over the wave at the end (a count kept on the scalar
                                                             // side would have to be updated under divergent control flow: per lane again)
    uint32_t carry_key = 0xffffffffu;      // EARLYOUT: (pair, row) of the previous chunk's last fragment ...
    bool carry_dead = false;               // ... and whether that row segment has already hit its `break`
#ifdef SWR_DEBUG_COUNTERS
    unsigned dbg_batches = 0, dbg_chunks = 0, dbg_chunk_lanes = 0, dbg_chain = 0, dbg_chain_c = 0, dbg_sum_r = 0, dbg_sum_c = 0, dbg_hidden = 0;
#endif

    // (A rolling batch -- the tail of a batch carried into the next one so that every chunk is full -- was built and measured
    //  twice: chunks -7.7 %, batches +19 %, kernel +4 %.  It lives in the history of this file, commit "Rolling batch ...".)
    DrawConsts dc = {};                    // per-draw constants of draw `dc_draw` (see DrawConsts)
    uint32_t dc_draw = 0xffffffffu;
    uint32_t batch_no = 0;
    bool fast_draw = false;                // the chunk's draw satisfies the per-draw conditions of shade_dust2_fast
    SWR_PHASE(0);
    for (uint32_t base = 0; base < n; ++batch_no) {
        // ---- batch: empty pairs (binning is conservative) and hidden ones are dropped, the first BATCH survivors
        //      of the window are staged in LDS ----
        const uint32_t pidx = start + base + (uint32_t)lane;
        const uint4 ref = ref_w;
        const bool in_window = lane < WINDOW && base + (uint32_t)lane < n;
        const bool simple_in = cnt_w < 0;                                // SWR_INFO_SIMPLE
        const int cnt_in = in_window ? (cnt_w & 0x7fffffff) : 0;
        int cnt = cnt_in;
        if (HIZ) {
            // Hierarchical Z.  Under Less / LessEqual the stored depth of a pixel only grows, so the minimum over the
            // tile at batch start bounds every later stored value from below.  A pair whose depth provably stays below
            // it at every pixel of bbox /\ tile (k_cover's bound, see there) fails the depth test everywhere: the
            // reference visits those fragments and writes nothing, so the pair is dropped here (its fragments still
            // count as tested).  Exact ties are kept (strict <).
            float zmin = 3.0e38f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int p = rr * 64 + lane;
                const bool inb = x0 + (p & 15) < W && y0 + (p >> 4) < H;
                zmin = fminf(zmin, inb ? L.z[p] : 3.0e38f);
            }
            zmin = wave_min(zmin);
#ifndef SWR_ABL_NOHIZ
            if (cnt > 0 && zmin > SWR_FLOAT_MINVALUE && zb_w < zmin) cnt = 0;
#endif
        }
        // take the first survivors -- at most BATCH pairs and SWR_BATCH_FRAGS fragments (the running sum is monotone, so the
        // taken set is a prefix of the survivors; the first one always fits); the list is consumed up to the first survivor left out
        int consumed;
        const int cscan = wave_incl_scan(cnt, lane);
        {
            const unsigned long long surv = SWR_BALLOT(cnt > 0);
            const int rank = __popcll(surv & ((1ull << lane) - 1ull));
            const unsigned long long left_out = SWR_BALLOT(cnt > 0 && (rank >= BATCH || cscan > SWR_BATCH_FRAGS));
            consumed = left_out ? __ffsll((long long)left_out) - 1 : min(WINDOW, (int)(n - base));
            if (lane >= consumed) cnt = 0;
        }
        const int cnt_seen = lane < consumed ? cnt_in : 0;
        n_tested += (unsigned)cnt_seen;
#ifdef SWR_DEBUG_COUNTERS
        dbg_hidden += (unsigned)(cnt_seen - cnt);
#endif
        // slide the window: every lane fetches its entry of the NEXT window (24 B per candidate, consecutive lanes consecutive addresses,
        // lines this wave touched a batch ago: L2 / L1 hits that return while this batch is rasterised).  Round 3 moved the unconsumed
        // entries down with six ds_bpermute and fetched only the freed lanes: six LDS-crossbar round trips on the batch's critical path
        // to save loads nobody waits for.
        base += (uint32_t)consumed;
        {
            const uint32_t e = base + (uint32_t)lane;
            ref_w = make_uint4(0u, 0u, 0u, 0u); cnt_w = 0; zb_w = 0.0f;
            if (lane < WINDOW && e < n) {
                ref_w = a.pair_refs[start + e];
                const uint2 pi = info[start + e];
                cnt_w = (int)pi.x; zb_w = __uint_as_float(pi.y);
            }
        }
        const int cincl = cscan;                                             // taken lanes precede every lane that was zeroed: their prefix sums stand
        const int total = consumed > 0 ? __builtin_amdgcn_readlane(cscan, min(consumed, 64) - 1) : 0;
        SWR_PHASE(1);
        if (total == 0) continue;
        const unsigned long long nzb = SWR_BALLOT(cnt > 0);
        const int ci = __popcll(nzb & ((1ull << lane) - 1ull));              // compacted index of this lane's pair
        if (lane < SWR_BATCH_FRAGS / 128 + 1) *reinterpret_cast<uint4*>(&L.head[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
        if (cnt > 0) {
            // one round trip: masks, TriRec and the three outputs of every surviving pair
            const uint4 m0 = masks[2 * (size_t)pidx], m1 = masks[2 * (size_t)pidx + 1];
            const float4* __restrict__ fq = reinterpret_cast<const float4*>(a.recs + ref.x);
            const float4 f0 = fq[0], f1 = fq[1], f2 = fq[2], f3 = fq[3];
            bool fastdiv;
            float wc_stage = 0.0f;                                               // VG: clip.w of outputs[2] (rides in the row-start entry)
            if (VG) {
                // only the three clip.w: the varyings themselves are fetched per fragment (load_varyings)
                const float wa = a.vout[ref.y].clip[3], wb = a.vout[ref.z].clip[3], wc = a.vout[ref.w].clip[3];
                fastdiv = div_operands_safe3(wa, wb, wc);
                L.stage[VG ? 4 : 0][ci] = make_float4(__uint_as_float(ref.y << 6), __uint_as_float(ref.z << 6), __uint_as_float(ref.w << 6), wa);
                L.stage[VG ? 5 : 0][ci] = make_float4(rcp_refined(wa), rcp_refined(wb), rcp_refined(wc), wb);
                L.stage[VG ? 6 : 0][ci] = make_float4(__uint_as_float(ref.x << 6), 0.0f, 0.0f, 0.0f);
                wc_stage = wc;
            } else {
                const float4* __restrict__ pa = reinterpret_cast<const float4*>(a.vout + ref.y);
                const float4* __restrict__ pb = reinterpret_cast<const float4*>(a.vout + ref.z);
                const float4* __restrict__ pc = reinterpret_cast<const float4*>(a.vout + ref.w);
                const float4 a3 = pa[3], b3 = pb[3], c3 = pc[3];
                float4 a0 = pa[0], b0 = pb[0], c0 = pc[0];
                a0.x = a3.x; b0.x = b3.x; c0.x = c3.x;                           // clip.x is not read by fragments: wn.z rides there
                // ... nor is clip.y: the refined reciprocal of clip.w rides there (Interpolate's divisions, see div_core)
                fastdiv = div_operands_safe3(a0.w, b0.w, c0.w);
                a0.y = rcp_refined(a0.w); b0.y = rcp_refined(b0.w); c0.y = rcp_refined(c0.w);
                SWR_VROW(VG ? 0 : 4, ci) = a0; SWR_VROW(VG ? 0 : 5, ci) = pa[1]; SWR_VROW(VG ? 0 : 6, ci) = pa[2];
                SWR_VROW(VG ? 0 : 7, ci) = b0; SWR_VROW(VG ? 0 : 8, ci) = pb[1]; SWR_VROW(VG ? 0 : 9, ci) = pb[2];
                SWR_VROW(VG ? 0 : 10, ci) = c0; SWR_VROW(VG ? 0 : 11, ci) = pc[1]; SWR_VROW(VG ? 0 : 12, ci) = pc[2];
                if (PHONG) { SWR_VROW((PHONG && !VG) ? 13 : 0, ci) = a3; SWR_VROW((PHONG && !VG) ? 14 : 0, ci) = b3; SWR_VROW((PHONG && !VG) ? 15 : 0, ci) = c3; }
            }
            {
                const float t0x = f0.x, t1x = f0.y, t2x = f0.z, t0y = f0.w, t1y = f1.x, t2y = f1.y;
                const uint32_t fbx = __float_as_uint(f3.y), fby = __float_as_uint(f3.z);
                const int fsX = max((int)(fbx & 0xffffu), x0), fsY = max((int)(fby & 0xffffu), y0);          // Rasterizer.cs:471-474
                const uint32_t fs = (uint32_t)(fsX - x0) | ((uint32_t)(fsY - y0) << 8);
                const float a01 = t0y - t1y, b01 = t1x - t0x;                                                 // :445-447
                const float a12 = t1y - t2y, b12 = t2x - t1x;
                const float a20 = t2y - t0y, b20 = t0x - t2x;
                float4 r0;
                if (LINES && (__float_as_uint(f3.w) & SWR_FLAG_LINE) != 0u) {
                    r0 = make_float4(t0x, t1x, t0y, t1y);
                } else {
                    const float fsx = (float)fsX, fsy = (float)fsY;
                    r0.x = a12 * (fsx - t1x) + b12 * (fsy - t1y);                                             // :481-483
                    r0.y = a20 * (fsx - t2x) + b20 * (fsy - t2y);
                    r0.z = a01 * (fsx - t0x) + b01 * (fsy - t0y);
                    r0.w = f2.y;
                }
                L.stage[0][ci] = r0;
                L.stage[1][ci] = make_float4(f1.z, f1.w, f2.x, __uint_as_float(__float_as_uint(f3.w) | (fastdiv ? SWR_FLAG_FASTDIV : 0u) |
                                                                                 (simple_in ? SWR_FLAG_SIMPLE : 0u)));
                const uint32_t pre = (uint32_t)(cincl - cnt);                    // stream position of the pair's first fragment
                L.stage[2][ci] = make_float4(a12, a20, a01, __uint_as_float(fs));
                L.stage[3][ci] = make_float4(b12, b20, b01, __uint_as_float(pre));
                if (ci > 0) atomicOr(&L.head[(pre - 1u) >> 5], 1u << ((pre - 1u) & 31u));
                // the reference's row chain from the pair's first row: the values at rows RT_ROWS, 2 RT_ROWS, ... are kept
                float rw0 = r0.x, rw1 = r0.y, rw2 = r0.z;
#pragma unroll
                for (int q = 0; q < WaveLdsC<PHONG>::RT_N; ++q) {
#pragma unroll
                    for (int i = 0; i < RT_ROWS; ++i) { rw0 += b12; rw1 += b20; rw2 += b01; }            // :532-534
                    L.rt_store(q, ci, rw0, rw1, rw2, wc_stage);
                }
            }
            *reinterpret_cast<uint4*>(&L.mask[ci][0]) = m0;
            *reinterpret_cast<uint4*>(&L.mask[ci][4]) = m1;
            const uint32_t p1 = (uint32_t)__popc(m0.x), p2 = p1 + (uint32_t)__popc(m0.y), p3 = p2 + (uint32_t)__popc(m0.z),
                           p4 = p3 + (uint32_t)__popc(m0.w), p5 = p4 + (uint32_t)__popc(m1.x), p6 = p5 + (uint32_t)__popc(m1.y),
                           p7 = p6 + (uint32_t)__popc(m1.z);
            *reinterpret_cast<uint4*>(&L.wpre[ci][0]) = make_uint4(p1 << 16, p2 | (p3 << 16), p4 | (p5 << 16), p6 | (p7 << 16));
        }
#ifdef SWR_DEBUG_COUNTERS
        ++dbg_batches;
#endif

        // the staging lanes' LDS writes above are read by OTHER lanes below: one wave, so the LDS unit already executes them
        // in order -- the fence pair only stops the compiler from ever moving a read above a write it cannot see aliasing
        SWR_WAVE_LDS_SYNC();
        // ---- fragment stream of the batch, 64 at a time ----
        int t0 = 0;                                     // pairs that start at or before fragment `pos`, minus one
        // The first two links of the lookup's dependent chain are taken off the chunk's critical path by running them one chunk
        // ahead: the three head words of the next window are fetched as soon as this chunk's cut is known, and between this
        // chunk's chain replay and its shading each lane's next pair is computed from them and that pair's word prefix counts
        // and stream position are fetched (`lookup_ahead`).  The next chunk then starts with its mask-word read.
        uint32_t hn0 = L.head[0], hn1 = L.head[1], hn2 = L.head[2];
        int t_n = 0;                           // next chunk: pair of this lane's fragment ...
        uint4 wp_n = make_uint4(0u, 0u, 0u, 0u);   // ... its word prefix counts ...
        uint32_t pre_n = 0u;                   // ... and the stream position of its first fragment
        auto lookup_ahead = [&](int pos_n) {
            const int hs_n = pos_n & 31;
            const uint32_t wl = __builtin_amdgcn_alignbit(hn1, hn0, hs_n), wh = __builtin_amdgcn_alignbit(hn2, hn1, hs_n);
            t_n = t0 + (int)__builtin_amdgcn_mbcnt_hi(wh, __builtin_amdgcn_mbcnt_lo(wl, 0u));
            wp_n = *reinterpret_cast<const uint4*>(&L.wpre[t_n][0]);
            pre_n = __float_as_uint(L.stage[3][t_n].w);
        };
        lookup_ahead(0);
        SWR_PHASE(2);
        for (int pos = 0; pos < total;) {
            const int g = pos + lane;
            const bool valid = g < total;
            const unsigned long long validmask = SWR_BALLOT(g < total);
            // pair of fragment g = number of head bits below position g: a 64-bit window of the bitmap at `pos`
            const int hs = pos & 31;
            const uint32_t h0 = hn0, h1 = hn1, h2 = hn2;
            const uint32_t win_lo = __builtin_amdgcn_alignbit(h1, h0, hs), win_hi = __builtin_amdgcn_alignbit(h2, h1, hs);
            const int t = t_n;
            const float4 f0 = L.stage[0][t], f1 = L.stage[1][t], f2 = L.stage[2][t], f3 = L.stage[3][t];
            TriVaryings Vg = TriVaryings();
            int k = valid ? g - (int)pre_n : 0;
            // k-th covered pixel of pair t in row-major order: the mask word by a 16-bit-field compare against the
            // word prefix counts, then a 5-level selection inside the word
            int pix;
            {
                const uint4 wp = wp_n;
                const uint32_t kk = ((uint32_t)k | ((uint32_t)k << 16)) | 0x80008000u;
                const int nle = __popc((kk - wp.x) & 0x80008000u) + __popc((kk - wp.y) & 0x80008000u) +
                                __popc((kk - wp.z) & 0x80008000u) + __popc((kk - wp.w) & 0x80008000u);    // fields <= k, >= 1
                const int wi = nle - 1;
                const uint32_t wsel = L.mask[t][wi];
                const int kw = k - (int)reinterpret_cast<const uint16_t*>(&L.wpre[t][0])[wi];
                const bool okk = valid;                              // (kw < popc(wsel) holds for every valid fragment)
                // the word holds two rows; when each is one run (SWR_FLAG_SIMPLE, nearly always) the k-th covered pixel
                // is first-pixel-of-the-run + k
                const bool simple = (__float_as_uint(f1.w) & SWR_FLAG_SIMPLE) != 0u;
                const uint32_t lo = wsel & 0xffffu;
                const int c0 = __popc(lo);
                const bool up = kw >= c0;
                const uint32_t half = up ? (wsel >> 16) : lo;
                int posw = (up ? 16 + (kw - c0) : kw) + (__ffs((int)half) - 1);
                // (a ballot of ONE compare is the compare's own mask; invalid lanes index some staged pair, at worst a spurious general pass)
                if (SWR_BALLOT((__float_as_uint(f1.w) & SWR_FLAG_SIMPLE) == 0u) != 0ull) {
                    if (!simple) posw = kth_set_bit32(wsel, okk ? kw : 0);
                }
                pix = (wi * 32 + posw) & 255;
            }
            // What the replay and the depth test will read from LDS depends on the pixel only: requested here, for every lane,
            // the reads return while the election's atomic is on its way (inside `if (act)` below they would be one more
            // dependent round trip; the compiler may not move them across the atomic itself).  The tile is as the previous chunk
            // left it and no two lanes that survive the cut share a pixel, so reading before the cut changes nothing.
            const uint32_t fs_early = __float_as_uint(f2.w);
            const int nrow_e = (pix >> 4) - (int)(fs_early >> 8);          // (lines, invalid lanes: any value; q stays in 0..3)
            const int q_e = nrow_e >> (RT_ROWS == 8 ? 3 : 2), qi_e = max(q_e, 1) - 1;
            // (not in the row-early-out kernels: they are at the register limit, and three spilled dwords cost more than the round trip)
            float t0r_e = 0.0f, t1r_e = 0.0f, t2r_e = 0.0f, z_old = 0.0f;
            if (!EARLYOUT) {
                L.rt_load(qi_e, t, t0r_e, t1r_e, t2r_e);
                z_old = L.z[pix];
            }
            // duplicate election: of the lanes that share a pixel in this chunk all but one must wait.  Which of them
            // wins does not matter for the result: the chunk is cut at the LOWEST loser, so no two lanes before the cut
            // share a pixel (a pixel has one winner; two lanes below the cut on one pixel would make one of them a loser
            // below the cut).  Lane 0 never loses and whoever shares its pixel always does, so the cut is >= 1.
            // (For speed the LOWER sharer should win -- later cut.  Measured, not guaranteed: the returning atomic below behaves
            //  that way (chunks average 57.8 of 64 fragments on cfg3, 60.2 on cfg2); an election by plain byte stores +
            //  read-back, 5 instructions cheaper, lets the HIGHEST lane win and halved cfg2's chunks, and with a second
            //  store round it cost two LDS round trips: cfg3 -1.5 %, cfg2 +6 %.)
            SWR_PHASE(3);
            const int pix_first = __builtin_amdgcn_readfirstlane(pix);
            // (every predicate is balloted where its compare is: the ballot of ONE compare is the compare's own lane mask, while a
            //  ballot of a combined or branch-carried bool is materialised in a VGPR and compared again)
            unsigned long long dupmask = 0ull;
#ifndef SWR_ABL_NOELECT
            {
                // the bitmap is cleared by eight 16-byte stores and every valid lane claims its pixel with one returning atomic:
                // the lanes of one instruction are served in some order, so exactly one sharer of a pixel sees its bit clear
                if (lane < 8) *reinterpret_cast<uint4*>(&L.touched[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
                SWR_WAVE_LDS_SYNC();      // the clearing store above -> the atomics below
                const uint32_t pbit = 1u << (pix >> 5);
                uint32_t before = 0u;
                if (valid) before = atomicOr(&L.touched[pix & 31], pbit);
                dupmask = (SWR_BALLOT((before & pbit) != 0u) | SWR_BALLOT(pix == pix_first)) & ~1ull;       // lane 0 never loses
            }
#endif
            const uint32_t dflags = __float_as_uint(f1.w);
            const uint32_t draw = dflags & SWR_DRAW_MASK;
            const uint32_t draw0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)draw);
            const unsigned long long stop = ~validmask | dupmask | SWR_BALLOT(draw != draw0);
            const int cut = stop ? (__ffsll((long long)stop) - 1) : 64;        // >= 1: lane 0 is valid, never dup, own draw
            {
                const unsigned long long win = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)win_hi) << 32) |
                                               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)win_lo);
                t0 += __popcll(cut >= 64 ? win : (win & ((1ull << cut) - 1ull)));
            }
            SWR_PHASE(4);
            const bool act = lane < cut;
            if (VG) { if (act) Vg = load_varyings(t, (dflags & SWR_FLAG_FASTDIV) != 0u, 1); }
            {   // (the head array has two spare words behind the last position: reading past the batch's end is in bounds)
                const int hw_next = (pos + cut) >> 5;
                hn0 = L.head[hw_next]; hn1 = L.head[hw_next + 1]; hn2 = L.head[hw_next + 2];
            }
#ifdef SWR_DEBUG_COUNTERS
            ++dbg_chunks; dbg_chunk_lanes += (unsigned)cut;
#endif

            const DrawParams* __restrict__ cdp = a.draws + draw0;
            if (draw0 != dc_draw) {                                                       // wave-uniform: the constants live in SGPRs across chunks
                dc_draw = draw0; dc = load_draw_consts(cdp);
                if (!PHONG && !LINES && PROG == SWR_PROG_DUST2_LAMBERT_FOG) fast_draw = dust2_fast_applies(dc);
                if (PHONG && !LINES && PROG == SWR_PROG_PHONG_4POINT) fast_draw = phong4_fast_applies(dc);
            }
            const int f_program = PROG >= 0 ? PROG : dc.program, f_blend = BLEND >= 0 ? BLEND : dc.blend, f_dt = DT >= 0 ? DT : dc.depth_test;
            // outputs[0].Interpolate: every program but FLAT_COLOR sets it (k_setup), and the clipper's vertices always do
            const bool f_interp = PROG > SWR_PROG_FLAT_COLOR ? true : (__float_as_uint(f1.w) & SWR_FLAG_INTERP) != 0u;
#ifdef SWR_DEBUG_COUNTERS
            int dbg_nrow = 0, dbg_ncol = 0;
#endif
            bool e_pass = false, e_alpha = false;          // EARLYOUT: results held until the row kills are known
            float e_d = 0.0f;
            float4 e_src = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool is_line = LINES && (dflags & SWR_FLAG_LINE) != 0u;
            float w0f = 0.0f, w1f = 0.0f, w2f = 0.0f, d = 0.0f;
            if (act) {
                const float d0 = f1.x, d1 = f1.y, d2 = f1.z;
                const uint32_t fs = __float_as_uint(f2.w);
                if (is_line) {
                    // DrawLine fragment, Rasterizer.cs:299-322: weights (1-t, t, 0) on outputs[0], outputs[1], outputs[0]
                    float t;
                    (void)line_test(f0.x, f0.z, f0.y, f0.w, x0 + (pix & 15), y0 + (pix >> 4), t);
                    w0f = 1.0f - t; w1f = t; w2f = 0.0f;
                    d = 1.0f / (d0 * (1.0f - t) + d1 * t);                                                // :315
                } else {
                    // replay of the reference's add chain from the pair's first pixel: rows first, then columns
                    float w0 = f0.x, w1 = f0.y, w2 = f0.z;
                    const float inv_area = f0.w;
                    const int nrow = (pix >> 4) - (int)(fs >> 8), ncol = (pix & 15) - (int)(fs & 0xffu);
#ifdef SWR_DEBUG_COUNTERS
                    dbg_nrow = nrow; dbg_ncol = ncol;
#endif
                    // rows: from the nearest staged row start (rowtab: rows 4, 8, 12 of the pair), at most 3 steps
                    {
                        const int q = nrow >> (RT_ROWS == 8 ? 3 : 2);
                        if (EARLYOUT) L.rt_load(qi_e, t, t0r_e, t1r_e, t2r_e);
                        if (q > 0) { w0 = t0r_e; w1 = t1r_e; w2 = t2r_e; }          // (read ahead of the election, see there)
                        const int rem = nrow & (RT_ROWS - 1);
                        replay_chain<RT_ROWS - 1>(w0, w1, w2, rem, f3.x, f3.y, f3.z);                      // :532-534
                    }
                    replay_chain<15>(w0, w1, w2, ncol, f2.x, f2.y, f2.z);                                  // :527-529
                    w0f = w0 * inv_area; w1f = w1 * inv_area; w2f = w2 * inv_area;                        // :498-500
                    d = (d0 * w0f + d1 * w1f) + d2 * w2f;                                                 // :502
                }
            }
            SWR_PHASE(5);
            lookup_ahead(pos + cut);           // all lanes; its LDS reads return during the shading below
            if (act) {
                if (!EARLYOUT) {
                    if (depth_func(f_dt, d, z_old)) {                                                      // :505 / :318
                        ++n_shaded;
                        const float4 dst = L.col[pix];        // read before the shading: off the dependent path of the blend
#ifdef SWR_ABL_NOSHADE
                        const float4 src = make_float4(w0f, w1f, w2f, 1.0f);
#else
                        float4 src = make_float4(0.f, 0.f, 0.f, 0.f);
                        constexpr bool FAST = !LINES && ((!PHONG && PROG == SWR_PROG_DUST2_LAMBERT_FOG) || (PHONG && PROG == SWR_PROG_PHONG_4POINT));
                        const TriVaryings Vs = VG ? load_varyings(t, (dflags & SWR_FLAG_FASTDIV) != 0u, 2, Vg) : load_varyings(t, (dflags & SWR_FLAG_FASTDIV) != 0u);
                        bool need_exact = !(FAST && fast_draw);           // wave-uniform
                        if (FAST && fast_draw) {
                            // speculate: the straight-line shader; verify: every shaded lane of the chunk took only legal shortcuts
                            bool safe;
                            if constexpr (PHONG) src = shade_phong4_fast(cdp, dc, Vs, w0f, w1f, w2f, safe);
                            else src = shade_dust2_fast(dc, Vs, w0f, w1f, w2f, safe);
                            need_exact = SWR_BALLOT(!safe) != 0ull;
                        }
                        if (need_exact)
                            src = (PHONG && PROG == SWR_PROG_DEBUG_VARYINGS) ? shade_debug(t, w0f, w1f, w2f) :
                                  shade_fragment<PHONG>(cdp, dc, f_program, f_interp, Vs, w0f, w1f, w2f);                           // :507-509 / :321-323
#endif
                        // triangles: W > 0 (:511); lines: W != 0 (:325)
                        if (is_line ? (src.w != 0.0f) : (src.w > 0.0f)) {
                            L.col[pix] = blend(src, dst, f_blend);                                         // :513-515
                            if (f_dt != SWR_DEPTH_DISABLED) L.z[pix] = d;                                  // :517-518
                            ++n_written;
                        }
                    }
                } else {
                    e_d = d;
                    e_pass = depth_func(f_dt, d, L.z[pix]);
                    if (e_pass) {
                        e_src = (PHONG && PROG == SWR_PROG_DEBUG_VARYINGS) ? shade_debug(t, w0f, w1f, w2f) :
                                shade_fragment<PHONG>(cdp, dc, f_program, f_interp,
                                                      VG ? load_varyings(t, (dflags & SWR_FLAG_FASTDIV) != 0u, 2, Vg) : load_varyings(t, (dflags & SWR_FLAG_FASTDIV) != 0u),
                                                      w0f, w1f, w2f);
                        e_alpha = is_line ? (e_src.w != 0.0f) : (e_src.w > 0.0f);
                    }
                }
            }
#ifdef SWR_DEBUG_COUNTERS
            {
                int mr = act ? dbg_nrow : 0, mc = act ? dbg_ncol : 0, sr = mr, sc = mc;
                for (int off = 32; off > 0; off >>= 1) { mr = max(mr, __shfl_xor(mr, off)); mc = max(mc, __shfl_xor(mc, off));
                                                         sr += __shfl_xor(sr, off); sc += __shfl_xor(sc, off); }
                dbg_chain += (unsigned)mr; dbg_chain_c += (unsigned)mc; dbg_sum_r += (unsigned)sr; dbg_sum_c += (unsigned)sc;
            }
#endif
            if (EARLYOUT) {
                // canEarlyOut (Rasterizer.cs:430,520-523): under BlendMode.None the first fragment of a row (within this
                // triangle and tile) that passes depth but fails alpha ends the row -- nothing to its right is visited.
                // Fragments of one (pair, row) are consecutive in the stream, so inside a chunk that is a segmented
                // "any failure to my left" over lanes; a segment cut by the chunk boundary carries its state over.
                bool killed = false;
                const bool none = f_blend == SWR_BLEND_NONE;                       // uniform: a chunk holds one draw
                const uint32_t key = ((batch_no * (uint32_t)BATCH + (uint32_t)t) << 4) | (uint32_t)(pix >> 4);   // unique per (pair, row)
                if (none) {
                    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
                    const bool head = act && (lane == 0 || key != prev);
                    const unsigned long long H = SWR_BALLOT(head);
                    const unsigned long long F = SWR_BALLOT(act && e_pass && !e_alpha);
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
            SWR_PHASE(6);
        }
    }

    // ---- write back: each wave store covers 4 rows x 256 B (colour) / 4 rows x 64 B (Z) ----
    // (the addresses are derived from a lane id the optimiser cannot connect with the one of the tile init: otherwise it keeps the
    //  init's four 64-bit pixel offsets, LDS addresses and bounds masks -- 25 VGPRs -- alive across the whole batch loop)
    int lane_wb = lane;
    asm volatile("" : "+v"(lane_wb));
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int p = rr * 64 + lane_wb;
        const int gx = x0 + (p & 15), gy = y0 + (p >> 4);
        if (gx < W && gy < H) {
            const size_t gi = (size_t)(ty_local * SWR_TILE + (p >> 4)) * (size_t)W + (size_t)gx;     // the band's buffers hold its tile rows consecutively
            a.color[gi] = L.col[p];
            a.depth[gi] = L.z[p];
        }
    }

    n_tested = (unsigned)__builtin_amdgcn_readlane(wave_incl_scan((int)n_tested, lane), 63);
    n_shaded = (unsigned)__builtin_amdgcn_readlane(wave_incl_scan((int)n_shaded, lane), 63);
    n_written = (unsigned)__builtin_amdgcn_readlane(wave_incl_scan((int)n_written, lane), 63);
    if (lane == 0) a.tile_work[tile] = n_tested;            // the next flush's scheduling weight (k_scan_apply)
    SWR_PHASE(7);
    SWR_PHASE_FLUSH(a.dbg);
    if (lane == 0 && n > 0) {
        uint32_t* ts = a.tile_stats + 3u * tile;
        atomicAdd(&ts[0], n_tested); atomicAdd(&ts[1], n_shaded); atomicAdd(&ts[2], n_written);     // no return value: no round trip
    }
#ifdef SWR_DEBUG_COUNTERS
    if (lane == 0 && a.dbg) {
        atomicAdd(&a.dbg[0], (unsigned long long)dbg_batches); atomicAdd(&a.dbg[1], (unsigned long long)dbg_chunks);
        atomicAdd(&a.dbg[4], (unsigned long long)dbg_chunk_lanes); atomicAdd(&a.dbg[3], (unsigned long long)dbg_chain);
        atomicAdd(&a.dbg[2], (unsigned long long)dbg_chain_c); atomicAdd(&a.dbg[5], (unsigned long long)dbg_sum_r);
        atomicAdd(&a.dbg[6], (unsigned long long)dbg_sum_c);
    }
    {
        unsigned h = dbg_hidden;
        for (int off = 32; off > 0; off >>= 1) h += (unsigned)__shfl_xor((int)h, off);
        if (lane == 0 && a.dbg) atomicAdd(&a.dbg[7], (unsigned long long)h);
    }
#endif
}

}  // namespace swr
