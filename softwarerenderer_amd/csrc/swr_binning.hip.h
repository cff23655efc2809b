// swr_binning.hip.h -- (triangle, 16x16 tile) pair binning that preserves submission order.
//
// The reference visits, for every triangle, every 16x16 tile of its clamped pixel bbox
// (Rasterizer.cs:449-476) and the per-pixel result depends on the ORDER triangles reach a
// pixel (depth ties under `>=`, blending).  The oracle of record is the serial order, so each
// tile's list must hold its triangles in ascending slot order:
//   k_bin<COUNT>  : tile_count[tile] += 1 per pair               (integer atomics, order-free)
//   k_scan        : exclusive scan -> tile_start, total pairs
//   k_bin<FILL>   : list[tile_start + atomic cursor] = slot       (arbitrary order inside a tile)
//   k_sort_tiles  : one wave per tile sorts its segment ascending (restores submission order)
#pragma once
#include "swr_device.h"

namespace swr {

struct BinArgs {
    const unsigned long long* __restrict__ slot_tb;
    uint32_t slot_lo, slot_hi;           // slots [lo, hi) are binned in this round
    int tiles_x;
    int band_ty0, band_ty1;
    uint32_t* __restrict__ tile_count;   // COUNT: incremented; FILL: used as cursor (zeroed again before)
    const uint32_t* __restrict__ tile_start;
    uint32_t* __restrict__ tile_list;
    uint32_t* __restrict__ pair_tile;    // FILL: band-local tile index of every pair (k_cover reads it)
    uint32_t list_capacity;
    Counters* __restrict__ counters;
};

template <bool FILL>
__device__ __forceinline__ void bin_one(const BinArgs& a, uint32_t tile, uint32_t slot) {
    if (FILL) {
        uint32_t pos = atomicAdd(&a.tile_count[tile], 1u);
        uint32_t at = a.tile_start[tile] + pos;
        if (at < a.list_capacity) { a.tile_list[at] = slot; a.pair_tile[at] = tile; }
        else a.counters->overflow = 1u;
    } else {
        atomicAdd(&a.tile_count[tile], 1u);
    }
}

// thread per slot; triangles spanning many tiles are spread over the whole wave
template <bool FILL>
__global__ __launch_bounds__(256) void k_bin(BinArgs a) {
    const uint32_t slot = a.slot_lo + blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int tminx = 0, tminy = 0, nx = 0, ny = 0;
    if (slot < a.slot_hi) {
        unsigned long long tb = a.slot_tb[slot];
        if (tb != SWR_TB_INVALID) {
            tminx = (int)(tb & 0xffff);
            int tmaxx = (int)((tb >> 16) & 0xffff);
            tminy = (int)((tb >> 32) & 0xffff);
            int tmaxy = (int)((tb >> 48) & 0xffff);
            tminy = max(tminy, a.band_ty0);
            tmaxy = min(tmaxy, a.band_ty1 - 1);
            nx = tmaxx - tminx + 1;
            ny = tmaxy - tminy + 1;
            if (ny <= 0) { nx = 0; ny = 0; }
        }
    }
    const int nt = nx * ny;
    const bool big = nt > 8;
    if (!big) {
        for (int i = 0; i < nt; ++i) {
            int ty = tminy + i / nx, tx = tminx + i % nx;
            bin_one<FILL>(a, (uint32_t)((ty - a.band_ty0) * a.tiles_x + tx), slot);
        }
    }
    unsigned long long m = __ballot(big);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int s_tminx = __shfl(tminx, src), s_tminy = __shfl(tminy, src);
        const int s_nx = __shfl(nx, src), s_nt = __shfl(nt, src);
        const uint32_t s_slot = (uint32_t)__shfl((int)slot, src);
        for (int i = lane; i < s_nt; i += 64) {
            int ty = s_tminy + i / s_nx, tx = s_tminx + i % s_nx;
            bin_one<FILL>(a, (uint32_t)((ty - a.band_ty0) * a.tiles_x + tx), s_slot);
        }
    }
}

// single-workgroup exclusive scan over the band's tiles (<= 512x512 tiles at 8192^2)
__global__ __launch_bounds__(1024) void k_scan(const uint32_t* __restrict__ count, uint32_t* __restrict__ start,
                                               uint32_t n, unsigned long long* __restrict__ total_out) {
    __shared__ unsigned long long s_sum[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(tid * per, n), hi = min(lo + per, n);
    unsigned long long sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += count[i];
    s_sum[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {        // Hillis-Steele inclusive scan
        unsigned long long v = (tid >= off) ? s_sum[tid - off] : 0ull;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    unsigned long long run = s_sum[tid] - sum;              // exclusive prefix of this thread's chunk
    for (uint32_t i = lo; i < hi; ++i) {
        start[i] = (uint32_t)min(run, 0xffffffffull);
        run += count[i];
    }
    if (tid == 1023) *total_out = s_sum[1023];
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

// one 64-thread block (one wave) per tile
__global__ __launch_bounds__(64) void k_sort_tiles(const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ tile_count,
                                                   uint32_t* __restrict__ tile_list, uint32_t n_tiles) {
    __shared__ uint32_t s_keys[SWR_SORT_LDS];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t n = tile_count[tile];
    if (n < 2) return;
    uint32_t* seg = tile_list + tile_start[tile];
    const uint32_t lane = threadIdx.x;

    if (n <= 64) {                                   // in registers, cross-lane shuffles
        uint32_t key = lane < n ? seg[lane] : 0xffffffffu;
        for (uint32_t k = 2; k <= 64; k <<= 1) {
            {   // flip step
                uint32_t partner = lane ^ (k - 1);
                uint32_t other = (uint32_t)__shfl((int)key, (int)partner);
                key = (lane < partner) ? min(key, other) : max(key, other);
            }
            for (uint32_t j = k >> 2; j > 0; j >>= 1) {
                uint32_t partner = lane ^ j;
                uint32_t other = (uint32_t)__shfl((int)key, (int)partner);
                key = (lane < partner) ? min(key, other) : max(key, other);
            }
        }
        if (lane < n) seg[lane] = key;
        return;
    }

    uint32_t m = 1;
    while (m < n) m <<= 1;
    const bool in_lds = n <= SWR_SORT_LDS;
    uint32_t* buf = in_lds ? s_keys : seg;
    if (in_lds) {
        for (uint32_t i = lane; i < n; i += 64) s_keys[i] = seg[i];
        __syncthreads();
    }
    for (uint32_t k = 2; k <= m; k <<= 1) {
        // flip: within each block of k, i in the lower half pairs with block_base + k-1 - offset
        for (uint32_t t = lane; t < (m >> 1); t += 64) {
            uint32_t blk = t / (k >> 1), off = t % (k >> 1);
            uint32_t i = blk * k + off, p = blk * k + (k - 1 - off);
            if (i < n) { if (in_lds) cmpx_lds(buf, i, p, n); else cmpx_glb(buf, i, p, n); }
        }
        if (!in_lds) __threadfence_block();
        __syncthreads();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (m >> 1); t += 64) {
                uint32_t i = 2 * j * (t / j) + (t % j), p = i + j;
                if (i < n) { if (in_lds) cmpx_lds(buf, i, p, n); else cmpx_glb(buf, i, p, n); }
            }
            if (!in_lds) __threadfence_block();
            __syncthreads();
        }
    }
    if (in_lds) {
        for (uint32_t i = lane; i < n; i += 64) seg[i] = s_keys[i];
    }
}

}  // namespace swr
