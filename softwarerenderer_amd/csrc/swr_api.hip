// swr_api.hip -- host side of libswr_hip.so: the C ABI of include/swr.h over the gfx950 kernels.
//
// A context records draws (Rasterizer.RenderMesh calls, Rasterizer.cs:163-174) in submission order
// and executes them as ONE batch per flush:
//     [k_frustum_cull ->] k_vertex -> k_setup -> k_bin<count> -> k_scan_sums/apply -> k_bin<fill> -> k_sort_tiles
//     (+ the tile order in its last blocks) -> k_cover -> k_raster_c
// There is no CPU fallback anywhere in this file: every pixel is produced by the HIP kernels.
//
// Frames in flight (swr_set_pipelining, on by default): the reference's loop renders frames back to back (Renderer.cs:404-419,
// Rasterizer.cs:163-230), and a frame here has two halves with different bounds -- a front end of record traffic and latency
// (vertex .. k_cover) and an issue-bound raster kernel.  The front end of flush N+1 therefore runs on a stream of its own
// (`front_stream`) beside the raster kernel of flush N: everything a raster kernel reads lives in one of two RasterSets that
// alternate per flush, two events per set order the hand-overs (front_done: F -> R, raster_done: R -> the front end that reuses
// the set two flushes later), the tile order's history comes from flush N-1, and the poison / replay protocol is per batch
// (batch_poisoned in swr_device.h).  The framebuffer is touched by the raster stream only, so everything a caller orders against
// `stream` (flatten, read-back, collectives) still sees frames in submission order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "swr.h"
#include "swr_device.h"
#include "swr_geometry.hip.h"
#include "swr_binning.hip.h"
#include "swr_raster.hip.h"
#include "swr_raster_c.hip.h"
#include "swr_cull.hip.h"

using namespace swr;

struct swr_mesh {
    float4* d_bounds = nullptr;               // Mesh.SphereBounds (ModelLoader.cs:291), computed on first use
    bool bounds_ready = false;
    swr_vertex* d_verts = nullptr;
    uint16_t* d_idx = nullptr;
    int n_verts = 0, n_idx = 0;
    size_t cap_verts = 0, cap_idx = 0;        // bytes allocated behind d_verts / d_idx (a recycled transient mesh may hold more than it uses)
    bool transient = false;
    float box_lo[3] = { 0, 0, 0 }, box_hi[3] = { 0, 0, 0 };   // exact model-space AABB of the vertices (host, at creation)
    bool has_box = false;
};
struct swr_texture {
    uint8_t* d_rgba = nullptr;
    uint8_t* d_blocked = nullptr;             // block-linear copy (4 x 4-texel blocks of 64 B) for the bilinear filter, made when it is first switched on
    int w = 0, h = 0;
    bool bilinear = false;                    // build-defined extension; the reference's Texture.Sample is nearest
};

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum Stage { ST_VERTEX = 0, ST_SETUP, ST_BIN, ST_SORT, ST_COVER, ST_RASTER, ST_CLEAR, ST_COUNT };

struct EventSpan { int stage; hipEvent_t a, b; };

#ifndef SWR_TIMING_EVENT_FLAGS
#define SWR_TIMING_EVENT_FLAGS hipEventDisableSystemFence
#endif
#ifndef SWR_HANDOVER_EVENT_FLAGS
#define SWR_HANDOVER_EVENT_FLAGS hipEventDisableTiming
#endif
#define SWR_SLOTS 3
struct FrameSlot { void* host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };

// Everything of a batch that its raster kernel reads (or that a front-end kernel and the raster kernel share).  With frames
// pipelined two sets alternate per flush, so the front end of flush N+1 never writes what the raster kernel of flush N reads;
// with pipelining off only set 0 is used.  Buffers only the front end touches (slot_tb, want, tile_list, pair_tile, the scan's
// totals) are single: front ends run in order on one stream.
struct RasterSet {
    DevBuf d_upload;         // draws | vertex block map | triangle block map [| bounds pointers | visibility words] of the batch
    DevBuf d_vout, d_vnorm, d_recs;
    DevBuf d_masks, d_pcounts, d_pair_refs;
    DevBuf d_tile_count, d_tile_start;
    DevBuf d_order;          // [tile_order n_tiles] uint4 {tile, list start, pairs, -}, [tile_work n_tiles][hist 256][cursor 256] u32, [tile_bucket n_tiles] u8: heaviest-first raster order
    uint32_t hist_tiles = 0; // tile count the fragment history in d_order (tile_work) belongs to (0: none yet)
    hipEvent_t front_done = nullptr, raster_done = nullptr;
    bool raster_pending = false;      // raster_done has been recorded and the stream has not been drained since
};

struct DrawCmd {
    DrawParams p;
    swr_mesh* mesh;
    bool frustum_cull = false;                 // render only if IsSphereInFrustum(mesh bounds, model, view, proj)
};

// one flush = one batch; kept until the host has seen that it fitted (optimistic execution, see swr::Ctrl)
struct Batch {
    std::vector<DrawCmd> draws;
    bool clear_color = false, clear_depth = false;
    float clear_rgba[4] = { 0, 0, 0, 0 };
    float near_clip = 0.1f;
    bool wireframe = false;                    // Rasterizer.RenderDebugMode == Wireframe for the whole batch
    uint32_t seq = 0;
    float4* color = nullptr;                   // the framebuffer bound when the batch was flushed: a replay (validate_locked) must
    float* depth = nullptr;                    // hit the same buffers even if the caller has bound others since (double buffering)
};

}  // namespace

struct swr_context {
    int device = 0;
    std::mutex mu;
    std::string err;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t front_stream = nullptr;    // front ends of pipelined flushes (and mesh uploads, which only front-end kernels read)
    int pipelining = 1;                    // swr_set_pipelining: 0 off, 1 every batch (default), 2 small frames / small batches only (see execute_batch)
    uint32_t pipeline_max_tris = 1u << 17; // mode 2: a batch is pipelined when it has at most this many triangles ...
    uint32_t pipeline_max_tiles = 1u << 15;   // ... or the band at most this many tiles (8 raster waves per wave slot of the chip)
    hipEvent_t f_tail_ev = nullptr, r_front_ev = nullptr;
    bool f_tail_pending = false;           // the front stream carries work (a front end, a mesh upload) the raster stream has not been ordered behind
    bool r_front_pending = false;          // an unpipelined batch ran its front end on the raster stream since the front stream last waited for it
    RasterSet sets[2];
    uint32_t raster_span_no = 0;           // profiling mode 3: raster launches seen since swr_profile_enable
    char dev_name[256] = { 0 };

    int W = 0, H = 0, tiles_x = 0, tiles_y = 0;
    bool geometry_applied = false;            // swr_resize has run at least once (a second call with the same size is a no-op)
    bool band_set = false;
    int band_first = 0, band_count = 0;       // as requested by swr_set_band
    int band_ty0 = 0, band_ty1 = 0;           // effective
    int il_k = 0, il_world = 1, il_rank = 0;  // swr_set_band_interleaved: stripes of il_k tile rows, stripe s belongs to rank s % il_world
    int band_tile_rows = 0;                   // tile rows this context stores (contiguous band or stripes)
    float4* color = nullptr;                  // band storage in use (own or external)
    float* depth = nullptr;
    DevBuf own_color, own_depth;
    void* ext_color = nullptr; void* ext_depth = nullptr;

    uint32_t nm_flags = SWR_NUMERICS_FMA ? (SWR_NM_TRANSFORM_FMA | SWR_NM_TRANSFORM_NORMAL_FMA) : 0u;   // swr_set_transform_fma
    float near_clip = 0.1f, far_clip = 1000.0f;   // Rasterizer.cs:20-21
    int debug_mode = SWR_DEBUG_NONE;               // Rasterizer.cs:22

    bool pend_clear_color = false, pend_clear_depth = false;
    float clear_rgba[4] = { 0, 0, 0, 0 };

    std::vector<DrawCmd> draws;
    uint64_t pend_verts = 0, pend_tris = 0;
    std::vector<swr_mesh*> garbage;           // transient meshes no recorded draw needs any more, possibly still read by kernels in flight
    std::vector<swr_mesh*> mesh_pool;         // transient meshes whose batches are KNOWN to be complete: their device buffers are handed to the
    size_t mesh_pool_bytes = 0;               // next swr_render_mesh_arrays call instead of hipFree / hipMalloc (both synchronise the device)
    uint64_t stale_dropped[2] = { 0, 0 };     // present tickets dropped by back-pressure whose pixels predate a replay (swr_present_wait reports them)
    std::vector<Batch> inflight;              // launched optimistically, not yet validated
    uint32_t next_seq = 1;
    bool sync_flush = false;                  // SWR_SYNC_FLUSH=1: read the pair total back in every flush
    uint32_t debug_fill_capacity = 0;         // SWR_DEBUG_FILL_CAPACITY=n: k_bin<FILL> of optimistic flushes sees a list of n entries (tests)

    DevBuf d_slot_tb;        // (the vertex-stage output, the records and the upload block live in the RasterSets)
    FrameSlot slots[SWR_SLOTS];
    uint32_t slot_next = 0;
    DevBuf d_pair_tile, d_ctrl;
    uint32_t* host_poison = nullptr;           // pinned, device-visible copy of Ctrl::poison
    DevBuf d_tile_list, d_tile_stats, d_counters, d_total, d_scratch;
    DevBuf d_want;           // 1 byte per slot: COUNT's pair_may_cover decisions, replayed by FILL
    size_t tile_stats_tiles = 0;
    swr_stats totals = {};
    unsigned long long host_tile_pairs = 0;   // rounds sized on the host (MODE_SYNC)
    unsigned long long replays = 0;           // times an optimistic batch did not fit and was replayed
    unsigned long long host_syncs = 0;        // times an entry point made the host wait for the stream (swr_sync_count)
    // asynchronous present (swr_present_rgb_async): two device staging buffers alternate; the flatten runs on `stream`, the copy to
    // the host on `copy_stream`, so the next frame renders while this one crosses PCIe
    hipStream_t copy_stream = nullptr;
    DevBuf present_buf[2];
    hipEvent_t present_flat[2] = { nullptr, nullptr }, present_done[2] = { nullptr, nullptr };
    uint64_t present_ticket[2] = { 0, 0 };    // ticket whose copy the slot carries (0 = none pending)
    uint32_t present_seq[2] = { 0, 0 };       // last batch flushed before that present: retired when the copy is known to be over
    uint64_t next_ticket = 0;

    int profiling = 0;                         // 0 off, 1 every stage, 2 only the raster kernel (2 events per flush)
    std::vector<EventSpan> spans;
    std::vector<float> raster_samples;         // duration of every raster launch that carried an event pair since swr_profile_reset (<= 65,536)
    std::vector<hipEvent_t> event_pool;
    swr_profile prof = {};
};

namespace {

#define SWR_HIP(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char b_[512];                                                                         \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (ctx)->err = b_;                                                                      \
            return e_ == hipErrorOutOfMemory ? SWR_ERR_OOM : SWR_ERR_HIP;                         \
        }                                                                                         \
    } while (0)

int fail(swr_context* c, int code, const char* msg) { c->err = msg; return code; }

int ensure(swr_context* c, DevBuf& b, size_t bytes, bool zero_new = false) {
    if (bytes <= b.cap) return SWR_OK;
    size_t want = std::max(bytes, b.cap + b.cap / 2);
    if (b.p) { SWR_HIP(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    SWR_HIP(c, hipMalloc(&b.p, want));
    b.cap = want;
    if (zero_new) SWR_HIP(c, hipMemsetAsync(b.p, 0, want, c->stream));
    return SWR_OK;
}

void release(DevBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

// Frame-slot ring: a batch takes the next slot = {pinned staging block for its upload, completion event}.
// Taking a slot first waits for the batch that used it SWR_SLOTS flushes ago, which (a) makes the pinned block
// safe to overwrite and (b) bounds how far the host runs ahead of the GPU (pageable uploads and unbounded
// queue depth both serialise the stream on some HIP runtimes).
void* slot_acquire(swr_context* c, size_t bytes) {
    FrameSlot& fs = c->slots[c->slot_next % SWR_SLOTS];
    if (fs.busy) { (void)hipEventSynchronize(fs.done); fs.busy = false; }
    if (fs.cap < bytes) {
        if (fs.host) (void)hipHostFree(fs.host);
        fs.host = nullptr; fs.cap = 0;
        size_t cap = std::max<size_t>(bytes + bytes / 2, 1 << 16);
        if (hipHostMalloc(&fs.host, cap, hipHostMallocDefault) != hipSuccess) { fs.host = nullptr; return nullptr; }
        fs.cap = cap;
    }
    return fs.host;
}
void slot_submit(swr_context* c) {            // call after the batch's last launch
    FrameSlot& fs = c->slots[c->slot_next % SWR_SLOTS];
    if (!fs.done) (void)hipEventCreateWithFlags(&fs.done, hipEventDisableTiming);
    if (fs.done && hipEventRecord(fs.done, c->stream) == hipSuccess) fs.busy = true;
    c->slot_next++;
}

BandMap host_band_map(const swr_context* c) {
    BandMap b; b.ty0 = c->band_ty0; b.ty1 = c->band_ty1; b.il_k = c->il_k; b.il_world = c->il_world; b.il_rank = c->il_rank;
    return b;
}
int band_y0(const swr_context* c) { return c->band_ty0 * SWR_TILE; }     // contiguous band only
// pixel rows stored: the band's tile rows, 16 pixel rows each, the frame's last tile row possibly partial
int band_rows(const swr_context* c) {
    if (c->band_tile_rows <= 0) return 0;
    const int last_global = band_global_row(host_band_map(c), c->band_tile_rows - 1);
    const int last_rows = std::min(SWR_TILE, c->H - last_global * SWR_TILE);
    return (c->band_tile_rows - 1) * SWR_TILE + std::max(0, last_rows);
}
// row of pixel row y in the band's buffers, -1 if the band does not hold it
int band_local_pixel_row(const swr_context* c, int y) {
    if (y < 0 || y >= c->H) return -1;
    const int lr = band_local_row(host_band_map(c), y / SWR_TILE);
    return lr < 0 ? -1 : lr * SWR_TILE + y % SWR_TILE;
}
size_t band_pixels(const swr_context* c) { return (size_t)std::max(0, c->W) * (size_t)band_rows(c); }

int apply_geometry(swr_context* c) {
    c->tiles_x = c->W > 0 ? (c->W + SWR_TILE - 1) / SWR_TILE : 0;     // Rasterizer.cs:76-77
    c->tiles_y = c->H > 0 ? (c->H + SWR_TILE - 1) / SWR_TILE : 0;
    if (c->il_k > 0) {
        c->band_ty0 = 0; c->band_ty1 = c->tiles_y;                     // ownership is decided row by row (BandMap)
        int rows = 0;
        for (int ty = 0; ty < c->tiles_y; ++ty) rows += band_local_row(host_band_map(c), ty) >= 0 ? 1 : 0;
        c->band_tile_rows = rows;
    } else {
        if (c->band_set) {
            c->band_ty0 = std::min(std::max(c->band_first, 0), c->tiles_y);
            c->band_ty1 = std::min(c->band_ty0 + std::max(c->band_count, 0), c->tiles_y);
        } else {
            c->band_ty0 = 0; c->band_ty1 = c->tiles_y;
        }
        c->band_tile_rows = c->band_ty1 - c->band_ty0;
    }
    size_t n = band_pixels(c);
    if (c->ext_color) {
        c->color = (float4*)c->ext_color; c->depth = (float*)c->ext_depth;
    } else {
        int rc;
        if ((rc = ensure(c, c->own_color, std::max<size_t>(n, 1) * sizeof(float4), false))) return rc;
        if ((rc = ensure(c, c->own_depth, std::max<size_t>(n, 1) * sizeof(float), false))) return rc;
        c->color = c->own_color.as<float4>(); c->depth = c->own_depth.as<float>();
    }
    return SWR_OK;
}

FrameParams frame_params(const swr_context* c) {
    FrameParams fp;
    fp.width = c->W; fp.height = c->H; fp.tiles_x = c->tiles_x; fp.tiles_y = c->tiles_y;
    fp.band_ty0 = c->band_ty0; fp.band_ty1 = c->band_ty1;
    fp.band_y0 = band_y0(c); fp.band_rows = band_rows(c);
    fp.il_k = c->il_k; fp.il_world = c->il_world; fp.il_rank = c->il_rank; fp.band_tile_rows = c->band_tile_rows;
    fp.near_clip = c->near_clip;
    return fp;
}

hipEvent_t get_event(swr_context* c) {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    // timing only, never ordering: no system-scope fence (the default event's cache write-back / invalidation would hit the OTHER
    // stream's kernels when frames are in flight -- measured: 20 timed frames 0.651 ms with default events, see SWR_EVENT_FLAGS)
    if (hipEventCreateWithFlags(&e, SWR_TIMING_EVENT_FLAGS) != hipSuccess) (void)hipEventCreate(&e);
    return e;
}
struct ScopedSpan {
    swr_context* c; int stage; hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    bool on = false;
    ScopedSpan(swr_context* c_, int st, hipStream_t s_ = nullptr) : c(c_), stage(st), s(s_ ? s_ : c_->stream) {
        // 1: every stage; 2: the raster kernel of every flush; 3: the raster kernel of every 4th flush (an event pair costs
        // about 10 us of stream time: sampling keeps a timed region within 0.5 % of its unobserved rate)
        on = c->profiling == 1 || (st == ST_RASTER && (c->profiling == 2 || (c->profiling == 3 && (c->raster_span_no++ & 3u) == 0u)));
        if (on) { a = get_event(c); b = get_event(c); (void)hipEventRecord(a, s); }
    }
    ~ScopedSpan() {
        if (on) { (void)hipEventRecord(b, s); c->spans.push_back({ stage, a, b }); }
    }
};

void collect_spans(swr_context* c) {      // stream must be idle
    for (auto& s : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            switch (s.stage) {
            case ST_VERTEX: c->prof.vertex_ms += ms; break;
            case ST_SETUP:  c->prof.setup_ms += ms; break;
            case ST_BIN:    c->prof.bin_ms += ms; break;
            case ST_SORT:   c->prof.sort_ms += ms; break;
            case ST_COVER:  c->prof.cover_ms += ms; break;
            case ST_RASTER: c->prof.raster_ms += ms; c->prof.raster_launches++;
                            if (c->raster_samples.size() < 65536) c->raster_samples.push_back(ms);
                            break;
            case ST_CLEAR:  c->prof.clear_ms += ms; break;
            }
            c->prof.total_ms += ms;
        }
        c->event_pool.push_back(s.a); c->event_pool.push_back(s.b);
    }
    c->spans.clear();
}

void destroy_mesh(swr_mesh* m) {
    if (m->d_verts) (void)hipFree(m->d_verts);
    if (m->d_idx) (void)hipFree(m->d_idx);
    if (m->d_bounds) (void)hipFree(m->d_bounds);
    delete m;
}
// A transient mesh (swr_render_mesh_arrays: the reference's RenderMesh(vertices, indices, ...) makes one per call) whose batch is
// complete goes to the pool, not to hipFree: freeing synchronises the whole device, which is exactly what an asynchronous present
// loop must not do, and the next call with arrays of that size needs the same buffers again.
const size_t kMeshPoolBytes = (size_t)1 << 30;
void trim_mesh_pool(swr_context* c) {      // keeps the newest meshes up to the byte budget (hipFree: a device-wide wait unless the streams are idle)
    while (c->mesh_pool_bytes > kMeshPoolBytes && !c->mesh_pool.empty()) {
        swr_mesh* m = c->mesh_pool.front();
        c->mesh_pool.erase(c->mesh_pool.begin());
        c->mesh_pool_bytes -= m->cap_verts + m->cap_idx;
        destroy_mesh(m);
    }
}
void pool_mesh(swr_context* c, swr_mesh* m) {
    c->mesh_pool.push_back(m);
    c->mesh_pool_bytes += m->cap_verts + m->cap_idx;
    // a present loop that never drains and whose array sizes never repeat would grow the pool for ever: past twice the budget the
    // oldest (unused: pooled meshes are complete) are freed here, at the price of one device-wide wait
    if (c->mesh_pool_bytes > 2 * kMeshPoolBytes) trim_mesh_pool(c);
}
void free_garbage(swr_context* c) {       // streams must be idle: nothing in flight reads the garbage any more
    for (swr_mesh* m : c->garbage) pool_mesh(c, m);
    c->garbage.clear();
    trim_mesh_pool(c);
}

int validate_locked(swr_context* c);
int check_list_overflow(swr_context* c);

// both streams idle (the raster stream waits for every front end it depends on; the front stream may carry mesh uploads beyond that)
int drain_streams(swr_context* c) {
    SWR_HIP(c, hipStreamSynchronize(c->stream));
    if (c->front_stream) SWR_HIP(c, hipStreamSynchronize(c->front_stream));
    for (auto& s : c->sets) s.raster_pending = false;
    c->f_tail_pending = c->r_front_pending = false;
    return SWR_OK;
}

int sync_locked(swr_context* c) {
    ++c->host_syncs;
    int rc = drain_streams(c);
    if (rc) return rc;
    rc = validate_locked(c);
    collect_spans(c);
    free_garbage(c);
    for (auto& fs : c->slots) fs.busy = false;     // stream idle: every slot is free
    return rc;
}

// the stream mesh uploads and front-end-only work go to: the front stream while frames are pipelined, else the context's stream
// (and notes that the front stream now carries work the raster stream has not been ordered behind)
hipStream_t use_front_stream(swr_context* c) {
    if (c->pipelining && c->front_stream) { c->f_tail_pending = true; return c->front_stream; }
    return c->stream;
}

// creates the front stream (once) and the hand-over events
int ensure_front_stream(swr_context* c) {
    if (!c->pipelining) return SWR_OK;
    if (c->front_stream) return SWR_OK;
    int least = 0, greatest = 0;
    SWR_HIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    // high priority: the front end's short kernels go first wherever a slot frees up, the raster kernel (65,536 one-wave workgroups)
    // fills the rest (default priority measured the same within noise, profiles/r04_frames_in_flight.md)
    (void)least;
    SWR_HIP(c, hipStreamCreateWithPriority(&c->front_stream, hipStreamNonBlocking, greatest));
    SWR_HIP(c, hipEventCreateWithFlags(&c->f_tail_ev, hipEventDisableTiming));
    SWR_HIP(c, hipEventCreateWithFlags(&c->r_front_ev, hipEventDisableTiming));
    for (auto& s : c->sets) {
        if (!s.front_done) SWR_HIP(c, hipEventCreateWithFlags(&s.front_done, SWR_HANDOVER_EVENT_FLAGS));
        if (!s.raster_done) SWR_HIP(c, hipEventCreateWithFlags(&s.raster_done, SWR_HANDOVER_EVENT_FLAGS));
    }
    return SWR_OK;
}

enum { MODE_SYNC = 0, MODE_ASYNC = 1 };
static bool b_has_debug_varyings(const Batch& b) {
    for (auto& d : b.draws) if (d.p.program == SWR_PROG_DEBUG_VARYINGS) return true;
    return false;
}
const unsigned long long kMaxPairs = 1ull << 30;       // list entries per round (4 GiB of slot ids)

// pairs every set in use can hold (the optimistic flush is checked against this on the device)
size_t pair_capacity(const swr_context* c) {
    size_t cap = std::min(c->d_tile_list.cap / 4, c->d_pair_tile.cap / 4);
    for (int i = 0; i < (c->pipelining ? 2 : 1); ++i) {
        const RasterSet& s = c->sets[i];
        cap = std::min(cap, std::min(std::min(s.d_masks.cap / 32, s.d_pcounts.cap / 8), s.d_pair_refs.cap / 16));
    }
    return cap;
}
int ensure_pairs(swr_context* c, size_t n) {
    int rc;
    if ((rc = ensure(c, c->d_tile_list, n * 4))) return rc;
    if ((rc = ensure(c, c->d_pair_tile, n * 4))) return rc;
    for (int i = 0; i < (c->pipelining ? 2 : 1); ++i) {
        RasterSet& s = c->sets[i];
        if ((rc = ensure(c, s.d_masks, n * 32))) return rc;
        if ((rc = ensure(c, s.d_pair_refs, n * 16))) return rc;
        if ((rc = ensure(c, s.d_pcounts, n * 8 + 64))) return rc;
    }
    return SWR_OK;
}

int run_clear(swr_context* c, const Batch& b, bool& cc, bool& cd, const float rgba_[4]) {
    size_t n = band_pixels(c);
    if (n && (cc || cd)) {
        ScopedSpan sp(c, ST_CLEAR);
        int blocks = (int)std::min<size_t>((n + 255) / 256, 2048 * 8);
        float4 rgba = make_float4(rgba_[0], rgba_[1], rgba_[2], rgba_[3]);
        hipLaunchKernelGGL(k_clear, dim3(blocks), dim3(256), 0, c->stream, b.color, b.depth, n, rgba,
                           cc ? 1 : 0, cd ? 1 : 0, (const Ctrl*)c->d_ctrl.as<Ctrl>(), b.seq);
        SWR_HIP(c, hipGetLastError());
    }
    cc = cd = false;
    return SWR_OK;
}

// bins slots [lo, hi) and rasterises them.  MODE_SYNC reads the pair total back (sizes buffers exactly, splits a
// range that would need more than kMaxPairs entries); MODE_ASYNC launches everything against the current capacity.
// the BinArgs of one round (slots [lo, hi) of batch b)
static BinArgs make_bin_args(swr_context* c, RasterSet& S, const Batch& b, uint32_t lo, uint32_t hi) {
    BinArgs ba;
    ba.slot_tb = c->d_slot_tb.as<unsigned long long>();
    ba.recs = S.d_recs.as<TriRec>();
    ba.slot_lo = lo; ba.slot_hi = hi;
    ba.spt = b.wireframe ? 6u : 2u;
    ba.width = c->W; ba.height = c->H;
    ba.tiles_x = c->tiles_x; ba.band_ty0 = c->band_ty0; ba.band_ty1 = c->band_ty1;
    ba.band = host_band_map(c);
    ba.tile_count = S.d_tile_count.as<uint32_t>();
    ba.tile_start = S.d_tile_start.as<uint32_t>();
    ba.tile_list = c->d_tile_list.as<uint32_t>();
    ba.list_capacity = (uint32_t)std::min<size_t>(pair_capacity(c), 0xffffffffu);
    ba.counters = c->d_counters.as<Counters>();
    ba.ctrl = c->d_ctrl.as<Ctrl>();
    ba.seq = b.seq; ba.replayable = 0;
    ba.total = c->d_total.as<unsigned long long>();
    ba.want = c->d_want.as<uint8_t>();
    ba.tpw = 64u;
    return ba;
}

// counts_clear: k_vertex of this batch cleared the per-tile counters and the tile order's histogram (no memsets here)
int bin_and_raster(swr_context* c, RasterSet& S, hipStream_t F, const Batch& b, bool& cc, bool& cd, uint32_t lo, uint32_t hi, int mode, bool counts_clear = false) {
    const uint32_t n_tiles = (uint32_t)c->tiles_x * (uint32_t)c->band_tile_rows;
    if (n_tiles == 0 || lo >= hi) return SWR_OK;
    int rc;
    const Ctrl* ctrl = c->d_ctrl.as<Ctrl>();
    unsigned long long* d_total = c->d_total.as<unsigned long long>();
    uint4* tile_order = S.d_order.as<uint4>();                    // (16-byte records first: alignment)
    uint32_t* tile_work = reinterpret_cast<uint32_t*>(tile_order + n_tiles);
    uint32_t* order_hist = tile_work + n_tiles;                   // [hist 256][cursor 256]
    uint8_t* tile_bucket = reinterpret_cast<uint8_t*>(order_hist + 2 * SWR_ORDER_BUCKETS);     // [n_tiles]
    BinArgs ba = make_bin_args(c, S, b, lo, hi);
    ba.replayable = mode == MODE_ASYNC ? 1u : 0u;
    const uint32_t bin_threads = (hi - lo + ba.spt - 1u) / ba.spt;          // = triangles
    ba.tpw = 64u;                                                            // aim for >= ~1000 waves
    while (ba.tpw > 4u && bin_threads < ba.tpw * 1024u) ba.tpw >>= 1;
    const uint32_t bin_blocks = (bin_threads + 4u * ba.tpw - 1u) / (4u * ba.tpw);
    {
        ScopedSpan sp(c, ST_BIN, F);
        if (!counts_clear) {
            SWR_HIP(c, hipMemsetAsync(ba.tile_count, 0, (size_t)n_tiles * 4, F));
            SWR_HIP(c, hipMemsetAsync(order_hist, 0, 2 * SWR_ORDER_BUCKETS * 4, F));
        }
        hipLaunchKernelGGL(k_bin<false>, dim3(bin_blocks), dim3(256), 0, F, ba);
        const unsigned scan_blocks = (n_tiles + (unsigned)SWR_SCAN_BLOCK - 1u) / (unsigned)SWR_SCAN_BLOCK;
        unsigned long long* sums = d_total + 32;      // room for 4096 block sums (2^20 tiles)
        hipLaunchKernelGGL(k_scan_sums, dim3(scan_blocks), dim3(SWR_SCAN_BLOCK), 0, F, (const uint32_t*)ba.tile_count, n_tiles, sums);
        // async: the device decides whether the batch fits; sync: the host does (capacity "infinite" here)
        const unsigned long long cap = mode == MODE_ASYNC ? (unsigned long long)ba.list_capacity : ~0ull;
        hipLaunchKernelGGL(k_scan_apply, dim3(scan_blocks), dim3(SWR_SCAN_BLOCK), 0, F, ba.tile_count,
                           S.d_tile_start.as<uint32_t>(), n_tiles, (const unsigned long long*)sums, d_total,
                           cap, b.seq, c->d_ctrl.as<Ctrl>(), c->d_counters.as<Counters>() + 64,
                           mode == MODE_ASYNC ? 1 : 0, (const uint32_t*)tile_work, order_hist, tile_bucket);
        SWR_HIP(c, hipGetLastError());
    }
    uint32_t cover_items = ba.list_capacity;          // async: grid covers the whole capacity, lanes beyond the total exit
    if (mode == MODE_SYNC) {
        unsigned long long total = 0;
        SWR_HIP(c, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, F));      // (MODE_SYNC: F is the context's stream)
        SWR_HIP(c, hipStreamSynchronize(F));
        if (total > kMaxPairs && hi - lo > 2) {
            // too many (triangle, tile) pairs for one round: split the slot range.  Order is preserved
            // because the framebuffer carries the state from one round to the next.
            uint32_t mid = lo + (((hi - lo) / 2u) & ~1u);
            if (mid == lo) mid = lo + 2;
            if ((rc = bin_and_raster(c, S, F, b, cc, cd, lo, mid, mode))) return rc;
            return bin_and_raster(c, S, F, b, cc, cd, mid, hi, mode);
        }
        Counters* tp = c->d_counters.as<Counters>() + 64;          // tile_pairs of a round that really runs
        if (total == 0) return run_clear(c, b, cc, cd, b.clear_rgba);
        if (total > 0xffffffffull) return fail(c, SWR_ERR_UNSUPPORTED, "a single triangle covers more tile pairs than one round can hold");
        c->host_tile_pairs += total; (void)tp;
        // a little headroom so that the next, similar frame can run without reading the total back
        if ((rc = ensure_pairs(c, (size_t)std::min<unsigned long long>(total + total / 4 + 4096, std::max(total, kMaxPairs))))) return rc;
        ba.tile_list = c->d_tile_list.as<uint32_t>();
        ba.list_capacity = (uint32_t)std::min<size_t>(pair_capacity(c), 0xffffffffu);
        cover_items = (uint32_t)total;
    }
    {
        ScopedSpan sp(c, ST_BIN, F);
        BinArgs bf = ba;
        // test hook: a FILL capacity below what COUNT was checked against forces the list-overflow path (bin_overflow)
        if (mode == MODE_ASYNC && c->debug_fill_capacity) bf.list_capacity = std::min(bf.list_capacity, c->debug_fill_capacity);
        // the grid's last blocks place the tiles in the raster kernel's dispatch order (tile_place_block)
        bf.bin_blocks = bin_blocks;
        bf.order_tiles_y = c->band_tile_rows;
        bf.tile_bucket = tile_bucket; bf.order_hist = order_hist; bf.order_cursor = order_hist + SWR_ORDER_BUCKETS; bf.tile_order = tile_order;
        hipLaunchKernelGGL(k_bin<true>, dim3(bin_blocks + order_blocks(c->tiles_x, c->band_tile_rows)), dim3(256), 0, F, bf);   // cursors were zeroed by k_scan_apply
        SWR_HIP(c, hipGetLastError());
    }
    {
        ScopedSpan sp(c, ST_SORT, F);
        hipLaunchKernelGGL(k_sort_tiles, dim3((n_tiles + SWR_SORT_TPB * SWR_SORT_TPW - 1u) / (SWR_SORT_TPB * SWR_SORT_TPW)), dim3(64 * SWR_SORT_TPB), 0, F, S.d_tile_start.as<uint32_t>(),
                           S.d_tile_count.as<uint32_t>(), c->d_tile_list.as<uint32_t>(), n_tiles, c->d_pair_tile.as<uint32_t>(), ctrl, b.seq,
                           (const uint4*)tile_order);
        SWR_HIP(c, hipGetLastError());
    }
    if (cover_items) {
        ScopedSpan sp(c, ST_COVER, F);
        CoverArgs ca;
        ca.recs = S.d_recs.as<TriRec>();
        ca.tile_list = c->d_tile_list.as<uint32_t>();
        ca.pair_tile = c->d_pair_tile.as<uint32_t>();
        ca.masks = S.d_masks.as<uint4>();
        ca.info = S.d_pcounts.as<uint2>();
        ca.refs = S.d_pair_refs.as<uint4>();
        ca.n_pairs = d_total;
        ca.ctrl = ctrl; ca.seq = b.seq;
        ca.fp = frame_params(c);
        ca.fp.near_clip = b.near_clip;
        ca.dbg = d_total + 8;
        // (a multiple of 8 blocks: the kernel hands XCD x the x-th contiguous eighth of the blocks that hold pairs)
        const dim3 cg((unsigned)((((cover_items + (uint32_t)SWR_COVER_BLOCK - 1u) / (uint32_t)SWR_COVER_BLOCK) + 7u) & ~7u)), cb(SWR_COVER_BLOCK);
        if (b.wireframe) hipLaunchKernelGGL(k_cover<true>, cg, cb, 0, F, ca);
        else hipLaunchKernelGGL(k_cover<false>, cg, cb, 0, F, ca);
        SWR_HIP(c, hipGetLastError());
    }
    if (F != c->stream) {
        // hand-over: the raster kernel (context's stream) starts when this batch's front end is complete
        SWR_HIP(c, hipEventRecord(S.front_done, F));
        SWR_HIP(c, hipStreamWaitEvent(c->stream, S.front_done, 0));
    } else if (c->pipelining && mode == MODE_ASYNC) {
        // a batch that is NOT pipelined (too big to gain, see execute_batch) while others are: the next pipelined front end shares
        // slot_tb / want / the lists with this one and may not start on the front stream before this point of the raster stream
        SWR_HIP(c, hipEventRecord(c->r_front_ev, c->stream));
        c->r_front_pending = true;
    }
    {
        ScopedSpan sp(c, ST_RASTER);
        RasterArgs ra;
        ra.fp = frame_params(c);
        ra.fp.near_clip = b.near_clip;
        ra.recs = S.d_recs.as<TriRec>();
        ra.vout = S.d_vout.as<VOut>();
        ra.vout_bytes = (uint32_t)std::min<size_t>(S.d_vout.cap, 0xfffffff0u);
        ra.vnorm = b_has_debug_varyings(b) ? S.d_vnorm.as<float4>() : nullptr;
        ra.draws = reinterpret_cast<const DrawParams*>(S.d_upload.p);
        ra.tile_start = S.d_tile_start.as<uint32_t>();
        ra.tile_count = S.d_tile_count.as<uint32_t>();
        ra.pair_refs = S.d_pair_refs.as<uint4>();
        ra.color = b.color; ra.depth = b.depth;
        ra.tile_stats = c->d_tile_stats.as<uint32_t>();
        memcpy(ra.clear_rgba, b.clear_rgba, 16);
        ra.clear_color_on = cc ? 1 : 0;
        ra.clear_depth_on = cd ? 1 : 0;
        ra.tile_order = tile_order;
        ra.tile_work = tile_work;
        ra.n_tiles = n_tiles;
        ra.dbg = d_total + 8;     // zero unless a SWR_DEBUG_COUNTERS build bumps it
        ra.ctrl = ctrl; ra.seq = b.seq;
        {
            const dim3 g((n_tiles + 511u) & ~511u), t(64);       // one wave per tile (grid in whole 8 x 64 XCD segments)
            const uint4* mk = (const uint4*)S.d_masks.as<uint4>();
            const uint2* pc = (const uint2*)S.d_pcounts.as<uint2>();
            bool phong = false, none = false, dust2_default = true, grows = true, phong_default = true, gouraud_default = true;
            for (auto& d : b.draws) {
                gouraud_default = gouraud_default && d.p.program == SWR_PROG_GOURAUD &&
                                  d.p.blend == SWR_BLEND_ALPHA && d.p.depth_test == SWR_DEPTH_LESSEQUAL;
                phong_default = phong_default && d.p.program == SWR_PROG_PHONG_4POINT &&
                                d.p.blend == SWR_BLEND_ALPHA && d.p.depth_test == SWR_DEPTH_LESSEQUAL;
                grows = grows && (d.p.depth_test == SWR_DEPTH_LESS || d.p.depth_test == SWR_DEPTH_LESSEQUAL);
                phong = phong || d.p.program == SWR_PROG_PHONG_4POINT;
                none = none || d.p.blend == SWR_BLEND_NONE;
                // the reference's own frame: every mesh drawn with Renderer's shader pair and the RenderMesh defaults
                dust2_default = dust2_default && d.p.program == SWR_PROG_DUST2_LAMBERT_FOG &&
                                d.p.blend == SWR_BLEND_ALPHA && d.p.depth_test == SWR_DEPTH_LESSEQUAL;
            }
            ra.depth_only_grows = grows ? 1 : 0;
            if (b_has_debug_varyings(b)) {        // (never mixed with other programs, never wireframe: swr_render_mesh / flush_locked)
                if (none) hipLaunchKernelGGL((k_raster_c<false, true, SWR_PROG_DEBUG_VARYINGS, -1, -1, true>), g, t, 0, c->stream, ra, mk, pc);
                else hipLaunchKernelGGL((k_raster_c<false, true, SWR_PROG_DEBUG_VARYINGS>), g, t, 0, c->stream, ra, mk, pc);
            }
            else if (b.wireframe) hipLaunchKernelGGL((k_raster_c<true, true>), g, t, 0, c->stream, ra, mk, pc);   // DrawLine has no early-out
            else if (none) hipLaunchKernelGGL((k_raster_c<false, true, -1, -1, -1, true>), g, t, 0, c->stream, ra, mk, pc);
            else if (phong_default)
                hipLaunchKernelGGL((k_raster_c<false, true, SWR_PROG_PHONG_4POINT, SWR_BLEND_ALPHA, SWR_DEPTH_LESSEQUAL>), g, t, 0, c->stream, ra, mk, pc);
            else if (phong) hipLaunchKernelGGL((k_raster_c<false, true>), g, t, 0, c->stream, ra, mk, pc);
            else if (dust2_default)
                hipLaunchKernelGGL((k_raster_c<false, false, SWR_PROG_DUST2_LAMBERT_FOG, SWR_BLEND_ALPHA, SWR_DEPTH_LESSEQUAL>), g, t, 0, c->stream, ra, mk, pc);
            else if (gouraud_default)
                hipLaunchKernelGGL((k_raster_c<false, false, SWR_PROG_GOURAUD, SWR_BLEND_ALPHA, SWR_DEPTH_LESSEQUAL>), g, t, 0, c->stream, ra, mk, pc);
            else hipLaunchKernelGGL((k_raster_c<false, false>), g, t, 0, c->stream, ra, mk, pc);
        }
        SWR_HIP(c, hipGetLastError());
        cc = cd = false;
    }
    if (c->pipelining && S.raster_done) {
        // ... and the front end that reuses this set (two flushes from now) starts when this raster kernel is complete
        SWR_HIP(c, hipEventRecord(S.raster_done, c->stream));
        S.raster_pending = true;
    }
    return SWR_OK;
}

// launches one batch (clears + draws) on the stream
int execute_batch(swr_context* c, const Batch& b, int mode, int count_stats) {
    bool cc = b.clear_color, cd = b.clear_depth;
    if (b.draws.empty()) return run_clear(c, b, cc, cd, b.clear_rgba);
    int rc;
    // pipelined flushes alternate between the two RasterSets and run their front end on the front stream; a synchronous batch (the
    // first frame, a replay, SWR_SYNC_FLUSH) reads the pair total back half way and runs on the context's stream alone
    if ((rc = ensure_front_stream(c))) return rc;
    // What it buys (profiles/r04_frames_in_flight.md).  A raster kernel holds every byte of LDS while it has tiles left to start, so a
    // front-end block runs beside it only in the slot of a retiring raster wave and only if it fits what four raster waves per SIMD
    // leave (SWR_FRONT_MAX_LDS / SWR_FRONT_MAX_VGPRS, swr_device.h).  With every front-end kernel inside that budget the whole front
    // end of a 4096^2 / 1 M-triangle frame runs beside the previous frame's raster kernel, which pays for the company (392 -> 491 us)
    // less than the front end costs alone: cfg3 -5.9 %, cfg4 -3.9 %, cfg5 -5.6 % in steady state, 1920x1080 -8 ... -16 %.  A burst of K
    // frames pays one un-overlapped front end to fill the pipe (+0.2 ms / K); mode 2 keeps big frames on one stream for callers whose
    // bursts are that short (frames of at most 2^15 tiles or batches of at most 2^17 triangles are pipelined regardless).
    uint64_t n_tris_batch = 0;
    for (auto& d : b.draws) n_tris_batch += d.p.n_tris;
    const uint64_t n_tiles_band = (uint64_t)c->tiles_x * (uint64_t)c->band_tile_rows;
    const bool piped = mode == MODE_ASYNC && (c->pipelining == 1 || (c->pipelining == 2 && (n_tris_batch <= c->pipeline_max_tris ||
                                                                                            n_tiles_band <= c->pipeline_max_tiles)));
    RasterSet& S = c->sets[c->pipelining ? (b.seq & 1u) : 0u];
    const hipStream_t F = piped ? c->front_stream : c->stream;
    if (c->pipelining && mode != MODE_ASYNC) SWR_HIP(c, hipStreamSynchronize(c->front_stream));     // earlier front ends and mesh uploads
    if (c->pipelining && mode == MODE_ASYNC) {
        // stream-level ordering between the two places a front end can run (no host wait)
        if (!piped && c->f_tail_pending) {
            SWR_HIP(c, hipEventRecord(c->f_tail_ev, c->front_stream));
            SWR_HIP(c, hipStreamWaitEvent(c->stream, c->f_tail_ev, 0));
            c->f_tail_pending = false;
        }
        if (piped && c->r_front_pending) {
            SWR_HIP(c, hipStreamWaitEvent(c->front_stream, c->r_front_ev, 0));
            c->r_front_pending = false;
        }
        if (piped) c->f_tail_pending = true;
    }
    const size_t nd = b.draws.size();
    std::vector<DrawParams> hp(nd);
    std::vector<BlockMap> vblocks, tblocks;
    std::vector<uint32_t> frag_reps;
    uint64_t V = 0, T = 0;
    for (size_t i = 0; i < nd; ++i) {
        DrawParams p = b.draws[i].p;
        p.vert_base = (uint32_t)V; p.tri_base = (uint32_t)T;
        for (uint32_t f = 0; f < p.n_verts; f += SWR_GEOM_BLOCK) vblocks.push_back({ (uint32_t)i, f });
        for (uint32_t f = 0; f < p.n_tris; f += SWR_GEOM_BLOCK) tblocks.push_back({ (uint32_t)i, f });
        V += p.n_verts; T += p.n_tris;
        // fragment-stage identity: draws that differ only in geometry / matrices / cull mode share one set of fragment constants
        p.frag_draw = (uint32_t)i;
        for (uint32_t j : frag_reps) {
            const DrawParams& q = hp[j];
            if (q.program == p.program && q.blend == p.blend && q.depth_test == p.depth_test && q.tex == p.tex && q.tex_w == p.tex_w &&
                q.tex_h == p.tex_h && memcmp(&q.u, &p.u, sizeof p.u) == 0) { p.frag_draw = j; break; }
        }
        // a representative must have its k_vertex block 0 run (it writes fog_r1 / fog_den): it has vertices and is not subject to the
        // device-side frustum test; the list is bounded so that a batch of thousands of distinct materials stays linear
        if (p.frag_draw == (uint32_t)i && p.n_verts > 0 && !b.draws[i].frustum_cull && frag_reps.size() < 64) frag_reps.push_back((uint32_t)i);
        hp[i] = p;
    }
    const uint64_t spt = b.wireframe ? 6 : 2;        // primitive slots per submitted triangle
    // (V + 4 T vertex records of 64 B: k_raster_c addresses them with 32-bit byte offsets)
    if (V + 4 * T >= (1ull << 26) || spt * T >= 0xffffffffull)
        return fail(c, SWR_ERR_UNSUPPORTED, "internal error: a batch beyond 2^26 vertex-stage records reached execute_batch (record_draw flushes before)");
    if (T == 0) return run_clear(c, b, cc, cd, b.clear_rgba);
    const uint32_t n_tiles = (uint32_t)c->tiles_x * (uint32_t)c->band_tile_rows;
    // the per-tile scan (k_scan_sums / k_scan_apply) holds 4096 block sums of SWR_SCAN_BLOCK = 256 tiles each
    if (n_tiles > (1u << 20))
        return fail(c, SWR_ERR_UNSUPPORTED, "more than 2^20 tiles in one band (a target beyond 16384 x 16384): render it in tile-row bands (swr_set_band)");

    const size_t off_vb = (nd * sizeof(DrawParams) + 255) & ~(size_t)255;
    const size_t off_tb = (off_vb + vblocks.size() * sizeof(BlockMap) + 255) & ~(size_t)255;
    bool any_cull = false;
    for (auto& d : b.draws) any_cull = any_cull || d.frustum_cull;
    const size_t off_bp = (off_tb + tblocks.size() * sizeof(BlockMap) + 255) & ~(size_t)255;     // per-draw bounds pointers
    const size_t off_vis = (off_bp + (any_cull ? nd * sizeof(void*) : 0) + 255) & ~(size_t)255;   // per-draw visibility words
    const size_t up_bytes = off_vis + (any_cull ? nd * 4 : 0);
    if ((rc = ensure(c, S.d_upload, up_bytes))) return rc;
    if ((rc = ensure(c, S.d_vout, (size_t)(V + 4 * T) * sizeof(VOut)))) return rc;
    const bool dbgv = b_has_debug_varyings(b);
    if (dbgv && (rc = ensure(c, S.d_vnorm, S.d_vout.cap / 4))) return rc;        // one float4 per VOut entry
    if ((rc = ensure(c, S.d_recs, (size_t)(spt * T) * sizeof(TriRec)))) return rc;
    if ((rc = ensure(c, c->d_slot_tb, (size_t)(spt * T) * 8))) return rc;
    if ((rc = ensure(c, c->d_want, (size_t)(spt * T) + 64))) return rc;
    if ((rc = ensure(c, S.d_tile_count, (size_t)n_tiles * 4))) return rc;
    if ((rc = ensure(c, S.d_tile_start, (size_t)n_tiles * 4))) return rc;
    if ((rc = ensure(c, S.d_order, (size_t)n_tiles * 25 + 2 * SWR_ORDER_BUCKETS * 4))) return rc;
    if (c->tile_stats_tiles != n_tiles) {
        // another tile count (resize / band change): the fragment counters gathered so far move into the carry words of d_total
        // (swr_get_stats adds them), in stream order, so totals survive a change of geometry
        if (c->tile_stats_tiles) {
            hipLaunchKernelGGL(k_reduce_tile_stats, dim3(1), dim3(1024), 0, c->stream, c->d_tile_stats.as<uint32_t>(),
                               (uint32_t)c->tile_stats_tiles, c->d_total.as<unsigned long long>() + 4, 1);
            SWR_HIP(c, hipGetLastError());
        }
        if ((rc = ensure(c, c->d_tile_stats, (size_t)n_tiles * 12))) return rc;
        SWR_HIP(c, hipMemsetAsync(c->d_tile_stats.p, 0, (size_t)n_tiles * 12, c->stream));
        c->tile_stats_tiles = n_tiles;
    }
    // the set's buffers were last read by the raster kernel of two flushes ago
    if (piped && S.raster_pending) SWR_HIP(c, hipStreamWaitEvent(F, S.raster_done, 0));
    if (S.hist_tiles != n_tiles) {
        SWR_HIP(c, hipMemsetAsync(S.d_order.as<uint4>() + n_tiles, 0, (size_t)n_tiles * 4, F));      // tile_work: no fragment history for this tiling in this set
        S.hist_tiles = n_tiles;
    }
    char* stage = (char*)slot_acquire(c, up_bytes);
    if (!stage) return fail(c, SWR_ERR_OOM, "hipHostMalloc failed for the upload staging block");
    memcpy(stage, hp.data(), nd * sizeof(DrawParams));
    if (!vblocks.empty()) memcpy(stage + off_vb, vblocks.data(), vblocks.size() * sizeof(BlockMap));
    memcpy(stage + off_tb, tblocks.data(), tblocks.size() * sizeof(BlockMap));
    if (any_cull) {
        const float4** bp = reinterpret_cast<const float4**>(stage + off_bp);
        for (size_t i = 0; i < nd; ++i) bp[i] = b.draws[i].frustum_cull ? b.draws[i].mesh->d_bounds : nullptr;
    }
    SWR_HIP(c, hipMemcpyAsync(S.d_upload.p, stage, up_bytes, hipMemcpyHostToDevice, F));
    const DrawParams* d_draws = reinterpret_cast<const DrawParams*>(S.d_upload.p);
    const BlockMap* d_vblocks = reinterpret_cast<const BlockMap*>((char*)S.d_upload.p + off_vb);
    const BlockMap* d_tblocks = reinterpret_cast<const BlockMap*>((char*)S.d_upload.p + off_tb);

    const uint32_t* d_visible = nullptr;
    if (any_cull) {
        ScopedSpan sp(c, ST_VERTEX, F);
        uint32_t* vis = reinterpret_cast<uint32_t*>((char*)S.d_upload.p + off_vis);
        hipLaunchKernelGGL(k_frustum_cull, dim3((unsigned)((nd + 63) / 64)), dim3(64), 0, F, d_draws,
                           reinterpret_cast<const float4* const*>((char*)S.d_upload.p + off_bp), (uint32_t)nd, vis);
        SWR_HIP(c, hipGetLastError());
        d_visible = vis;
    }
    FrameParams fp = frame_params(c);
    fp.near_clip = b.near_clip;
    if (!vblocks.empty()) {
        ScopedSpan sp(c, ST_VERTEX, F);
        hipLaunchKernelGGL(k_vertex, dim3((unsigned)vblocks.size()), dim3(SWR_GEOM_BLOCK), 0, F,
                           d_draws, d_vblocks, S.d_vout.as<VOut>(), d_visible,
                           reinterpret_cast<float*>((char*)S.d_upload.p + offsetof(DrawParams, fog_r1)),
                           dbgv ? S.d_vnorm.as<float4>() : (float4*)nullptr, S.d_tile_count.as<uint32_t>(), n_tiles,
                           reinterpret_cast<uint32_t*>(S.d_order.as<uint4>() + n_tiles) + n_tiles);
        SWR_HIP(c, hipGetLastError());
    }
    const bool counts_clear = !vblocks.empty() && n_tiles != 0;      // k_vertex cleared the per-tile counters and the order histogram
    {
        ScopedSpan sp(c, ST_SETUP, F);
        hipLaunchKernelGGL(k_setup, dim3((unsigned)tblocks.size()), dim3(SWR_GEOM_BLOCK), 0, F,
                           d_draws, d_tblocks, (const VOut*)S.d_vout.as<VOut>(),
                           S.d_vout.as<VOut>() + V, (uint32_t)V, S.d_recs.as<TriRec>(),
                           c->d_slot_tb.as<unsigned long long>(), fp, c->d_counters.as<Counters>(),
                           (const Ctrl*)c->d_ctrl.as<Ctrl>(), b.seq, count_stats, b.wireframe ? 1 : 0, d_visible,
                           dbgv ? S.d_vnorm.as<float4>() : (float4*)nullptr);
        SWR_HIP(c, hipGetLastError());
    }
    rc = bin_and_raster(c, S, F, b, cc, cd, 0, (uint32_t)(spt * T), mode, counts_clear);
    slot_submit(c);
    if (rc) return rc;
    // a synchronous batch ran its front end on the context's stream: the next pipelined front end shares slot_tb / want / the lists
    // with it, so it may not start before this one is over (rare path: the first frame and replays)
    if (c->pipelining && mode != MODE_ASYNC) SWR_HIP(c, hipStreamSynchronize(c->stream));
    if (cc || cd) return run_clear(c, b, cc, cd, b.clear_rgba);   // nothing was binned
    return SWR_OK;
}

// complete = the batch's kernels are known to have finished (the stream was drained, or an event behind them has been waited for)
void retire_batch(swr_context* c, Batch& b, bool complete = false) {
    for (auto& d : b.draws)
        if (d.mesh && d.mesh->transient) { if (complete) pool_mesh(c, d.mesh); else c->garbage.push_back(d.mesh); }
    b.draws.clear();
}

// Counters::overflow (mirrored in the pinned word host_poison[1]): a list position beyond the capacity in a batch that
// could not poison itself (synchronous flush: the lists were sized from the COUNT pass, so COUNT and FILL disagreed).
// The dropped pair cannot be recovered there: report it instead of returning a silently wrong image.
int check_list_overflow(swr_context* c) {
    if (!c->host_poison || !((volatile uint32_t*)c->host_poison)[1]) return SWR_OK;
    c->host_poison[1] = 0;
    return fail(c, SWR_ERR_HIP, "internal error: a (triangle, tile) pair did not fit its tile list and was dropped (Counters::overflow)");
}

// stream must be idle: looks at the control block, replays what did not fit, retires the in-flight batches
int validate_locked(swr_context* c) {
    if (c->inflight.empty()) return check_list_overflow(c);
    int rc = SWR_OK;
    if (*(volatile uint32_t*)c->host_poison) {          // set by k_scan_apply / bin_overflow together with Ctrl::poison (stream is idle here)
        Ctrl h;
        SWR_HIP(c, hipMemcpy(&h, c->d_ctrl.p, sizeof h, hipMemcpyDeviceToHost));
        *c->host_poison = 0;
        c->host_poison[1] = 0;                              // an overflow inside an optimistic batch is cured by the replay below
        Ctrl fresh; fresh.poison = 0; fresh.first_bad = 0xffffffffu; fresh.need = 0; fresh.host_flag = c->host_poison;
        SWR_HIP(c, hipMemcpy(c->d_ctrl.p, &fresh, sizeof fresh, hipMemcpyHostToDevice));
        std::vector<Batch> todo;
        todo.swap(c->inflight);
        c->totals.flushes += 0;
        c->replays++;
        for (auto& b : todo) {
            if (!rc && b.seq >= h.first_bad) {
                rc = execute_batch(c, b, MODE_SYNC, b.seq != h.first_bad);   // the first bad batch already counted its triangles
                retire_batch(c, b);                                         // (replayed: its kernels are in flight again)
            } else {
                retire_batch(c, b, true);
            }
        }
        if (!rc) rc = drain_streams(c);
        if (!rc) rc = check_list_overflow(c);
        return rc;
    }
    for (auto& b : c->inflight) retire_batch(c, b, true);
    c->inflight.clear();
    return check_list_overflow(c);
}

int flush_locked(swr_context* c) {
    if (c->W <= 0 || c->H <= 0 || band_pixels(c) == 0) {          // Rasterizer.cs:176: silently skip
        for (auto& d : c->draws) if (d.mesh && d.mesh->transient) c->garbage.push_back(d.mesh);
        c->draws.clear(); c->pend_verts = c->pend_tris = 0;
        c->pend_clear_color = c->pend_clear_depth = false;
        return SWR_OK;
    }
    if (c->draws.empty() && !c->pend_clear_color && !c->pend_clear_depth) return SWR_OK;
    Batch b;
    b.draws.swap(c->draws);
    b.clear_color = c->pend_clear_color; b.clear_depth = c->pend_clear_depth;
    memcpy(b.clear_rgba, c->clear_rgba, 16);
    b.near_clip = c->near_clip;
    b.wireframe = c->debug_mode == SWR_DEBUG_WIREFRAME;
    if (b.wireframe && b_has_debug_varyings(b)) {
        // DrawLine hands Interpolate the TRIANGLE's outputs[0..1] for all three edges (Rasterizer.cs:421-423); a line record keeps the
        // edge's end points, not those two vertices' screen positions, so the program's ScreenCoords term is not available there
        for (auto& d : b.draws) if (d.mesh && d.mesh->transient) c->garbage.push_back(d.mesh);
        c->pend_verts = c->pend_tris = 0;
        return fail(c, SWR_ERR_UNSUPPORTED, "SWR_PROG_DEBUG_VARYINGS is not available in DebugMode.Wireframe");
    }
    b.seq = c->next_seq++;
    b.color = c->color; b.depth = c->depth;
    c->pend_clear_color = c->pend_clear_depth = false;
    c->pend_verts = c->pend_tris = 0;
    if (!b.draws.empty()) c->totals.flushes++;
    // optimistic once an earlier (synchronous) batch has sized the pair buffers
    const bool optimistic = !c->sync_flush && pair_capacity(c) > 0;
    int rc = execute_batch(c, b, optimistic ? MODE_ASYNC : MODE_SYNC, 1);
    if (optimistic && !rc) {
        c->inflight.push_back(std::move(b));
        if (c->inflight.size() >= 64) {                            // bound the replay log
            if ((rc = drain_streams(c))) return rc;
            rc = validate_locked(c);
        }
    } else {
        retire_batch(c, b);
    }
    return rc;
}

int ensure_bounds(swr_context* c, swr_mesh* m) {
    if (m->bounds_ready) return SWR_OK;
    if (!m->d_bounds) SWR_HIP(c, hipMalloc((void**)&m->d_bounds, sizeof(float4)));
    // (on the stream the mesh was uploaded on and k_frustum_cull will read the result on)
    hipLaunchKernelGGL(k_bounding_sphere, dim3(1), dim3(1024), 0, use_front_stream(c), (const swr_vertex*)m->d_verts, (uint32_t)m->n_verts, m->d_bounds);
    SWR_HIP(c, hipGetLastError());
    m->bounds_ready = true;
    return SWR_OK;
}

// Multi-GPU bands: a mesh whose exact bounding box projects entirely above or below this context's band of tile
// rows produces no fragment here, so the draw is not recorded at all (every rank would otherwise run the vertex, setup
// and count stages for all triangles).  Conservative: the corners are projected in double arithmetic and must all be in
// front of the camera plane (w > 0: then every point of the box projects inside the hull of the projected corners); the
// margin covers what the DEVICE's float32 evaluation of a vertex can differ from that: the three chained row-vector
// products of k_vertex (Renderer.cs:832-834) round 4 products + 3 sums per component and stage, so
// |fl(clip_k) - clip_k| <= 16 * 2^-24 * (|p| |model| |view| |proj|)_k (the running-error bound with absolute values,
// maximal at a corner of the box), which moves the screen row by at most H/2 * (E_y + |ndc_y| E_w) / (w - E_w); plus
// two pixels for the viewport arithmetic and the floor / ceil of the pixel bbox.  With large cancelling translations
// that bound exceeds any fixed margin -- then the draw is simply kept.
static bool band_rejects(const swr_context* c, const swr_mesh* m, const float* model, const float* view, const float* proj) {
    if (!m->has_box || c->il_k > 0 || (c->band_ty0 <= 0 && c->band_ty1 >= c->tiles_y)) return false;    // stripes: every rank sees the whole frame
    double mv[16], M[16], amv[16], A[16];
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 4; ++k) {
            double a = 0, b = 0;
            for (int j = 0; j < 4; ++j) { a += (double)model[r * 4 + j] * (double)view[j * 4 + k]; b += std::fabs((double)model[r * 4 + j]) * std::fabs((double)view[j * 4 + k]); }
            mv[r * 4 + k] = a; amv[r * 4 + k] = b;
        }
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 4; ++k) {
            double a = 0, b = 0;
            for (int j = 0; j < 4; ++j) { a += mv[r * 4 + j] * (double)proj[j * 4 + k]; b += amv[r * 4 + j] * std::fabs((double)proj[j * 4 + k]); }
            M[r * 4 + k] = a; A[r * 4 + k] = b;
        }
    const double u16 = 16.0 / 16777216.0;
    double smin = 1e300, smax = -1e300, ey = 0, ew = 0, wmin = 1e300, ndc_abs = 0;
    for (int corner = 0; corner < 8; ++corner) {
        const double x = (corner & 1) ? m->box_hi[0] : m->box_lo[0];
        const double y = (corner & 2) ? m->box_hi[1] : m->box_lo[1];
        const double z = (corner & 4) ? m->box_hi[2] : m->box_lo[2];
        const double cy = x * M[1] + y * M[5] + z * M[9] + M[13];
        const double cw = x * M[3] + y * M[7] + z * M[11] + M[15];
        const double sy_abs = std::fabs(x) * A[1] + std::fabs(y) * A[5] + std::fabs(z) * A[9] + A[13];
        const double sw_abs = std::fabs(x) * A[3] + std::fabs(y) * A[7] + std::fabs(z) * A[11] + A[15];
        if (!std::isfinite(cy) || !std::isfinite(cw) || !std::isfinite(sy_abs) || !std::isfinite(sw_abs)) return false;
        ey = std::max(ey, u16 * sy_abs); ew = std::max(ew, u16 * sw_abs);
        if (!(cw > 0)) return false;                                                             // not in front: keep
        wmin = std::min(wmin, cw);
        const double ndc = cy / cw;
        ndc_abs = std::max(ndc_abs, std::fabs(ndc));
        const double sy = (1.0 - (ndc * 0.5 + 0.5)) * (double)c->H;                              // Rasterizer.cs:385-386
        if (!std::isfinite(sy)) return false;
        smin = std::min(smin, sy); smax = std::max(smax, sy);
    }
    if (!(wmin - ew > 1e-6 * wmin)) return false;                       // the float32 w of some vertex may not be safely positive: keep
    const double margin = 2.0 + 0.5 * (double)c->H * (ey + ndc_abs * ew) / (wmin - ew);
    if (!std::isfinite(margin)) return false;
    const double y0 = (double)band_y0(c), y1 = (double)std::min(c->H, c->band_ty1 * SWR_TILE);   // band = pixel rows [y0, y1)
    return smax + margin < y0 || smin - margin > y1;
}

int record_draw(swr_context* c, swr_mesh* mesh, const float* model, const float* view, const float* proj,
                int program, const swr_uniforms* u, const swr_texture* tex, int cull, int depth_test, int blend, bool frustum_cull = false) {
    if (!mesh || !model || !view || !proj) return fail(c, SWR_ERR_INVALID_ARG, "null argument to render_mesh");
    if (program < SWR_PROG_FLAT_COLOR || program > SWR_PROG_DEBUG_VARYINGS)
        return fail(c, SWR_ERR_INVALID_ARG, "unknown program id");
    if ((program == SWR_PROG_DUST2_LAMBERT_FOG || program == SWR_PROG_PHONG_4POINT) && !u)
        return fail(c, SWR_ERR_INVALID_ARG, "this program needs a uniform block");
    if (cull < 0 || cull > 2 || depth_test < 0 || depth_test > 7 || blend < 0 || blend > 3)
        return fail(c, SWR_ERR_INVALID_ARG, "enum value out of range");
    if (c->W <= 0 || c->H <= 0) return SWR_OK;                      // Rasterizer.cs:176
    const int n_tris = mesh->n_idx / 3;                             // Rasterizer.cs:180
    if (n_tris == 0) return SWR_OK;
    if (band_rejects(c, mesh, model, view, proj)) return SWR_OK;      // nothing of it can land in this band
    // one batch holds fewer than 2^26 vertex-stage records (one per vertex + four per triangle for the clipper: k_raster_c addresses
    // them with 32-bit byte offsets); a batch of several draws is flushed in time below, a SINGLE draw beyond it cannot be helped
    if ((uint64_t)mesh->n_verts + 4 * (uint64_t)n_tris >= (1ull << 26))
        return fail(c, SWR_ERR_UNSUPPORTED, "mesh exceeds 2^26 vertex-stage records (vertices + 4 x triangles: about 16.7 M triangles): split it");
    // keep a batch within the 32-bit slot / vertex numbering
    // (and the vertex-stage output -- one record per vertex plus four per triangle for the clipper -- below 4 GiB)
    if (c->pend_verts + (uint64_t)mesh->n_verts + 4 * (c->pend_tris + (uint64_t)n_tris) >= (1ull << 26)) {
        int rc = flush_locked(c);
        if (rc) return rc;
    }
    // SWR_PROG_DEBUG_VARYINGS has kernels of its own: a batch holds either only such draws or none (submission order is kept)
    if (!c->draws.empty() && (c->draws.back().p.program == SWR_PROG_DEBUG_VARYINGS) != (program == SWR_PROG_DEBUG_VARYINGS)) {
        int rc = flush_locked(c);
        if (rc) return rc;
    }
    DrawCmd d;
    memset(&d.p, 0, sizeof d.p);
    memcpy(d.p.model, model, 64); memcpy(d.p.view, view, 64); memcpy(d.p.proj, proj, 64);
    if (u) d.p.u = *u;
    d.p.verts = mesh->d_verts; d.p.idx = mesh->d_idx;
    d.p.tex = tex ? tex->d_rgba : nullptr;
    d.p.tex_w = tex ? tex->w : 0; d.p.tex_h = tex ? (tex->bilinear ? -tex->h : tex->h) : 0;
    if (tex && tex->bilinear && tex->d_blocked) { d.p.tex = tex->d_blocked; d.p.tex_w = -tex->w; }      // (texture_fetch in swr_device.h)
    d.p.tex_wf = tex ? (float)tex->w : 0.0f; d.p.tex_hf = tex ? (float)tex->h : 0.0f;
    d.p.program = program; d.p.cull = cull; d.p.depth_test = depth_test; d.p.blend = blend;
    d.p.n_verts = (uint32_t)mesh->n_verts; d.p.n_tris = (uint32_t)n_tris;
    d.p.nm_flags = c->nm_flags;
    d.mesh = mesh;
    d.frustum_cull = frustum_cull;
    if (frustum_cull) { int rc = ensure_bounds(c, mesh); if (rc) return rc; }
    c->draws.push_back(d);
    c->pend_verts += mesh->n_verts; c->pend_tris += n_tris;
    return SWR_OK;
}

int make_mesh(swr_context* c, const swr_vertex* v, int nv, const uint16_t* idx, int ni, bool transient, swr_mesh** out) {
    if (nv < 0 || ni < 0 || (nv > 0 && !v) || (ni > 0 && !idx) || !out) return fail(c, SWR_ERR_INVALID_ARG, "bad mesh arguments");
    const int used = (ni / 3) * 3;
    for (int i = 0; i < used; ++i)
        if ((int)idx[i] >= nv) return fail(c, SWR_ERR_INVALID_ARG, "index out of range (C#: IndexOutOfRangeException)");
    swr_mesh* m = nullptr;
    if (transient) {                               // smallest pooled mesh whose buffers hold these arrays
        size_t best = (size_t)-1;
        for (size_t i = 0; i < c->mesh_pool.size(); ++i) {
            swr_mesh* q = c->mesh_pool[i];
            if (q->cap_verts >= (size_t)nv * sizeof(swr_vertex) && q->cap_idx >= (size_t)ni * 2 + 8 &&
                (best == (size_t)-1 || q->cap_verts + q->cap_idx < c->mesh_pool[best]->cap_verts + c->mesh_pool[best]->cap_idx)) best = i;
        }
        if (best != (size_t)-1) {
            m = c->mesh_pool[best];
            c->mesh_pool.erase(c->mesh_pool.begin() + (std::ptrdiff_t)best);
            c->mesh_pool_bytes -= m->cap_verts + m->cap_idx;
            m->bounds_ready = false; m->has_box = false;
        }
    }
    if (!m) m = new swr_mesh();
    m->n_verts = nv; m->n_idx = ni; m->transient = transient;
    if (nv > 0) {                                  // exact AABB (min / max only): input to band_rejects
        bool ok = true;
        for (int k = 0; k < 3; ++k) { m->box_lo[k] = v[0].position[k]; m->box_hi[k] = v[0].position[k]; }
        for (int i = 0; i < nv; ++i)
            for (int k = 0; k < 3; ++k) {
                const float p = v[i].position[k];
                if (!(p == p)) ok = false;
                m->box_lo[k] = std::min(m->box_lo[k], p); m->box_hi[k] = std::max(m->box_hi[k], p);
            }
        m->has_box = ok;
    }
    hipError_t e = hipSuccess;
    if (nv && m->cap_verts < (size_t)nv * sizeof(swr_vertex)) {
        e = hipMalloc((void**)&m->d_verts, (size_t)nv * sizeof(swr_vertex));
        if (e == hipSuccess) m->cap_verts = (size_t)nv * sizeof(swr_vertex);
    }
    if (e == hipSuccess && ni && m->cap_idx < (size_t)ni * 2 + 8) {
        e = hipMalloc((void**)&m->d_idx, (size_t)ni * 2 + 8);
        if (e == hipSuccess) m->cap_idx = (size_t)ni * 2 + 8;
    }
    // (only front-end kernels read a mesh: the upload goes to their stream, so a pipelined frame does not wait for the raster stream)
    if (e == hipSuccess && nv) e = hipMemcpyAsync(m->d_verts, v, (size_t)nv * sizeof(swr_vertex), hipMemcpyHostToDevice, use_front_stream(c));
    if (e == hipSuccess && ni) e = hipMemcpyAsync(m->d_idx, idx, (size_t)ni * 2, hipMemcpyHostToDevice, use_front_stream(c));
    if (e != hipSuccess) {
        destroy_mesh(m);
        c->err = std::string("mesh upload failed: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? SWR_ERR_OOM : SWR_ERR_HIP;
    }
    *out = m;
    return SWR_OK;
}

// Were the pixels of present slot `slot` rendered by a batch that has to be replayed?  Only if a batch flushed at or before that
// present poisoned itself: Ctrl::first_bad (one small blocking copy, and only when the pinned poison word is up at all).
int present_is_stale(swr_context* c, int slot, bool& stale) {
    stale = false;
    if (!*(volatile uint32_t*)c->host_poison) return SWR_OK;
    Ctrl h;
    SWR_HIP(c, hipMemcpy(&h, c->d_ctrl.p, sizeof h, hipMemcpyDeviceToHost));
    stale = h.first_bad <= c->present_seq[slot];
    return SWR_OK;
}

bool in_band(const swr_context* c, int x, int y) {
    return x >= 0 && x < c->W && band_local_pixel_row(c, y) >= 0;
}

}  // namespace

extern "C" {

int swr_abi_version(void) { return SWR_ABI_VERSION; }

// set by csrc/Makefile: the compiler's version line and bench.py --print-src-hash (sha256 over the kernel sources)
#ifndef SWR_BUILD_HIPCC
#define SWR_BUILD_HIPCC "unknown"
#endif
#ifndef SWR_BUILD_SRC_SHA
#define SWR_BUILD_SRC_SHA "unknown"
#endif
#ifndef SWR_BUILD_EXTRA
#define SWR_BUILD_EXTRA ""
#endif
#define SWR_STR2(x) #x
#define SWR_STR(x) SWR_STR2(x)
const char* swr_build_info(void) {
    return "hipcc=" SWR_BUILD_HIPCC "; csrc_sha256=" SWR_BUILD_SRC_SHA "; fma=" SWR_STR(SWR_NUMERICS_FMA) "; dot=" SWR_STR(SWR_DOT_PAIRWISE) "; extra=" SWR_BUILD_EXTRA;
}

int swr_numerics_mode(int* fma, int* dot_order) {
    if (!fma || !dot_order) return SWR_ERR_INVALID_ARG;
    *fma = SWR_NUMERICS_FMA; *dot_order = SWR_DOT_PAIRWISE;
    return SWR_OK;
}

int swr_set_transform_fma(swr_context* c, int transform_fused, int transform_normal_fused) {
    if (!c) return SWR_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock_(c->mu);
    // (draws recorded so far keep the model they were recorded under: DrawParams::nm_flags)
    c->nm_flags = (transform_fused ? SWR_NM_TRANSFORM_FMA : 0u) | (transform_normal_fused ? SWR_NM_TRANSFORM_NORMAL_FMA : 0u);
    return SWR_OK;
}

int swr_get_transform_fma(swr_context* c, int* transform_fused, int* transform_normal_fused) {
    if (!c || !transform_fused || !transform_normal_fused) return SWR_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock_(c->mu);
    *transform_fused = (c->nm_flags & SWR_NM_TRANSFORM_FMA) ? 1 : 0;
    *transform_normal_fused = (c->nm_flags & SWR_NM_TRANSFORM_NORMAL_FMA) ? 1 : 0;
    return SWR_OK;
}

const char* swr_last_error(const swr_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int swr_create(int device_id, swr_context** out) {
    if (!out) { g_create_error = "out is null"; return SWR_ERR_INVALID_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e) + " (this backend has no CPU fallback)";
        return SWR_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) { g_create_error = "device id out of range"; return SWR_ERR_INVALID_ARG; }
    if ((e = hipSetDevice(device_id)) != hipSuccess) { g_create_error = hipGetErrorString(e); return SWR_ERR_HIP; }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) { g_create_error = hipGetErrorString(e); return SWR_ERR_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + "; libswr_hip.so carries gfx950 code objects only";
        return SWR_ERR_NO_DEVICE;
    }
    swr_context* c = new swr_context();
    c->device = device_id;
    snprintf(c->dev_name, sizeof c->dev_name, "%s (%s)", prop.name, prop.gcnArchName);
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = hipGetErrorString(e); delete c; return SWR_ERR_HIP;
    }
    c->stream = c->own_stream;
    if (ensure_front_stream(c)) { g_create_error = c->err; (void)hipStreamDestroy(c->own_stream); delete c; return SWR_ERR_HIP; }
    { const char* sf = getenv("SWR_SYNC_FLUSH"); c->sync_flush = sf && sf[0] == '1'; }
#ifdef SWR_TEST_HOOKS       // libswr_hip_test.so only: the FILL pass of optimistic flushes sees a list of n entries (forces the list-overflow path)
    { const char* df = getenv("SWR_DEBUG_FILL_CAPACITY"); c->debug_fill_capacity = df ? (uint32_t)strtoul(df, nullptr, 10) : 0u; }
#endif
    int rc = ensure(c, c->d_counters, 65 * sizeof(Counters));
    if (!rc) rc = ensure(c, c->d_ctrl, 64);
    if (!rc && hipHostMalloc((void**)&c->host_poison, 64, hipHostMallocDefault) != hipSuccess) rc = SWR_ERR_OOM;
    if (!rc) { memset(c->host_poison, 0, 64); Ctrl fresh; fresh.poison = 0; fresh.first_bad = 0xffffffffu; fresh.need = 0; fresh.host_flag = c->host_poison; if (hipMemcpy(c->d_ctrl.p, &fresh, sizeof fresh, hipMemcpyHostToDevice) != hipSuccess) rc = SWR_ERR_HIP; }
    if (!rc) rc = ensure(c, c->d_total, 256 + 4096 * 8);
    if (!rc && hipMemsetAsync(c->d_total.p, 0, 256 + 4096 * 8, c->stream) != hipSuccess) rc = SWR_ERR_HIP;
    if (!rc && hipMemsetAsync(c->d_counters.p, 0, 65 * sizeof(Counters), c->stream) != hipSuccess) rc = SWR_ERR_HIP;
    if (rc) { g_create_error = c->err; swr_destroy(c); return rc; }
    *out = c;
    return SWR_OK;
}

void swr_destroy(swr_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->front_stream) (void)hipStreamSynchronize(c->front_stream);
    for (auto& d : c->draws) if (d.mesh && d.mesh->transient) c->garbage.push_back(d.mesh);
    for (auto& b : c->inflight) retire_batch(c, b);
    c->inflight.clear();
    collect_spans(c);
    free_garbage(c);
    for (swr_mesh* m : c->mesh_pool) destroy_mesh(m);
    c->mesh_pool.clear();
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    for (auto& fs : c->slots) { if (fs.host) (void)hipHostFree(fs.host); if (fs.done) (void)hipEventDestroy(fs.done); }
    if (c->host_poison) (void)hipHostFree(c->host_poison);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (int i = 0; i < 2; ++i) {
        if (c->present_flat[i]) (void)hipEventDestroy(c->present_flat[i]);
        if (c->present_done[i]) (void)hipEventDestroy(c->present_done[i]);
        release(c->present_buf[i]);
    }
    DevBuf* bufs[] = { &c->own_color, &c->own_depth, &c->d_slot_tb, &c->d_want, &c->d_ctrl, &c->d_pair_tile, &c->d_tile_list, &c->d_tile_stats,
                       &c->d_counters, &c->d_total, &c->d_scratch };
    for (DevBuf* b : bufs) release(*b);
    for (auto& s : c->sets) {
        DevBuf* sb[] = { &s.d_upload, &s.d_vout, &s.d_vnorm, &s.d_recs, &s.d_masks, &s.d_pcounts, &s.d_pair_refs, &s.d_tile_count, &s.d_tile_start, &s.d_order };
        for (DevBuf* b : sb) release(*b);
        if (s.front_done) (void)hipEventDestroy(s.front_done);
        if (s.raster_done) (void)hipEventDestroy(s.raster_done);
    }
    if (c->front_stream) (void)hipStreamDestroy(c->front_stream);
    if (c->f_tail_ev) (void)hipEventDestroy(c->f_tail_ev);
    if (c->r_front_ev) (void)hipEventDestroy(c->r_front_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

#define SWR_ENTER(c)                                   \
    if (!(c)) return SWR_ERR_INVALID_ARG;              \
    std::lock_guard<std::mutex> lock_((c)->mu);        \
    (void)hipSetDevice((c)->device)

int swr_resize(swr_context* c, int width, int height) {
    SWR_ENTER(c);
    if (width > 65535 || height > 65535) return fail(c, SWR_ERR_INVALID_ARG, "render target larger than 65535");
    if (c->geometry_applied && width == c->W && height == c->H) return SWR_OK;      // nothing changes: no flush, no wait
    int rc = flush_locked(c);
    if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    c->W = width; c->H = height;
    c->geometry_applied = true;
    return apply_geometry(c);
}

int swr_set_band(swr_context* c, int first_tile_row, int n_tile_rows) {
    SWR_ENTER(c);
    {
        const bool want_band = !(first_tile_row < 0 || n_tile_rows < 0);
        if (c->il_k == 0 && want_band == c->band_set && (!want_band || (first_tile_row == c->band_first && n_tile_rows == c->band_count)))
            return SWR_OK;                                                         // the same band again: no flush, no wait
    }
    int rc = flush_locked(c);
    if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    c->il_k = 0; c->il_world = 1; c->il_rank = 0;
    if (first_tile_row < 0 || n_tile_rows < 0) { c->band_set = false; }
    else { c->band_set = true; c->band_first = first_tile_row; c->band_count = n_tile_rows; }
    return apply_geometry(c);
}

int swr_set_band_interleaved(swr_context* c, int rank, int world, int stripe_tile_rows) {
    SWR_ENTER(c);
    if (world < 1 || rank < 0 || rank >= world || stripe_tile_rows < 1) return fail(c, SWR_ERR_INVALID_ARG, "bad interleaved band arguments");
    if (c->il_k == stripe_tile_rows && c->il_world == world && c->il_rank == rank) return SWR_OK;     // unchanged: no flush, no wait
    int rc = flush_locked(c);
    if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    c->band_set = false;
    c->il_k = stripe_tile_rows; c->il_world = world; c->il_rank = rank;
    return apply_geometry(c);
}

int swr_bind_framebuffer(swr_context* c, void* color, void* depth) {
    SWR_ENTER(c);
    if ((color == nullptr) != (depth == nullptr)) return fail(c, SWR_ERR_INVALID_ARG, "bind both colour and depth, or neither");
    // draws recorded so far belong to the buffers bound so far: launch them (asynchronously -- a batch remembers its
    // target, so neither the switch nor a later replay needs the stream to drain; a frame loop that alternates two band
    // buffers pays no host round trip here)
    int rc = flush_locked(c);
    if (rc) return rc;
    // back to internal storage: apply_geometry may have to (re)allocate it, which must not happen under batches that could
    // still be replayed against the old allocation -- drain and validate first (binding caller memory never allocates)
    if (!color && (rc = sync_locked(c))) return rc;
    c->ext_color = color; c->ext_depth = depth;
    return apply_geometry(c);
}

int swr_set_stream(swr_context* c, void* s) {
    SWR_ENTER(c);
    int rc = flush_locked(c);
    if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return SWR_OK;
}

int swr_clear_color(swr_context* c, const float rgba[4]) {
    SWR_ENTER(c);
    if (!rgba) return fail(c, SWR_ERR_INVALID_ARG, "rgba is null");
    if (!c->draws.empty()) { int rc = flush_locked(c); if (rc) return rc; }
    memcpy(c->clear_rgba, rgba, 16);
    c->pend_clear_color = true;
    return SWR_OK;
}

int swr_clear_depth(swr_context* c) {
    SWR_ENTER(c);
    if (!c->draws.empty()) { int rc = flush_locked(c); if (rc) return rc; }
    c->pend_clear_depth = true;
    return SWR_OK;
}

int swr_flush(swr_context* c) { SWR_ENTER(c); return flush_locked(c); }

int swr_set_pipelining(swr_context* c, int mode) {
    SWR_ENTER(c);
    if (mode < 0 || mode > 2) return fail(c, SWR_ERR_INVALID_ARG, "pipelining mode must be 0 (off), 1 (every batch) or 2 (only frames of up to 2^15 tiles or batches of up to 2^17 triangles)");
    if (mode == c->pipelining) return SWR_OK;
    int rc = flush_locked(c);
    if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;          // nothing in flight while the streams and the set in use change
    c->pipelining = mode;
    return ensure_front_stream(c);
}

int swr_get_pipelining(swr_context* c, int* mode) {
    SWR_ENTER(c);
    if (!mode) return SWR_ERR_INVALID_ARG;
    *mode = c->pipelining;
    return SWR_OK;
}

int swr_sync(swr_context* c) {
    SWR_ENTER(c);
    int rc = flush_locked(c);
    if (rc) return rc;
    return sync_locked(c);
}

int swr_readback(swr_context* c, float* color, float* depth) {
    SWR_ENTER(c);
    int rc = flush_locked(c);
    if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;          // validates optimistic batches BEFORE pixels are observed
    size_t n = band_pixels(c);
    if (n) {
        if (color) SWR_HIP(c, hipMemcpyAsync(color, c->color, n * 16, hipMemcpyDeviceToHost, c->stream));
        if (depth) SWR_HIP(c, hipMemcpyAsync(depth, c->depth, n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    return sync_locked(c);
}

int swr_readback_rgb(swr_context* c, float* rgb) {
    SWR_ENTER(c);
    if (!rgb) return fail(c, SWR_ERR_INVALID_ARG, "rgb is null");
    int rc = flush_locked(c);
    if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    size_t n = band_pixels(c);
    if (!n) return SWR_OK;
    if ((rc = ensure(c, c->d_scratch, n * 12))) return rc;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 2048 * 8);
    hipLaunchKernelGGL(k_flatten_rgb, dim3(blocks), dim3(256), 0, c->stream, (const float4*)c->color, c->d_scratch.as<float>(), n);
    SWR_HIP(c, hipGetLastError());
    SWR_HIP(c, hipMemcpyAsync(rgb, c->d_scratch.p, n * 12, hipMemcpyDeviceToHost, c->stream));
    return sync_locked(c);
}

int swr_present_rgb_async(swr_context* c, float* rgb, uint64_t* ticket) {
    SWR_ENTER(c);
    if (!rgb || !ticket) return fail(c, SWR_ERR_INVALID_ARG, "rgb or ticket is null");
    int rc = flush_locked(c);
    if (rc) return rc;
    const size_t n = band_pixels(c);
    const int slot = (int)(c->next_ticket & 1ull);
    if (!c->copy_stream) SWR_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!c->present_flat[i]) SWR_HIP(c, hipEventCreateWithFlags(&c->present_flat[i], hipEventDisableTiming));
        if (!c->present_done[i]) SWR_HIP(c, hipEventCreateWithFlags(&c->present_done[i], hipEventDisableTiming));
    }
    // the slot's previous copy (two presents ago) must be over before its staging buffer is overwritten: back-pressure, not
    // steady-state waiting -- a caller that waits for ticket i before it presents frame i + 2 never blocks here
    if (c->present_ticket[slot]) {
        SWR_HIP(c, hipEventSynchronize(c->present_done[slot]));
        // the ticket is dropped unwaited: if a batch at or before that present poisoned itself, its pixels were stale -- remember it
        // for the swr_present_wait that may still come
        bool stale = false;
        if ((rc = present_is_stale(c, slot, stale))) return rc;
        if (stale) c->stale_dropped[slot] = c->present_ticket[slot];
        c->present_ticket[slot] = 0;
    }
    if (n) {
        if (c->present_buf[slot].cap < n * 12) {
            // growing frees the old block: make sure neither stream still uses it (first frame / after a resize only)
            SWR_HIP(c, hipStreamSynchronize(c->copy_stream));
            if ((rc = ensure(c, c->present_buf[slot], n * 12))) return rc;
        }
        const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048 * 8);
        hipLaunchKernelGGL(k_flatten_rgb, dim3(blocks), dim3(256), 0, c->stream, (const float4*)c->color, c->present_buf[slot].as<float>(), n);
        SWR_HIP(c, hipGetLastError());
    }
    SWR_HIP(c, hipEventRecord(c->present_flat[slot], c->stream));
    SWR_HIP(c, hipStreamWaitEvent(c->copy_stream, c->present_flat[slot], 0));
    if (n) SWR_HIP(c, hipMemcpyAsync(rgb, c->present_buf[slot].p, n * 12, hipMemcpyDeviceToHost, c->copy_stream));
    SWR_HIP(c, hipEventRecord(c->present_done[slot], c->copy_stream));
    c->present_ticket[slot] = ++c->next_ticket;
    c->present_seq[slot] = c->next_seq - 1u;
    *ticket = c->present_ticket[slot];
    return SWR_OK;
}

int swr_present_wait(swr_context* c, uint64_t ticket) {
    SWR_ENTER(c);
    int slot = -1;
    for (int i = 0; i < 2; ++i) if (c->present_ticket[i] == ticket && ticket) slot = i;
    if (slot < 0) {
        if (!ticket || ticket > c->next_ticket) return fail(c, SWR_ERR_INVALID_ARG, "unknown present ticket");
        for (int i = 0; i < 2; ++i)
            if (c->stale_dropped[i] == ticket) { c->stale_dropped[i] = 0; return SWR_STALE; }     // dropped by back-pressure, and it was stale
        return SWR_OK;                                                                             // already waited for
    }
    SWR_HIP(c, hipEventSynchronize(c->present_done[slot]));
    c->present_ticket[slot] = 0;
    bool stale = false;
    int rc = present_is_stale(c, slot, stale);
    if (rc) return rc;
    if (stale) {
        // a batch flushed at or before this present did not fit its pair buffers and poisoned itself (and everything after it): the
        // copied pixels predate it.  Drain, replay (validate_locked) and tell the caller to present again.
        rc = sync_locked(c);
        return rc ? rc : SWR_STALE;
    }
    // Every batch flushed before this present has completed (the flatten ran behind them): retire them without draining the stream,
    // and without freeing anything -- their transient meshes go to the pool (hipFree would wait for frame i + 1, which is rendering).
    // (A batch poisoned AFTER this present leaves this frame good: it is replayed at the next synchronisation point.)
    const uint32_t upto = c->present_seq[slot];
    size_t k = 0;
    while (k < c->inflight.size() && c->inflight[k].seq <= upto) { retire_batch(c, c->inflight[k], true); ++k; }
    c->inflight.erase(c->inflight.begin(), c->inflight.begin() + (std::ptrdiff_t)k);
    return SWR_OK;
}

int swr_host_register(swr_context* c, void* ptr, size_t bytes) {
    SWR_ENTER(c);
    if (!ptr || !bytes) return fail(c, SWR_ERR_INVALID_ARG, "null or empty host buffer");
    SWR_HIP(c, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return SWR_OK;
}

int swr_host_unregister(swr_context* c, void* ptr) {
    SWR_ENTER(c);
    if (!ptr) return fail(c, SWR_ERR_INVALID_ARG, "null host buffer");
    int rc = flush_locked(c);                       // nothing in flight may still target the buffer
    if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    SWR_HIP(c, hipHostUnregister(ptr));
    return SWR_OK;
}

int swr_flatten_rgb_device(swr_context* c, float* d_rgb) {
    SWR_ENTER(c);
    if (!d_rgb) return fail(c, SWR_ERR_INVALID_ARG, "d_rgb is null");
    int rc = flush_locked(c);
    if (rc) return rc;
    size_t n = band_pixels(c);
    if (!n) return SWR_OK;
    // stream order puts this after the frame's kernels; a batch that has to be replayed (optimistic flush) is replayed
    // by the caller's swr_sync BEFORE the result is consumed -- and then the flatten must run again, which sync does
    // not know about: so validate first (cheap when nothing overflowed: one pinned-flag read after the stream drains)
    if ((rc = sync_locked(c))) return rc;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 2048 * 8);
    hipLaunchKernelGGL(k_flatten_rgb, dim3(blocks), dim3(256), 0, c->stream, (const float4*)c->color, d_rgb, n);
    SWR_HIP(c, hipGetLastError());
    return SWR_OK;
}

int swr_flatten_rgb_device_async(swr_context* c, float* d_rgb) {
    SWR_ENTER(c);
    if (!d_rgb) return fail(c, SWR_ERR_INVALID_ARG, "d_rgb is null");
    int rc = flush_locked(c);
    if (rc) return rc;
    size_t n = band_pixels(c);
    if (!n) return SWR_OK;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 2048 * 8);
    hipLaunchKernelGGL(k_flatten_rgb, dim3(blocks), dim3(256), 0, c->stream, (const float4*)c->color, d_rgb, n);
    SWR_HIP(c, hipGetLastError());
    return SWR_OK;
}

int swr_replay_count(swr_context* c, uint64_t* out) {
    SWR_ENTER(c);
    if (!out) return SWR_ERR_INVALID_ARG;
    *out = c->replays;
    return SWR_OK;
}

int swr_sync_count(swr_context* c, uint64_t* out) {
    SWR_ENTER(c);
    if (!out) return SWR_ERR_INVALID_ARG;
    *out = c->host_syncs;
    return SWR_OK;
}

int swr_upload(swr_context* c, const float* color, const float* depth) {
    SWR_ENTER(c);
    int rc = flush_locked(c);
    if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    size_t n = band_pixels(c);
    if (n) {
        if (color) SWR_HIP(c, hipMemcpyAsync(c->color, color, n * 16, hipMemcpyHostToDevice, c->stream));
        if (depth) SWR_HIP(c, hipMemcpyAsync(c->depth, depth, n * 4, hipMemcpyHostToDevice, c->stream));
    }
    return sync_locked(c);
}

int swr_color_device_ptr(swr_context* c, void** out) { SWR_ENTER(c); if (!out) return SWR_ERR_INVALID_ARG; *out = c->color; return SWR_OK; }
int swr_depth_device_ptr(swr_context* c, void** out) { SWR_ENTER(c); if (!out) return SWR_ERR_INVALID_ARG; *out = c->depth; return SWR_OK; }

int swr_get_pixel(swr_context* c, int x, int y, float rgba[4]) {
    SWR_ENTER(c);
    if (!rgba) return SWR_ERR_INVALID_ARG;
    rgba[0] = rgba[1] = rgba[2] = rgba[3] = 0.0f;                 // Vector4.Zero out of bounds, MainWindow.cs:397
    if (!in_band(c, x, y)) return SWR_OK;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    SWR_HIP(c, hipMemcpyAsync(rgba, c->color + (size_t)band_local_pixel_row(c, y) * c->W + x, 16, hipMemcpyDeviceToHost, c->stream));
    return sync_locked(c);
}
int swr_set_pixel(swr_context* c, int x, int y, const float rgba[4]) {
    SWR_ENTER(c);
    if (!rgba) return SWR_ERR_INVALID_ARG;
    if (!in_band(c, x, y)) return SWR_OK;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    SWR_HIP(c, hipMemcpyAsync(c->color + (size_t)band_local_pixel_row(c, y) * c->W + x, rgba, 16, hipMemcpyHostToDevice, c->stream));
    return sync_locked(c);
}
int swr_get_depth(swr_context* c, int x, int y, float* d) {
    SWR_ENTER(c);
    if (!d) return SWR_ERR_INVALID_ARG;
    *d = SWR_FLOAT_MINVALUE;                                       // MainWindow.cs:425
    if (!in_band(c, x, y)) return SWR_OK;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    SWR_HIP(c, hipMemcpyAsync(d, c->depth + (size_t)band_local_pixel_row(c, y) * c->W + x, 4, hipMemcpyDeviceToHost, c->stream));
    return sync_locked(c);
}
int swr_set_depth(swr_context* c, int x, int y, float d) {
    SWR_ENTER(c);
    if (!in_band(c, x, y)) return SWR_OK;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    SWR_HIP(c, hipMemcpyAsync(c->depth + (size_t)band_local_pixel_row(c, y) * c->W + x, &d, 4, hipMemcpyHostToDevice, c->stream));
    return sync_locked(c);
}

int swr_texture_create(swr_context* c, const uint8_t* rgba8, int w, int h, swr_texture** out) {
    SWR_ENTER(c);
    if (!rgba8 || w <= 0 || h <= 0 || !out) return fail(c, SWR_ERR_INVALID_ARG, "bad texture arguments");
    if ((uint64_t)w * (uint64_t)h >= (1ull << 30)) return fail(c, SWR_ERR_UNSUPPORTED, "textures of 2^30 texels or more are not supported (32-bit texel index)");
    swr_texture* t = new swr_texture();
    t->w = w; t->h = h;
    hipError_t e = hipMalloc((void**)&t->d_rgba, (size_t)w * h * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_rgba, rgba8, (size_t)w * h * 4, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) {
        if (t->d_rgba) (void)hipFree(t->d_rgba);
        delete t;
        c->err = std::string("texture upload failed: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? SWR_ERR_OOM : SWR_ERR_HIP;
    }
    *out = t;
    return SWR_OK;
}

int swr_texture_set_filter(swr_context* c, swr_texture* t, int bilinear) {
    SWR_ENTER(c);
    if (!t) return fail(c, SWR_ERR_INVALID_ARG, "texture is null");
    t->bilinear = bilinear != 0;          // read when a draw is recorded
    if (t->bilinear && !t->d_blocked && (t->w & 3) == 0 && (t->h & 3) == 0) {
        // the four taps of a bilinear fetch: one 64-byte line 9 times out of 16 from a block-linear copy, two lines always from the
        // row-major original (same stream as the upload and as the raster kernels that will read it)
        SWR_HIP(c, hipMalloc((void**)&t->d_blocked, (size_t)t->w * t->h * 4));
        const int blocks = (int)std::min<size_t>(((size_t)t->w * t->h + 255) / 256, 2048 * 8);
        hipLaunchKernelGGL(k_block_texture, dim3(blocks), dim3(256), 0, c->stream, (const uint32_t*)t->d_rgba, (uint32_t*)t->d_blocked, t->w, t->h);
        SWR_HIP(c, hipGetLastError());
    }
    return SWR_OK;
}

int swr_texture_destroy(swr_context* c, swr_texture* t) {
    SWR_ENTER(c);
    if (!t) return SWR_OK;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    if (t->d_rgba) (void)hipFree(t->d_rgba);
    if (t->d_blocked) (void)hipFree(t->d_blocked);
    delete t;
    return SWR_OK;
}

int swr_texture_sample(swr_context* c, const swr_texture* t, const float* uv, int n, float* out) {
    SWR_ENTER(c);
    if (!t || !uv || !out || n < 0) return fail(c, SWR_ERR_INVALID_ARG, "bad texture_sample arguments");
    if (n == 0) return SWR_OK;
    const size_t off_out = ((size_t)n * 8 + 15) & ~(size_t)15;
    int rc = ensure(c, c->d_scratch, off_out + (size_t)n * 16);
    if (rc) return rc;
    float2* d_uv = c->d_scratch.as<float2>();
    float4* d_out = reinterpret_cast<float4*>(reinterpret_cast<char*>(c->d_scratch.p) + off_out);
    SWR_HIP(c, hipMemcpyAsync(d_uv, uv, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_texture_sample, dim3((n + 255) / 256), dim3(256), 0, c->stream, t->d_rgba, t->w, t->h, d_uv, n, d_out);
    SWR_HIP(c, hipGetLastError());
    SWR_HIP(c, hipMemcpyAsync(out, d_out, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
    return sync_locked(c);
}

int swr_mesh_create(swr_context* c, const swr_vertex* v, int nv, const uint16_t* idx, int ni, swr_mesh** out) {
    SWR_ENTER(c);
    return make_mesh(c, v, nv, idx, ni, false, out);
}

int swr_mesh_destroy(swr_context* c, swr_mesh* m) {
    SWR_ENTER(c);
    if (!m) return SWR_OK;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    if (m->d_verts) (void)hipFree(m->d_verts);
    if (m->d_idx) (void)hipFree(m->d_idx);
    if (m->d_bounds) (void)hipFree(m->d_bounds);
    delete m;
    return SWR_OK;
}

int swr_set_state(swr_context* c, float near_clip, float far_clip, int debug_mode) {
    SWR_ENTER(c);
    if (debug_mode != SWR_DEBUG_NONE && debug_mode != SWR_DEBUG_WIREFRAME) return fail(c, SWR_ERR_INVALID_ARG, "bad debug mode");
    if (!c->draws.empty() && (near_clip != c->near_clip || debug_mode != c->debug_mode)) {
        int rc = flush_locked(c); if (rc) return rc;     // NearClip is read at clip time (Rasterizer.cs:112)
    }
    c->near_clip = near_clip; c->far_clip = far_clip; c->debug_mode = debug_mode;
    return SWR_OK;
}

int swr_initialize_tile_locks(swr_context* c, int width, int height) {
    SWR_ENTER(c);
    if (width <= 0 || height <= 0)                                 // Rasterizer.cs:71-74
        return fail(c, SWR_ERR_INVALID_ARG, "Width and height must be positive non-zero values.");
    return SWR_OK;    // tiles need no locks here: one wave owns one tile
}

int swr_render_mesh(swr_context* c, const swr_mesh* mesh, const float model[16], const float view[16], const float proj[16],
                    int program, const swr_uniforms* u, const swr_texture* tex, int cull, int depth_test, int blend) {
    SWR_ENTER(c);
    return record_draw(c, const_cast<swr_mesh*>(mesh), model, view, proj, program, u, tex, cull, depth_test, blend);
}

int swr_render_mesh_culled(swr_context* c, const swr_mesh* mesh, const float model[16], const float view[16], const float proj[16],
                           int program, const swr_uniforms* u, const swr_texture* tex, int cull, int depth_test, int blend) {
    SWR_ENTER(c);
    return record_draw(c, const_cast<swr_mesh*>(mesh), model, view, proj, program, u, tex, cull, depth_test, blend, true);
}

int swr_mesh_bounds(swr_context* c, const swr_mesh* mesh, float center_radius[4]) {
    SWR_ENTER(c);
    if (!mesh || !center_radius) return fail(c, SWR_ERR_INVALID_ARG, "bad mesh_bounds arguments");
    int rc = ensure_bounds(c, const_cast<swr_mesh*>(mesh));
    if (rc) return rc;
    SWR_HIP(c, hipMemcpyAsync(center_radius, mesh->d_bounds, 16, hipMemcpyDeviceToHost, use_front_stream(c)));
    SWR_HIP(c, hipStreamSynchronize(use_front_stream(c)));
    return SWR_OK;
}

int swr_is_sphere_in_frustum(swr_context* c, const float center_radius[4], const float model[16], const float view[16],
                             const float proj[16], int* inside) {
    SWR_ENTER(c);
    if (!center_radius || !model || !view || !proj || !inside) return fail(c, SWR_ERR_INVALID_ARG, "bad is_sphere_in_frustum arguments");
    int rc = ensure(c, c->d_scratch, 256);
    if (rc) return rc;
    float h[48];
    memcpy(h, model, 64); memcpy(h + 16, view, 64); memcpy(h + 32, proj, 64);
    char* base = reinterpret_cast<char*>(c->d_scratch.p);
    SWR_HIP(c, hipMemcpyAsync(base, h, sizeof h, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_frustum_test, dim3(1), dim3(64), 0, c->stream,
                       make_float4(center_radius[0], center_radius[1], center_radius[2], center_radius[3]),
                       (const float*)base, (uint32_t*)(base + 192), c->nm_flags);
    SWR_HIP(c, hipGetLastError());
    uint32_t r = 0;
    SWR_HIP(c, hipMemcpyAsync(&r, base + 192, 4, hipMemcpyDeviceToHost, c->stream));
    SWR_HIP(c, hipStreamSynchronize(c->stream));
    *inside = (int)r;
    return SWR_OK;
}

int swr_render_mesh_arrays(swr_context* c, const swr_vertex* v, int nv, const uint16_t* idx, int ni,
                           const float model[16], const float view[16], const float proj[16],
                           int program, const swr_uniforms* u, const swr_texture* tex, int cull, int depth_test, int blend) {
    SWR_ENTER(c);
    if (c->W <= 0 || c->H <= 0) return SWR_OK;                     // Rasterizer.cs:176
    swr_mesh* m = nullptr;
    int rc = make_mesh(c, v, nv, idx, ni, true, &m);
    if (rc) return rc;
    rc = record_draw(c, m, model, view, proj, program, u, tex, cull, depth_test, blend);
    // by identity (record_draw may have flushed the pending list first): a draw references m iff it is the last one recorded
    if (rc || c->draws.empty() || c->draws.back().mesh != m) c->garbage.push_back(m);
    return rc;
}

int swr_interpolate(swr_context* c, const float* verts60, const float* w, int n, int interpolate, float* out) {
    SWR_ENTER(c);
    if (!verts60 || !w || !out || n < 0) return fail(c, SWR_ERR_INVALID_ARG, "bad interpolate arguments");
    if (n == 0) return SWR_OK;
    const size_t off_w = 256, off_o = 256 + (((size_t)n * 12 + 15) & ~(size_t)15);
    int rc = ensure(c, c->d_scratch, off_o + (size_t)n * 96);
    if (rc) return rc;
    char* base = reinterpret_cast<char*>(c->d_scratch.p);
    SWR_HIP(c, hipMemcpyAsync(base, verts60, 240, hipMemcpyHostToDevice, c->stream));
    SWR_HIP(c, hipMemcpyAsync(base + off_w, w, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_interpolate, dim3((n + 255) / 256), dim3(256), 0, c->stream,
                       (const float*)base, (const float*)(base + off_w), n, interpolate, (float*)(base + off_o));
    SWR_HIP(c, hipGetLastError());
    SWR_HIP(c, hipMemcpyAsync(out, base + off_o, (size_t)n * 96, hipMemcpyDeviceToHost, c->stream));
    return sync_locked(c);
}

int swr_get_stats(swr_context* c, swr_stats* out) {
    SWR_ENTER(c);
    if (!out) return SWR_ERR_INVALID_ARG;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;          // validate (and replay) optimistic batches BEFORE the counters are gathered
    Counters host[65];
    unsigned long long frag[3] = { 0, 0, 0 };
    if (c->tile_stats_tiles) {
        unsigned long long* d3 = c->d_total.as<unsigned long long>() + 1;
        hipLaunchKernelGGL(k_reduce_tile_stats, dim3(1), dim3(1024), 0, c->stream, c->d_tile_stats.as<uint32_t>(),
                           (uint32_t)c->tile_stats_tiles, d3, 0);
        SWR_HIP(c, hipGetLastError());
        SWR_HIP(c, hipMemcpyAsync(frag, d3, 24, hipMemcpyDeviceToHost, c->stream));
    }
    unsigned long long carry[3] = { 0, 0, 0 };      // counters of earlier geometries (see the tile_stats_tiles change in the flush)
    SWR_HIP(c, hipMemcpyAsync(carry, c->d_total.as<unsigned long long>() + 4, 24, hipMemcpyDeviceToHost, c->stream));
    SWR_HIP(c, hipMemcpyAsync(host, c->d_counters.p, sizeof host, hipMemcpyDeviceToHost, c->stream));
    if ((rc = sync_locked(c))) return rc;
    swr_stats s = {};
    for (int i = 0; i < 65; ++i) {
        s.triangles_in += host[i].triangles_in; s.triangles_setup += host[i].triangles_setup;
        s.triangles_clipped += host[i].triangles_clipped; s.tile_pairs += host[i].tile_pairs;
    }
    s.fragments_tested = frag[0] + carry[0]; s.fragments_shaded = frag[1] + carry[1]; s.fragments_written = frag[2] + carry[2];
    s.tile_pairs += c->host_tile_pairs;
    s.flushes = c->totals.flushes;
    *out = s;
    return SWR_OK;
}

int swr_reset_stats(swr_context* c) {
    SWR_ENTER(c);
    int rc = flush_locked(c); if (rc) return rc;
    SWR_HIP(c, hipMemsetAsync(c->d_counters.p, 0, 65 * sizeof(Counters), c->stream));
    if (c->tile_stats_tiles) SWR_HIP(c, hipMemsetAsync(c->d_tile_stats.p, 0, c->tile_stats_tiles * 12, c->stream));
    SWR_HIP(c, hipMemsetAsync(c->d_total.as<unsigned long long>() + 4, 0, 24, c->stream));
    c->totals = {}; c->host_tile_pairs = 0;
    return sync_locked(c);
}

int swr_profile_enable(swr_context* c, int on) {
    SWR_ENTER(c);
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    c->profiling = on < 0 ? 0 : (on > 3 ? 1 : on);
    c->raster_span_no = 0;
    return SWR_OK;
}
int swr_profile_get(swr_context* c, swr_profile* out) {
    SWR_ENTER(c);
    if (!out) return SWR_ERR_INVALID_ARG;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    c->prof.flushes = c->totals.flushes;
    *out = c->prof;
    return SWR_OK;
}
int swr_profile_reset(swr_context* c) {
    SWR_ENTER(c);
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    c->prof = {};
    c->raster_samples.clear();
    return SWR_OK;
}
int swr_profile_raster_samples(swr_context* c, float* out_ms, int capacity, int* n) {
    SWR_ENTER(c);
    if (!n || capacity < 0 || (capacity > 0 && !out_ms)) return SWR_ERR_INVALID_ARG;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    const int have = (int)c->raster_samples.size();
    for (int i = 0; i < std::min(have, capacity); ++i) out_ms[i] = c->raster_samples[(size_t)i];
    *n = have;
    return SWR_OK;
}

int swr_debug_counters(swr_context* c, uint64_t out[8]) {
    SWR_ENTER(c);
    if (!out) return SWR_ERR_INVALID_ARG;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = sync_locked(c))) return rc;
    SWR_HIP(c, hipMemcpyAsync(out, c->d_total.as<unsigned long long>() + 8, 64, hipMemcpyDeviceToHost, c->stream));
    if ((rc = sync_locked(c))) return rc;
    SWR_HIP(c, hipMemsetAsync(c->d_total.as<unsigned long long>() + 8, 0, 64, c->stream));
    return SWR_OK;
}

int swr_selftest_division(swr_context* c, uint64_t samples, uint64_t seed, uint64_t out[8]) {
    SWR_ENTER(c);
    if (!out) return SWR_ERR_INVALID_ARG;
    int rc = flush_locked(c); if (rc) return rc;
    if ((rc = ensure(c, c->d_scratch, 64))) return rc;
    SWR_HIP(c, hipMemsetAsync(c->d_scratch.p, 0, 64, c->stream));
    const int iters = 1024;
    const uint64_t per_block = 256ull * (uint64_t)iters;
    const unsigned blocks = (unsigned)std::min<uint64_t>((samples + per_block - 1) / per_block, 1u << 20);
    if (blocks) hipLaunchKernelGGL(k_selftest_division, dim3(blocks), dim3(256), 0, c->stream, (unsigned long long)seed, iters,
                                   c->d_scratch.as<unsigned long long>());
    SWR_HIP(c, hipGetLastError());
    // reciprocals (recip_core): exhaustive over its whole range, whatever `samples` says
    hipLaunchKernelGGL(k_selftest_recip, dim3(16384), dim3(256), 0, c->stream, c->d_scratch.as<unsigned long long>());
    SWR_HIP(c, hipGetLastError());
    SWR_HIP(c, hipMemcpyAsync(out, c->d_scratch.p, 64, hipMemcpyDeviceToHost, c->stream));
    return sync_locked(c);
}

int swr_device_name(swr_context* c, char* buf, int buflen) {
    SWR_ENTER(c);
    if (!buf || buflen <= 0) return SWR_ERR_INVALID_ARG;
    snprintf(buf, (size_t)buflen, "%s", c->dev_name);
    return SWR_OK;
}

}  // extern "C"
