/*
 * swr.h -- C ABI of the MI355X-native rasterizer backend (libswr_hip.so).
 *
 * Drop-in boundary for the raster hot path of OCSYT/SoftwareRenderer.  The
 * reference has no FFI of its own: the C# host calls
 * `Rasterizer.RenderMesh` directly with delegates and a MainWindow object.  Each
 * entry point below names the reference interface it replaces (file:line under
 * the C# repo); INTEGRATION.md shows the P/Invoke stub a maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types; every function returns an
 *    int status (SWR_OK = 0, negative = error) and never unwinds across the ABI.
 *    swr_last_error() gives the text of the last failure on that context.
 *  - matrices are 16 floats M11..M44, row-major, row-vector convention
 *    (v' = v * M) exactly as System.Numerics.Matrix4x4 is laid out in memory.
 *  - caller owns every input array for the duration of the call; retained meshes
 *    and textures are copied to HBM and owned by the context until destroyed.
 *  - draw calls are RECORDED in submission order under a context mutex and
 *    executed on the GPU at swr_flush / swr_readback / any accessor that needs
 *    pixels, so concurrent callers (Renderer.cs:444 Parallel.ForEach) are safe and
 *    the result equals the serial schedule mesh 0,1,2... triangle 0,1,2...
 *  - one context drives one GPU (one process per GPU); for multi-GPU frames each
 *    rank owns a band of 16-pixel tile rows (swr_set_band) and the host gathers
 *    the colour bands (RCCL) -- see softwarerenderer_amd/multigpu.py.
 *  - there is NO CPU fallback: without a usable HIP device swr_create fails.
 */
#ifndef SWR_H
#define SWR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWR_ABI_VERSION 3     /* 2: swr_bind_framebuffer no longer drains (lifetime rule below); swr_resize / swr_set_band* are no-ops when
                               * nothing changes; new: swr_sync_count, swr_build_info, swr_numerics_mode, swr_present_rgb_async / swr_present_wait
                               * 3: frames in flight (swr_set_pipelining / swr_get_pipelining, on by default); swr_set_transform_fma;
                               *    one batch holds fewer than 2^26 vertex-stage records (see swr_render_mesh) */

/* status codes */
#define SWR_OK                 0
#define SWR_ERR_INVALID_ARG   (-1)  /* C#: ArgumentException (Rasterizer.cs:71-74) / IndexOutOfRangeException */
#define SWR_ERR_HIP           (-2)  /* a HIP runtime call failed */
#define SWR_ERR_OOM           (-3)  /* hipMalloc failed */
#define SWR_ERR_NO_DEVICE     (-4)  /* no gfx950 device / runtime missing */
#define SWR_ERR_UNSUPPORTED   (-5)
#define SWR_STALE              1    /* swr_present_wait only (not an error): the copied frame predates a replayed batch, present again */

/* Rasterizer.DebugMode, Rasterizer.cs:14-18 */
enum { SWR_DEBUG_NONE = 0, SWR_DEBUG_WIREFRAME = 1 };
/* Rasterizer.BlendMode, Rasterizer.cs:25-31 */
enum { SWR_BLEND_NONE = 0, SWR_BLEND_ALPHA = 1, SWR_BLEND_ADDITIVE = 2, SWR_BLEND_MULTIPLY = 3 };
/* Rasterizer.DepthTest, Rasterizer.cs:33-43 */
enum { SWR_DEPTH_DISABLED = 0, SWR_DEPTH_LESS = 1, SWR_DEPTH_LESSEQUAL = 2, SWR_DEPTH_GREATER = 3,
       SWR_DEPTH_GREATEREQUAL = 4, SWR_DEPTH_EQUAL = 5, SWR_DEPTH_NOTEQUAL = 6, SWR_DEPTH_ALWAYS = 7 };
/* Rasterizer.CullMode, Rasterizer.cs:45-50 */
enum { SWR_CULL_NONE = 0, SWR_CULL_BACK = 1, SWR_CULL_FRONT = 2 };

/* Built-in shader programs: C# delegates (Shaders.cs:97-98) cannot cross the ABI, so a
 * program id selects a (vertex, fragment) pair compiled into the backend. */
enum {
    SWR_PROG_FLAT_COLOR = 0,        /* VS of Renderer.cs:830-846 with Interpolate=false; FS = input.Color */
    SWR_PROG_GOURAUD = 1,           /* same VS, Interpolate=true; FS = input.Color */
    SWR_PROG_DUST2_LAMBERT_FOG = 2, /* exactly Renderer.VertexShader/FragmentShader, Renderer.cs:830-860 */
    SWR_PROG_PHONG_4POINT = 3,      /* build-defined 4-point-light Phong (no reference semantics) */
    SWR_PROG_DEBUG_VARYINGS = 4     /* build-defined: FS returns the varyings of Shaders.VertexOutput that no other built-in reads, as
                                     * Rasterizer.Interpolate delivers them -- (ScreenCoords.x + Normal.x, ScreenCoords.y + Normal.y,
                                     * Barycentric.x + Normal.z, Barycentric.y + 0.5); Rasterizer.cs:390,598-613,638 */
};

/* Shaders.VertexInput, Shaders.cs:10-24 -- 48 bytes, identical memory layout */
typedef struct swr_vertex {
    float position[3];
    float uv[2];
    float normal[3];
    float color[4];
} swr_vertex;

typedef struct swr_point_light {      /* subset of Light.cs:9-17 used by PHONG_4POINT */
    float position[3]; float range;
    float color[3];    float intensity;
} swr_point_light;

/* Uniform block: the fields Renderer.FragmentShader closes over, Renderer.cs:39-44 */
typedef struct swr_uniforms {
    float light_direction[3]; float _pad0;
    float light_color[4];
    float fog_color[4];
    float fog_start, fog_end;
    float shininess; float _pad1;
    float camera_position[3]; float _pad2;
    swr_point_light lights[4];
} swr_uniforms;

typedef struct swr_stats {            /* counters for the last flushed batch and totals since reset */
    uint64_t triangles_in;            /* index triples submitted to RenderMesh */
    uint64_t triangles_setup;         /* triangles that reached the tile loop (after clip/cull/reject) */
    uint64_t triangles_clipped;       /* went through the near-plane clipper */
    uint64_t fragments_tested;        /* passed coverage (Rasterizer.cs:493-496) */
    uint64_t fragments_shaded;        /* passed the depth test */
    uint64_t fragments_written;       /* passed alpha, pixel written (Rasterizer.cs:511-519) */
    uint64_t tile_pairs;              /* (triangle, 16x16 tile) pairs binned */
    uint64_t flushes;
} swr_stats;

/* per-stage GPU time of flushes since swr_profile_reset, from hipEvents on the context's stream */
typedef struct swr_profile {
    double vertex_ms, setup_ms, bin_ms, sort_ms, cover_ms, raster_ms, clear_ms, total_ms;
    uint64_t raster_launches, flushes;
} swr_profile;

typedef struct swr_context swr_context;
typedef struct swr_mesh swr_mesh;
typedef struct swr_texture swr_texture;

int  swr_abi_version(void);
/* Identity of this build: "hipcc=<compiler version>; csrc_sha256=<hash of the kernel sources>; fma=<0|1>; dot=<0|1|2>; extra=<extra
 * compile switches, empty for the product>".
 * The lane-to-lane LDS hand-offs of k_cover / k_raster_c are verified per compiler + source pair: the
 * pair the parity sweeps ran on is committed in profiles/verified_build.json and tests/test_gpu_api.py compares. */
const char* swr_build_info(void);
/* The System.Numerics model this library was compiled with (the reference's .NET 9 SIMD paths are not pinned by anything in
 * its repository): *fma = 1 when Vector4.Lerp (Shaders.cs:52-55, Renderer.cs:858) is modelled with fused
 * multiply-adds (SWR_NUMERICS_FMA), *dot_order = summation order of Vector3.Dot / LengthSquared (SWR_DOT_PAIRWISE: 0 sequential,
 * 1 dpps, 2 shuffle-adds).  csharp/RasterizerNative.cs probes the running .NET at start-up and loads the library that matches. */
int  swr_numerics_mode(int* fma, int* dot_order);
const char* swr_last_error(const swr_context* ctx);   /* ctx may be NULL: last swr_create failure */

/* lifetime -------------------------------------------------------------------------------- */
int  swr_create(int device_id, swr_context** out);
/* The other half of the System.Numerics model, per context and at run time (it only touches the vertex stage and the frustum
 * test): whether Vector4.Transform / Vector3.Transform / Matrix4x4.Multiply (Renderer.cs:832-834, FrustumCuller.cs:203,213) and
 * Vector3.TransformNormal (Renderer.cs:835) fuse their multiply-adds.  Default: both = the library's compile-time *fma.  Draws
 * recorded before the call keep the model they were recorded under.  The C# start-up probe sets it from what it observes, so
 * every (Transform, TransformNormal, Lerp, Dot) combination a .NET runtime can show is served by one of the six libraries. */
int  swr_set_transform_fma(swr_context* ctx, int transform_fused, int transform_normal_fused);
int  swr_get_transform_fma(swr_context* ctx, int* transform_fused, int* transform_normal_fused);
void swr_destroy(swr_context* ctx);

/* framebuffer == MainWindow.ColorBuffer / DepthBuffer (MainWindow.cs:25-31) ------------------ */
/* allocates Vector4[W*H] + float[W*H] in HBM; ≙ MainWindow.HandleResize (MainWindow.cs:320-321).
 * W or H <= 0 gives a zero-size target on which every draw is silently skipped (Rasterizer.cs:176). */
int  swr_resize(swr_context* ctx, int width, int height);
/* multi-GPU: this context renders only tile rows [first_tile_row, first_tile_row + n_tile_rows) of the
 * W x H frame; its buffers hold just those rows.  Default = whole frame. */
int  swr_set_band(swr_context* ctx, int first_tile_row, int n_tile_rows);
/* multi-GPU, interleaved variant (load balance for clustered scenes): the frame's tile rows are cut into stripes of
 * `stripe_tile_rows` rows and stripe s belongs to rank s % world; this context renders rank's stripes and its buffers hold them
 * one after the other in ascending order (16 pixel rows per tile row).  swr_set_band returns to a contiguous band / the whole frame. */
int  swr_set_band_interleaved(swr_context* ctx, int rank, int world, int stripe_tile_rows);
/* use caller-provided device memory (e.g. a torch tensor that RCCL will gather) for the band's
 * colour (float4 per pixel) and depth (float per pixel); NULL returns to internal storage.  Draws recorded before the
 * call are launched against the buffers bound before it; the call itself does not wait for the GPU.
 * LIFETIME RULE (ABI 2): a buffer that has been bound stays referenced by the batches flushed against it until the next
 * VALIDATING call on this context (swr_sync, swr_readback*, swr_get_stats, any pixel accessor, swr_destroy): an optimistic
 * batch that did not fit its pair buffers is replayed into the buffer it was flushed against at that point.  Keep previously
 * bound buffers allocated, and do not hand their contents to a consumer as final, until such a call has returned (the caller's
 * own stream synchronisation is not enough).  swr_replay_count tells whether a replay happened. */
int  swr_bind_framebuffer(swr_context* ctx, void* color_device_ptr, void* depth_device_ptr);
int  swr_set_stream(swr_context* ctx, void* hip_stream);      /* hipStream_t; NULL = context's own stream */
int  swr_clear_color(swr_context* ctx, const float rgba[4]);   /* MainWindow.ClearColorBuffer, MainWindow.cs:400-407 */
int  swr_clear_depth(swr_context* ctx);                        /* MainWindow.ClearDepthBuffer -> float.MinValue, :429-436 */
int  swr_get_pixel(swr_context* ctx, int x, int y, float rgba[4]);       /* MainWindow.GetPixel :391-398 (OOB -> 0) */
int  swr_set_pixel(swr_context* ctx, int x, int y, const float rgba[4]); /* MainWindow.SetPixel :382-388 (OOB ignored) */
int  swr_get_depth(swr_context* ctx, int x, int y, float* depth);        /* MainWindow.GetDepth :420-426 (OOB -> MinValue) */
int  swr_set_depth(swr_context* ctx, int x, int y, float depth);         /* MainWindow.SetDepth :411-417 */
/* flush, then copy the band into caller memory (either pointer may be NULL); ≙ the host reading
 * ColorBuffer/DepthBuffer in MainWindow.OnRender (MainWindow.cs:226-263) */
int  swr_readback(swr_context* ctx, float* color_rgba, float* depth);
/* flush, then copy the colour band as packed RGB floats (12 B per pixel): the Vector4 -> Vector3 flatten of
 * MainWindow.OnRender (MainWindow.cs:234-240) done on the GPU, ready for glTexSubImage2D(RGB, FLOAT) */
int  swr_readback_rgb(swr_context* ctx, float* rgb);
/* ASYNCHRONOUS PRESENT (the read-back of a 4096 x 4096 frame takes 7 x as long as rendering it): flatten on the
 * GPU behind the recorded draws and copy the band's RGB floats into `rgb` WITHOUT waiting -- the copy runs on a second stream of the
 * context, so the next frame renders while this one crosses PCIe; two device staging buffers alternate, i.e. two presents may be
 * in flight (a third first waits for the oldest).  `rgb` should be page-locked (swr_host_register) and must stay untouched until
 * swr_present_wait(ticket) returns: SWR_OK = rgb holds the frame; SWR_STALE (1) = an optimistic batch was replayed meanwhile, the
 * pixels predate it: present the frame again.  ≙ MainWindow.OnRender's upload of flatColorBuffer (MainWindow.cs:226-263), double
 * buffered: a host loop `render i+1; present_async(i+1); wait(i); upload i` costs max(render, copy) per frame, not their sum. */
int  swr_present_rgb_async(swr_context* ctx, float* rgb, uint64_t* ticket);
int  swr_present_wait(swr_context* ctx, uint64_t ticket);
/* same flatten, but into DEVICE memory supplied by the caller (band rows x W x 3 floats), enqueued after the recorded
 * draws on the context's stream and NOT synchronised: swr_sync (or the caller's own stream order) completes it.  This is
 * the present payload a multi-GPU frame gathers over xGMI (12 instead of 16 B per pixel). */
int  swr_flatten_rgb_device(swr_context* ctx, float* d_rgb);
/* The same flatten WITHOUT validating the optimistic flushes first: nothing here waits for the GPU, so a frame loop can
 * chain render -> flatten -> its own consumer (an RCCL gather, a peer-mapped store target) in stream order.  The caller
 * validates later -- swr_sync at a point where it waits anyway -- and compares swr_replay_count before / after: if it grew,
 * a batch had not fitted its pair buffers, was replayed by that swr_sync, and payloads flattened in between are stale
 * (flatten and send them again).  Steady-state frames never replay. */
int  swr_flatten_rgb_device_async(swr_context* ctx, float* d_rgb);
int  swr_replay_count(swr_context* ctx, uint64_t* out);
/* how many times an entry point has made the calling thread wait for the stream so far (hipStreamSynchronize): lets a frame
 * loop assert that its steady state never blocks (swr_bind_framebuffer, swr_flush, swr_flatten_rgb_device_async, and
 * swr_resize / swr_set_band* with unchanged arguments do not) */
int  swr_sync_count(swr_context* ctx, uint64_t* out);
/* Page-lock a long-lived host buffer (the C# side's pinned ColorBuffer / flatColorBuffer arrays) so that swr_readback /
 * swr_readback_rgb / swr_upload DMA straight into it at PCIe rate instead of going through a pageable staging copy.
 * Optional: unregistered buffers work, only slower.  Unregister before freeing the memory. */
int  swr_host_register(swr_context* ctx, void* ptr, size_t bytes);
int  swr_host_unregister(swr_context* ctx, void* ptr);
/* upload caller memory into the band (tests: resume from a known framebuffer state) */
int  swr_upload(swr_context* ctx, const float* color_rgba, const float* depth);
int  swr_color_device_ptr(swr_context* ctx, void** out);
int  swr_depth_device_ptr(swr_context* ctx, void** out);

/* retained resources -------------------------------------------------------------------------- */
/* Texture(Image<Rgba32>) ctor / Dispose, Texture.cs:31-41,65-68: RGBA8 row-major, w*h*4 bytes */
int  swr_texture_create(swr_context* ctx, const uint8_t* rgba8, int width, int height, swr_texture** out);
int  swr_texture_destroy(swr_context* ctx, swr_texture* tex);
/* BUILD-DEFINED extension (row N4): 0 = the reference's nearest filter (Texture.cs:43-63, default), 1 = bilinear with wrap
 * (formula in oracle/swr_oracle.c:oswr_texture_sample_bilinear; no reference semantics).  Applies to draws recorded later. */
int  swr_texture_set_filter(swr_context* ctx, swr_texture* tex, int bilinear);
/* Texture.Sample, Texture.cs:43-63, batched: n uv pairs -> n RGBA float4 (runs on the GPU) */
int  swr_texture_sample(swr_context* ctx, const swr_texture* tex, const float* uv, int n, float* out_rgba);
/* mesh.Vertices / mesh.Indices (ModelLoader.cs:45-47): u16 indices, 3 per triangle; an index >= n_vertices
 * is SWR_ERR_INVALID_ARG (C#: IndexOutOfRangeException at Rasterizer.cs:187) */
int  swr_mesh_create(swr_context* ctx, const swr_vertex* vertices, int n_vertices,
                     const uint16_t* indices, int n_indices, swr_mesh** out);
int  swr_mesh_destroy(swr_context* ctx, swr_mesh* mesh);

/* state ≙ public statics Rasterizer.NearClip / FarClip / RenderDebugMode, Rasterizer.cs:20-22 */
int  swr_set_state(swr_context* ctx, float near_clip, float far_clip, int debug_mode);
/* Rasterizer.InitializeTileLocks, Rasterizer.cs:69-93: width/height <= 0 -> SWR_ERR_INVALID_ARG */
int  swr_initialize_tile_locks(swr_context* ctx, int width, int height);

/* draw ≙ Rasterizer.RenderMesh, Rasterizer.cs:163-174 --------------------------------------- */
int  swr_render_mesh(swr_context* ctx, const swr_mesh* mesh,
                     const float model[16], const float view[16], const float projection[16],
                     int program, const swr_uniforms* uniforms, const swr_texture* texture /* NULL -> white */,
                     int cull_mode, int depth_test, int blend_mode);
/* same, taking the arrays of the C# signature directly (copied to HBM for this draw only) */
int  swr_render_mesh_arrays(swr_context* ctx, const swr_vertex* vertices, int n_vertices,
                            const uint16_t* indices, int n_indices,
                            const float model[16], const float view[16], const float projection[16],
                            int program, const swr_uniforms* uniforms, const swr_texture* texture,
                            int cull_mode, int depth_test, int blend_mode);
/* FrustumCuller.cs on the GPU (row N3) ------------------------------------------------------- */
/* Mesh.SphereBounds = FrustumCuller.CalculateBoundingSphere(vertices) (FrustumCuller.cs:59-151, ModelLoader.cs:291),
 * computed once per retained mesh in the serial schedule of the reference's loops; out = {cx, cy, cz, radius} */
int  swr_mesh_bounds(swr_context* ctx, const swr_mesh* mesh, float center_radius[4]);
/* FrustumCuller.IsSphereInFrustum(bounds, model, view, projection), FrustumCuller.cs:201-218 */
int  swr_is_sphere_in_frustum(swr_context* ctx, const float center_radius[4], const float model[16], const float view[16],
                              const float projection[16], int* inside);
/* `if (!IsSphereInFrustum(mesh.SphereBounds, ...)) return; RenderMesh(...)` of Renderer.cs:446-459, with the test
 * evaluated on the device at flush time (one thread per draw): a culled mesh costs no host round trip */
/* LIMIT (ABI 3, all three render calls): one draw holds fewer than 2^26 vertex-stage records = vertices + 4 x triangles (about
 * 16.7 M triangles of a u16-indexed mesh); a larger one is refused with SWR_ERR_UNSUPPORTED when it is recorded.  Draws are batched
 * up to that same bound and flushed by themselves beyond it. */
int  swr_render_mesh_culled(swr_context* ctx, const swr_mesh* mesh,
                            const float model[16], const float view[16], const float projection[16],
                            int program, const swr_uniforms* uniforms, const swr_texture* texture,
                            int cull_mode, int depth_test, int blend_mode);
int  swr_flush(swr_context* ctx);    /* execute recorded draws (asynchronous on the stream) */
int  swr_sync(swr_context* ctx);     /* flush + wait for the stream */
/* Frames in flight.  The reference's loop renders frames back to back (Renderer.cs:404-419: RenderScene per frame; Rasterizer.cs:
 * 163-230: RenderMesh per mesh).  A pipelined flush runs its front end -- vertex stage, clip, setup, binning, coverage -- on a second
 * stream beside the raster kernel of the flush before it (double-buffered intermediates, event-ordered hand-over); pixels, counters
 * and every ordering guarantee against the context's stream are unchanged.  mode: 0 = never (one stream: what kernel timings are
 * quoted on), 1 = every batch (default: 4096^2 / 1 M triangles -6 %, 1920x1080 -8 ... -16 % in steady state; a burst of K frames from
 * an idle context pays one un-overlapped front end, +0.2 ms / K), 2 = only frames of up to 2^15 tiles (2896^2 pixels) or batches of
 * up to 2^17 triangles (bigger ones run on the context's stream, ordered against pipelined neighbours by events).  Switching drains. */
int  swr_set_pipelining(swr_context* ctx, int mode);
int  swr_get_pipelining(swr_context* ctx, int* mode);

/* Rasterizer.Interpolate (public, Rasterizer.cs:566-640), batched on the GPU for API coverage:
 * verts = 3 vertex records of 20 floats {clip4,color4,uv2,normal3,screen2,worldNormal3,pad2};
 * w = n * 3 barycentric weights; out = n records of 24 floats
 * {clip4,color4,uv2,normal3,screen2,worldNormal3,bary3,pad3} */
int  swr_interpolate(swr_context* ctx, const float* verts60, const float* w, int n, int interpolate, float* out);

/* counters / profiling ------------------------------------------------------------------------ */
int  swr_get_stats(swr_context* ctx, swr_stats* out);   /* syncs */
int  swr_reset_stats(swr_context* ctx);
int  swr_profile_enable(swr_context* ctx, int on);       /* 0 off; 1 hipEvent pairs around every stage of a flush; 2 around the raster kernel only;
                                                           * 3 around the raster kernel of every 4th flush (a pair costs about 10 us of stream time) */
int  swr_profile_get(swr_context* ctx, swr_profile* out); /* syncs */
int  swr_profile_reset(swr_context* ctx);
/* the duration (ms) of every raster-kernel launch that carried an event pair since swr_profile_reset, in launch order: *n = how
 * many there are (at most 65,536 are kept), the first min(*n, capacity) are copied (bench.py: median / min / max).  syncs */
int  swr_profile_raster_samples(swr_context* ctx, float* out_ms, int capacity, int* n);
int  swr_device_name(swr_context* ctx, char* buf, int buflen);
/* Self-test of the kernels' exact-division shortcut (csrc/swr_device.h: div_core / rcp_refined / sqrt_core): about
 * `samples` random, near-midpoint, boundary and renderer-shaped operand pairs are divided both ways on the GPU and compared bit
 * for bit with the compiler's correctly rounded `/` and sqrtf (the IEEE results the reference's C# computes, Rasterizer.cs:576-582,
 * 684-686, Renderer.cs:855); the reciprocal core (recip_core, n = 1) is checked on EVERY float of its range, both signs, on top
 * of `samples`.  out = {divisions tested, mismatches, sqrts tested, mismatches, first bad n, d, got, want}. */
int  swr_selftest_division(swr_context* ctx, uint64_t samples, uint64_t seed, uint64_t out[8]);
/* kernel-internal work counters; all zero unless the library was built with -DSWR_DEBUG_COUNTERS (tools/) */
int  swr_debug_counters(swr_context* ctx, uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* SWR_H */
