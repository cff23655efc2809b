/* abi_smoke.c -- a plain-C caller of the drop-in boundary: dlopen()s libswr_hip.so, resolves every entry point that
 * include/swr.h declares and renders BASELINE cfg1 (256x256, one flat-shaded triangle, identity matrices, no depth,
 * Rasterizer.cs:163-174 defaults otherwise) through swr_render_mesh_arrays -- the array form of the reference's own
 * RenderMesh signature.  Expected: the analytic coverage of the triangle (8321 pixels, tests/test_oracle_kat.py), every
 * covered pixel = the colour of outputs[0] = the LAST vertex (flat: Rasterizer.cs:367,622-627), depth untouched.
 * usage: abi_smoke /path/to/libswr_hip.so      exit code 0 = ok.   Built with gcc (tests/c/Makefile), no HIP headers. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "swr.h"

#define RESOLVE(name) do { *(void**)(&p_##name) = dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; } } while (0)

static const char* const all_symbols[] = {
    "swr_abi_version", "swr_build_info", "swr_numerics_mode", "swr_set_transform_fma", "swr_get_transform_fma", "swr_set_pipelining", "swr_get_pipelining", "swr_sync_count", "swr_present_rgb_async", "swr_present_wait", "swr_last_error", "swr_create", "swr_destroy", "swr_resize", "swr_set_band", "swr_set_band_interleaved", "swr_bind_framebuffer",
    "swr_set_stream", "swr_clear_color", "swr_clear_depth", "swr_get_pixel", "swr_set_pixel", "swr_get_depth", "swr_set_depth",
    "swr_readback", "swr_readback_rgb", "swr_flatten_rgb_device", "swr_flatten_rgb_device_async", "swr_replay_count", "swr_host_register", "swr_host_unregister", "swr_upload",
    "swr_color_device_ptr", "swr_depth_device_ptr", "swr_texture_create", "swr_texture_destroy", "swr_texture_set_filter",
    "swr_texture_sample", "swr_mesh_create", "swr_mesh_destroy", "swr_set_state", "swr_initialize_tile_locks", "swr_render_mesh",
    "swr_render_mesh_arrays", "swr_mesh_bounds", "swr_is_sphere_in_frustum", "swr_render_mesh_culled", "swr_flush", "swr_sync",
    "swr_interpolate", "swr_get_stats", "swr_reset_stats", "swr_profile_enable", "swr_profile_get", "swr_profile_reset", "swr_profile_raster_samples",
    "swr_device_name", "swr_debug_counters", "swr_selftest_division" };

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s libswr_hip.so\n", argv[0]); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    for (size_t i = 0; i < sizeof all_symbols / sizeof all_symbols[0]; ++i)
        if (!dlsym(lib, all_symbols[i])) { fprintf(stderr, "missing symbol %s\n", all_symbols[i]); return 2; }

    int (*p_swr_abi_version)(void);
    const char* (*p_swr_last_error)(const swr_context*);
    int (*p_swr_create)(int, swr_context**);
    void (*p_swr_destroy)(swr_context*);
    int (*p_swr_resize)(swr_context*, int, int);
    int (*p_swr_clear_color)(swr_context*, const float[4]);
    int (*p_swr_clear_depth)(swr_context*);
    int (*p_swr_initialize_tile_locks)(swr_context*, int, int);
    int (*p_swr_render_mesh_arrays)(swr_context*, const swr_vertex*, int, const uint16_t*, int, const float[16], const float[16],
                                    const float[16], int, const swr_uniforms*, const swr_texture*, int, int, int);
    int (*p_swr_readback)(swr_context*, float*, float*);
    int (*p_swr_get_stats)(swr_context*, swr_stats*);
    RESOLVE(swr_abi_version); RESOLVE(swr_last_error); RESOLVE(swr_create); RESOLVE(swr_destroy); RESOLVE(swr_resize);
    RESOLVE(swr_clear_color); RESOLVE(swr_clear_depth); RESOLVE(swr_initialize_tile_locks); RESOLVE(swr_render_mesh_arrays);
    RESOLVE(swr_readback); RESOLVE(swr_get_stats);

    if (p_swr_abi_version() != SWR_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 3; }
    swr_context* ctx = NULL;
    int rc = p_swr_create(0, &ctx);
    if (rc != SWR_OK) { fprintf(stderr, "swr_create: %d %s\n", rc, p_swr_last_error(NULL)); return 4; }   /* no CPU fallback */
#define CK(call) do { rc = (call); if (rc != SWR_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc, p_swr_last_error(ctx)); return 5; } } while (0)
    const int W = 256, H = 256;
    CK(p_swr_resize(ctx, W, H));
    if (p_swr_initialize_tile_locks(ctx, 0, 16) != SWR_ERR_INVALID_ARG) { fprintf(stderr, "tile-lock argument check missing\n"); return 6; }
    const float black[4] = { 0.f, 0.f, 0.f, 1.f };
    CK(p_swr_clear_color(ctx, black));
    CK(p_swr_clear_depth(ctx));
    swr_vertex v[3];
    memset(v, 0, sizeof v);
    const float pos[3][3] = { { -0.5f, -0.5f, 0.f }, { 0.5f, -0.5f, 0.f }, { 0.f, 0.5f, 0.f } };
    const float col[3][4] = { { 1, 0, 0, 1 }, { 0, 1, 0, 1 }, { 0, 0, 1, 1 } };
    for (int i = 0; i < 3; ++i) { memcpy(v[i].position, pos[i], 12); memcpy(v[i].color, col[i], 16); v[i].normal[2] = 1.f; }
    const uint16_t idx[3] = { 0, 1, 2 };
    const float I[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    CK(p_swr_render_mesh_arrays(ctx, v, 3, idx, 3, I, I, I, SWR_PROG_FLAT_COLOR, NULL, NULL, SWR_CULL_NONE, SWR_DEPTH_DISABLED, SWR_BLEND_ALPHA));
    float* color = (float*)malloc((size_t)W * H * 16);
    float* depth = (float*)malloc((size_t)W * H * 4);
    CK(p_swr_readback(ctx, color, depth));
    long covered = 0, wrong = 0, depth_touched = 0;
    for (long i = 0; i < (long)W * H; ++i) {
        const float* c = color + 4 * i;
        if (c[0] != 0.f || c[1] != 0.f || c[2] != 0.f) {
            ++covered;
            if (!(c[0] == 0.f && c[1] == 0.f && c[2] == 1.f && c[3] == 1.f)) ++wrong;      /* outputs[0] = v2 = blue */
        }
        if (depth[i] != -3.40282347e+38f) ++depth_touched;                                   /* DepthTest.Disabled: no Z write, :517 */
    }
    swr_stats st;
    CK(p_swr_get_stats(ctx, &st));
    printf("abi_smoke: covered=%ld wrong=%ld depth_touched=%ld fragments_written=%llu triangles_in=%llu\n", covered, wrong, depth_touched,
           (unsigned long long)st.fragments_written, (unsigned long long)st.triangles_in);
    p_swr_destroy(ctx);
    free(color); free(depth);
    return (covered == 8321 && wrong == 0 && depth_touched == 0 && st.fragments_written == 8321 && st.triangles_in == 1) ? 0 : 7;
}
