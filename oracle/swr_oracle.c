/*
 * swr_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see swr_oracle.h header).
 *
 * Plain-C, strict-IEEE restatement of the reference's raster hot path.  Every
 * function cites the reference file:line it follows (paths under the C# repo
 * OCSYT/SoftwareRenderer).  Build: gcc -O2 -ffp-contract=off -fno-fast-math
 * (RyuJIT never contracts a*b+c written as separate operations).
 *
 * PARITY UNPINNED by the reference (it holds no tests / vectors); pinned by the
 * analytic known-answer tests in tests/test_oracle_kat.py only.
 *
 * Order of record: meshes, triangles and tiles are visited serially in index
 * order -- a legal schedule of the reference's Parallel.For loops
 * (Rasterizer.cs:200,462) that fixes the result of depth ties and blending.
 * oswr_set_threads(n>1) runs the same arithmetic under per-tile locks like the
 * reference does; it is used for CPU timing only (its tie order is not fixed).
 */
#include "swr_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#ifndef SWR_NUMERICS_FMA
#define SWR_NUMERICS_FMA 0
#endif

#define TILE_SIZE 16            /* Rasterizer.cs:53 */
#define EPSILON   1e-6f         /* Rasterizer.cs:52 */
#define FLOAT_MINVALUE (-3.40282347e+38f) /* float.MinValue, MainWindow.cs:425,434 */

struct oswr_context {
    int width, height;
    float* color;   /* Vector4[] ColorBuffer, MainWindow.cs:30 */
    float* depth;   /* float[]  DepthBuffer, MainWindow.cs:31 */
    float near_clip, far_clip;  /* Rasterizer.cs:20-21 */
    int debug_mode;             /* Rasterizer.cs:22 */
    int tex_bilinear;           /* build-defined extension: bilinear instead of the reference's nearest filter */
    int n_threads;
    int tiles_x, tiles_y;       /* Rasterizer.cs:55 */
    pthread_mutex_t* tile_locks; /* Rasterizer.cs:54 (monitor per tile): a mutex each, like the C# `lock` -- a spinning flag starves
                                  * once there are more threads than a few tiles' worth of contention */
    oswr_stats stats;
    struct oswr_pool* pool;     /* persistent workers (the reference's Parallel.For runs on the .NET thread pool) */
};

/* ---------- .NET scalar semantics ---------- */

/* (int)float in .NET 9 on x64: saturating, NaN -> 0 */
static inline int f2i(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT_MAX;
    if (f <= -2147483648.0f) return INT_MIN;
    return (int)f;
}
static inline int is_neg(float f) { uint32_t u; memcpy(&u, &f, 4); return (int)(u >> 31); }
/* MathF.Min / MathF.Max (.NET Core 3.0+): NaN-propagating, -0 < +0 */
static inline float mathf_min(float a, float b) {
    if (a != b) { if (!(a != a)) return a < b ? a : b; return a; }
    return is_neg(a) ? a : b;
}
static inline float mathf_max(float a, float b) {
    if (a != b) { if (!(a != a)) return b < a ? a : b; return a; }
    return is_neg(b) ? a : b;
}
/* Math.Clamp(float,float,float) */
static inline float math_clamp(float v, float lo, float hi) {
    if (v < lo) return lo;
    else if (v > hi) return hi;
    return v;
}
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int is_nan_or_inf(float f) { return (f != f) || f == INFINITY || f == -INFINITY; }

/* ---------- System.Numerics restatement (see header for the semantics assumed) ---------- */
/* Two families, modelled separately (nothing says .NET 9 treats them alike): Lerp is a compile-time switch (SWR_NUMERICS_FMA,
 * like the HIP side, where it sits in the hot kernels); Transform / TransformNormal are run-time switches of this library
 * instance (oswr_set_transform_fma; default = SWR_NUMERICS_FMA for both), mirroring swr_set_transform_fma. */
static int g_transform_fma = SWR_NUMERICS_FMA, g_transform_normal_fma = SWR_NUMERICS_FMA;
void oswr_set_transform_fma(int transform_fused, int transform_normal_fused) {
    g_transform_fma = transform_fused != 0; g_transform_normal_fma = transform_normal_fused != 0;
}
void oswr_get_transform_fma(int* transform_fused, int* transform_normal_fused) {
    *transform_fused = g_transform_fma; *transform_normal_fused = g_transform_normal_fma;
}
static inline float nm_madd(float a, float b, float c, int fused) {
    if (fused) return fmaf(a, b, c);      /* (libm's correctly rounded fmaf when the build has no -mfma) */
    float p = a * b;
    return p + c;
}
int oswr_numerics_fma(void) { return SWR_NUMERICS_FMA; }

/* Vector4.Transform(Vector4, Matrix4x4): row-vector convention, M = M11..M44 row-major */
static void vec4_transform(const float v[4], const float m[16], float out[4]) {
    for (int j = 0; j < 4; ++j) {
        float r = m[0 + j] * v[0];
        r = nm_madd(m[4 + j], v[1], r, g_transform_fma);
        r = nm_madd(m[8 + j], v[2], r, g_transform_fma);
        r = nm_madd(m[12 + j], v[3], r, g_transform_fma);
        out[j] = r;
    }
}
/* Vector3.TransformNormal(Vector3, Matrix4x4) */
static void vec3_transform_normal(const float n[3], const float m[16], float out[3]) {
    for (int j = 0; j < 3; ++j) {
        float r = m[0 + j] * n[0];
        r = nm_madd(m[4 + j], n[1], r, g_transform_normal_fma);
        r = nm_madd(m[8 + j], n[2], r, g_transform_normal_fma);
        out[j] = r;
    }
}
/* Vector3.Dot / LengthSquared on the (x, y, z, 0) lanes of a Vector128: the summation order is the second open
 * System.Numerics ambiguity: 0 sequential, 1 dpps order, 2 two shuffle-adds (Vector128.Sum) */
#ifndef SWR_DOT_PAIRWISE
#define SWR_DOT_PAIRWISE 0
#endif
int oswr_dot_pairwise(void) { return SWR_DOT_PAIRWISE; }
static inline float vec3_dot(const float a[3], const float b[3]) {
#if SWR_DOT_PAIRWISE == 1
    float p = a[0] * b[0] + a[1] * b[1], q = a[2] * b[2] + 0.0f;
    return p + q;
#elif SWR_DOT_PAIRWISE == 2
    float p = a[0] * b[0] + a[2] * b[2], q = a[1] * b[1] + 0.0f;
    return p + q;
#else
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
#endif
}
/* Vector3.Normalize(v) = v / v.Length() */
static void vec3_normalize(const float v[3], float out[3]) {
    float len = sqrtf(vec3_dot(v, v));
    out[0] = v[0] / len; out[1] = v[1] / len; out[2] = v[2] / len;
}
/* VectorN.Lerp(a,b,t) = a*(1-t) + b*t */
static inline float nm_lerp(float a, float b, float t) {
#if SWR_NUMERICS_FMA
    return fmaf(a, 1.0f - t, b * t);
#else
    float x = a * (1.0f - t);
    float y = b * t;
    return x + y;
#endif
}
/* The three System.Numerics operations whose SIMD implementation nothing in the reference pins, exposed one by one: the start-up
 * probe of csharp/RasterizerNative.cs evaluates Vector4.Transform / Vector4.Lerp / Vector3.Dot on operands chosen (by
 * tools/make_numerics_probe.py, from these functions in every build of this file) so that the models give different bits. */
void oswr_nm_transform4(const float v[4], const float m[16], float out[4]) { vec4_transform(v, m, out); }
void oswr_nm_transform_normal3(const float n[3], const float m[16], float out[3]) { vec3_transform_normal(n, m, out); }
float oswr_nm_lerp(float a, float b, float t) { return nm_lerp(a, b, t); }
float oswr_nm_dot3(const float a[3], const float b[3]) { return vec3_dot(a, b); }

/* ---------- framebuffer, MainWindow.cs:378-436 ---------- */
static inline int fb_index(const oswr_context* c, int x, int y) { return y * c->width + x; } /* :379 */
static inline float fb_get_depth(const oswr_context* c, int x, int y) {                      /* :420-426 */
    if (x >= 0 && x < c->width && y >= 0 && y < c->height) return c->depth[fb_index(c, x, y)];
    return FLOAT_MINVALUE;
}
static inline void fb_set_depth(oswr_context* c, int x, int y, float d) {                    /* :411-417 */
    if (x >= 0 && x < c->width && y >= 0 && y < c->height) c->depth[fb_index(c, x, y)] = d;
}
static inline void fb_get_pixel(const oswr_context* c, int x, int y, float out[4]) {         /* :391-398 */
    if (x >= 0 && x < c->width && y >= 0 && y < c->height) {
        memcpy(out, c->color + 4 * (size_t)fb_index(c, x, y), 16);
    } else { out[0] = out[1] = out[2] = out[3] = 0.0f; }
}
static inline void fb_set_pixel(oswr_context* c, int x, int y, const float v[4]) {           /* :382-388 */
    if (x >= 0 && x < c->width && y >= 0 && y < c->height)
        memcpy(c->color + 4 * (size_t)fb_index(c, x, y), v, 16);
}

void oswr_clear_color(oswr_context* c, const float rgba[4]) {   /* MainWindow.cs:400-407 */
    size_t n = (size_t)c->width * c->height;
    for (size_t i = 0; i < n; ++i) memcpy(c->color + 4 * i, rgba, 16);
}
void oswr_clear_depth(oswr_context* c) {                         /* MainWindow.cs:429-436 */
    size_t n = (size_t)c->width * c->height;
    for (size_t i = 0; i < n; ++i) c->depth[i] = FLOAT_MINVALUE;
}

static void free_tile_locks(oswr_context* c) {
    if (c->tile_locks) for (int i = 0; i < c->tiles_x * c->tiles_y; ++i) pthread_mutex_destroy(&c->tile_locks[i]);
    free(c->tile_locks);
    c->tile_locks = NULL;
}
static void pool_destroy(oswr_context* c);

/* Rasterizer.InitializeTileLocks, Rasterizer.cs:69-93 */
static int init_tile_locks(oswr_context* c, int width, int height) {
    if (width <= 0 || height <= 0) return -1;   /* C#: throws ArgumentException */
    int tx = (width + TILE_SIZE - 1) / TILE_SIZE;
    int ty = (height + TILE_SIZE - 1) / TILE_SIZE;
    if (tx == c->tiles_x && ty == c->tiles_y && c->tile_locks) return 0;
    free_tile_locks(c);
    c->tiles_x = tx; c->tiles_y = ty;
    c->tile_locks = (pthread_mutex_t*)malloc(sizeof(pthread_mutex_t) * (size_t)tx * ty);
    for (int i = 0; i < tx * ty; ++i) pthread_mutex_init(&c->tile_locks[i], NULL);
    return 0;
}

int oswr_resize(oswr_context* c, int width, int height) {        /* MainWindow.cs:320-321 */
    if (width < 0 || height < 0) return -1;
    free(c->color); free(c->depth);
    c->width = width; c->height = height;
    size_t n = (size_t)width * height;
    c->color = (float*)calloc(n ? n : 1, 16);
    c->depth = (float*)calloc(n ? n : 1, 4);
    free_tile_locks(c); c->tiles_x = c->tiles_y = 0;
    return (c->color && c->depth) ? 0 : -2;
}
oswr_context* oswr_create(int width, int height) {
    oswr_context* c = (oswr_context*)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->near_clip = 0.1f; c->far_clip = 1000.0f; c->debug_mode = OSWR_DEBUG_NONE;  /* Rasterizer.cs:20-22 */
    c->n_threads = 1;
    if (oswr_resize(c, width, height) != 0) { oswr_destroy(c); return NULL; }
    return c;
}
void oswr_destroy(oswr_context* c) {
    if (!c) return;
    pool_destroy(c);
    free(c->color); free(c->depth); free_tile_locks(c); free(c);
}
void oswr_set_state(oswr_context* c, float near_clip, float far_clip, int debug_mode) {
    c->near_clip = near_clip; c->far_clip = far_clip; c->debug_mode = debug_mode;
}
void oswr_set_threads(oswr_context* c, int n) { n = n < 1 ? 1 : n; if (n != c->n_threads) pool_destroy(c); c->n_threads = n; }
float* oswr_color_buffer(oswr_context* c) { return c->color; }
float* oswr_depth_buffer(oswr_context* c) { return c->depth; }
int oswr_width(oswr_context* c) { return c->width; }
int oswr_height(oswr_context* c) { return c->height; }
void oswr_get_stats(oswr_context* c, oswr_stats* out) { *out = c->stats; }
void oswr_reset_stats(oswr_context* c) { memset(&c->stats, 0, sizeof(c->stats)); }

/* ---------- Blend, Rasterizer.cs:58-65 ---------- */
void oswr_blend(const float src[4], const float dst[4], int mode, float out[4]) {
    switch (mode) {
    case OSWR_BLEND_ALPHA: {                       /* src * src.W + dst * (1 - src.W), all four channels */
        float a = src[3], ia = 1.0f - src[3];
        for (int i = 0; i < 4; ++i) { float x = src[i] * a; float y = dst[i] * ia; out[i] = x + y; }
        break; }
    case OSWR_BLEND_ADDITIVE:                      /* Vector4.Min(src + dst, Vector4.One) */
        for (int i = 0; i < 4; ++i) out[i] = mathf_min(src[i] + dst[i], 1.0f);
        break;
    case OSWR_BLEND_MULTIPLY:
        for (int i = 0; i < 4; ++i) out[i] = src[i] * dst[i];
        break;
    case OSWR_BLEND_NONE:
    default:
        for (int i = 0; i < 4; ++i) out[i] = src[i];
        break;
    }
}

/* ---------- GetDepthTestFunction, Rasterizer.cs:543-559 (names inverted as written) ---------- */
int oswr_depth_func(int test, float nd, float od) {
    switch (test) {
    case OSWR_DEPTH_LESSEQUAL:    return nd >= od;
    case OSWR_DEPTH_DISABLED:     return 1;
    case OSWR_DEPTH_LESS:         return nd > od;
    case OSWR_DEPTH_GREATER:      return nd < od;
    case OSWR_DEPTH_GREATEREQUAL: return nd <= od;
    case OSWR_DEPTH_EQUAL:        return fabsf(nd - od) < EPSILON;
    case OSWR_DEPTH_NOTEQUAL:     return fabsf(nd - od) >= EPSILON;
    case OSWR_DEPTH_ALWAYS:       return 1;
    default:                      return 1;
    }
}

/* EdgeFunction, Rasterizer.cs:562-563 */
float oswr_edge_function(const float a[2], const float b[2], const float c[2]) {
    float p = (c[0] - a[0]) * (b[1] - a[1]);
    float q = (c[1] - a[1]) * (b[0] - a[0]);
    return p - q;
}

/* ---------- Texture.Sample, Texture.cs:43-63 (nearest, wrap) ---------- */
void oswr_texture_sample(const uint8_t* rgba8, int w, int h, const float uv[2], float out[4]) {
    float u = uv[0] - (float)f2i(uv[0]);
    float v = uv[1] - (float)f2i(uv[1]);
    u += (u < 0) ? 1.0f : 0.0f;
    v += (v < 0) ? 1.0f : 0.0f;
    int x = f2i(u * (float)w) % w;
    int y = f2i(v * (float)h) % h;
    if (x < 0) x += w;
    if (y < 0) y += h;
    const uint8_t* p = rgba8 + 4 * ((size_t)y * w + x);
    const float inv255 = 1.0f / 255.0f;
    out[0] = (float)p[0] * inv255; out[1] = (float)p[1] * inv255;
    out[2] = (float)p[2] * inv255; out[3] = (float)p[3] * inv255;
}

/* BUILD-DEFINED bilinear filter (row N4; the reference samples nearest).  Texel centres at (i + 0.5) / size, wrap
 * addressing, float32 throughout, lerps as a*(1-t) + b*t (unfused):
 *   x = u*W - 0.5, y = v*H - 0.5; x0 = floor(x), fx = x - x0 (same in y); taps wrap((int)x0), wrap((int)x0 + 1);
 *   out = lerp(lerp(c00, c10, fx), lerp(c01, c11, fx), fy), each c = byte * (1f/255f). */
static inline int wrap_idx(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }
void oswr_texture_sample_bilinear(const uint8_t* rgba8, int w, int h, const float uv[2], float out[4]) {
    float x = uv[0] * (float)w - 0.5f, y = uv[1] * (float)h - 0.5f;
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    int ix0 = wrap_idx(f2i(x0), w), iy0 = wrap_idx(f2i(y0), h);
    int ix1 = ix0 + 1 == w ? 0 : ix0 + 1, iy1 = iy0 + 1 == h ? 0 : iy0 + 1;
    const float inv255 = 1.0f / 255.0f;
    const uint8_t* p00 = rgba8 + 4 * ((size_t)iy0 * w + ix0); const uint8_t* p10 = rgba8 + 4 * ((size_t)iy0 * w + ix1);
    const uint8_t* p01 = rgba8 + 4 * ((size_t)iy1 * w + ix0); const uint8_t* p11 = rgba8 + 4 * ((size_t)iy1 * w + ix1);
    for (int c = 0; c < 4; ++c) {
        float c00 = (float)p00[c] * inv255, c10 = (float)p10[c] * inv255, c01 = (float)p01[c] * inv255, c11 = (float)p11[c] * inv255;
        float top = c00 * (1.0f - fx) + c10 * fx;
        float bot = c01 * (1.0f - fx) + c11 * fx;
        out[c] = top * (1.0f - fy) + bot * fy;
    }
}
void oswr_set_texture_filter(oswr_context* c, int bilinear) { c->tex_bilinear = bilinear; }

/* ---------- Renderer.VertexShader, Renderer.cs:830-846 ---------- */
void oswr_vertex_shader(const oswr_vertex_input* in, const float model[16], const float view[16],
                        const float projection[16], int program, oswr_vertex_output* out) {
    memset(out, 0, sizeof(*out));                    /* VertexOutput() ctor, Shaders.cs:37-46 */
    float p[4] = { in->position[0], in->position[1], in->position[2], 1.0f };
    float world[4], viewp[4], tn[3];
    vec4_transform(p, model, world);                 /* :832 */
    vec4_transform(world, view, viewp);              /* :833 */
    vec4_transform(viewp, projection, out->clip);    /* :834 */
    vec3_transform_normal(in->normal, model, tn);    /* :835 */
    vec3_normalize(tn, out->world_normal);
    out->has_data = 1;                               /* Data = { ["WorldNormal"] = worldNormal } :840 */
    memcpy(out->texcoord, in->uv, 8);
    memcpy(out->color, in->color, 16);
    memcpy(out->normal, in->normal, 12);
    out->interpolate = (program == OSWR_PROG_FLAT_COLOR) ? 0 : 1;   /* :844 (true); FLAT is the build's flat variant */
    if (program == OSWR_PROG_PHONG_4POINT) memcpy(out->world_pos, world, 16);  /* build-defined extra varying */
}

/* ---------- Shaders.Lerp, Shaders.cs:50-95 ---------- */
void oswr_lerp(const oswr_vertex_output* a, const oswr_vertex_output* b, float t, int interpolate,
               oswr_vertex_output* out) {
    oswr_vertex_output r;
    memset(&r, 0, sizeof(r));                        /* ScreenCoords/Barycentric stay default */
    for (int i = 0; i < 4; ++i) r.clip[i] = nm_lerp(a->clip[i], b->clip[i], t);
    for (int i = 0; i < 2; ++i) r.texcoord[i] = nm_lerp(a->texcoord[i], b->texcoord[i], t);
    for (int i = 0; i < 4; ++i) r.color[i] = interpolate ? nm_lerp(a->color[i], b->color[i], t) : a->color[i];
    for (int i = 0; i < 3; ++i) r.normal[i] = interpolate ? nm_lerp(a->normal[i], b->normal[i], t) : a->normal[i];
    if (interpolate && a->has_data && b->has_data) { /* :60-80, Vector3/Vector4 keys: Lerp, no renormalisation */
        for (int i = 0; i < 3; ++i) r.world_normal[i] = nm_lerp(a->world_normal[i], b->world_normal[i], t);
        for (int i = 0; i < 4; ++i) r.world_pos[i] = nm_lerp(a->world_pos[i], b->world_pos[i], t);
        r.has_data = 1;
    } else if (!interpolate && a->has_data) {        /* :81-84 */
        memcpy(r.world_normal, a->world_normal, 12);
        memcpy(r.world_pos, a->world_pos, 16);
        r.has_data = 1;
    }
    r.interpolate = interpolate;
    *out = r;
}

/* ---------- Rasterizer.Interpolate + InterpolateData, Rasterizer.cs:566-707 ---------- */
static inline float persp3(float A, float B, float C, float ra, float rb, float rc, float w) {
    float s = A * ra;              /* Multiply(a, rcpWa) */
    s = s + B * rb;                /* Add(., Multiply(b, rcpWb)) */
    s = s + C * rc;
    return s * w;
}
static inline float bary3(float A, float B, float C, float wa, float wb, float wc) {
    float s = A * wa;              /* Multiply(va, w0) + Multiply(vb, w1) + Multiply(vc, w2) */
    s = s + B * wb;
    s = s + C * wc;
    return s;
}
void oswr_interpolate(const oswr_vertex_output* a, const oswr_vertex_output* b, const oswr_vertex_output* c,
                      float w0, float w1, float w2, int interpolate, oswr_vertex_output* out) {
    oswr_vertex_output r;
    memset(&r, 0, sizeof(r));
    float ra = w0 / a->clip[3];    /* :576-578 */
    float rb = w1 / b->clip[3];
    float rc = w2 / c->clip[3];
    float inv_sum = ra + rb + rc;  /* :579 */
    float w = 1.0f / inv_sum;      /* :582 */
    float wa = ra * w, wb = rb * w, wc = rc * w;   /* :583-585 */
    for (int i = 0; i < 4; ++i) r.clip[i] = persp3(a->clip[i], b->clip[i], c->clip[i], ra, rb, rc, w);
    for (int i = 0; i < 2; ++i) r.texcoord[i] = persp3(a->texcoord[i], b->texcoord[i], c->texcoord[i], ra, rb, rc, w);
    for (int i = 0; i < 2; ++i) r.screen[i] = persp3(a->screen[i], b->screen[i], c->screen[i], ra, rb, rc, w);
    if (interpolate) {
        for (int i = 0; i < 3; ++i) r.normal[i] = persp3(a->normal[i], b->normal[i], c->normal[i], ra, rb, rc, w);
        for (int i = 0; i < 4; ++i) r.color[i] = persp3(a->color[i], b->color[i], c->color[i], ra, rb, rc, w);
        /* InterpolateData(a.Data, b.Data, c.Data, wa, wb, wc), :643-707 */
        if (!a->has_data) {                          /* :652 return bData ?? cData */
            const oswr_vertex_output* s = b->has_data ? b : c;
            memcpy(r.world_normal, s->world_normal, 12); memcpy(r.world_pos, s->world_pos, 16);
            r.has_data = s->has_data;
        } else if (!b->has_data || !c->has_data) {   /* :653 return aData */
            memcpy(r.world_normal, a->world_normal, 12); memcpy(r.world_pos, a->world_pos, 16);
            r.has_data = 1;
        } else {
            float v[3];
            for (int i = 0; i < 3; ++i) v[i] = bary3(a->world_normal[i], b->world_normal[i], c->world_normal[i], wa, wb, wc);
            float len_sq = vec3_dot(v, v);           /* LengthSquared(), :684 */
            if (len_sq > 1e-6f) {
                float s = 1.0f / sqrtf(len_sq);      /* :686 */
                v[0] = v[0] * s; v[1] = v[1] * s; v[2] = v[2] * s;
            }
            memcpy(r.world_normal, v, 12);
            for (int i = 0; i < 4; ++i)              /* Vector4 key: no normalisation, :690-693 */
                r.world_pos[i] = bary3(a->world_pos[i], b->world_pos[i], c->world_pos[i], wa, wb, wc);
            r.has_data = 1;
        }
    } else {                                         /* :622-627 flat: from a */
        memcpy(r.normal, a->normal, 12);
        memcpy(r.color, a->color, 16);
        memcpy(r.world_normal, a->world_normal, 12);
        memcpy(r.world_pos, a->world_pos, 16);
        r.has_data = a->has_data;
    }
    r.interpolate = interpolate;
    r.barycentric[0] = wa; r.barycentric[1] = wb; r.barycentric[2] = wc;
    *out = r;
}

/* ---------- fragment programs ---------- */
/* Renderer.FragmentShader, Renderer.cs:848-860 */
static void fs_dust2(const oswr_uniforms* u, const oswr_vertex_output* in,
                     const uint8_t* tex, int tw, int th, float out[4]) {
    float neg_l[3] = { -u->light_direction[0], -u->light_direction[1], -u->light_direction[2] };
    float diffuse = mathf_max(0.25f, vec3_dot(in->world_normal, neg_l));          /* :851 */
    float tc[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
    if (tex) { if (th < 0) oswr_texture_sample_bilinear(tex, tw, -th, in->texcoord, tc); else oswr_texture_sample(tex, tw, th, in->texcoord, tc); }   /* :852; th < 0 encodes the build-defined bilinear filter */
    float base[4];
    for (int i = 0; i < 4; ++i) base[i] = in->color[i] * tc[i];                      /* :853 */
    float depth = in->clip[2];                                                       /* :854 */
    float fog = math_clamp((u->fog_end - depth) / (u->fog_end - u->fog_start), 0.0f, 1.0f);  /* :855 */
    fog = (fog * fog) * (3.0f - 2.0f * fog);                                         /* :856 */
    float s = 0.1f + 0.9f * diffuse;
    for (int i = 0; i < 4; ++i) {
        float lit = (base[i] * s) * u->light_color[i];
        out[i] = nm_lerp(u->fog_color[i], lit, fog);                                 /* :858 */
    }
    out[3] = base[3];                                                                /* :859 */
}

/* PHONG_4POINT: BUILD-DEFINED (the reference has no point-light shader).
 * Uses only + - * / sqrt so that CPU and GPU can agree bit for bit (no pow/exp):
 *   N = WorldNormal, P = WorldPos.xyz, V = normalize(camera - P), base = Color * tex
 *   rgb = 0.1*base.rgb + sum_l [ (base.rgb*max(0,N.L) + max(0,N.H)^16) * light.color*intensity*att ]
 *   att = clamp(1 - dist/range, 0, 1)^2 ; alpha = base.a */
static void fs_phong4(const oswr_uniforms* u, const oswr_vertex_output* in,
                      const uint8_t* tex, int tw, int th, float out[4]) {
    float tc[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
    if (tex) { if (th < 0) oswr_texture_sample_bilinear(tex, tw, -th, in->texcoord, tc); else oswr_texture_sample(tex, tw, th, in->texcoord, tc); }
    float base[4];
    for (int i = 0; i < 4; ++i) base[i] = in->color[i] * tc[i];
    const float* N = in->world_normal;
    float P[3] = { in->world_pos[0], in->world_pos[1], in->world_pos[2] };
    float Vd[3] = { u->camera_position[0] - P[0], u->camera_position[1] - P[1], u->camera_position[2] - P[2] };
    float V[3]; vec3_normalize(Vd, V);
    float acc[3] = { 0.1f * base[0], 0.1f * base[1], 0.1f * base[2] };
    for (int l = 0; l < 4; ++l) {
        const oswr_point_light* L = &u->lights[l];
        float Ld[3] = { L->position[0] - P[0], L->position[1] - P[1], L->position[2] - P[2] };
        float dist = sqrtf(vec3_dot(Ld, Ld));
        float Ln[3] = { Ld[0] / dist, Ld[1] / dist, Ld[2] / dist };
        float ndotl = mathf_max(0.0f, vec3_dot(N, Ln));
        float att = math_clamp(1.0f - dist / L->range, 0.0f, 1.0f);
        att = att * att;
        float Hd[3] = { Ln[0] + V[0], Ln[1] + V[1], Ln[2] + V[2] };
        float H[3]; vec3_normalize(Hd, H);
        float sp = mathf_max(0.0f, vec3_dot(N, H));
        sp = sp * sp; sp = sp * sp; sp = sp * sp; sp = sp * sp;   /* ^16 */
        float k = L->intensity * att;
        for (int i = 0; i < 3; ++i) {
            float term = (base[i] * ndotl + sp) * (L->color[i] * k);
            acc[i] = acc[i] + term;
        }
    }
    out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2]; out[3] = base[3];
}

/* DEBUG_VARYINGS (build-defined program; its INPUTS are the reference's: Rasterizer.Interpolate's Normal :610-613, ScreenCoords
 * :390,598-601 and Barycentric :638): rgba = (ScreenCoords.x + Normal.x, ScreenCoords.y + Normal.y, Barycentric.x + Normal.z,
 * Barycentric.y + 0.5) */
static void fs_debug_varyings(const oswr_vertex_output* in, float out[4]) {
    out[0] = in->screen[0] + in->normal[0];
    out[1] = in->screen[1] + in->normal[1];
    out[2] = in->barycentric[0] + in->normal[2];
    out[3] = in->barycentric[1] + 0.5f;
}

/* returns 1 when the delegate would return a value (built-ins always do) */
int oswr_fragment_shader(int program, const oswr_uniforms* u, const oswr_vertex_output* in,
                         const uint8_t* tex, int tw, int th, float out[4]) {
    switch (program) {
    case OSWR_PROG_DUST2_LAMBERT_FOG: fs_dust2(u, in, tex, tw, th, out); return 1;
    case OSWR_PROG_PHONG_4POINT:      fs_phong4(u, in, tex, tw, th, out); return 1;
    case OSWR_PROG_DEBUG_VARYINGS:    fs_debug_varyings(in, out); return 1;
    case OSWR_PROG_FLAT_COLOR:
    case OSWR_PROG_GOURAUD:
    default: memcpy(out, in->color, 16); return 1;
    }
}

/* ---------- per-draw state handed down the call chain ---------- */
typedef struct {
    oswr_context* ctx;
    int program; const oswr_uniforms* uniforms;
    const uint8_t* tex; int tw, th;
    int cull, depth_test, blend;
    oswr_stats* stats;   /* per-thread accumulator */
    int threaded;
} draw_state;

static inline void tile_lock(draw_state* ds, int idx) {
    if (ds->threaded) pthread_mutex_lock(&ds->ctx->tile_locks[idx]);
}
static inline void tile_unlock(draw_state* ds, int idx) {
    if (ds->threaded) pthread_mutex_unlock(&ds->ctx->tile_locks[idx]);
}

#ifdef OSWR_DEAD_COUNT
/* tools/dead_fragments.py only (a build of its own, never the checker the tests load): how many written fragments are later made
 * irrelevant, bit for bit, by a written fragment of the same pixel whose alpha is exactly 1.0f under BlendMode.Alpha
 * (Rasterizer.cs:58-65: src * 1 + dst * 0 == src when dst is finite and no component of src is a zero whose sign dst * 0 could flip) */
static unsigned* dc_pending; static int dc_w, dc_h;
unsigned long long oswr_dc_written, oswr_dc_dead, oswr_dc_alpha_one, oswr_dc_killers;
void oswr_dead_count_reset(void) { free(dc_pending); dc_pending = NULL; oswr_dc_written = oswr_dc_dead = oswr_dc_alpha_one = oswr_dc_killers = 0; }
static void oswr_dead_count_note(oswr_context* c, int x, int y, const float src[4], const float dst[4]) {
    if (!dc_pending || dc_w != c->width || dc_h != c->height) {
        free(dc_pending); dc_w = c->width; dc_h = c->height;
        dc_pending = (unsigned*)calloc((size_t)dc_w * (size_t)dc_h, sizeof(unsigned));
    }
    unsigned* pend = &dc_pending[(size_t)y * (size_t)dc_w + (size_t)x];
    ++oswr_dc_written;
    if (src[3] == 1.0f) ++oswr_dc_alpha_one;
    int finite = 1, nonzero = 1;
    for (int i = 0; i < 4; ++i) { if (is_nan_or_inf(dst[i]) || is_nan_or_inf(src[i])) finite = 0; if (src[i] == 0.0f) nonzero = 0; }
    if (src[3] == 1.0f && finite && nonzero) { oswr_dc_dead += *pend; *pend = 0; ++oswr_dc_killers; }
    ++*pend;
}
#endif

/* fragment tail shared by RasterizeTriangle and DrawLine; alpha_rule: 0 => W > 0 (:511), 1 => W != 0 (:325)
 * returns 1 if the pixel was written */
static int shade_and_write(draw_state* ds, int x, int y, float depth, const oswr_vertex_output* frag, int alpha_rule) {
    oswr_context* c = ds->ctx;
    float col[4];
    int has = oswr_fragment_shader(ds->program, ds->uniforms, frag, ds->tex, ds->tw, ds->th, col);
    int ok = has && (alpha_rule ? (col[3] != 0.0f) : (col[3] > 0.0f));
    if (!ok) return 0;
    float dst[4], out[4];
    fb_get_pixel(c, x, y, dst);
#ifdef OSWR_DEAD_COUNT
    oswr_dead_count_note(c, x, y, col, dst);
#endif
    oswr_blend(col, dst, ds->blend, out);
    fb_set_pixel(c, x, y, out);
    if (ds->depth_test != OSWR_DEPTH_DISABLED) fb_set_depth(c, x, y, depth);
    return 1;
}

/* ---------- DrawLine, Rasterizer.cs:232-340 (wireframe debug mode) ---------- */
static void draw_line(draw_state* ds, const float p0[2], const float p1[2], const float depths[3],
                      const oswr_vertex_output outputs[3]) {
    oswr_context* c = ds->ctx;
    int minX = f2i(mathf_max(mathf_min(p0[0], p1[0]), 0.0f));                         /* :242 */
    int maxX = f2i(mathf_min(mathf_max(p0[0], p1[0]), (float)(c->width - 1)));        /* :243 */
    int minY = f2i(mathf_max(mathf_min(p0[1], p1[1]), 0.0f));
    int maxY = f2i(mathf_min(mathf_max(p0[1], p1[1]), (float)(c->height - 1)));
    if (minX > maxX || minY > maxY) return;
    int tileMinX = minX / TILE_SIZE, tileMaxX = maxX / TILE_SIZE;
    int tileMinY = minY / TILE_SIZE, tileMaxY = maxY / TILE_SIZE;
    float dx = p1[0] - p0[0], dy = p1[1] - p0[1];
    float len_sq = dx * dx + dy * dy;                                                 /* :259 */
    for (int tileY = tileMinY; tileY <= tileMaxY; ++tileY) {
        for (int tileX = tileMinX; tileX <= tileMaxX; ++tileX) {
            int tsx = tileX * TILE_SIZE, tex_ = imin(tsx + TILE_SIZE - 1, c->width - 1);
            int tsy = tileY * TILE_SIZE, tey = imin(tsy + TILE_SIZE - 1, c->height - 1);
            int startX = imax(minX, tsx), endX = imin(maxX, tex_);
            int startY = imax(minY, tsy), endY = imin(maxY, tey);
            if (startX > endX || startY > endY) continue;
            int lock = tileY * c->tiles_x + tileX;
            tile_lock(ds, lock);
            for (int y = startY; y <= endY; ++y) {
                for (int x = startX; x <= endX; ++x) {
                    float px = (float)x + 0.5f - p0[0];                               /* :296 */
                    float py = (float)y + 0.5f - p0[1];
                    float t = 0;
                    if (len_sq > 0) t = (px * dx + py * dy) / len_sq;                 /* :301 */
                    t = mathf_max(0.0f, mathf_min(1.0f, t));                          /* :303 */
                    float cx = p0[0] + t * dx, cy = p0[1] + t * dy;
                    float ddx = ((float)x + 0.5f) - cx, ddy = ((float)y + 0.5f) - cy;
                    float dist_sq = ddx * ddx + ddy * ddy;
                    if (dist_sq <= 0.5f * 0.5f) {
                        ds->stats->fragments_tested++;
                        float depth = 1.0f / (depths[0] * (1.0f - t) + depths[1] * t);  /* :315 */
                        float old = fb_get_depth(c, x, y);
                        if (!oswr_depth_func(ds->depth_test, depth, old)) continue;
                        ds->stats->fragments_shaded++;
                        oswr_vertex_output frag;                                        /* :321-322 */
                        oswr_interpolate(&outputs[0], &outputs[1], &outputs[0], 1.0f - t, t, 0.0f,
                                         outputs[0].interpolate, &frag);
                        if (shade_and_write(ds, x, y, depth, &frag, 1)) ds->stats->fragments_written++;
                    }
                }
            }
            tile_unlock(ds, lock);
        }
    }
}

/* ---------- RasterizeTriangle, Rasterizer.cs:401-539 ---------- */
static void rasterize_triangle(draw_state* ds, float screen[3][2], const float depths[3],
                               const oswr_vertex_output outputs[3]) {
    oswr_context* c = ds->ctx;
    float area = oswr_edge_function(screen[0], screen[1], screen[2]);                 /* :411 */
    if (area == 0) return;
    int front = area < 0;                                                             /* :414 */
    if ((ds->cull == OSWR_CULL_BACK && !front) || (ds->cull == OSWR_CULL_FRONT && front)) return;

    if (c->debug_mode == OSWR_DEBUG_WIREFRAME) {                                      /* :419-425 */
        draw_line(ds, screen[0], screen[1], depths, outputs);
        draw_line(ds, screen[1], screen[2], depths, outputs);
        draw_line(ds, screen[2], screen[0], depths, outputs);
        return;
    }

    float inv_area = 1.0f / area;                                                     /* :427 */
    int needs_depth = ds->depth_test != OSWR_DEPTH_DISABLED;
    int can_early_out = ds->blend == OSWR_BLEND_NONE;

    float minXf = mathf_min(mathf_min(screen[0][0], screen[1][0]), screen[2][0]);     /* :432-435 */
    float maxXf = mathf_max(mathf_max(screen[0][0], screen[1][0]), screen[2][0]);
    float minYf = mathf_min(mathf_min(screen[0][1], screen[1][1]), screen[2][1]);
    float maxYf = mathf_max(mathf_max(screen[0][1], screen[1][1]), screen[2][1]);
    int minX = imax(f2i(floorf(minXf)), 0);                                           /* :437-440 */
    int maxX = imin(f2i(ceilf(maxXf)), c->width - 1);
    int minY = imax(f2i(floorf(minYf)), 0);
    int maxY = imin(f2i(ceilf(maxYf)), c->height - 1);
    if (minX > maxX || minY > maxY) return;

    float a01 = screen[0][1] - screen[1][1], b01 = screen[1][0] - screen[0][0];       /* :445-447 */
    float a12 = screen[1][1] - screen[2][1], b12 = screen[2][0] - screen[1][0];
    float a20 = screen[2][1] - screen[0][1], b20 = screen[0][0] - screen[2][0];

    int tileMinX = minX / TILE_SIZE, tileMaxX = maxX / TILE_SIZE;                     /* :449-452 */
    int tileMinY = minY / TILE_SIZE, tileMaxY = maxY / TILE_SIZE;
    float depth0 = depths[0], depth1 = depths[1], depth2 = depths[2];
    ds->stats->triangles_setup++;

    for (int tileY = tileMinY; tileY <= tileMaxY; ++tileY) {                          /* :462 (serial schedule) */
        for (int tileX = tileMinX; tileX <= tileMaxX; ++tileX) {
            int tileStartX = tileX * TILE_SIZE;
            int tileEndX = imin(tileStartX + TILE_SIZE - 1, c->width - 1);
            int tileStartY = tileY * TILE_SIZE;
            int tileEndY = imin(tileStartY + TILE_SIZE - 1, c->height - 1);
            int startX = imax(minX, tileStartX), endX = imin(maxX, tileEndX);
            int startY = imax(minY, tileStartY), endY = imin(maxY, tileEndY);
            if (startX > endX || startY > endY) continue;

            int lock = tileY * c->tiles_x + tileX;
            tile_lock(ds, lock);                                                      /* :478-479 */
            float w0Row = a12 * ((float)startX - screen[1][0]) + b12 * ((float)startY - screen[1][1]);  /* :481-483 */
            float w1Row = a20 * ((float)startX - screen[2][0]) + b20 * ((float)startY - screen[2][1]);
            float w2Row = a01 * ((float)startX - screen[0][0]) + b01 * ((float)startY - screen[0][1]);
            for (int y = startY; y <= endY; ++y) {
                float w0 = w0Row, w1 = w1Row, w2 = w2Row;
                for (int x = startX; x <= endX; ++x) {
                    int inside = (w0 >= 0 && w1 >= 0 && w2 >= 0) || (w0 <= 0 && w1 <= 0 && w2 <= 0);   /* :493-494 */
                    if (inside) {
                        ds->stats->fragments_tested++;
                        float w0f = w0 * inv_area, w1f = w1 * inv_area, w2f = w2 * inv_area;
                        float depth = depth0 * w0f + depth1 * w1f + depth2 * w2f;     /* :502, left to right */
                        float old = fb_get_depth(c, x, y);
                        if (oswr_depth_func(ds->depth_test, depth, old)) {
                            ds->stats->fragments_shaded++;
                            oswr_vertex_output frag;
                            oswr_interpolate(&outputs[0], &outputs[1], &outputs[2], w0f, w1f, w2f,
                                             outputs[0].interpolate, &frag);          /* :507-508 */
                            if (shade_and_write(ds, x, y, depth, &frag, 0)) {
                                ds->stats->fragments_written++;
                            } else if (can_early_out) {
                                break;                                                /* :520-523 */
                            }
                        }
                    }
                    w0 += a12; w1 += a20; w2 += a01;                                  /* :527-529 */
                }
                w0Row += b12; w1Row += b20; w2Row += b01;                             /* :532-534 */
            }
            (void)needs_depth;
            tile_unlock(ds, lock);
        }
    }
}

/* ---------- DrawTriangle, Rasterizer.cs:342-399 ---------- */
static void draw_triangle(draw_state* ds, const oswr_vertex_output* v0, const oswr_vertex_output* v1,
                          const oswr_vertex_output* v2) {
    oswr_context* c = ds->ctx;
    if (c->width <= 0 || c->height <= 0) return;
    int rw = c->width, rh = c->height;
    float inv_w = 1.0f / (float)(rw - 1);                                             /* :362-363 */
    float inv_h = 1.0f / (float)(rh - 1);
    float screen[3][2]; float depths[3];
    oswr_vertex_output outputs[3] = { *v2, *v1, *v0 };                                /* :367 */
    for (int i = 0; i < 3; ++i) {
        float invW = 1.0f / outputs[i].clip[3];                                       /* :371 */
        float nx = outputs[i].clip[0] * invW, ny = outputs[i].clip[1] * invW, nz = outputs[i].clip[2] * invW;
        if (is_nan_or_inf(nx) || is_nan_or_inf(ny) || is_nan_or_inf(nz)) return;      /* :378-380 */
        screen[i][0] = (nx * 0.5f + 0.5f) * (float)rw;                                /* :383-386 */
        screen[i][1] = (1.0f - (ny * 0.5f + 0.5f)) * (float)rh;
        depths[i] = (nz + 1.0f) * 0.5f;                                               /* :388 */
        outputs[i].screen[0] = screen[i][0] * inv_w;                                  /* :390 */
        outputs[i].screen[1] = screen[i][1] * inv_h;
    }
    if (v0->clip[3] == 0 || v1->clip[3] == 0 || v2->clip[3] == 0) return;             /* :393 */
    if (oswr_edge_function(screen[0], screen[1], screen[2]) == 0) return;             /* :396 */
    rasterize_triangle(ds, screen, depths, outputs);
}

/* ---------- ClipTriangleAgainstNearPlane, Rasterizer.cs:95-160 ---------- */
static int clip_near(const oswr_context* c, const oswr_vertex_output in[3], oswr_vertex_output out[4]) {
    int n = 0;
    float nearc = c->near_clip;
    for (int i = 0; i < 3; ++i) {
        const oswr_vertex_output* cur = &in[i];
        const oswr_vertex_output* nxt = &in[(i + 1) % 3];
        int cur_in = cur->clip[2] >= nearc * cur->clip[3];                            /* :112-113 */
        int nxt_in = nxt->clip[2] >= nearc * nxt->clip[3];
        if (cur_in) out[n++] = *cur;
        if (cur_in != nxt_in) {
            float t;
            float z0 = cur->clip[2], w0 = cur->clip[3], z1 = nxt->clip[2], w1 = nxt->clip[3];
            float denom = (z1 - z0) - nearc * (w1 - w0);                              /* :131 */
            if (fabsf(denom) < EPSILON) {
                t = 0.5f;
            } else {
                t = (z0 - nearc * w0) / (nearc * (w1 - w0) - (z1 - z0));              /* :138 */
                t = math_clamp(t, 0.0f, 1.0f);
            }
            oswr_lerp(cur, nxt, t, 1, &out[n++]);                                     /* :142 */
        }
    }
    return n;
}

/* one triangle of RenderMesh's Parallel.For body, Rasterizer.cs:200-229 */
static void process_triangle(draw_state* ds, const oswr_vertex_input* vertices, const uint16_t* indices, int i,
                             const float* model, const float* view, const float* proj) {
    oswr_vertex_output v[3];
    for (int k = 0; k < 3; ++k)
        oswr_vertex_shader(&vertices[indices[i * 3 + k]], model, view, proj, ds->program, &v[k]);
    ds->stats->triangles_in++;
    int b0 = v[0].clip[3] <= 0, b1 = v[1].clip[3] <= 0, b2 = v[2].clip[3] <= 0;      /* :208-210 */
    if (b0 && b1 && b2) return;
    if (b0 || b1 || b2) {
        oswr_vertex_output poly[4];
        ds->stats->triangles_clipped++;
        int n = clip_near(ds->ctx, v, poly);
        if (n < 3) return;                                                            /* :148 */
        for (int k = 1; k < n - 1; ++k) draw_triangle(ds, &poly[0], &poly[k], &poly[k + 1]);  /* :154-157, :219-223 */
    } else {
        draw_triangle(ds, &v[0], &v[1], &v[2]);
    }
}

/* One RenderMesh call = one job for the persistent pool (the reference: Parallel.For over the triangles of the mesh on the
 * .NET thread pool, Rasterizer.cs:200).  Triangles are handed out in chunks; consecutive grabs take chunks that lie far apart in
 * the mesh (chunk j -> (j % lanes) * per_lane + j / lanes), because neighbouring triangles share tiles and would queue on
 * the same tile locks. */
#define OSWR_CHUNK 64
typedef struct {
    draw_state ds;
    const oswr_vertex_input* vertices; const uint16_t* indices; int n_tris;
    const float *model, *view, *proj;
    atomic_int next;
    int n_chunks, lanes, per_lane;
} oswr_job;

typedef struct oswr_pool {
    int n; pthread_t* th; oswr_stats* stats;
    pthread_mutex_t mu; pthread_cond_t cv_job, cv_done;
    unsigned gen; int running, shutdown;
    oswr_job job;
    oswr_context* ctx;
} oswr_pool;

typedef struct { oswr_pool* pool; int id; } pool_arg;

static void run_job(oswr_job* j, oswr_stats* st) {
    draw_state ds = j->ds;
    ds.stats = st;
    for (;;) {
        int g = atomic_fetch_add(&j->next, 1);
        if (g >= j->n_chunks) break;
        int chunk = (g % j->lanes) * j->per_lane + g / j->lanes;
        int base = chunk * OSWR_CHUNK, end = imin(base + OSWR_CHUNK, j->n_tris);       /* chunk ids past the mesh give an empty range */
        for (int i = base; i < end; ++i)
            process_triangle(&ds, j->vertices, j->indices, i, j->model, j->view, j->proj);
    }
}

static void* pool_main(void* p) {
    pool_arg* pa = (pool_arg*)p;
    oswr_pool* pool = pa->pool; int id = pa->id;
    free(pa);
    unsigned seen = 0;
    for (;;) {
        pthread_mutex_lock(&pool->mu);
        while (pool->gen == seen && !pool->shutdown) pthread_cond_wait(&pool->cv_job, &pool->mu);
        if (pool->shutdown) { pthread_mutex_unlock(&pool->mu); return NULL; }
        seen = pool->gen;
        pthread_mutex_unlock(&pool->mu);
        run_job(&pool->job, &pool->stats[id]);
        pthread_mutex_lock(&pool->mu);
        if (--pool->running == 0) pthread_cond_signal(&pool->cv_done);
        pthread_mutex_unlock(&pool->mu);
    }
}

static oswr_pool* pool_get(oswr_context* c) {
    if (c->pool) return c->pool;
    oswr_pool* pool = (oswr_pool*)calloc(1, sizeof(*pool));
    pool->n = c->n_threads; pool->ctx = c;
    pool->th = (pthread_t*)calloc((size_t)pool->n, sizeof(pthread_t));
    pool->stats = (oswr_stats*)calloc((size_t)pool->n, sizeof(oswr_stats));
    pthread_mutex_init(&pool->mu, NULL); pthread_cond_init(&pool->cv_job, NULL); pthread_cond_init(&pool->cv_done, NULL);
    for (int t = 0; t < pool->n; ++t) {
        pool_arg* pa = (pool_arg*)malloc(sizeof(*pa));
        pa->pool = pool; pa->id = t;
        pthread_create(&pool->th[t], NULL, pool_main, pa);
    }
    c->pool = pool;
    return pool;
}

static void pool_destroy(oswr_context* c) {
    oswr_pool* pool = c->pool;
    if (!pool) return;
    pthread_mutex_lock(&pool->mu);
    pool->shutdown = 1;
    pthread_cond_broadcast(&pool->cv_job);
    pthread_mutex_unlock(&pool->mu);
    for (int t = 0; t < pool->n; ++t) pthread_join(pool->th[t], NULL);
    pthread_mutex_destroy(&pool->mu); pthread_cond_destroy(&pool->cv_job); pthread_cond_destroy(&pool->cv_done);
    free(pool->th); free(pool->stats); free(pool);
    c->pool = NULL;
}

static void stats_add(oswr_stats* d, const oswr_stats* s) {
    d->triangles_in += s->triangles_in; d->triangles_setup += s->triangles_setup;
    d->triangles_clipped += s->triangles_clipped; d->fragments_tested += s->fragments_tested;
    d->fragments_shaded += s->fragments_shaded; d->fragments_written += s->fragments_written;
}

/* ---------- RenderMesh, Rasterizer.cs:163-230 ---------- */
int oswr_render_mesh(oswr_context* c,
                     const oswr_vertex_input* vertices, int n_vertices,
                     const uint16_t* indices, int n_indices,
                     const float model[16], const float view[16], const float projection[16],
                     int program, const oswr_uniforms* uniforms,
                     const uint8_t* tex, int tw, int th,
                     int cull_mode, int depth_test, int blend_mode) {
    if (c->width <= 0 || c->height <= 0) return 0;                                    /* :176 */
    init_tile_locks(c, c->width, c->height);                                          /* :178 */
    int n_tris = n_indices / 3;                                                       /* :180 */
    for (int i = 0; i < n_tris * 3; ++i) if ((int)indices[i] >= n_vertices) return -1;
    /* the avgDepth pre-pass (:184-197) has no observable effect (result discarded at :202) */
    draw_state ds;
    memset(&ds, 0, sizeof(ds));
    ds.ctx = c; ds.program = program; ds.uniforms = uniforms;
    ds.tex = (tex && tw > 0 && th > 0) ? tex : NULL; ds.tw = tw; ds.th = c->tex_bilinear ? -th : th;
    ds.cull = cull_mode; ds.depth_test = depth_test; ds.blend = blend_mode;

    if (c->n_threads <= 1) {
        ds.stats = &c->stats; ds.threaded = 0;
        for (int i = 0; i < n_tris; ++i) process_triangle(&ds, vertices, indices, i, model, view, projection);
        return 0;
    }
    oswr_pool* pool = pool_get(c);
    ds.threaded = 1;
    oswr_job* j = &pool->job;
    j->ds = ds; j->vertices = vertices; j->indices = indices; j->n_tris = n_tris;
    j->model = model; j->view = view; j->proj = projection;
    atomic_store(&j->next, 0);
    {
        /* chunk ids are laid out on a (lanes x per_lane) grid and grabbed column by column, so that simultaneous grabs are far
         * apart in the mesh; the grid can exceed the real chunk count by up to lanes - 1 ids, which give empty ranges */
        int real = (n_tris + OSWR_CHUNK - 1) / OSWR_CHUNK;
        j->lanes = pool->n < real ? pool->n : (real > 0 ? real : 1);
        j->per_lane = (real + j->lanes - 1) / j->lanes;
        j->n_chunks = j->lanes * j->per_lane;          /* grabs g in [0, n_chunks) */
    }
    memset(pool->stats, 0, sizeof(oswr_stats) * (size_t)pool->n);
    pthread_mutex_lock(&pool->mu);
    pool->running = pool->n;
    pool->gen++;
    pthread_cond_broadcast(&pool->cv_job);
    while (pool->running > 0) pthread_cond_wait(&pool->cv_done, &pool->mu);
    pthread_mutex_unlock(&pool->mu);
    for (int t = 0; t < pool->n; ++t) stats_add(&c->stats, &pool->stats[t]);
    return 0;
}

/* ---------- FrustumCuller.cs ---------- */
static inline float dist_sq3(const float a[3], const float b[3]) {       /* Vector3.DistanceSquared */
    float d[3] = { a[0] - b[0], a[1] - b[1], a[2] - b[2] };
    return vec3_dot(d, d);
}
void oswr_bounding_sphere(const oswr_vertex_input* v, int n, float out[4]) {          /* FrustumCuller.cs:59-151 */
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (n == 0) return;                                                               /* :65-66 */
    if (n == 1) { memcpy(out, v[0].position, 12); return; }                           /* :67-68 */
    float p0[3], p1[3], p2[3];
    memcpy(p0, v[0].position, 12); memcpy(p1, p0, 12);
    float max_sq = 0.0f;
    for (int i = 1; i < n; ++i) {                                                     /* :75-93 */
        float d = dist_sq3(v[i].position, p0);
        if (d > max_sq) { max_sq = d; memcpy(p1, v[i].position, 12); }
    }
    memcpy(p2, p1, 12); max_sq = 0.0f;
    for (int i = 0; i < n; ++i) {                                                     /* :98-116 */
        float d = dist_sq3(v[i].position, p1);
        if (d > max_sq) { max_sq = d; memcpy(p2, v[i].position, 12); }
    }
    float c[3] = { (p1[0] + p2[0]) * 0.5f, (p1[1] + p2[1]) * 0.5f, (p1[2] + p2[2]) * 0.5f };   /* :118 */
    float r = sqrtf(max_sq) * 0.5f;                                                   /* :119 */
    float nc[3] = { c[0], c[1], c[2] }, nr = r;
    int found = 0; float fpos[3] = { 0, 0, 0 }, fdist = 0.0f;
    for (int i = 0; i < n; ++i) {                                                     /* :124-131: local keeps the LAST one */
        float dist = sqrtf(dist_sq3(v[i].position, c));
        if (dist > r) { found = 1; fdist = dist; memcpy(fpos, v[i].position, 12); }
    }
    if (found && fdist > nr) {                                                        /* :133-147 */
        float upd = (nr + fdist) * 0.5f;
        float k = (upd - nr) / fdist;
        for (int j = 0; j < 3; ++j) { float t = (fpos[j] - nc[j]) * k; nc[j] = nc[j] + t; }
        nr = upd;
    }
    out[0] = nc[0]; out[1] = nc[1]; out[2] = nc[2]; out[3] = nr;
}

/* Plane ctor normalises again (FrustumCuller.cs:25-29) after NormalizePlane (:189-199) */
static void make_plane(float x, float y, float z, float w, float pl[4]) {
    float mag = sqrtf((x * x + y * y) + z * z);
    float n[3] = { x / mag, y / mag, z / mag }, nn[3];
    vec3_normalize(n, nn);
    pl[0] = nn[0]; pl[1] = nn[1]; pl[2] = nn[2]; pl[3] = w / mag;
}
int oswr_is_sphere_in_frustum(const float sphere[4], const float model[16], const float view[16], const float proj[16]) {
    float c4[4] = { sphere[0], sphere[1], sphere[2], 1.0f }, wc[4];
    vec4_transform(c4, model, wc);                                                    /* Vector3.Transform(center, model) :203 */
    const float* m = model;
    float s0 = sqrtf((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]);                       /* :204-209 */
    float s1 = sqrtf((m[4] * m[4] + m[5] * m[5]) + m[6] * m[6]);
    float s2 = sqrtf((m[8] * m[8] + m[9] * m[9]) + m[10] * m[10]);
    float max_scale = mathf_max(mathf_max(s0, s1), s2);
    float wr = sphere[3] * max_scale;                                                 /* :211 */
    float vp[16];                                                                     /* Matrix4x4.Multiply(view, proj) :212 */
    for (int i = 0; i < 4; ++i) vec4_transform(view + 4 * i, proj, vp + 4 * i);
    /* planes in the reference's test order: Left, Right, Top, Bottom, Near, Far (:213-218); M_rc = vp[(r-1)*4 + (c-1)] */
    const int col[6] = { 0, 0, 1, 1, 2, 2 };
    const float sgn[6] = { 1.0f, -1.0f, 1.0f, -1.0f, 1.0f, -1.0f };
    for (int k = 0; k < 6; ++k) {
        float co[4];
        for (int r = 0; r < 4; ++r) co[r] = sgn[k] > 0 ? vp[r * 4 + 3] + vp[r * 4 + col[k]] : vp[r * 4 + 3] - vp[r * 4 + col[k]];
        float pl[4];
        make_plane(co[0], co[1], co[2], co[3], pl);
        float dist = vec3_dot(pl, wc) + pl[3];                                        /* GetDistanceToPoint :31-34 */
        if (!(dist > -wr)) return 0;                                                  /* :221-224 */
    }
    return 1;
}
