/*
 * swr_oracle.h -- CPU oracle for the rasterizer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm in the C# reference
 * (OCSYT/SoftwareRenderer: Rasterizer.cs, Shaders.cs, Texture.cs:43-63,
 * Renderer.cs:830-860, MainWindow.cs:378-436).  It exists to CHECK the HIP
 * backend; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product library (libswr_hip.so) neither links nor calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 *  and its C# cannot be built in this image (no .NET).
 * This restatement is therefore pinned only by hand-derivable known-answer
 * tests (tests/test_oracle_kat.py) -- not by outputs of the reference itself.
 * System.Numerics (.NET 9 shared framework, patch version unpinned by the
 * reference) is restated from its published semantics:
 *   Vector4.Transform(v,M) = ((x*row1 + y*row2) + z*row3) + w*row4
 *   Vector3.TransformNormal = (x*row1 + y*row2) + z*row3
 *   Vector3.Normalize(v)    = v / sqrt((x*x + y*y) + z*z)
 *   VectorN.Lerp(a,b,t)     = a*(1-t) + b*t
 * unfused by default; build with -DSWR_NUMERICS_FMA=1 to model a fused
 * MultiplyAddEstimate in Transform/TransformNormal/Lerp, and with -DSWR_DOT_PAIRWISE=1|2
 * to model the two SIMD summation orders of Vector3.Dot / LengthSquared
 * ((xx + yy) + (zz + 0) as dpps sums, (xx + zz) + (yy + 0) as two shuffle-adds do).
 * oracle/Makefile builds every combination the HIP side is tested in.
 */
#ifndef SWR_ORACLE_H
#define SWR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Shaders.VertexInput, Shaders.cs:10-24 : 48 bytes, sequential floats */
typedef struct {
    float position[3];
    float uv[2];
    float normal[3];
    float color[4];
} oswr_vertex_input;

/* Shaders.VertexOutput, Shaders.cs:26-47. Data dictionary is modelled as the one
 * key the reference ever stores ("WorldNormal" -> Vector3, Renderer.cs:840). */
typedef struct {
    float clip[4];
    float color[4];
    float texcoord[2];
    float normal[3];
    float screen[2];
    float world_normal[3];
    float world_pos[4];  /* Data["WorldPos"] (Vector4) -- only the build-defined PHONG_4POINT program stores it */
    int   has_data;      /* Data != null */
    int   interpolate;
    float barycentric[3];
} oswr_vertex_output;

/* enums keep the reference's ordinals, Rasterizer.cs:14-50 */
enum { OSWR_DEBUG_NONE = 0, OSWR_DEBUG_WIREFRAME = 1 };
enum { OSWR_BLEND_NONE = 0, OSWR_BLEND_ALPHA = 1, OSWR_BLEND_ADDITIVE = 2, OSWR_BLEND_MULTIPLY = 3 };
enum { OSWR_DEPTH_DISABLED = 0, OSWR_DEPTH_LESS = 1, OSWR_DEPTH_LESSEQUAL = 2, OSWR_DEPTH_GREATER = 3,
       OSWR_DEPTH_GREATEREQUAL = 4, OSWR_DEPTH_EQUAL = 5, OSWR_DEPTH_NOTEQUAL = 6, OSWR_DEPTH_ALWAYS = 7 };
enum { OSWR_CULL_NONE = 0, OSWR_CULL_BACK = 1, OSWR_CULL_FRONT = 2 };

/* built-in shader programs standing in for the C# delegates */
enum {
    OSWR_PROG_FLAT_COLOR = 0,        /* VS = Renderer.cs:830-846 with Interpolate=false; FS returns input.Color */
    OSWR_PROG_GOURAUD = 1,           /* same VS, Interpolate=true; FS returns input.Color */
    OSWR_PROG_DUST2_LAMBERT_FOG = 2, /* exactly Renderer.cs:830-860 */
    OSWR_PROG_PHONG_4POINT = 3,      /* build-defined (no reference semantics): see swr_oracle.c */
    OSWR_PROG_DEBUG_VARYINGS = 4     /* build-defined: returns the varyings no other built-in reads -- Normal, ScreenCoords, Barycentric */
};

typedef struct {
    float position[3]; float range;      /* Light.cs fields used by the build-defined Phong program */
    float color[3];    float intensity;
} oswr_point_light;

/* uniforms of Renderer.FragmentShader, Renderer.cs:39-44 */
typedef struct {
    float light_direction[3]; float _pad0;
    float light_color[4];
    float fog_color[4];
    float fog_start, fog_end;
    float shininess; float _pad1;
    float camera_position[3]; float _pad2;
    oswr_point_light lights[4];
} oswr_uniforms;

typedef struct {
    uint64_t triangles_in;       /* index triples submitted */
    uint64_t triangles_setup;    /* triangles that reached RasterizeTriangle's tile loop */
    uint64_t triangles_clipped;  /* went through ClipTriangleAgainstNearPlane */
    uint64_t fragments_tested;   /* passed the coverage test (Rasterizer.cs:493-496) */
    uint64_t fragments_shaded;   /* passed the depth test */
    uint64_t fragments_written;  /* passed alpha (Rasterizer.cs:511) */
} oswr_stats;

typedef struct oswr_context oswr_context;

oswr_context* oswr_create(int width, int height);
void  oswr_destroy(oswr_context* c);
int   oswr_resize(oswr_context* c, int width, int height);
void  oswr_set_state(oswr_context* c, float near_clip, float far_clip, int debug_mode);
void  oswr_set_threads(oswr_context* c, int n_threads);  /* 1 = deterministic serial order (oracle of record) */
void  oswr_clear_color(oswr_context* c, const float rgba[4]);
void  oswr_clear_depth(oswr_context* c);
float* oswr_color_buffer(oswr_context* c);   /* W*H*4 floats, idx = y*W + x  (MainWindow.cs:379) */
float* oswr_depth_buffer(oswr_context* c);   /* W*H floats */
int   oswr_width(oswr_context* c);
int   oswr_height(oswr_context* c);

/* Rasterizer.RenderMesh, Rasterizer.cs:163-230. texture may be NULL (-> white, Renderer.cs:852).
 * Returns 0, or -1 when an index is out of range (C# would throw IndexOutOfRangeException). */
int   oswr_render_mesh(oswr_context* c,
                       const oswr_vertex_input* vertices, int n_vertices,
                       const uint16_t* indices, int n_indices,
                       const float model[16], const float view[16], const float projection[16],
                       int program, const oswr_uniforms* uniforms,
                       const uint8_t* texture_rgba8, int tex_width, int tex_height,
                       int cull_mode, int depth_test, int blend_mode);

void  oswr_get_stats(oswr_context* c, oswr_stats* out);
void  oswr_reset_stats(oswr_context* c);

/* pieces exposed for unit tests */
void  oswr_vertex_shader(const oswr_vertex_input* in, const float model[16], const float view[16],
                         const float projection[16], int program, oswr_vertex_output* out);
void  oswr_interpolate(const oswr_vertex_output* a, const oswr_vertex_output* b, const oswr_vertex_output* c,
                       float w0, float w1, float w2, int interpolate, oswr_vertex_output* out);
void  oswr_lerp(const oswr_vertex_output* a, const oswr_vertex_output* b, float t, int interpolate,
                oswr_vertex_output* out);
void  oswr_texture_sample(const uint8_t* rgba8, int w, int h, const float uv[2], float out[4]);
/* BUILD-DEFINED bilinear filter with wrap (no reference semantics: Texture.Sample is nearest) */
void  oswr_texture_sample_bilinear(const uint8_t* rgba8, int w, int h, const float uv[2], float out[4]);
void  oswr_set_texture_filter(oswr_context* c, int bilinear);   /* applies to the textures of subsequent oswr_render_mesh calls */
int   oswr_fragment_shader(int program, const oswr_uniforms* u, const oswr_vertex_output* in,
                           const uint8_t* tex, int tw, int th, float out[4]);
void  oswr_blend(const float src[4], const float dst[4], int mode, float out[4]);
int   oswr_depth_func(int test, float new_depth, float old_depth);
float oswr_edge_function(const float a[2], const float b[2], const float c[2]);
int   oswr_numerics_fma(void);
int   oswr_dot_pairwise(void);   /* SWR_DOT_PAIRWISE of this build: 0 sequential, 1 dpps order, 2 two shuffle-adds */
/* Vector4.Transform / VectorN.Lerp / Vector3.Dot as THIS build models them (numerics start-up probe, tools/make_numerics_probe.py) */
void  oswr_nm_transform4(const float v[4], const float m[16], float out[4]);
void  oswr_nm_transform_normal3(const float n[3], const float m[16], float out[3]);
/* run-time half of the model: do Vector4.Transform (+ Vector3.Transform, Matrix4x4.Multiply) / Vector3.TransformNormal fuse their
 * multiply-adds?  Per library instance; default = oswr_numerics_fma() for both (mirrors swr_set_transform_fma of include/swr.h) */
void  oswr_set_transform_fma(int transform_fused, int transform_normal_fused);
void  oswr_get_transform_fma(int* transform_fused, int* transform_normal_fused);
float oswr_nm_lerp(float a, float b, float t);
float oswr_nm_dot3(const float a[3], const float b[3]);

/* FrustumCuller.cs (row N3 of SURVEY.md section 8f) */
/* CalculateBoundingSphere, FrustumCuller.cs:59-151, in the serial schedule of its Parallel.For loops (one partition:
 * the third pass then applies ONE update with the LAST vertex found outside the first sphere). out = {cx,cy,cz,r} */
void  oswr_bounding_sphere(const oswr_vertex_input* vertices, int n_vertices, float out[4]);
/* IsSphereInFrustum, FrustumCuller.cs:201-218 */
int   oswr_is_sphere_in_frustum(const float sphere[4], const float model[16], const float view[16], const float projection[16]);

#ifdef __cplusplus
}
#endif
#endif
