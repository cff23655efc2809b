// RasterizerNative.cs -- the C# side of the drop-in boundary: P/Invoke binding of libswr_hip.so (include/swr.h) and the
// forwards that put it behind the reference's own API surface (OCSYT/SoftwareRenderer, net9.0).
//
// A maintainer adds this ONE file to the project and
//   * renames the body of Rasterizer.RenderMesh (Rasterizer.cs:163-230) to RenderMeshManaged (kept as the path for
//     shader delegates the backend has no built-in program for) and makes `Rasterizer` a `partial` class;
//   * replaces the six accessor bodies of MainWindow (MainWindow.cs:382-436) and its buffer allocation
//     (MainWindow.cs:320-321) with the one-line forwards of `MainWindowNative` below;
//   * lets `Texture` (Texture.cs:31-68) hold a `TextureNative` next to its Image<Rgba32>.
// Renderer.cs, Camera.cs, FrustumCuller.cs, ModelLoader.cs, Material.cs, Light.cs stay as they are: RenderMesh keeps its
// exact signature and defaults, and the one shader pair the application uses (Renderer.VertexShader / FragmentShader,
// Renderer.cs:830-860) is recognised and mapped to SWR_PROG_DUST2_LAMBERT_FOG with its closed-over fields as uniforms.
//
// There is no .NET toolchain in the build image, so this file is NOT compiled there; tests/test_abi.py parses it and
// checks it mechanically against include/swr.h and the ctypes binding (every entry point present with the right arity,
// struct field order and sizes).  The compiled, tested callers of the same ABI are softwarerenderer_amd/rasterizer.py
// (ctypes), softwarerenderer_amd/cpp/Rasterizer.hpp (C++) and tests/c/abi_smoke.c (plain C, dlopen).
using System;
using System.Collections.Generic;
using System.Numerics;
using System.Reflection;
using System.Runtime.CompilerServices;
using System.Runtime.InteropServices;
using SixLabors.ImageSharp;
using SixLabors.ImageSharp.PixelFormats;

namespace SoftwareRenderer
{
    // ---------------------------------------------------------------- structs of include/swr.h ----
    [StructLayout(LayoutKind.Sequential)]
    public struct SwrPointLight            // swr_point_light (subset of Light.cs:9-17 used by PHONG_4POINT)
    {
        public Vector3 Position;
        public float Range;
        public Vector3 Color;
        public float Intensity;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SwrUniforms              // swr_uniforms: the fields Renderer.FragmentShader closes over, Renderer.cs:39-44
    {
        public Vector3 LightDirection;
        public float Pad0;
        public Vector4 LightColor;
        public Vector4 FogColor;
        public float FogStart;
        public float FogEnd;
        public float Shininess;
        public float Pad1;
        public Vector3 CameraPosition;
        public float Pad2;
        public SwrPointLight Light0;
        public SwrPointLight Light1;
        public SwrPointLight Light2;
        public SwrPointLight Light3;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SwrStats                 // swr_stats
    {
        public ulong TrianglesIn;
        public ulong TrianglesSetup;
        public ulong TrianglesClipped;
        public ulong FragmentsTested;
        public ulong FragmentsShaded;
        public ulong FragmentsWritten;
        public ulong TilePairs;
        public ulong Flushes;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SwrProfile               // swr_profile
    {
        public double VertexMs;
        public double SetupMs;
        public double BinMs;
        public double SortMs;
        public double CoverMs;
        public double RasterMs;
        public double ClearMs;
        public double TotalMs;
        public ulong RasterLaunches;
        public ulong Flushes;
    }

    public enum SwrProgram { FlatColor = 0, Gouraud = 1, Dust2LambertFog = 2, Phong4Point = 3 }

    // ---------------------------------------------------------------- the 49 entry points ----
    // Shaders.VertexInput (Shaders.cs:10-24) IS swr_vertex: four sequential System.Numerics fields, 48 bytes, blittable.
    // Matrix4x4 is 16 sequential floats M11..M44 (row-major, row-vector convention): passed by address, no marshalling.
    internal static unsafe class Native
    {
        const string Lib = "swr_hip";      // libswr_hip.so
        [DllImport(Lib)] public static extern int swr_abi_version();
        [DllImport(Lib)] public static extern IntPtr swr_build_info();
        [DllImport(Lib)] public static extern int swr_numerics_mode(out int fma, out int dotOrder);
        [DllImport(Lib)] public static extern IntPtr swr_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_create(int deviceId, out IntPtr ctx);
        [DllImport(Lib)] public static extern void swr_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_resize(IntPtr ctx, int width, int height);
        [DllImport(Lib)] public static extern int swr_set_band(IntPtr ctx, int firstTileRow, int nTileRows);
        [DllImport(Lib)] public static extern int swr_set_band_interleaved(IntPtr ctx, int rank, int world, int stripeTileRows);
        [DllImport(Lib)] public static extern int swr_bind_framebuffer(IntPtr ctx, IntPtr colorDevicePtr, IntPtr depthDevicePtr);
        [DllImport(Lib)] public static extern int swr_set_stream(IntPtr ctx, IntPtr hipStream);
        [DllImport(Lib)] public static extern int swr_clear_color(IntPtr ctx, Vector4* rgba);
        [DllImport(Lib)] public static extern int swr_clear_depth(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_get_pixel(IntPtr ctx, int x, int y, Vector4* rgba);
        [DllImport(Lib)] public static extern int swr_set_pixel(IntPtr ctx, int x, int y, Vector4* rgba);
        [DllImport(Lib)] public static extern int swr_get_depth(IntPtr ctx, int x, int y, float* depth);
        [DllImport(Lib)] public static extern int swr_set_depth(IntPtr ctx, int x, int y, float depth);
        [DllImport(Lib)] public static extern int swr_readback(IntPtr ctx, Vector4* colorRgba, float* depth);
        [DllImport(Lib)] public static extern int swr_readback_rgb(IntPtr ctx, Vector3* rgb);
        [DllImport(Lib)] public static extern int swr_flatten_rgb_device(IntPtr ctx, IntPtr deviceRgb);
        [DllImport(Lib)] public static extern int swr_flatten_rgb_device_async(IntPtr ctx, IntPtr deviceRgb);
        [DllImport(Lib)] public static extern int swr_replay_count(IntPtr ctx, out ulong replays);
        [DllImport(Lib)] public static extern int swr_sync_count(IntPtr ctx, out ulong syncs);
        [DllImport(Lib)] public static extern int swr_host_register(IntPtr ctx, void* ptr, nuint bytes);
        [DllImport(Lib)] public static extern int swr_host_unregister(IntPtr ctx, void* ptr);
        [DllImport(Lib)] public static extern int swr_upload(IntPtr ctx, Vector4* colorRgba, float* depth);
        [DllImport(Lib)] public static extern int swr_color_device_ptr(IntPtr ctx, out IntPtr devicePtr);
        [DllImport(Lib)] public static extern int swr_depth_device_ptr(IntPtr ctx, out IntPtr devicePtr);
        [DllImport(Lib)] public static extern int swr_texture_create(IntPtr ctx, byte* rgba8, int width, int height, out IntPtr texture);
        [DllImport(Lib)] public static extern int swr_texture_destroy(IntPtr ctx, IntPtr texture);
        [DllImport(Lib)] public static extern int swr_texture_set_filter(IntPtr ctx, IntPtr texture, int bilinear);
        [DllImport(Lib)] public static extern int swr_texture_sample(IntPtr ctx, IntPtr texture, Vector2* uv, int n, Vector4* outRgba);
        [DllImport(Lib)] public static extern int swr_mesh_create(IntPtr ctx, Shaders.VertexInput* vertices, int nVertices, ushort* indices, int nIndices, out IntPtr mesh);
        [DllImport(Lib)] public static extern int swr_mesh_destroy(IntPtr ctx, IntPtr mesh);
        [DllImport(Lib)] public static extern int swr_set_state(IntPtr ctx, float nearClip, float farClip, int debugMode);
        [DllImport(Lib)] public static extern int swr_initialize_tile_locks(IntPtr ctx, int width, int height);
        [DllImport(Lib)] public static extern int swr_render_mesh(IntPtr ctx, IntPtr mesh, Matrix4x4* model, Matrix4x4* view, Matrix4x4* projection, int program, SwrUniforms* uniforms, IntPtr texture, int cullMode, int depthTest, int blendMode);
        [DllImport(Lib)] public static extern int swr_render_mesh_arrays(IntPtr ctx, Shaders.VertexInput* vertices, int nVertices, ushort* indices, int nIndices, Matrix4x4* model, Matrix4x4* view, Matrix4x4* projection, int program, SwrUniforms* uniforms, IntPtr texture, int cullMode, int depthTest, int blendMode);
        [DllImport(Lib)] public static extern int swr_mesh_bounds(IntPtr ctx, IntPtr mesh, Vector4* centerRadius);
        [DllImport(Lib)] public static extern int swr_is_sphere_in_frustum(IntPtr ctx, Vector4* centerRadius, Matrix4x4* model, Matrix4x4* view, Matrix4x4* projection, out int inside);
        [DllImport(Lib)] public static extern int swr_render_mesh_culled(IntPtr ctx, IntPtr mesh, Matrix4x4* model, Matrix4x4* view, Matrix4x4* projection, int program, SwrUniforms* uniforms, IntPtr texture, int cullMode, int depthTest, int blendMode);
        [DllImport(Lib)] public static extern int swr_flush(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_sync(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_interpolate(IntPtr ctx, float* verts60, float* w, int n, int interpolate, float* outRecords);
        [DllImport(Lib)] public static extern int swr_get_stats(IntPtr ctx, out SwrStats stats);
        [DllImport(Lib)] public static extern int swr_reset_stats(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_profile_enable(IntPtr ctx, int on);
        [DllImport(Lib)] public static extern int swr_profile_get(IntPtr ctx, out SwrProfile profile);
        [DllImport(Lib)] public static extern int swr_profile_reset(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_device_name(IntPtr ctx, byte* buf, int buflen);
        [DllImport(Lib)] public static extern int swr_selftest_division(IntPtr ctx, ulong samples, ulong seed, ulong* out8);
        [DllImport(Lib)] public static extern int swr_debug_counters(IntPtr ctx, ulong* out8);
    }

    // ---------------------------------------------------------------- context ----
    public static class SwrContext
    {
        static IntPtr ctx;
        static readonly object createLock = new object();

        public static IntPtr Handle
        {
            get
            {
                if (ctx != IntPtr.Zero) return ctx;
                lock (createLock)
                {
                    if (ctx != IntPtr.Zero) return ctx;
                    if (Native.swr_abi_version() != 2) throw new InvalidOperationException("libswr_hip.so: ABI version mismatch");
                    int rc = Native.swr_create(0, out IntPtr c);       // one context drives one GPU; there is NO CPU fallback
                    if (rc != 0) throw new InvalidOperationException($"swr_create failed ({rc}): {Marshal.PtrToStringAnsi(Native.swr_last_error(IntPtr.Zero))}");
                    ctx = c;
                    return ctx;
                }
            }
        }

        // status codes of swr.h: -1 maps to the reference's own exception type (Rasterizer.cs:71-74)
        public static void Check(int rc)
        {
            if (rc == 0) return;
            string msg = Marshal.PtrToStringAnsi(Native.swr_last_error(ctx)) ?? "";
            if (rc == -1 && msg.Contains("index out of range")) throw new IndexOutOfRangeException(msg);
            if (rc == -1) throw new ArgumentException(msg);
            if (rc == -3) throw new OutOfMemoryException(msg);
            throw new InvalidOperationException($"swr error {rc}: {msg}");
        }
    }

    // ---------------------------------------------------------------- Rasterizer ----
    public static partial class Rasterizer
    {
        // Retained device copies of ModelLoader meshes (Mesh.Vertices / Mesh.Indices, ModelLoader.cs:45-47), keyed by the
        // Mesh object: uploaded on first use, freed with the Mesh.  The array overload below cannot retain anything
        // (Renderer.cs:452-453 passes a fresh ToArray() copy every frame) and uploads per call.
        static readonly ConditionalWeakTable<Mesh, MeshHandle> retained = new ConditionalWeakTable<Mesh, MeshHandle>();

        sealed class MeshHandle
        {
            public IntPtr Ptr;
            ~MeshHandle() { if (Ptr != IntPtr.Zero) Native.swr_mesh_destroy(SwrContext.Handle, Ptr); }
        }

        /// Rasterizer.RenderMesh with the reference's exact signature and defaults (Rasterizer.cs:163-174).
        public static unsafe void RenderMesh(
            MainWindow window,
            Shaders.VertexInput[] vertices,
            ushort[] indices,
            Matrix4x4 model,
            Matrix4x4 view,
            Matrix4x4 projection,
            Shaders.VertexShader vertexShader,
            Shaders.FragmentShader fragmentShader,
            CullMode cullMode = CullMode.Back,
            DepthTest depthTest = DepthTest.LessEqual,
            BlendMode blendMode = BlendMode.Alpha)
        {
            if (window.RenderWidth <= 0 || window.RenderHeight <= 0) return;                    // Rasterizer.cs:176
            if (!ShaderMap.TryResolve(vertexShader, fragmentShader, out SwrProgram program, out SwrUniforms uniforms, out IntPtr texture))
            {
                // a shader pair the backend has no built-in program for: the reference's own CPU path
                RenderMeshManaged(window, vertices, indices, model, view, projection, vertexShader, fragmentShader, cullMode, depthTest, blendMode);
                return;
            }
            IntPtr ctx = SwrContext.Handle;
            SwrContext.Check(Native.swr_initialize_tile_locks(ctx, window.RenderWidth, window.RenderHeight));   // :178 (argument check)
            SwrContext.Check(Native.swr_set_state(ctx, NearClip, FarClip, (int)RenderDebugMode));               // :20-22
            fixed (Shaders.VertexInput* v = vertices)
            fixed (ushort* idx = indices)
            {
                SwrContext.Check(Native.swr_render_mesh_arrays(ctx, v, vertices.Length, idx, indices.Length, &model, &view, &projection,
                                                               (int)program, &uniforms, texture, (int)cullMode, (int)depthTest, (int)blendMode));
            }
        }

        /// Same draw from a retained ModelLoader mesh: no per-frame upload (what Renderer.RenderDust2 can call with `mesh`
        /// instead of `mesh.Vertices.ToArray(), mesh.Indices.ToArray()`); frustumCull folds the
        /// `if (!FrustumCuller.IsSphereInFrustum(mesh.SphereBounds, ...)) return;` of Renderer.cs:446 into the GPU batch.
        public static unsafe void RenderMesh(
            MainWindow window,
            Mesh mesh,
            Matrix4x4 model,
            Matrix4x4 view,
            Matrix4x4 projection,
            Shaders.VertexShader vertexShader,
            Shaders.FragmentShader fragmentShader,
            CullMode cullMode = CullMode.Back,
            DepthTest depthTest = DepthTest.LessEqual,
            BlendMode blendMode = BlendMode.Alpha,
            bool frustumCull = false)
        {
            if (window.RenderWidth <= 0 || window.RenderHeight <= 0) return;
            if (!ShaderMap.TryResolve(vertexShader, fragmentShader, out SwrProgram program, out SwrUniforms uniforms, out IntPtr texture))
            {
                RenderMeshManaged(window, mesh.Vertices.ToArray(), mesh.Indices.ToArray(), model, view, projection, vertexShader, fragmentShader, cullMode, depthTest, blendMode);
                return;
            }
            IntPtr ctx = SwrContext.Handle;
            MeshHandle h = retained.GetValue(mesh, m =>
            {
                var va = m.Vertices.ToArray();
                var ia = m.Indices.ToArray();
                var mh = new MeshHandle();
                fixed (Shaders.VertexInput* v = va)
                fixed (ushort* idx = ia)
                    SwrContext.Check(Native.swr_mesh_create(ctx, v, va.Length, idx, ia.Length, out mh.Ptr));
                return mh;
            });
            SwrContext.Check(Native.swr_set_state(ctx, NearClip, FarClip, (int)RenderDebugMode));
            int rc = frustumCull
                ? Native.swr_render_mesh_culled(ctx, h.Ptr, &model, &view, &projection, (int)program, &uniforms, texture, (int)cullMode, (int)depthTest, (int)blendMode)
                : Native.swr_render_mesh(ctx, h.Ptr, &model, &view, &projection, (int)program, &uniforms, texture, (int)cullMode, (int)depthTest, (int)blendMode);
            SwrContext.Check(rc);
        }
    }

    // Delegates cannot cross the ABI.  The application has exactly one shader pair (Renderer.VertexShader and the lambda
    // `input => FragmentShader(input, texture)`, Renderer.cs:450-459,830-860): it is recognised by its methods, and the
    // fields it closes over -- the Renderer's light / fog fields (Renderer.cs:39-44) and the captured Texture -- are read
    // through reflection, so Renderer.cs needs no edit.  Anything else returns false (managed path).
    static class ShaderMap
    {
        static readonly BindingFlags Any = BindingFlags.Instance | BindingFlags.Public | BindingFlags.NonPublic;

        public static bool TryResolve(Shaders.VertexShader vs, Shaders.FragmentShader fs, out SwrProgram program, out SwrUniforms uniforms, out IntPtr texture)
        {
            program = SwrProgram.Dust2LambertFog; uniforms = default; texture = IntPtr.Zero;
            if (vs?.Target is not Renderer renderer || vs.Method.Name != "VertexShader") return false;
            object closure = fs?.Target;
            if (closure == null) return false;
            Texture captured = null;
            bool sameRenderer = ReferenceEquals(closure, renderer);
            foreach (FieldInfo f in closure.GetType().GetFields(Any))
            {
                object v = f.GetValue(closure);
                if (v is Texture t) captured = t;
                if (ReferenceEquals(v, renderer)) sameRenderer = true;
            }
            if (!sameRenderer) return false;                                   // a fragment lambda of some other object
            uniforms.LightDirection = Get<Vector3>(renderer, "LightDirection");
            uniforms.LightColor = Get<Vector4>(renderer, "LightColor");
            uniforms.FogColor = Get<Vector4>(renderer, "FogColor");
            uniforms.FogStart = Get<float>(renderer, "FogStart");
            uniforms.FogEnd = Get<float>(renderer, "FogEnd");
            texture = captured?.Native?.Handle ?? IntPtr.Zero;                 // null texture -> Vector4.One (Renderer.cs:852)
            return true;
        }

        static T Get<T>(object o, string field) => (T)o.GetType().GetField(field, Any).GetValue(o);
    }

    // ---------------------------------------------------------------- MainWindow forwards ----
    // MainWindow.ColorBuffer / DepthBuffer (MainWindow.cs:30-31) live in HBM.  The bodies of the accessors become:
    //   SetPixel(x, y, c)      -> MainWindowNative.SetPixel(x, y, c);             (MainWindow.cs:382-388)
    //   GetPixel(x, y)         -> return MainWindowNative.GetPixel(x, y);         (:391-398, Vector4.Zero out of bounds)
    //   ClearColorBuffer(c)    -> MainWindowNative.ClearColorBuffer(c);           (:400-407; fused into the tile pass)
    //   SetDepth(x, y, d)      -> MainWindowNative.SetDepth(x, y, d);             (:411-417)
    //   GetDepth(x, y)         -> return MainWindowNative.GetDepth(x, y);         (:420-426, float.MinValue out of bounds)
    //   ClearDepthBuffer()     -> MainWindowNative.ClearDepthBuffer();            (:429-436)
    //   HandleResize           -> MainWindowNative.Resize(RenderWidth, RenderHeight) instead of allocating the arrays (:320-321)
    //   OnRender               -> MainWindowNative.Present(flatColorBuffer) instead of the Vector4 -> Vector3 loop (:234-240)
    public static unsafe class MainWindowNative
    {
        public static void Resize(int renderWidth, int renderHeight) => SwrContext.Check(Native.swr_resize(SwrContext.Handle, renderWidth, renderHeight));
        public static void SetPixel(int x, int y, Vector4 color) => SwrContext.Check(Native.swr_set_pixel(SwrContext.Handle, x, y, &color));
        public static Vector4 GetPixel(int x, int y) { Vector4 c; SwrContext.Check(Native.swr_get_pixel(SwrContext.Handle, x, y, &c)); return c; }
        public static void ClearColorBuffer(Vector4 clearColor) => SwrContext.Check(Native.swr_clear_color(SwrContext.Handle, &clearColor));
        public static void SetDepth(int x, int y, float depth) => SwrContext.Check(Native.swr_set_depth(SwrContext.Handle, x, y, depth));
        public static float GetDepth(int x, int y) { float d; SwrContext.Check(Native.swr_get_depth(SwrContext.Handle, x, y, &d)); return d; }
        public static void ClearDepthBuffer() => SwrContext.Check(Native.swr_clear_depth(SwrContext.Handle));

        /// The present payload of MainWindow.OnRender: executes the recorded draws and fills the caller's Vector3[] (RGB float,
        /// flattened on the GPU).  Register a long-lived pinned array once with Pin() and the copy runs at PCIe rate.
        public static void Present(Vector3[] flatColorBuffer)
        {
            fixed (Vector3* p = flatColorBuffer) SwrContext.Check(Native.swr_readback_rgb(SwrContext.Handle, p));
        }
        /// Full-precision read-back into the reference's own arrays (tools, screenshots).
        public static void Readback(Vector4[] colorBuffer, float[] depthBuffer)
        {
            fixed (Vector4* c = colorBuffer) fixed (float* d = depthBuffer) SwrContext.Check(Native.swr_readback(SwrContext.Handle, c, d));
        }
        public static GCHandle Pin(Array longLivedBuffer, nuint bytes)
        {
            GCHandle h = GCHandle.Alloc(longLivedBuffer, GCHandleType.Pinned);
            SwrContext.Check(Native.swr_host_register(SwrContext.Handle, (void*)h.AddrOfPinnedObject(), bytes));
            return h;
        }
        public static SwrStats Stats() { SwrContext.Check(Native.swr_get_stats(SwrContext.Handle, out SwrStats s)); return s; }
    }

    // ---------------------------------------------------------------- Texture ----
    // Texture(Image<Rgba32>) / Width / Height / Sample / Dispose (Texture.cs:31-68).  `Texture` keeps its public surface and
    // gains `internal TextureNative Native` created in its constructor; Sample(uv) stays available on the host (one call per
    // sample: tools only -- the fragment programs sample on the GPU).
    public sealed unsafe class TextureNative : IDisposable
    {
        public IntPtr Handle { get; private set; }
        public int Width { get; }
        public int Height { get; }

        public TextureNative(Image<Rgba32> image)
        {
            Width = image.Width; Height = image.Height;
            var pixels = new byte[Width * Height * 4];
            image.CopyPixelDataTo(pixels);                                     // RGBA8 row-major, as Texture.cs:55-56 indexes it
            fixed (byte* p = pixels)
            {
                SwrContext.Check(Native.swr_texture_create(SwrContext.Handle, p, Width, Height, out IntPtr h));
                Handle = h;
            }
        }

        public Vector4 Sample(Vector2 uv)                                       // Texture.cs:43-63: nearest, wrap, byte * (1f / 255f)
        {
            Vector4 o;
            SwrContext.Check(Native.swr_texture_sample(SwrContext.Handle, Handle, &uv, 1, &o));
            return o;
        }

        public void Dispose()                                                   // Texture.cs:65-68
        {
            if (Handle == IntPtr.Zero) return;
            Native.swr_texture_destroy(SwrContext.Handle, Handle);
            Handle = IntPtr.Zero;
        }
    }

    // FrustumCuller.CalculateBoundingSphere / IsSphereInFrustum stay managed (FrustumCuller.cs is untouched); the GPU versions
    // (swr_mesh_bounds, swr_is_sphere_in_frustum -- bit-identical, tests/test_gpu_api.py) are reachable through
    // Rasterizer.RenderMesh(window, mesh, ..., frustumCull: true).
}
