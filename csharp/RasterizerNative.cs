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

    public enum SwrProgram { FlatColor = 0, Gouraud = 1, Dust2LambertFog = 2, Phong4Point = 3, DebugVaryings = 4 }

    // ---------------------------------------------------------------- the 59 entry points ----
    // Shaders.VertexInput (Shaders.cs:10-24) IS swr_vertex: four sequential System.Numerics fields, 48 bytes, blittable.
    // Matrix4x4 is 16 sequential floats M11..M44 (row-major, row-vector convention): passed by address, no marshalling.
    internal static unsafe class Native
    {
        const string Lib = "swr_hip";      // libswr_hip.so
        [DllImport(Lib)] public static extern int swr_abi_version();
        [DllImport(Lib)] public static extern IntPtr swr_build_info();
        [DllImport(Lib)] public static extern int swr_numerics_mode(out int fma, out int dotOrder);
        [DllImport(Lib)] public static extern int swr_set_transform_fma(IntPtr ctx, int transformFused, int transformNormalFused);
        [DllImport(Lib)] public static extern int swr_get_transform_fma(IntPtr ctx, out int transformFused, out int transformNormalFused);
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
        [DllImport(Lib)] public static extern int swr_present_rgb_async(IntPtr ctx, Vector3* rgb, out ulong ticket);
        [DllImport(Lib)] public static extern int swr_present_wait(IntPtr ctx, ulong ticket);
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
        [DllImport(Lib)] public static extern int swr_set_pipelining(IntPtr ctx, int mode);
        [DllImport(Lib)] public static extern int swr_get_pipelining(IntPtr ctx, out int mode);
        [DllImport(Lib)] public static extern int swr_sync(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_interpolate(IntPtr ctx, float* verts60, float* w, int n, int interpolate, float* outRecords);
        [DllImport(Lib)] public static extern int swr_get_stats(IntPtr ctx, out SwrStats stats);
        [DllImport(Lib)] public static extern int swr_reset_stats(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_profile_enable(IntPtr ctx, int on);
        [DllImport(Lib)] public static extern int swr_profile_get(IntPtr ctx, out SwrProfile profile);
        [DllImport(Lib)] public static extern int swr_profile_reset(IntPtr ctx);
        [DllImport(Lib)] public static extern int swr_profile_raster_samples(IntPtr ctx, float* outMs, int capacity, out int n);
        [DllImport(Lib)] public static extern int swr_device_name(IntPtr ctx, byte* buf, int buflen);
        [DllImport(Lib)] public static extern int swr_selftest_division(IntPtr ctx, ulong samples, ulong seed, ulong* out8);
        [DllImport(Lib)] public static extern int swr_debug_counters(IntPtr ctx, ulong* out8);
    }

    // ---------------------------------------------------------------- numerics probe ----
    // Nothing in the reference's repository pins how .NET 9 evaluates Vector4.Transform / Vector4.Lerp (with fused multiply-adds or
    // without) and Vector3.Dot (in which order the lanes are summed), and a third of a real frame's depth words depend on the first
    // question (DESIGN.md section 3).  The three fused-or-not questions (Transform, TransformNormal, Lerp) and the dot order are
    // modelled INDEPENDENTLY: Lerp x dot order picks one of six builds (libswr_hip.so = unfused + sequential, _fma, _dotpw,
    // _fma_dotpw, _dpps, _fma_dpps), Transform / TransformNormal are run-time flags of the context (swr_set_transform_fma).  At
    // start-up the operations are evaluated with the RUNNING System.Numerics on operands on which the models give different float32
    // bits; the library built for the observed (Lerp, Dot) model is the one `swr_hip` resolves to and the context gets the observed
    // Transform flags: all 24 combinations are served.  Only a bit pattern that is NEITHER model's throws.
    public static class NumericsProbe
    {
        // <generated by tools/make_numerics_probe.py -- do not edit; tests/test_abi.py compares with csharp/numerics_probe.json>
        const uint LerpA = 0xC00CEF08u, LerpB = 0xBED29EFEu, LerpT = 0x3F478017u, LerpUnfused = 0xBF4E7C50u, LerpFused = 0xBF4E7C4Fu;
        static readonly uint[] TransformV = { 0xC06E43E7u, 0xC00C65BDu, 0x3EF222FCu, 0xBF9FF980u }, TransformColumn = { 0x3FA4C6C1u, 0x3F79740Cu, 0x3FC9D8E3u, 0xC06C7478u };
        const uint TransformUnfused = 0xBFC88EB0u, TransformFused = 0xBFC88EACu;
        static readonly uint[] TransformNormalN = { 0x404E9543u, 0xBE0E5810u, 0x3FC0BF7Cu }, TransformNormalColumn = { 0x3F68F07Eu, 0x3EE1EC4Bu, 0xC03AC7F0u };
        const uint TransformNormalUnfused = 0xBFC26DE8u, TransformNormalFused = 0xBFC26DE7u;
        static readonly uint[] DotA = { 0xC0230FD7u, 0x3FC580AFu, 0x3F8C55D4u }, DotB = { 0xC02E1034u, 0x4069FD33u, 0x3FE786D7u };
        const uint DotSequential = 0x4168DCA9u, DotShuffle = 0x4168DCA8u;
        static readonly uint[] DotZeroA = { 0x80000000u, 0x80000000u, 0x80000000u }, DotZeroB = { 0x3F800000u, 0x3F800000u, 0x3F800000u };
        const uint DotZeroSequential = 0x80000000u, DotZeroDpps = 0x00000000u;
        // </generated>

        static volatile uint salt = 0;      // read at run time, so that the JIT cannot fold the probe expressions at compile time
        static float F(uint bits) => BitConverter.UInt32BitsToSingle(bits ^ salt);
        static uint B(float f) => BitConverter.SingleToUInt32Bits(f);

        /// The running System.Numerics in the terms of swr_numerics_mode (lerpFused = *fma, dotOrder) and of swr_set_transform_fma
        /// (transformFused, transformNormalFused); throws on a bit pattern that neither model of an operation produces.
        [MethodImpl(MethodImplOptions.NoInlining)]
        public static (int lerpFused, int transformFused, int transformNormalFused, int dotOrder) Observe()
        {
            uint lerp = B(Vector4.Lerp(new Vector4(F(LerpA)), new Vector4(F(LerpB)), F(LerpT)).X);
            var m = new Matrix4x4(F(TransformColumn[0]), 0, 0, 0, F(TransformColumn[1]), 0, 0, 0, F(TransformColumn[2]), 0, 0, 0, F(TransformColumn[3]), 0, 0, 0);
            uint tr = B(Vector4.Transform(new Vector4(F(TransformV[0]), F(TransformV[1]), F(TransformV[2]), F(TransformV[3])), m).X);
            var mn = new Matrix4x4(F(TransformNormalColumn[0]), 0, 0, 0, F(TransformNormalColumn[1]), 0, 0, 0, F(TransformNormalColumn[2]), 0, 0, 0, 0, 0, 0, 1);
            uint tn = B(Vector3.TransformNormal(new Vector3(F(TransformNormalN[0]), F(TransformNormalN[1]), F(TransformNormalN[2])), mn).X);
            uint dot = B(Vector3.Dot(new Vector3(F(DotA[0]), F(DotA[1]), F(DotA[2])), new Vector3(F(DotB[0]), F(DotB[1]), F(DotB[2]))));
            uint dz = B(Vector3.Dot(new Vector3(F(DotZeroA[0]), F(DotZeroA[1]), F(DotZeroA[2])), new Vector3(F(DotZeroB[0]), F(DotZeroB[1]), F(DotZeroB[2]))));
            int lerpFused = lerp == LerpFused ? 1 : lerp == LerpUnfused ? 0 : -1;
            int trFused = tr == TransformFused ? 1 : tr == TransformUnfused ? 0 : -1;
            int tnFused = tn == TransformNormalFused ? 1 : tn == TransformNormalUnfused ? 0 : -1;
            if (lerpFused < 0 || trFused < 0 || tnFused < 0)
                throw new NotSupportedException($"System.Numerics model not built: Lerp -> 0x{lerp:X8}, Transform -> 0x{tr:X8}, TransformNormal -> 0x{tn:X8} (see csharp/numerics_probe.json)");
            int order;
            if (dot == DotShuffle) order = 2;
            else if (dot == DotSequential) order = dz == DotZeroDpps ? 1 : dz == DotZeroSequential ? 0 : -1;
            else order = -1;
            if (order < 0) throw new NotSupportedException($"System.Numerics model not built: Dot -> 0x{dot:X8} / 0x{dz:X8} (see csharp/numerics_probe.json)");
            return (lerpFused, trFused, tnFused, order);
        }

        /// File name of the build that models the running System.Numerics' Lerp and Dot (six builds: every combination exists).
        public static string SelectLibrary()
        {
            var (lerpFused, _, _, order) = Observe();
            string suffix = (lerpFused == 1 ? "_fma" : "") + (order == 2 ? "_dotpw" : order == 1 ? "_dpps" : "");
            return "libswr_hip" + suffix + ".so";
        }

        /// Makes every [DllImport("swr_hip")] of this assembly load the selected build; checks that the library agrees.
        public static void Install()
        {
            string file = SelectLibrary();
            NativeLibrary.SetDllImportResolver(typeof(NumericsProbe).Assembly, (name, assembly, path) =>
                name == "swr_hip" ? NativeLibrary.Load(System.IO.Path.Combine(AppContext.BaseDirectory, file)) : IntPtr.Zero);
            var (lerpFused, _, _, order) = Observe();
            if (Native.swr_numerics_mode(out int libFma, out int libOrder) != 0 || libFma != lerpFused || libOrder != order)
                throw new InvalidOperationException($"{file} was built for fma={libFma}, dot={libOrder}; the probe observed fma={lerpFused}, dot={order}");
        }

        /// The run-time half of the model: the observed Transform / TransformNormal flags go to the context (after swr_create).
        public static void Configure(IntPtr ctx)
        {
            var (_, trFused, tnFused, _) = Observe();
            if (Native.swr_set_transform_fma(ctx, trFused, tnFused) != 0) throw new InvalidOperationException("swr_set_transform_fma failed");
        }
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
                    NumericsProbe.Install();                          // before the first P/Invoke: picks the build that models this .NET's System.Numerics
                    if (Native.swr_abi_version() != 3) throw new InvalidOperationException("libswr_hip.so: ABI version mismatch");
                    int rc = Native.swr_create(0, out IntPtr c);       // one context drives one GPU; there is NO CPU fallback
                    if (rc != 0) throw new InvalidOperationException($"swr_create failed ({rc}): {Marshal.PtrToStringAnsi(Native.swr_last_error(IntPtr.Zero))}");
                    NumericsProbe.Configure(c);                       // Transform / TransformNormal fused or not: run-time flags of the context
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
        /// Double-buffered present for a host that keeps MainWindow.OnRender (MainWindow.cs:226-263): two PINNED flat buffers
        /// alternate; PresentAsync(i) starts frame i's flatten + copy and returns the buffer of frame i - 1, which has finished
        /// crossing PCIe while frame i was being rendered (null on the very first call).  Per frame the host then pays
        /// max(render, copy) instead of render + copy: at 4096 x 4096 the copy is 4.6 ms and the render 0.7 ms (DESIGN.md section 5).
        static readonly Vector3[]?[] presentBuffers = new Vector3[]?[2];
        static readonly GCHandle[] presentPins = new GCHandle[2];
        static readonly ulong[] presentTickets = new ulong[2];
        static int presentNext;
        public static Vector3[]? PresentAsync(int width, int height)
        {
            int cur = presentNext, prev = presentNext ^ 1;
            presentNext = prev;
            int n = Math.Max(width, 0) * Math.Max(height, 0);
            if (presentBuffers[cur] == null || presentBuffers[cur]!.Length != n)
            {
                if (presentBuffers[cur] != null) { SwrContext.Check(Native.swr_host_unregister(SwrContext.Handle, (void*)presentPins[cur].AddrOfPinnedObject())); presentPins[cur].Free(); }
                presentBuffers[cur] = new Vector3[n];
                presentPins[cur] = Pin(presentBuffers[cur]!, (nuint)n * 12);
            }
            SwrContext.Check(Native.swr_present_rgb_async(SwrContext.Handle, (Vector3*)presentPins[cur].AddrOfPinnedObject(), out presentTickets[cur]));
            if (presentTickets[prev] == 0) return null;
            int rc = Native.swr_present_wait(SwrContext.Handle, presentTickets[prev]);
            presentTickets[prev] = 0;
            if (rc == 1)        // SWR_STALE: a batch was replayed while the pair buffers were still growing -- take that frame synchronously
            {
                Present(presentBuffers[prev]!);
                return presentBuffers[prev];
            }
            SwrContext.Check(rc);
            return presentBuffers[prev];
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
