// Rasterizer.hpp -- C++ host-side mirror of the reference's raster API over the C ABI of include/swr.h.
//
// The reference is C# (no .NET toolchain in this image), so the compiled-language host above the ABI is C++.
// Same names, argument meaning and error behaviour as the reference (file:line under OCSYT/SoftwareRenderer):
//   SoftwareRenderer::Rasterizer::RenderMesh / InitializeTileLocks, enums, statics   Rasterizer.cs:14-50,69,163
//   SoftwareRenderer::Shaders::VertexInput                                           Shaders.cs:10-24
//   SoftwareRenderer::Texture (Width/Height/Sample/Dispose)                          Texture.cs:31-68
//   SoftwareRenderer::MainWindow (RenderWidth/Height, buffers, Get/Set*, Clear*)     MainWindow.cs:25-31,378-436
// C# delegates cannot cross the ABI: a ShaderProgram {program id, uniforms, texture} stands in for the
// (VertexShader, FragmentShader) pair.  Header-only; link with -lswr_hip (softwarerenderer_amd/libswr_hip.so).
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "swr.h"

namespace SoftwareRenderer {

struct SwrError : std::runtime_error {
    int code;
    SwrError(int c, const std::string& m) : std::runtime_error("swr error " + std::to_string(c) + ": " + m), code(c) {}
};

using Matrix4x4 = std::array<float, 16>;     // M11..M44 row-major, row-vector convention (System.Numerics layout)
using Vector4 = std::array<float, 4>;

class Device {                                // one swr_context = one GPU
public:
    explicit Device(int device_id = 0) {
        int rc = swr_create(device_id, &ctx_);
        if (rc != SWR_OK) throw SwrError(rc, swr_last_error(nullptr));
    }
    ~Device() { if (ctx_) swr_destroy(ctx_); }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    swr_context* ctx() const { return ctx_; }
    void check(int rc) const {
        if (rc == SWR_OK) return;
        std::string msg = swr_last_error(ctx_);
        if (rc == SWR_ERR_INVALID_ARG) throw std::invalid_argument(msg);   // C#: ArgumentException / IndexOutOfRangeException
        throw SwrError(rc, msg);
    }
    void Flush() const { check(swr_flush(ctx_)); }
    void Sync() const { check(swr_sync(ctx_)); }
    swr_stats Stats() const { swr_stats s; check(swr_get_stats(ctx_, &s)); return s; }
    void ResetStats() const { check(swr_reset_stats(ctx_)); }
private:
    swr_context* ctx_ = nullptr;
};

struct Shaders {
    using VertexInput = swr_vertex;           // Shaders.cs:10-24, 48 bytes
    enum class Program { FlatColor = SWR_PROG_FLAT_COLOR, Gouraud = SWR_PROG_GOURAUD,
                         Dust2LambertFog = SWR_PROG_DUST2_LAMBERT_FOG, Phong4Point = SWR_PROG_PHONG_4POINT };
};
static_assert(sizeof(Shaders::VertexInput) == 48, "VertexInput must match Shaders.cs:10-24");

class Texture {                               // Texture.cs:31-68
public:
    Texture(const Device& dev, const uint8_t* rgba8, int width, int height) : dev_(dev), Width(width), Height(height) {
        dev_.check(swr_texture_create(dev_.ctx(), rgba8, width, height, &h_));
    }
    ~Texture() { Dispose(); }
    Vector4 Sample(float u, float v) const {  // Texture.cs:43-63 (runs on the GPU)
        float uv[2] = { u, v }; Vector4 out{};
        dev_.check(swr_texture_sample(dev_.ctx(), h_, uv, 1, out.data()));
        return out;
    }
    void Dispose() { if (h_) { swr_texture_destroy(dev_.ctx(), h_); h_ = nullptr; } }
    swr_texture* handle() const { return h_; }
    const int Width, Height;
private:
    const Device& dev_;
    swr_texture* h_ = nullptr;
};

struct ShaderProgram {                        // stands in for (VertexShader, FragmentShader), Shaders.cs:97-98
    Shaders::Program program = Shaders::Program::Gouraud;
    swr_uniforms uniforms{};
    const Texture* texture = nullptr;
};

inline swr_uniforms DefaultUniforms() {      // Renderer.cs:39-44
    swr_uniforms u{};
    u.light_direction[0] = 0.5f; u.light_direction[1] = -0.70710678f; u.light_direction[2] = -0.5f;
    for (int i = 0; i < 4; ++i) u.light_color[i] = 1.0f;
    u.fog_color[0] = 1.0f; u.fog_color[1] = 0.62f; u.fog_color[2] = 0.5f; u.fog_color[3] = 1.0f;
    u.fog_start = 1.0f; u.fog_end = 25.0f; u.shininess = 16.0f;
    return u;
}

class MainWindow {                            // framebuffer part of MainWindow.cs
public:
    MainWindow(const Device& dev, int renderWidth, int renderHeight) : dev_(dev) { Resize(renderWidth, renderHeight); }
    void Resize(int w, int h) { dev_.check(swr_resize(dev_.ctx(), w, h)); RenderWidth = w; RenderHeight = h; }
    void ClearColorBuffer(const Vector4& c) { dev_.check(swr_clear_color(dev_.ctx(), c.data())); }     // MainWindow.cs:400-407
    void ClearDepthBuffer() { dev_.check(swr_clear_depth(dev_.ctx())); }                                // :429-436
    Vector4 GetPixel(int x, int y) const { Vector4 o{}; dev_.check(swr_get_pixel(dev_.ctx(), x, y, o.data())); return o; }
    void SetPixel(int x, int y, const Vector4& c) { dev_.check(swr_set_pixel(dev_.ctx(), x, y, c.data())); }
    float GetDepth(int x, int y) const { float d; dev_.check(swr_get_depth(dev_.ctx(), x, y, &d)); return d; }
    void SetDepth(int x, int y, float d) { dev_.check(swr_set_depth(dev_.ctx(), x, y, d)); }
    // ColorBuffer / DepthBuffer: read the HBM buffers back (Vector4[W*H], float[W*H], idx = y*W + x)
    void ReadBuffers(std::vector<float>* color, std::vector<float>* depth) const {
        size_t n = (size_t)(RenderWidth > 0 ? RenderWidth : 0) * (size_t)(RenderHeight > 0 ? RenderHeight : 0);
        if (color) color->resize(n * 4);
        if (depth) depth->resize(n);
        dev_.check(swr_readback(dev_.ctx(), color ? color->data() : nullptr, depth ? depth->data() : nullptr));
    }
    const Device& device() const { return dev_; }
    int RenderWidth = 0, RenderHeight = 0;
private:
    const Device& dev_;
};

class Rasterizer {                            // public static class Rasterizer, Rasterizer.cs:12
public:
    enum class DebugMode { None = 0, Wireframe = 1 };                                              // :14-18
    enum class BlendMode { None = 0, Alpha = 1, Additive = 2, Multiply = 3 };                      // :25-31
    enum class DepthTest { Disabled = 0, Less, LessEqual, Greater, GreaterEqual, Equal, NotEqual, Always };   // :33-43
    enum class CullMode { None = 0, Back = 1, Front = 2 };                                         // :45-50
    static inline float NearClip = 0.1f;                                                           // :20
    static inline float FarClip = 1000.0f;                                                         // :21
    static inline DebugMode RenderDebugMode = DebugMode::None;                                     // :22

    static void InitializeTileLocks(const MainWindow& w, int width, int height) {                  // :69-93
        w.device().check(swr_initialize_tile_locks(w.device().ctx(), width, height));
    }
    // Rasterizer.RenderMesh, Rasterizer.cs:163-174 (same defaults)
    static void RenderMesh(MainWindow& window, const std::vector<Shaders::VertexInput>& vertices,
                           const std::vector<uint16_t>& indices, const Matrix4x4& model, const Matrix4x4& view,
                           const Matrix4x4& projection, const ShaderProgram& shader,
                           CullMode cullMode = CullMode::Back, DepthTest depthTest = DepthTest::LessEqual,
                           BlendMode blendMode = BlendMode::Alpha) {
        const Device& d = window.device();
        d.check(swr_set_state(d.ctx(), NearClip, FarClip, (int)RenderDebugMode));
        d.check(swr_render_mesh_arrays(d.ctx(), vertices.data(), (int)vertices.size(), indices.data(), (int)indices.size(),
                                       model.data(), view.data(), projection.data(), (int)shader.program, &shader.uniforms,
                                       shader.texture ? shader.texture->handle() : nullptr,
                                       (int)cullMode, (int)depthTest, (int)blendMode));
    }
};

}  // namespace SoftwareRenderer
