// swr_cull.hip.h -- FrustumCuller.cs on the GPU (row N3): bounding spheres of retained meshes and the per-draw
// sphere / frustum test, evaluated on the device so that a culled RenderMesh costs no host round trip.
//
// Reference (file:line under the C# repo):
//   FrustumCuller.CalculateBoundingSphere   FrustumCuller.cs:59-151
//   CreateFrustumFromMatrix / NormalizePlane / Plane ctor   :153-199, :25-29
//   IsSphereInFrustum / TestSphereAgainstPlane              :201-224
// The reference's three Parallel.For passes are order dependent (ties; the third pass keeps the LAST outside
// vertex of every partition); the serial one-partition schedule is reproduced: pass 1/2 = argmax with the lowest
// index on ties, pass 3 = the highest index whose distance exceeds the first radius.
#pragma once
#include "swr_device.h"

namespace swr {

__device__ __forceinline__ float dist_sq3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;     // Vector3.DistanceSquared
    return (dx * dx + dy * dy) + dz * dz;
}

// block-wide max of a 64-bit key (1024 threads)
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, unsigned long long* s_part) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)hi, off) << 32) | (unsigned)__shfl_xor((int)lo, off);
        v = o > v ? o : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long t = 0;
    for (int w = 0; w < 16; ++w) t = s_part[w] > t ? s_part[w] : t;
    return t;
}
// key = (distance bits, ~index): larger distance wins, then the LOWER index; NaN never "exceeds" anything -> key 0
__device__ __forceinline__ unsigned long long far_key(float d, uint32_t i, float floor_excl) {
    if (!(d > floor_excl)) return 0ull;
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xffffffffu - i);
}

__global__ __launch_bounds__(1024) void k_bounding_sphere(const swr_vertex* __restrict__ v, uint32_t n, float4* __restrict__ out) {
    __shared__ unsigned long long s_part[16];
    if (n == 0) { if (threadIdx.x == 0) *out = make_float4(0.f, 0.f, 0.f, 0.f); return; }                 // :65-66
    const float p0x = v[0].position[0], p0y = v[0].position[1], p0z = v[0].position[2];
    if (n == 1) { if (threadIdx.x == 0) *out = make_float4(p0x, p0y, p0z, 0.f); return; }                  // :67-68
    // pass 1 (:75-93): farthest from p0 among i >= 1, strict '>' against 0 and earlier maxima
    unsigned long long key = 0;
    for (uint32_t i = 1 + threadIdx.x; i < n; i += 1024u) {
        const unsigned long long k = far_key(dist_sq3(v[i].position[0], v[i].position[1], v[i].position[2], p0x, p0y, p0z), i, 0.0f);
        key = k > key ? k : key;
    }
    key = block_max_u64(key, s_part);
    uint32_t i1 = key ? 0xffffffffu - (uint32_t)key : 0u;                     // nothing farther than 0: p1 stays p0
    const float p1x = v[i1].position[0], p1y = v[i1].position[1], p1z = v[i1].position[2];
    // pass 2 (:98-116): farthest from p1 among all
    key = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
        const unsigned long long k = far_key(dist_sq3(v[i].position[0], v[i].position[1], v[i].position[2], p1x, p1y, p1z), i, 0.0f);
        key = k > key ? k : key;
    }
    key = block_max_u64(key, s_part);
    const uint32_t i2 = key ? 0xffffffffu - (uint32_t)key : i1;
    const float max_sq = key ? __uint_as_float((uint32_t)(key >> 32)) : 0.0f;
    const float p2x = v[i2].position[0], p2y = v[i2].position[1], p2z = v[i2].position[2];
    const float cx = (p1x + p2x) * 0.5f, cy = (p1y + p2y) * 0.5f, cz = (p1z + p2z) * 0.5f;               // :118
    const float r = sqrtf(max_sq) * 0.5f;                                                                  // :119
    // pass 3 (:124-131, one partition): the LAST vertex whose distance exceeds r
    unsigned long long last = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
        const float dist = sqrtf(dist_sq3(v[i].position[0], v[i].position[1], v[i].position[2], cx, cy, cz));
        if (dist > r) last = (unsigned long long)i + 1ull;
    }
    last = block_max_u64(last, s_part);
    if (threadIdx.x == 0) {
        float ncx = cx, ncy = cy, ncz = cz, nr = r;
        if (last) {
            const uint32_t i = (uint32_t)(last - 1ull);
            const float fx = v[i].position[0], fy = v[i].position[1], fz = v[i].position[2];
            const float dist = sqrtf(dist_sq3(fx, fy, fz, cx, cy, cz));
            if (dist > nr) {                                                                               // :139-145
                const float upd = (nr + dist) * 0.5f;
                const float k = (upd - nr) / dist;
                float t;
                t = (fx - ncx) * k; ncx = ncx + t;
                t = (fy - ncy) * k; ncy = ncy + t;
                t = (fz - ncz) * k; ncz = ncz + t;
                nr = upd;
            }
        }
        *out = make_float4(ncx, ncy, ncz, nr);
    }
}

// IsSphereInFrustum, FrustumCuller.cs:201-224 (planes tested in the reference's order Left, Right, Top, Bottom, Near, Far)
__device__ __forceinline__ bool sphere_in_frustum(float4 sphere, const float* __restrict__ model, const float* __restrict__ view,
                                                  const float* __restrict__ proj, uint32_t nm_flags) {
    const bool fma_t = (nm_flags & SWR_NM_TRANSFORM_FMA) != 0u;           // Vector3.Transform and Matrix4x4.Multiply: the Transform family
    const float c4[4] = { sphere.x, sphere.y, sphere.z, 1.0f };
    float wc[4];
    vec4_transform(c4, model, wc, fma_t);                                                // Vector3.Transform(center, model)
    const float s0 = sqrtf((model[0] * model[0] + model[1] * model[1]) + model[2] * model[2]);    // :204-209
    const float s1 = sqrtf((model[4] * model[4] + model[5] * model[5]) + model[6] * model[6]);
    const float s2 = sqrtf((model[8] * model[8] + model[9] * model[9]) + model[10] * model[10]);
    const float wr = sphere.w * mathf_max(mathf_max(s0, s1), s2);                         // :211
    float vp[16];                                                                         // Matrix4x4.Multiply(view, proj)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float row[4] = { view[4 * i], view[4 * i + 1], view[4 * i + 2], view[4 * i + 3] };
        vec4_transform(row, proj, vp + 4 * i, fma_t);
    }
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int col = k >> 1;
        float co[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) co[rr] = (k & 1) ? vp[rr * 4 + 3] - vp[rr * 4 + col] : vp[rr * 4 + 3] + vp[rr * 4 + col];
        const float mag = sqrtf((co[0] * co[0] + co[1] * co[1]) + co[2] * co[2]);       // NormalizePlane :189-199
        const float nx = co[0] / mag, ny = co[1] / mag, nz = co[2] / mag, dd = co[3] / mag;
        const float len = sqrtf(dot3(nx, ny, nz, nx, ny, nz));                            // Plane ctor normalises again :25-29
        const float px = nx / len, py = ny / len, pz = nz / len;
        const float dist = dot3(px, py, pz, wc[0], wc[1], wc[2]) + dd;                   // GetDistanceToPoint :31-34
        inside = inside && (dist > -wr);                                                  // :221-224
    }
    return inside;
}

// one thread per draw of the batch: 1 = render, 0 = skipped by the frustum test (draws without a cull request are 1)
__global__ __launch_bounds__(64) void k_frustum_cull(const DrawParams* __restrict__ draws, const float4* const* __restrict__ bounds,
                                                     uint32_t n_draws, uint32_t* __restrict__ visible) {
    SWR_FRONT_ENTER();
    const uint32_t d = blockIdx.x * 64u + threadIdx.x;
    if (d >= n_draws) return;
    const float4* b = bounds[d];
    visible[d] = b ? (sphere_in_frustum(*b, draws[d].model, draws[d].view, draws[d].proj, draws[d].nm_flags) ? 1u : 0u) : 1u;
}

__global__ void k_frustum_test(float4 sphere, const float* __restrict__ mvp48, uint32_t* __restrict__ out, uint32_t nm_flags) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = sphere_in_frustum(sphere, mvp48, mvp48 + 16, mvp48 + 32, nm_flags) ? 1u : 0u;
}

}  // namespace swr
