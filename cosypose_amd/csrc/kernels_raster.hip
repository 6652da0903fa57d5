// On-device mesh rasteriser behind renderer.render: the reference renders every crop with PyBullet's
// OpenGL pipeline in a pool of worker processes (cosypose/rendering/bullet_batch_renderer.py:46-90,
// bullet_scene_renderer.py:38-60) and ships the images host -> device in EVERY iteration of the loop; this keeps the
// whole loop on the GPU.  Taken from the reference: the camera model (pixel i spans [i, i+1) in K coordinates, samples
// at i + 0.5: proj_from_K, simulator/camera.py:9-33), near plane 0.01, black background, float (B,3,H,W) in [0,1],
// non-finite poses -> black image (bullet_batch_renderer.py:25-36).  NOT reproducible: PyBullet's shading
// (third-party OpenGL renderer) -> pixel values are PARITY UNPINNED; the shading here is vertex colours x
// (ambient + diffuse |n.l|), flat per face.
//
// Pipeline per call, all crops at once (meshes of a few 10^3..10^4 triangles, a few pixels each at crop resolution):
//   1. project: one thread per (crop, vertex) -> (u, v, z_cam)
//   2. z-buffer: one thread per (crop, triangle) walks the triangle's pixel bounding box; edge functions at pixel
//      centres; 64-bit atomicMin of (depth bits << 32 | face id): order-independent, hence deterministic
//   3. resolve: one thread per pixel re-derives the barycentrics of the winning face, perspective-correct colour
//      interpolation, Lambert term from the camera-space face normal, clamps, writes planar RGB (+ depth).
// fp32 with contraction off: oracle/cosy_oracle.c:cosy_oracle_rasterize is the same arithmetic in scalar loops and the
// GPU tests require identical face ids / depths.
#include "cosy_common.h"

#pragma clang fp contract(off)

namespace cosy {
namespace {

__device__ __forceinline__ float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
__device__ __forceinline__ bool pose_finite(const float* T, const float* K) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 16; ++i) ok = ok && isfinite(T[i]);
#pragma unroll
    for (int i = 0; i < 9; ++i) ok = ok && isfinite(K[i]);
    return ok;
}

__global__ __launch_bounds__(256) void raster_clear_kernel(unsigned long long* __restrict__ zbuf, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) zbuf[i] = ~0ull;
}

__global__ __launch_bounds__(256) void raster_project_kernel(const float* __restrict__ verts, const int* __restrict__ obj,
                                                             const float* __restrict__ TCO, const float* __restrict__ K, int V,
                                                             float* __restrict__ uvz) {
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float* T = TCO + (size_t)b * 16;
    const float* Kb = K + (size_t)b * 9;
    const float* p = verts + ((size_t)obj[b] * V + v) * 3;
    float c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) c[i] = ((T[i * 4] * p[0] + T[i * 4 + 1] * p[1]) + T[i * 4 + 2] * p[2]) + T[i * 4 + 3];
    float* o = uvz + ((size_t)b * V + v) * 3;
    o[0] = Kb[0] * c[0] / c[2] + Kb[2];
    o[1] = Kb[4] * c[1] / c[2] + Kb[5];
    o[2] = c[2];
}

__global__ __launch_bounds__(256) void raster_tri_kernel(const float* __restrict__ uvz, const int* __restrict__ faces,
                                                         const int* __restrict__ n_faces, const int* __restrict__ obj,
                                                         const float* __restrict__ TCO, const float* __restrict__ K, int V, int F, int H,
                                                         int W, unsigned long long* __restrict__ zbuf) {
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    const int o = obj[b];
    if (f >= n_faces[o]) return;
    if (!pose_finite(TCO + (size_t)b * 16, K + (size_t)b * 9)) return;
    const int* tri = faces + ((size_t)o * F + f) * 3;
    const float* base = uvz + (size_t)b * V * 3;
    const float ax = base[tri[0] * 3], ay = base[tri[0] * 3 + 1], az = base[tri[0] * 3 + 2];
    const float bx = base[tri[1] * 3], by = base[tri[1] * 3 + 1], bz = base[tri[1] * 3 + 2];
    const float cx = base[tri[2] * 3], cy = base[tri[2] * 3 + 1], cz = base[tri[2] * 3 + 2];
    const float near = 0.01f;
    if (!(az > near && bz > near && cz > near)) return;
    const float area = edge_fn(ax, ay, bx, by, cx, cy);
    if (area == 0.f || !isfinite(area)) return;
    const float xmin = fminf(ax, fminf(bx, cx)), xmax = fmaxf(ax, fmaxf(bx, cx));
    const float ymin = fminf(ay, fminf(by, cy)), ymax = fmaxf(ay, fmaxf(by, cy));
    int x0 = (int)floorf(xmin - 0.5f), x1 = (int)ceilf(xmax - 0.5f), y0 = (int)floorf(ymin - 0.5f), y1 = (int)ceilf(ymax - 0.5f);
    x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, W - 1); y1 = min(y1, H - 1);
    const float inv_area = 1.f / area;
    unsigned long long* zb = zbuf + (size_t)b * H * W;
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            const float px = (float)x + 0.5f, py = (float)y + 0.5f;
            const float w0 = edge_fn(bx, by, cx, cy, px, py) * inv_area;
            const float w1 = edge_fn(cx, cy, ax, ay, px, py) * inv_area;
            const float w2 = edge_fn(ax, ay, bx, by, px, py) * inv_area;
            if (!(w0 >= 0.f && w1 >= 0.f && w2 >= 0.f)) continue;
            const float iz = (w0 / az + w1 / bz) + w2 / cz;
            const float z = 1.f / iz;
            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned int)f;
            atomicMin(zb + (size_t)y * W + x, key);
        }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const unsigned long long* __restrict__ zbuf, const float* __restrict__ uvz,
                                                             const float* __restrict__ verts, const float* __restrict__ colors,
                                                             const int* __restrict__ faces, const int* __restrict__ obj,
                                                             const float* __restrict__ TCO, int V, int F, int H, int W, float ambient,
                                                             float diffuse, float lx, float ly, float lz, float* __restrict__ rgb,
                                                             float* __restrict__ depth) {
    const int b = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int x = pix % W, y = pix / W;
    const unsigned long long key = zbuf[(size_t)b * H * W + pix];
    float out[3] = {0.f, 0.f, 0.f}, zo = 0.f;
    if (key != ~0ull) {
        const int o = obj[b], f = (int)(key & 0xffffffffu);
        const int* tri = faces + ((size_t)o * F + f) * 3;
        const int i0 = tri[0], i1 = tri[1], i2 = tri[2];
        const float* base = uvz + (size_t)b * V * 3;
        const float ax = base[i0 * 3], ay = base[i0 * 3 + 1], az = base[i0 * 3 + 2];
        const float bx = base[i1 * 3], by = base[i1 * 3 + 1], bz = base[i1 * 3 + 2];
        const float cx = base[i2 * 3], cy = base[i2 * 3 + 1], cz = base[i2 * 3 + 2];
        const float px = (float)x + 0.5f, py = (float)y + 0.5f;
        const float inv_area = 1.f / edge_fn(ax, ay, bx, by, cx, cy);
        const float w0 = edge_fn(bx, by, cx, cy, px, py) * inv_area;
        const float w1 = edge_fn(cx, cy, ax, ay, px, py) * inv_area;
        const float w2 = edge_fn(ax, ay, bx, by, px, py) * inv_area;
        const float q0 = w0 / az, q1 = w1 / bz, q2 = w2 / cz;
        const float z = 1.f / ((q0 + q1) + q2);
        // camera-space vertices of the face (same expression as the projection kernel) -> flat two-sided Lambert term
        const float* T = TCO + (size_t)b * 16;
        float P[3][3];
        const int idx[3] = {i0, i1, i2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* p = verts + ((size_t)o * V + idx[k]) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) P[k][i] = ((T[i * 4] * p[0] + T[i * 4 + 1] * p[1]) + T[i * 4 + 2] * p[2]) + T[i * 4 + 3];
        }
        const float e1[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
        const float e2[3] = {P[2][0] - P[0][0], P[2][1] - P[0][1], P[2][2] - P[0][2]};
        const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const float nn = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
        const float lam = nn > 0.f ? fabsf((n[0] * lx + n[1] * ly) + n[2] * lz) / nn : 0.f;
        const float shade = ambient + diffuse * lam;
        const float* ca = colors + ((size_t)o * V + i0) * 3;
        const float* cb = colors + ((size_t)o * V + i1) * 3;
        const float* cc = colors + ((size_t)o * V + i2) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float col = (((q0 * ca[k] + q1 * cb[k]) + q2 * cc[k]) * z) * shade;
            out[k] = fminf(fmaxf(col, 0.f), 1.f);
        }
        zo = z;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[((size_t)b * 3 + k) * H * W + pix] = out[k];
    if (depth) depth[(size_t)b * H * W + pix] = zo;
}

}  // namespace
}  // namespace cosy

using namespace cosy;

extern "C" {

size_t cosy_render_scratch_bytes(int B, int V, int H, int W) {
    return (size_t)B * H * W * sizeof(unsigned long long) + (size_t)B * V * 3 * sizeof(float);
}

int cosy_render_meshes(const float* verts, const float* colors, const int* faces, const int* n_faces, const int* obj_id,
                       const float* TCO, const float* K, int B, int V, int F, int H, int W, float ambient, float diffuse,
                       float light_x, float light_y, float light_z, float* rgb, float* depth, void* scratch, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(B >= 0 && V > 0 && F > 0 && H > 0 && W > 0, "render_meshes: B=%d V=%d F=%d H=%d W=%d", B, V, F, H, W);
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(verts && colors && faces && n_faces && obj_id && TCO && K && rgb && scratch, "render_meshes: null pointer");
    unsigned long long* zbuf = (unsigned long long*)scratch;
    float* uvz = (float*)(zbuf + (size_t)B * H * W);
    const long npx = (long)B * H * W;
    hipLaunchKernelGGL(raster_clear_kernel, dim3(cdiv(npx, 256)), dim3(256), 0, s, zbuf, npx);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(raster_project_kernel, dim3(cdiv(V, 256), B), dim3(256), 0, s, verts, obj_id, TCO, K, V, uvz);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(raster_tri_kernel, dim3(cdiv(F, 256), B), dim3(256), 0, s, (const float*)uvz, faces, n_faces, obj_id, TCO, K, V, F, H, W,
                       zbuf);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(raster_resolve_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, s, (const unsigned long long*)zbuf, (const float*)uvz,
                       verts, colors, faces, obj_id, TCO, V, F, H, W, ambient, diffuse, light_x, light_y, light_z, rgb, depth);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

}  // extern "C"
