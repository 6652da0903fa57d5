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
//   2. z-buffer: one thread per (crop, triangle) walks the triangle's pixel bounding box (boxes of more than 64 pixels are
//      shared by the 64 lanes of the wave); edge functions at pixel centres; 64-bit atomicMin of (depth bits << 32 | face
//      id): order-independent, hence deterministic
//   3. resolve: one thread per pixel re-derives the barycentrics of the winning face, perspective-correct colour
//      interpolation, Lambert term from the camera-space face normal, clamps, writes planar RGB (+ depth).
// fp32 with contraction off: oracle/cosy_oracle.c:cosy_oracle_rasterize is the same arithmetic in scalar loops and the
// GPU tests require identical face ids / depths.
#include "cosy_common.h"
#include "raster_device.h"

#pragma clang fp contract(off)

namespace cosy {
namespace {

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

// one pixel of one triangle: edge functions at the pixel centre, perspective-correct depth, 64-bit atomicMin of (depth | face)
__device__ __forceinline__ void raster_pixel(float ax, float ay, float az, float bx, float by, float bz, float cx, float cy, float cz,
                                             float inv_area, int f, int x, int y, int W, unsigned long long* zb) {
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    const float w0 = edge_fn(bx, by, cx, cy, px, py) * inv_area;
    const float w1 = edge_fn(cx, cy, ax, ay, px, py) * inv_area;
    const float w2 = edge_fn(ax, ay, bx, by, px, py) * inv_area;
    if (!(w0 >= 0.f && w1 >= 0.f && w2 >= 0.f)) return;
    const float iz = (w0 / az + w1 / bz) + w2 / cz;
    const float z = 1.f / iz;
    const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned int)f;
    atomicMin(zb + (size_t)y * W + x, key);
}

// Triangles whose pixel box holds more than BIG pixels are not walked by their own thread (one thread per triangle serialises
// a coarse mesh: a cube that fills a 256x256 crop is 12 triangles of ~10^4 pixels each): the wave takes them one after the
// other (ballot + readlane broadcast of the triangle) and its 64 lanes share the box.  Same per-pixel arithmetic, and the
// z-buffer merge is an order-independent min: results are identical to the one-thread walk.
__global__ __launch_bounds__(256) void raster_tri_kernel(const float* __restrict__ uvz, const int* __restrict__ faces,
                                                         const int* __restrict__ n_faces, const int* __restrict__ obj,
                                                         const float* __restrict__ TCO, const float* __restrict__ K, int V, int F, int H,
                                                         int W, unsigned long long* __restrict__ zbuf) {
    constexpr int BIG = 64;
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const int o = obj[b];
    unsigned long long* zb = zbuf + (size_t)b * H * W;
    bool live = f < n_faces[o] && pose_finite(TCO + (size_t)b * 16, K + (size_t)b * 9);
    float ax = 0.f, ay = 0.f, az = 1.f, bx = 0.f, by = 0.f, bz = 1.f, cx = 0.f, cy = 0.f, cz = 1.f, inv_area = 0.f;
    int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
    if (live) {
        const int* tri = faces + ((size_t)o * F + f) * 3;
        const float* base = uvz + (size_t)b * V * 3;
        ax = base[tri[0] * 3]; ay = base[tri[0] * 3 + 1]; az = base[tri[0] * 3 + 2];
        bx = base[tri[1] * 3]; by = base[tri[1] * 3 + 1]; bz = base[tri[1] * 3 + 2];
        cx = base[tri[2] * 3]; cy = base[tri[2] * 3 + 1]; cz = base[tri[2] * 3 + 2];
        const float near = 0.01f;
        const float area = edge_fn(ax, ay, bx, by, cx, cy);
        live = (az > near && bz > near && cz > near) && !(area == 0.f || !isfinite(area));
        if (live) {
            const float xmin = fminf(ax, fminf(bx, cx)), xmax = fmaxf(ax, fmaxf(bx, cx));
            const float ymin = fminf(ay, fminf(by, cy)), ymax = fmaxf(ay, fmaxf(by, cy));
            x0 = max((int)floorf(xmin - 0.5f), 0); y0 = max((int)floorf(ymin - 0.5f), 0);
            x1 = min((int)ceilf(xmax - 0.5f), W - 1); y1 = min((int)ceilf(ymax - 0.5f), H - 1);
            inv_area = 1.f / area;
            live = x1 >= x0 && y1 >= y0;
        }
    }
    const bool big = live && (long)(x1 - x0 + 1) * (y1 - y0 + 1) > BIG;
    if (live && !big) {
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) raster_pixel(ax, ay, az, bx, by, bz, cx, cy, cz, inv_area, f, x, y, W, zb);
    }
    unsigned long long todo = __ballot(big);
    while (todo) {                                   // wave-uniform loop over the wave's big triangles
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        auto bc = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); };
        const float tax = bc(ax), tay = bc(ay), taz = bc(az), tbx = bc(bx), tby = bc(by), tbz = bc(bz), tcx = bc(cx), tcy = bc(cy), tcz = bc(cz);
        const float tinv = bc(inv_area);
        const int tx0 = __builtin_amdgcn_readlane(x0, src), tx1 = __builtin_amdgcn_readlane(x1, src);
        const int ty0 = __builtin_amdgcn_readlane(y0, src), ty1 = __builtin_amdgcn_readlane(y1, src);
        const int tf = __builtin_amdgcn_readlane(f, src);
        const int bw = tx1 - tx0 + 1, npx = bw * (ty1 - ty0 + 1);
        for (int i = lane; i < npx; i += 64) {
            const int yy = i / bw, xx = i - yy * bw;
            raster_pixel(tax, tay, taz, tbx, tby, tbz, tcx, tcy, tcz, tinv, tf, tx0 + xx, ty0 + yy, W, zb);
        }
    }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const unsigned long long* __restrict__ zbuf, const float* __restrict__ uvz,
                                                             MeshView m, const int* __restrict__ obj, const float* __restrict__ TCO, int H,
                                                             int W, ShadeParams sp, float* __restrict__ rgb, float* __restrict__ depth) {
    const int b = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    float out[3], zo;
    resolve_pixel(zbuf[(size_t)b * H * W + pix], uvz + (size_t)b * m.V * 3, m, obj[b], TCO + (size_t)b * 16, pix % W, pix / W, sp, out, zo);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[((size_t)b * 3 + k) * H * W + pix] = out[k];
    if (depth) depth[(size_t)b * H * W + pix] = zo;
}

}  // namespace
}  // namespace cosy

using namespace cosy;

extern "C" {

size_t cosy_render_scratch_bytes(int B, int V, int H, int W) {
    // [z-buffer | projected vertices (padded to 32 bytes) | roi_align tap tables of the fused render + crop kernel]
    return (size_t)B * H * W * sizeof(unsigned long long) + (((size_t)B * V * 3 + 7) & ~(size_t)7) * sizeof(float) + crop_taps_bytes(B, H, W);
}

}  // extern "C"

namespace cosy {
// clear + project + z-buffer: fills `scratch` = [zbuf (B,H,W) u64 | uvz (B,V,3)]
int launch_render_zbuffer(const float* verts, const int* faces, const int* n_faces, const int* obj_id, const float* TCO, const float* K, int B,
                          int V, int F, int H, int W, void* scratch, hipStream_t s) {
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
    return COSY_OK;
}
int check_mesh_shade(const cosy_mesh_t* mesh, const cosy_shade_t* shade, MeshView* m, ShadeParams* sp) {
    COSY_REQUIRE(mesh && shade, "render: null mesh / shade");
    COSY_REQUIRE(mesh->verts && mesh->colors && mesh->faces && mesh->n_faces && mesh->V > 0 && mesh->F > 0, "render: incomplete mesh set");
    COSY_REQUIRE(!shade->smooth || mesh->normals, "render: smooth shading needs vertex normals");
    COSY_REQUIRE(!mesh->tex || (mesh->uvs && mesh->TH > 0 && mesh->TW > 0), "render: a texture needs uvs and its size");
    *m = MeshView{mesh->verts, mesh->colors, mesh->normals, mesh->uvs, mesh->tex, mesh->faces, mesh->V, mesh->F, mesh->TH, mesh->TW};
    *sp = ShadeParams{shade->ambient, shade->diffuse, shade->specular, shade->shininess, shade->light[0], shade->light[1], shade->light[2],
                      shade->light_frame, shade->smooth, shade->quantize};
    return COSY_OK;
}
int render_crop_pack(void* x, int dtype, const cosy_mesh_t* mesh, const cosy_shade_t* shade, const int* obj_id, const float* TCO,
                     const float* K_crop, const float* frames4, const int* im_id, const float* boxes, int B, int N, int h, int w, int H, int W,
                     void* scratch, hipStream_t s) {
    (void)N;
    MeshView m; ShadeParams sp;
    int rc;
    if ((rc = check_mesh_shade(mesh, shade, &m, &sp))) return rc;
    COSY_REQUIRE(B >= 0 && H > 0 && W > 0 && h > 0 && w > 0, "render_crop_pack: B=%d H=%d W=%d h=%d w=%d", B, H, W, h, w);
    COSY_REQUIRE(dtype == COSY_F32 || dtype == COSY_BF16 || dtype == COSY_F16, "render_crop_pack: dtype %d", dtype);
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(x && obj_id && TCO && K_crop && frames4 && boxes && scratch, "render_crop_pack: null pointer");
    if ((rc = launch_render_zbuffer(m.verts, m.faces, mesh->n_faces, obj_id, TCO, K_crop, B, m.V, m.F, H, W, scratch, s))) return rc;
    return launch_render_crop_pack(x, dtype, frames4, im_id, boxes, scratch, m, obj_id, TCO, sp, B, h, w, H, W, s);
}
}  // namespace cosy

extern "C" {

int cosy_render_crop_pack_to(void* x_nhwc8, int dtype, const cosy_mesh_t* mesh, const cosy_shade_t* shade, const int* obj_id,
                             const float* TCO, const float* K_crop, const float* frames_nhwc4, const int* im_id, const float* boxes_crop,
                             int B, int N, int h, int w, int H, int W, void* scratch, cosy_stream_t stream) {
    return render_crop_pack(x_nhwc8, dtype, mesh, shade, obj_id, TCO, K_crop, frames_nhwc4, im_id, boxes_crop, B, N, h, w, H, W, scratch,
                            (hipStream_t)stream);
}

int cosy_render_meshes_ex(const cosy_mesh_t* mesh, const cosy_shade_t* shade, const int* obj_id, const float* TCO, const float* K, int B,
                          int H, int W, float* rgb, float* depth, void* scratch, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    MeshView m; ShadeParams sp;
    int rc;
    if ((rc = check_mesh_shade(mesh, shade, &m, &sp))) return rc;
    COSY_REQUIRE(B >= 0 && H > 0 && W > 0, "render_meshes: B=%d H=%d W=%d", B, H, W);
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(obj_id && TCO && K && rgb && scratch, "render_meshes: null pointer");
    if ((rc = launch_render_zbuffer(m.verts, m.faces, mesh->n_faces, obj_id, TCO, K, B, m.V, m.F, H, W, scratch, s))) return rc;
    const unsigned long long* zbuf = (const unsigned long long*)scratch;
    const float* uvz = (const float*)(zbuf + (size_t)B * H * W);
    hipLaunchKernelGGL(raster_resolve_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, s, zbuf, uvz, m, obj_id, TCO, H, W, sp, rgb, depth);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

int cosy_render_meshes(const float* verts, const float* colors, const int* faces, const int* n_faces, const int* obj_id,
                       const float* TCO, const float* K, int B, int V, int F, int H, int W, float ambient, float diffuse,
                       float light_x, float light_y, float light_z, float* rgb, float* depth, void* scratch, cosy_stream_t stream) {
    COSY_REQUIRE(B >= 0 && V > 0 && F > 0 && H > 0 && W > 0, "render_meshes: B=%d V=%d F=%d H=%d W=%d", B, V, F, H, W);
    cosy_mesh_t mesh{verts, colors, nullptr, nullptr, nullptr, faces, n_faces, V, F, 0, 0};
    cosy_shade_t shade{ambient, diffuse, 0.f, 1.f, {light_x, light_y, light_z}, 0, 0, 0};
    return cosy_render_meshes_ex(&mesh, &shade, obj_id, TCO, K, B, H, W, rgb, depth, scratch, stream);
}

}  // extern "C"
