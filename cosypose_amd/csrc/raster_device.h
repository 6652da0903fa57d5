// Per-pixel resolve of the mesh rasteriser (shared by kernels_raster.hip and the fused render + crop kernel in
// kernels_geom.hip).  fp32 with contraction off: oracle/cosy_oracle.c restates the same arithmetic in scalar loops.
#pragma once
#include "cosy_common.h"

namespace cosy {

// Shading of the resolve pass.  Two models:
//   flat (smooth = 0)   : vertex colour x (ambient + diffuse |n.l|), n = camera-space face normal, two-sided.  Round 1's model.
//   OpenGL-like (smooth): what PyBullet's hardware renderer does in structure (bullet_scene_renderer.py:38-60 ->
//                         getCameraImage with ER_BULLET_HARDWARE_OPENGL): texture x vertex colour, interpolated vertex
//                         normals, one-sided Lambert + Blinn-Phong highlight, light fixed in the WORLD frame -- which is the
//                         object's frame, the reference renders every object at TWO = identity
//                         (bullet_batch_renderer.py:56-59) -- and 8-bit output (`images.float() / 255`, :83-84).
//                         The shader's constants are third-party and not in the reference: defaults are placeholders,
//                         pixel values stay parity-unpinned.
struct ShadeParams {
    float ambient, diffuse, specular, shininess;
    float lx, ly, lz;     // unit vector from the surface towards the light
    int light_frame;      // 0: camera frame (a headlight), 1: object frame (= PyBullet's world frame)
    int smooth;           // 0: flat two-sided face normal, 1: interpolated vertex normals (needs `normals`)
    int quantize;         // 1: round to multiples of 1/255
};

struct MeshView {
    const float* verts;    // (n_obj, V, 3)
    const float* colors;   // (n_obj, V, 3)
    const float* normals;  // (n_obj, V, 3) unit, object frame, or null
    const float* uvs;      // (n_obj, V, 2) or null
    const float* tex;      // (n_obj, TH, TW, 4) RGB + pad in [0,1], or null
    const int* faces;      // (n_obj, F, 3)
    int V, F, TH, TW;
};

__device__ __forceinline__ float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
#pragma clang fp contract(off)
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}

// colour (3, in [0,1]) and depth of pixel (x, y) of crop b given its z-buffer key; black / 0 for background
__device__ __forceinline__ void resolve_pixel(unsigned long long key, const float* __restrict__ uvz_b, const MeshView& m, int o,
                                              const float* __restrict__ T, int x, int y, const ShadeParams& sp, float* out, float& zo) {
#pragma clang fp contract(off)
    out[0] = out[1] = out[2] = 0.f; zo = 0.f;
    if (key == ~0ull) return;
    const int f = (int)(key & 0xffffffffu);
    const int* tri = m.faces + ((size_t)o * m.F + f) * 3;
    const int i0 = tri[0], i1 = tri[1], i2 = tri[2];
    const float ax = uvz_b[i0 * 3], ay = uvz_b[i0 * 3 + 1], az = uvz_b[i0 * 3 + 2];
    const float bx = uvz_b[i1 * 3], by = uvz_b[i1 * 3 + 1], bz = uvz_b[i1 * 3 + 2];
    const float cx = uvz_b[i2 * 3], cy = uvz_b[i2 * 3 + 1], cz = uvz_b[i2 * 3 + 2];
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    const float inv_area = 1.f / edge_fn(ax, ay, bx, by, cx, cy);
    const float w0 = edge_fn(bx, by, cx, cy, px, py) * inv_area;
    const float w1 = edge_fn(cx, cy, ax, ay, px, py) * inv_area;
    const float w2 = edge_fn(ax, ay, bx, by, px, py) * inv_area;
    const float q0 = w0 / az, q1 = w1 / bz, q2 = w2 / cz;
    const float z = 1.f / ((q0 + q1) + q2);
    // camera-space vertices of the face (same expression as the projection kernel)
    float P[3][3];
    const int idx[3] = {i0, i1, i2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* p = m.verts + ((size_t)o * m.V + idx[k]) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) P[k][i] = ((T[i * 4] * p[0] + T[i * 4 + 1] * p[1]) + T[i * 4 + 2] * p[2]) + T[i * 4 + 3];
    }
    // light in the camera frame
    float l[3] = {sp.lx, sp.ly, sp.lz};
    if (sp.light_frame == 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i) l[i] = (T[i * 4] * sp.lx + T[i * 4 + 1] * sp.ly) + T[i * 4 + 2] * sp.lz;
    }
    float n[3], lam;
    if (sp.smooth && m.normals) {
        const float* na = m.normals + ((size_t)o * m.V + i0) * 3;
        const float* nb = m.normals + ((size_t)o * m.V + i1) * 3;
        const float* nc = m.normals + ((size_t)o * m.V + i2) * 3;
        float no[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) no[k] = (q0 * na[k] + q1 * nb[k]) + q2 * nc[k];
#pragma unroll
        for (int i = 0; i < 3; ++i) n[i] = (T[i * 4] * no[0] + T[i * 4 + 1] * no[1]) + T[i * 4 + 2] * no[2];
        const float nn = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
        const float inv = nn > 0.f ? 1.f / nn : 0.f;
        n[0] *= inv; n[1] *= inv; n[2] *= inv;
        lam = fmaxf((n[0] * l[0] + n[1] * l[1]) + n[2] * l[2], 0.f);
    } else {
        const float e1[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
        const float e2[3] = {P[2][0] - P[0][0], P[2][1] - P[0][1], P[2][2] - P[0][2]};
        n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const float nn = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
        lam = nn > 0.f ? fabsf((n[0] * l[0] + n[1] * l[1]) + n[2] * l[2]) / nn : 0.f;
        const float inv = nn > 0.f ? 1.f / nn : 0.f;
        n[0] *= inv; n[1] *= inv; n[2] *= inv;
    }
    const float shade = sp.ambient + sp.diffuse * lam;
    float spec = 0.f;
    if (sp.specular > 0.f && lam > 0.f) {
        // Blinn-Phong: half vector between the light and the viewer (the camera sits at the origin of its frame)
        float pc[3], v[3], hv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pc[i] = ((q0 * P[0][i] + q1 * P[1][i]) + q2 * P[2][i]) * z;
        const float pn = sqrtf((pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] = pn > 0.f ? -pc[i] / pn : 0.f; hv[i] = l[i] + v[i]; }
        const float hn = sqrtf((hv[0] * hv[0] + hv[1] * hv[1]) + hv[2] * hv[2]);
        float nh = hn > 0.f ? ((n[0] * hv[0] + n[1] * hv[1]) + n[2] * hv[2]) / hn : 0.f;
        if (!sp.smooth) nh = fabsf(nh);
        spec = nh > 0.f ? sp.specular * powf(nh, sp.shininess) : 0.f;
    }
    const float* ca = m.colors + ((size_t)o * m.V + i0) * 3;
    const float* cb = m.colors + ((size_t)o * m.V + i1) * 3;
    const float* cc = m.colors + ((size_t)o * m.V + i2) * 3;
    float texel[3] = {1.f, 1.f, 1.f};
    if (m.tex && m.uvs) {
        // bilinear, repeat wrap; texel centres at (i + 0.5) / size, v = 0 at the first row
        const float* ua = m.uvs + ((size_t)o * m.V + i0) * 2;
        const float* ub = m.uvs + ((size_t)o * m.V + i1) * 2;
        const float* uc = m.uvs + ((size_t)o * m.V + i2) * 2;
        float u = ((q0 * ua[0] + q1 * ub[0]) + q2 * uc[0]) * z, vv = ((q0 * ua[1] + q1 * ub[1]) + q2 * uc[1]) * z;
        u = u - floorf(u); vv = vv - floorf(vv);
        const float fx = u * (float)m.TW - 0.5f, fy = vv * (float)m.TH - 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const float tx = fx - x0f, ty = fy - y0f;
        int x0 = (int)x0f, y0 = (int)y0f;
        int x1 = x0 + 1, y1 = y0 + 1;
        x0 = (x0 % m.TW + m.TW) % m.TW; x1 = (x1 % m.TW + m.TW) % m.TW;
        y0 = (y0 % m.TH + m.TH) % m.TH; y1 = (y1 % m.TH + m.TH) % m.TH;
        const float* tb = m.tex + (size_t)o * m.TH * m.TW * 4;
        const float* t00 = tb + ((size_t)y0 * m.TW + x0) * 4; const float* t01 = tb + ((size_t)y0 * m.TW + x1) * 4;
        const float* t10 = tb + ((size_t)y1 * m.TW + x0) * 4; const float* t11 = tb + ((size_t)y1 * m.TW + x1) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            texel[k] = ((1.f - ty) * ((1.f - tx) * t00[k] + tx * t01[k])) + (ty * ((1.f - tx) * t10[k] + tx * t11[k]));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float col = ((((q0 * ca[k] + q1 * cb[k]) + q2 * cc[k]) * z) * texel[k]) * shade + spec;
        col = fminf(fmaxf(col, 0.f), 1.f);
        if (sp.quantize) col = floorf(col * 255.f + 0.5f) / 255.f;
        out[k] = col;
    }
    zo = z;
}


// host side (kernels_raster.hip / kernels_geom.hip)
int launch_render_zbuffer(const float* verts, const int* faces, const int* n_faces, const int* obj_id, const float* TCO, const float* K, int B,
                          int V, int F, int H, int W, void* scratch, hipStream_t s);
int check_mesh_shade(const cosy_mesh_t* mesh, const cosy_shade_t* shade, MeshView* m, ShadeParams* sp);
int launch_render_crop_pack(void* x, int dtype, const float* frames4, const int* im_id, const float* boxes, const void* scratch,
                            const MeshView& m, const int* obj, const float* TCO, const ShadeParams& sp, int B, int h, int w, int H, int W,
                            hipStream_t s);
int render_crop_pack(void* x, int dtype, const cosy_mesh_t* mesh, const cosy_shade_t* shade, const int* obj_id, const float* TCO,
                     const float* K_crop, const float* frames4, const int* im_id, const float* boxes, int B, int N, int h, int w, int H, int W,
                     void* scratch, hipStream_t s);

}  // namespace cosy
