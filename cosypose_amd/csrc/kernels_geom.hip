// Geometry / crop kernels of the pose-refinement loop (gfx950, wave64).
// fp32 throughout; FMA contraction is disabled so that results are bit-comparable with the
// CPU oracle (oracle/cosy_oracle.c), which restates the reference's torch arithmetic.
#include "cosy_common.h"
#include "raster_device.h"
#include <math.h>

#pragma clang fp contract(off)

namespace cosy {

// ----------------------------------------------------------------------------------------
// crop geometry: one wavefront per object.
// project_points_robust + boxes_from_uv (camera_geometry.py:18-42), deepim_boxes
// (cropping.py:7-47), get_K_crop_resize (camera_geometry.py:45-87).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ void k_times_t(const float* K, const float* T, float* P) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) acc += K[i * 3 + k] * T[k * 4 + j];
            P[i * 4 + j] = acc;
        }
}

__device__ __forceinline__ void project1(const float* P, float x, float y, float z, float z_min, float& u, float& v) {
    float s0 = P[0] * x + P[1] * y + P[2] * z + P[3] * 1.0f;
    float s1 = P[4] * x + P[5] * y + P[6] * z + P[7] * 1.0f;
    float s2 = P[8] * x + P[9] * y + P[10] * z + P[11] * 1.0f;
    float zz = s2 > z_min ? s2 : z_min;
    if (s2 != s2) zz = s2;
    u = s0 / zz;
    v = s1 / zz;
}

__global__ __launch_bounds__(64) void crop_geometry_kernel(const float* __restrict__ pts_table, const int* __restrict__ obj_id,
                                                           const float* __restrict__ K, const int* __restrict__ im_id,
                                                           const float* __restrict__ TCO, int P, float z_min, int im_h, int im_w,
                                                           int out_h, int out_w, float lamb, float* __restrict__ boxes_rend,
                                                           float* __restrict__ boxes_crop, float* __restrict__ K_crop) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* Kb = K + (size_t)(im_id ? im_id[b] : b) * 9;
    const float* T = TCO + (size_t)b * 16;
    float Kl[9], Tl[12], Pm[12];
    for (int i = 0; i < 9; ++i) Kl[i] = Kb[i];
    for (int i = 0; i < 12; ++i) Tl[i] = T[i];
    k_times_t(Kl, Tl, Pm);
    const float* pts = pts_table + (size_t)obj_id[b] * P * 3;
    float x1 = INFINITY, y1 = INFINITY, x2 = -INFINITY, y2 = -INFINITY;
    int nan = 0;
    for (int p = lane; p < P; p += 64) {
        float u, v;
        project1(Pm, pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2], z_min, u, v);
        nan |= (u != u) | (v != v);
        x1 = fminf(x1, u); x2 = fmaxf(x2, u);
        y1 = fminf(y1, v); y2 = fmaxf(y2, v);
    }
    x1 = wave_min(x1); y1 = wave_min(y1); x2 = wave_max(x2); y2 = wave_max(y2);
    nan = __any(nan);
    if (lane != 0) return;
    if (nan) x1 = y1 = x2 = y2 = NAN;
    float* br = boxes_rend + (size_t)b * 4;
    br[0] = x1; br[1] = y1; br[2] = x2; br[3] = y2;
    float xc, yc;
    project1(Pm, 0.f, 0.f, 0.f, z_min, xc, yc);
    const int wmax = im_h > im_w ? im_h : im_w, hmin = im_h > im_w ? im_w : im_h;
    const float r = (float)((double)wmax / (double)hmin);
    // obs_boxes == rend_boxes on this path (pose.py:55, cropping.py:69-72)
    float xd = fmaxf(fabsf(x1 - xc), fabsf(x2 - xc));
    float yd = fmaxf(fabsf(y1 - yc), fabsf(y2 - yc));
    if (nan) xd = yd = NAN;
    float width = fmaxf(xd, yd * r) * 2.f * lamb;
    float height = fmaxf(xd / r, yd) * 2.f * lamb;
    if (xd != xd || yd != yd) width = height = NAN;
    float bx0 = xc - width / 2.f, by0 = yc - height / 2.f, bx1 = xc + width / 2.f, by1 = yc + height / 2.f;
    float* bc = boxes_crop + (size_t)b * 4;
    bc[0] = bx0; bc[1] = by0; bc[2] = bx1; bc[3] = by1;
    // get_K_crop_resize: final_width = max(crop_resize), final_height = min(crop_resize)
    const float fw = (float)(out_h > out_w ? out_h : out_w), fh = (float)(out_h > out_w ? out_w : out_h);
    float cw = bx1 - bx0, ch = by1 - by0;
    float cj = (bx0 + bx1) / 2.f, ci = (by0 + by1) / 2.f;
    float cx = Kl[2] + (cw - 1.f) / 2.f - cj;
    float cy = Kl[5] + (ch - 1.f) / 2.f - ci;
    float center_x = (cw - 1.f) / 2.f, center_y = (ch - 1.f) / 2.f;
    float dx = cx - center_x, dy = cy - center_y;
    float sx = fw / cw, sy = fh / ch;
    float scx = (fw - 1.f) / 2.f, scy = (fh - 1.f) / 2.f;
    float* ko = K_crop + (size_t)b * 9;
    for (int i = 0; i < 9; ++i) ko[i] = Kl[i];
    ko[0] = sx * Kl[0];
    ko[4] = sy * Kl[4];
    ko[2] = scx + sx * dx;
    ko[5] = scy + sy * dy;
}

int launch_crop_geometry(const float* pts_table, const int* obj_id, const float* K, const int* im_id, const float* TCO,
                         int B, int P, float z_min, int im_h, int im_w, int out_h, int out_w, float lamb,
                         float* boxes_rend, float* boxes_crop, float* K_crop, hipStream_t s) {
    if (B == 0) return COSY_OK;
    hipLaunchKernelGGL(crop_geometry_kernel, dim3(B), dim3(64), 0, s, pts_table, obj_id, K, im_id, TCO, P, z_min, im_h,
                       im_w, out_h, out_w, lamb, boxes_rend, boxes_crop, K_crop);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// ----------------------------------------------------------------------------------------
// roi_align (torchvision 0.4.2 semantics: no `aligned`, roi size clamped to >= 1, taps outside
// [-1, size] contribute 0, clamp at the borders).  One thread per output pixel, C channels.
// ----------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void roi_pixel(const float* __restrict__ img /* C planes of h*w */, int h, int w, float x1, float y1,
                                          float bin_h, float bin_w, int ph, int pw, int g, float* acc) {
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    const size_t plane = (size_t)h * w;
    for (int iy = 0; iy < g; ++iy) {
        const float yy = y1 + ph * bin_h + ((float)iy + .5f) * bin_h / (float)g;
        for (int ix = 0; ix < g; ++ix) {
            const float xx = x1 + pw * bin_w + ((float)ix + .5f) * bin_w / (float)g;
            float x = xx, y = yy;
            if (y < -1.0f || y > (float)h || x < -1.0f || x > (float)w) continue;
            if (y <= 0) y = 0;
            if (x <= 0) x = 0;
            int yl = (int)y, xl = (int)x, yh, xh;
            if (yl >= h - 1) { yh = yl = h - 1; y = (float)yl; } else yh = yl + 1;
            if (xl >= w - 1) { xh = xl = w - 1; x = (float)xl; } else xh = xl + 1;
            const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
            const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
            const int o1 = yl * w + xl, o2 = yl * w + xh, o3 = yh * w + xl, o4 = yh * w + xh;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float* p = img + c * plane;
                acc[c] += w1 * p[o1] + w2 * p[o2] + w3 * p[o3] + w4 * p[o4];
            }
        }
    }
    const float count = (float)(g * g);
    for (int c = 0; c < C; ++c) acc[c] = acc[c] / count;
}

__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ images, const int* __restrict__ im_id,
                                                        const float* __restrict__ boxes, int C, int h, int w, int PH, int PW,
                                                        int g, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= PH * PW) return;
    const int ph = pix / PW, pw = pix % PW;
    const float* bx = boxes + (size_t)b * 4;
    const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
    const float roi_w = fmaxf(x2 - x1, 1.f), roi_h = fmaxf(y2 - y1, 1.f);
    const float bin_h = roi_h / (float)PH, bin_w = roi_w / (float)PW;
    const float* img = images + (size_t)(im_id ? im_id[b] : b) * C * h * w;
    for (int c = 0; c < C; ++c) {
        float a[1];
        roi_pixel<1>(img + (size_t)c * h * w, h, w, x1, y1, bin_h, bin_w, ph, pw, g, a);
        out[(((size_t)b * C + c) * PH + ph) * PW + pw] = a[0];
    }
}

int launch_roi_align(const float* images, const int* im_id, const float* boxes, int B, int N, int C, int h, int w,
                     int out_h, int out_w, int sampling, float* out, hipStream_t s) {
    (void)N;
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(sampling > 0, "roi_align: sampling_ratio must be > 0 (the reference uses 4)");
    hipLaunchKernelGGL(roi_align_kernel, dim3(cdiv(out_h * out_w, 256), B), dim3(256), 0, s, images, im_id, boxes, C, h, w,
                       out_h, out_w, sampling, out);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// crop (3 ch) + render (3 ch) -> one NHWC8 pixel of the network input (channels 6,7 = 0)
template <typename T>
__device__ __forceinline__ void store_px8(T* dst, const float* v);
template <>
__device__ __forceinline__ void store_px8<f16_t>(f16_t* dst, const float* v) {
    f16x8 o = {(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3], (f16_t)v[4], (f16_t)v[5], (f16_t)0.f, (f16_t)0.f};
    *(f16x8*)dst = o;
}
template <>
__device__ __forceinline__ void store_px8<float>(float* dst, const float* v) {
    ((f32x4*)dst)[0] = f32x4{v[0], v[1], v[2], v[3]};
    ((f32x4*)dst)[1] = f32x4{v[4], v[5], 0.f, 0.f};
}
template <>
__device__ __forceinline__ void store_px8<bf16_t>(bf16_t* dst, const float* v) {
    bf16x8 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3], (bf16_t)v[4], (bf16_t)v[5], (bf16_t)0.f, (bf16_t)0.f};
    *(bf16x8*)dst = o;
}

// frames (N,3,h,w) fp32 planar -> (N,h,w,4) fp32 interleaved: one 16-byte load fetches a pixel's RGB
__global__ __launch_bounds__(256) void frames_to_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, int hw) {
    const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const float* s = src + (size_t)n * 3 * hw + i;
    ((f32x4*)dst)[(size_t)n * hw + i] = f32x4{s[0], s[hw], s[2 * (size_t)hw], 0.f};
}
// the same from uint8 frames (what the datasets deliver): value / 255.f, the arithmetic of the reference's `images.float() / 255.`
// (training/pose_forward_loss.py:24) without the two fp32 passes over the frames
__global__ __launch_bounds__(256) void frames_u8_to_nhwc4_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int hw) {
    const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const unsigned char* s = src + (size_t)n * 3 * hw + i;
    ((f32x4*)dst)[(size_t)n * hw + i] = f32x4{(float)s[0] / 255.f, (float)s[hw] / 255.f, (float)s[2 * (size_t)hw] / 255.f, 0.f};
}
int launch_frames_u8_to_nhwc4(const unsigned char* images, float* out, int N, int h, int w, hipStream_t s) {
    if (N == 0) return COSY_OK;
    hipLaunchKernelGGL(frames_u8_to_nhwc4_kernel, dim3(cdiv(h * w, 256), N), dim3(256), 0, s, images, out, h * w);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int launch_frames_to_nhwc4(const float* images, float* out, int N, int h, int w, hipStream_t s) {
    if (N == 0) return COSY_OK;
    hipLaunchKernelGGL(frames_to_nhwc4_kernel, dim3(cdiv(h * w, 256), N), dim3(256), 0, s, images, out, h * w);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// One axis of the 4 roi_align sample points of an output pixel (torchvision 0.4.2 rules, see roi_pixel):
// validity, clamped low/high pixel and the two bilinear weights of every sample.
struct AxisTaps { int lo[4], hi[4]; float wl[4], wh[4]; int first, last; };
// sample i (0..3) of output position p along one axis; returns validity (weights are 0 when invalid)
__device__ __forceinline__ bool axis_sample(float start, float bin, int p, int size, int i, int& lo, int& hi, float& wl, float& wh) {
    float c = start + p * bin + ((float)i + .5f) * bin / 4.f;
    const bool valid = c >= -1.0f && c <= (float)size;   // NaN coordinates (non-finite poses) count as outside
    if (!valid || c <= 0) c = 0;
    lo = (int)c;
    if (lo >= size - 1) { hi = lo = size - 1; c = (float)lo; } else hi = lo + 1;
    const float l = c - lo, h = 1.f - l;
    wl = valid ? h : 0.f; wh = valid ? l : 0.f;
    return valid;
}
__device__ __forceinline__ void axis_taps(float start, float bin, int p, int size, AxisTaps& t) {
    t.first = 0x7fffffff; t.last = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool valid = axis_sample(start, bin, p, size, i, t.lo[i], t.hi[i], t.wl[i], t.wh[i]);
        if (valid) { t.first = t.lo[i] < t.first ? t.lo[i] : t.first; t.last = t.hi[i] > t.last ? t.hi[i] : t.last; }
    }
}
__device__ __forceinline__ float axis_weight(const AxisTaps& t, int q) {
    float w = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) w += (t.lo[i] == q ? t.wl[i] : 0.f) + (t.hi[i] == q ? t.wh[i] : 0.f);
    return w;
}

// roi_align(sampling 4) of one output pixel from an NHWC4 frame.  The 16 bilinear samples are separable
// (validity, clamping and weights factor per axis), so the pixel is a (<= 6 x 6) window of frame pixels weighted by
// the per-axis tap sums: 9-16 16-byte loads instead of 64 taps x 3 channels.  Falls back to the sample loop for
// huge bins.  Mathematically identical to the reference sum; fp32 rounding differs at the 1e-7 level.
template <bool WINDOW4 = true>
__device__ __forceinline__ void roi_pixel_nhwc4(const f32x4* __restrict__ img, int h, int w, float x1, float y1, float bin_h,
                                                float bin_w, int ph, int pw, float* acc) {
    AxisTaps ty, tx;
    axis_taps(y1, bin_h, ph, h, ty);
    axis_taps(x1, bin_w, pw, w, tx);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (ty.last >= 0 && tx.last >= 0) {
        // WINDOW4 = false (the tiled kernel's rare fallback): windows of up to 4x4 take the row loop below -- the same weights in the
        // same order, 64 fewer live registers
        if (WINDOW4 && ty.last - ty.first < 4 && tx.last - tx.first < 4) {
            // common case (bins up to ~2.6 frame pixels): a 4x4 window, all 16 loads issued back to back so that their
            // latencies overlap; positions past `last` carry weight 0 and read a clamped (valid) address
            float ax[4], ay[4];
            int ox[4], oy[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                ax[d] = axis_weight(tx, tx.first + d);
                ay[d] = axis_weight(ty, ty.first + d);
                ox[d] = min(tx.first + d, w - 1);
                oy[d] = min(ty.first + d, h - 1) * w;
            }
            f32x4 p[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < 4; ++d) p[r][d] = img[oy[r] + ox[d]];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float wgt = ay[r] * ax[d];
                    a0 += wgt * p[r][d][0]; a1 += wgt * p[r][d][1]; a2 += wgt * p[r][d][2];
                }
        } else if (!WINDOW4 && ty.last - ty.first < 6 && tx.last - tx.first < 6) {
            // rolled twin of the branch below (same weights, same order)
#pragma unroll 1
            for (int Y = ty.first; Y <= ty.last; ++Y) {
                const float ay = axis_weight(ty, Y);
                const f32x4* row = img + (size_t)Y * w;
#pragma unroll 1
                for (int X = tx.first; X <= tx.last; ++X) {
                    const f32x4 p = row[X];
                    const float wgt = ay * axis_weight(tx, X);
                    a0 += wgt * p[0]; a1 += wgt * p[1]; a2 += wgt * p[2];
                }
            }
        } else if (ty.last - ty.first < 6 && tx.last - tx.first < 6) {
            float ax[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) ax[d] = axis_weight(tx, tx.first + d);
#pragma unroll 1
            for (int Y = ty.first; Y <= ty.last; ++Y) {
                const float ay = axis_weight(ty, Y);
                const f32x4* row = img + (size_t)Y * w + tx.first;
#pragma unroll
                for (int d = 0; d < 6; ++d)
                    if (tx.first + d <= tx.last) {
                        const f32x4 p = row[d];
                        const float wgt = ay * ax[d];
                        a0 += wgt * p[0]; a1 += wgt * p[1]; a2 += wgt * p[2];
                    }
            }
        } else {
            // huge bins: the plain 16-sample loop.  Taps are recomputed per sample (no dynamically indexed arrays:
            // those would push the tap tables of every path into scratch)
#pragma unroll 1
            for (int iy = 0; iy < 4; ++iy) {
                int ylo, yhi; float ywl, ywh;
                axis_sample(y1, bin_h, ph, h, iy, ylo, yhi, ywl, ywh);
#pragma unroll 1
                for (int ix = 0; ix < 4; ++ix) {
                    int xlo, xhi; float xwl, xwh;
                    axis_sample(x1, bin_w, pw, w, ix, xlo, xhi, xwl, xwh);
                    const f32x4 p1 = img[(size_t)ylo * w + xlo], p2 = img[(size_t)ylo * w + xhi];
                    const f32x4 p3 = img[(size_t)yhi * w + xlo], p4 = img[(size_t)yhi * w + xhi];
                    const float w1 = ywl * xwl, w2 = ywl * xwh, w3 = ywh * xwl, w4 = ywh * xwh;
                    a0 += w1 * p1[0] + w2 * p2[0] + w3 * p3[0] + w4 * p4[0];
                    a1 += w1 * p1[1] + w2 * p2[1] + w3 * p3[1] + w4 * p4[1];
                    a2 += w1 * p1[2] + w2 * p2[2] + w3 * p3[2] + w4 * p4[2];
                }
            }
        }
    }
    acc[0] = a0 / 16.f; acc[1] = a1 / 16.f; acc[2] = a2 / 16.f;
}

// Per-crop tap tables: the roi_align taps of an output pixel are separable, and along one axis they depend on the output
// row (or column) only -- 256 + 256 entries per crop instead of 65,536 x 2 evaluations of ~200 instructions each (the crop
// kernel was VALU-bound: ~400 instructions per pixel, three quarters of them this arithmetic).  Entry = first frame pixel of
// the window, its extent, and the summed weights of the window's 4 positions (same functions, same values as the on-the-fly
// path); entries with an extent > 4 (bins above ~2.6 frame pixels) send the pixel to the on-the-fly path.
struct CropTap { int first, span; float w[4]; int pad[2]; };      // span = last - first; -1: no valid sample
static_assert(sizeof(CropTap) == 32, "one tap entry = two 16-byte loads");
__global__ __launch_bounds__(256) void crop_taps_kernel(const float* __restrict__ boxes, int B, int h, int w, int PH, int PW,
                                                        CropTap* __restrict__ taps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * (PH + PW)) return;
    const int b = i / (PH + PW), q = i - b * (PH + PW);
    const float* bx = boxes + (size_t)b * 4;
    const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
    const float roi_w = fmaxf(x2 - x1, 1.f), roi_h = fmaxf(y2 - y1, 1.f);
    const bool yaxis = q < PH;
    AxisTaps t;
    if (yaxis) axis_taps(y1, roi_h / (float)PH, q, h, t); else axis_taps(x1, roi_w / (float)PW, q - PH, w, t);
    CropTap o;
    o.first = t.last < 0 ? 0 : t.first;
    o.span = t.last < 0 ? -1 : t.last - t.first;
#pragma unroll
    for (int d = 0; d < 4; ++d) o.w[d] = (t.last >= 0 && o.span < 4) ? axis_weight(t, t.first + d) : 0.f;
    o.pad[0] = o.pad[1] = 0;
    taps[i] = o;
}
// the 4x4-window path of roi_pixel_nhwc4 driven by two table entries; returns false when the pixel needs the general path
__device__ __forceinline__ bool roi_pixel_table(const f32x4* __restrict__ img, int h, int w, const CropTap& ty, const CropTap& tx, float* acc) {
    if (ty.span >= 4 || tx.span >= 4) return false;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (ty.span >= 0 && tx.span >= 0) {
        int ox[4], oy[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) { ox[d] = min(tx.first + d, w - 1); oy[d] = min(ty.first + d, h - 1) * w; }
        // rows / columns of the window that ANY lane of the wave needs (wave-uniform: whole gather instructions drop out; a
        // crop that magnifies its box has 2-3 pixel windows, not 4).  Skipped positions carry weight 0.
        const int rows = __ballot(ty.span > 2) ? 4 : __ballot(ty.span > 1) ? 3 : 2;
        const int cols = __ballot(tx.span > 2) ? 4 : __ballot(tx.span > 1) ? 3 : 2;
        f32x4 p[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < rows) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (d < cols) p[r][d] = img[oy[r] + ox[d]];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < rows) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (d < cols) {
                        const float wgt = ty.w[r] * tx.w[d];
                        a0 += wgt * p[r][d][0]; a1 += wgt * p[r][d][1]; a2 += wgt * p[r][d][2];
                    }
            }
    }
    acc[0] = a0 / 16.f; acc[1] = a1 / 16.f; acc[2] = a2 / 16.f;
    return true;
}
// observed-crop channels of one output pixel: table-driven when a table is given, else (or for huge bins) on the fly
__device__ __forceinline__ void crop_pixel(const f32x4* __restrict__ img, const float* __restrict__ bx, const CropTap* __restrict__ taps, int b,
                                           int h, int w, int PH, int PW, int ph, int pw, float* v) {
    if (taps) {
        const CropTap* tb = taps + (size_t)b * (PH + PW);
        const CropTap ty = tb[ph], tx = tb[PH + pw];
        if (roi_pixel_table(img, h, w, ty, tx, v)) return;
    }
    const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
    const float roi_w = fmaxf(x2 - x1, 1.f), roi_h = fmaxf(y2 - y1, 1.f);
    roi_pixel_nhwc4(img, h, w, x1, y1, roi_h / (float)PH, roi_w / (float)PW, ph, pw, v);
}
int launch_crop_taps(const float* boxes, int B, int h, int w, int H, int W, void* taps, hipStream_t s) {
    if (B == 0) return COSY_OK;
    hipLaunchKernelGGL(crop_taps_kernel, dim3(cdiv((long)B * (H + W), 256)), dim3(256), 0, s, boxes, B, h, w, H, W, (CropTap*)taps);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
size_t crop_taps_bytes(int B, int H, int W) { return (size_t)B * (H + W) * sizeof(CropTap); }

template <typename T>
__global__ __launch_bounds__(256) void crop_pack_kernel(T* __restrict__ x, const float* __restrict__ frames4,
                                                        const int* __restrict__ im_id, const float* __restrict__ boxes,
                                                        const float* __restrict__ renders, const CropTap* __restrict__ taps, int B, int h,
                                                        int w, int PH, int PW, int dbg) {
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2.  All pixel blocks of
    // one crop get ids of the same residue, so the frame region under a crop's box is fetched into ONE L2, once
    // (crop-major ids had every XCD fetch every region: 1.2 GB of L2 fills per launch for 0.36 GB of distinct data).
    const int id = blockIdx.x, xcd = id & 7, bpc = (PH * PW + 255) / 256;
    const int j = id >> 3, b = (j / bpc) * 8 + xcd;
    if (b >= B) return;
    const int pix = (j % bpc) * 256 + threadIdx.x;
    if (pix >= PH * PW) return;
    const int ph = pix / PW, pw = pix % PW;
    const f32x4* img = (const f32x4*)frames4 + (size_t)(im_id ? im_id[b] : b) * h * w;
    float v[6];
    if (COSY_DBG(dbg & 1)) { const f32x4 q = img[pix % (h * w)]; v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; }   // dbg 1: one coalesced load, no window
    else crop_pixel(img, boxes + (size_t)b * 4, taps, b, h, w, PH, PW, ph, pw, v);
    const float* r = renders + (size_t)b * 3 * PH * PW + pix;
    if (COSY_DBG(dbg & 2)) v[3] = v[4] = v[5] = 0.f;                                                      // dbg 2: no render loads
    else { v[3] = r[0]; v[4] = r[(size_t)PH * PW]; v[5] = r[(size_t)2 * PH * PW]; }
    if (COSY_DBG(dbg & 4) && v[0] != 123.f) return;                                                      // dbg 4: no stores
    store_px8<T>(x + ((size_t)b * PH * PW + pix) * 8, v);
}

// ----------------------------------------------------------------------------------------
// Tiled crop + pack (round 6): the frame window under an output tile goes through the LDS.
// crop_pack_kernel gathers every output pixel's 2x2 .. 4x4 window straight from the frame: 9-16 sixteen-byte gathers per
// pixel whose addresses repeat between neighbouring lanes and rows (a crop that magnifies its box reads every frame pixel
// ~20 times), behind a dependent table load -- the launch is bound by the vector-memory address path and by two memory
// latencies per pixel, not by bytes.  Here a workgroup owns a tile of 16 output rows x 64 columns (wave = 4 rows, lane =
// column): the tile's frame window -- rows / columns [first, first + span] of the tile's tap entries -- is loaded ONCE,
// coalesced, into the LDS, and every window position is then an LDS read.  Same tap tables, same weights, same order of
// the sum as roi_pixel_table -> bit-identical results.  Tiles whose window does not fit (bins above ~1.2 frame pixels) are
// walked in 2 or 4 passes of fewer rows; tap entries wider than 4 pixels (bins above ~2.6) take the per-pixel path.
// ----------------------------------------------------------------------------------------
constexpr int CROP_TW = 64, CROP_TH = 16, CROP_LDS_PX = 2048;      // 32 KB of frame pixels (fp32 RGB + pad) per workgroup
__device__ __forceinline__ int wave_min_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
template <typename T>
__global__ __launch_bounds__(256, 4) void crop_pack_tile_kernel(T* __restrict__ x, const float* __restrict__ frames4,
                                                                const int* __restrict__ im_id, const float* __restrict__ boxes,
                                                                const float* __restrict__ renders, const CropTap* __restrict__ taps, int B, int h,
                                                                int w, int PH, int PW) {
    __shared__ f32x4 tile[CROP_LDS_PX];
    const int tcols = (PW + CROP_TW - 1) / CROP_TW, tpc = tcols * ((PH + CROP_TH - 1) / CROP_TH);
    // XCD-aware order as in crop_pack_kernel: all tiles of one crop on one XCD (one L2 holds the frame region under its box)
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3, b = (j / tpc) * 8 + xcd;
    if (b >= B) return;
    const int t = j % tpc, tr = t / tcols, tc = t - tr * tcols;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: the rows' tap entries come by scalar loads
    const int pw = tc * CROP_TW + lane;
    const bool colok = pw < PW;
    const f32x4* img = (const f32x4*)frames4 + (size_t)(im_id ? im_id[b] : b) * h * w;
    const CropTap* tb = taps + (size_t)b * (PH + PW);
    const CropTap tx = tb[PH + min(pw, PW - 1)];
    // the tile's columns of the frame: [xf, xl] over the valid entries (the same in every wave of the workgroup).  (Extents derived from the box
    // instead -- a superset with a pixel of margin, so that the window's loads wait for no tap entry -- were measured SLOWER: 206 vs 180 us per
    // launch, the margins push more tiles over the LDS budget into two passes; profiles/r06_crop_tiled.txt)
    const bool xvalid = colok && tx.span >= 0;
    const int xf = wave_min_i(xvalid ? tx.first : 0x7fffffff), xl = wave_max_i(xvalid ? tx.first + tx.span : -1);
    const bool xwide = __ballot(colok && tx.span >= 4) != 0;        // entries wider than 4 pixels: per-pixel path
    const int ncols = xl >= 0 ? xl - xf + 1 : 0;       // (no valid column in the tile: nothing is loaded, every pixel of it is zero)
    const float* bx = boxes + (size_t)b * 4;
    const float bin_h = fmaxf(bx[3] - bx[1], 1.f) / (float)PH;
    // passes: the fewest of 1 / 2 / 4 whose rows' window is expected to fit (checked per pass against the real entries)
    int np = 1;
    while (np < 4 && ((int)fminf((float)(CROP_TH / np) * bin_h, 65536.f) + 5) * ncols > CROP_LDS_PX) np *= 2;      // (huge boxes: bounded before the conversion)
    const int rp = CROP_TH / np, rpw = rp / 4;                      // rows per pass, rows per wave and pass (4 / 2 / 1)
    const size_t HW = (size_t)PH * PW;
    for (int p = 0; p < np; ++p) {
        const int row0 = tr * CROP_TH + p * rp;                     // first output row of the pass
        if (row0 >= PH) break;
        // the pass's rows of the frame
        const int rq = row0 + lane;
        const bool rok = lane < rp && rq < PH;
        const CropTap tq = tb[min(rq, PH - 1)];
        const bool yvalid = rok && tq.span >= 0;
        const int yf = wave_min_i(yvalid ? tq.first : 0x7fffffff), yl = wave_max_i(yvalid ? tq.first + tq.span : -1);
        const bool ywide = __ballot(rok && tq.span >= 4) != 0;
        const int nrows = yl >= 0 ? yl - yf + 1 : 0;
        const bool any = xl >= 0 && yl >= 0;                        // else: no valid sample in the pass -> zeros
        const bool fits = any && nrows * ncols <= CROP_LDS_PX;
        // this lane's pixels: rows ph0 .. ph0 + rpw - 1 of column pw; their tap entries and render channels are requested first
        const int ph0 = row0 + wv * rpw;
        float rv[4][3];
        CropTap tys[4];
        const bool slow = any && (!fits || xwide || ywide);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < rpw && ph0 + i < PH) {
                tys[i] = tb[ph0 + i];
                if (colok) {
                    const float* r = renders + (size_t)b * 3 * HW + (size_t)(ph0 + i) * PW + pw;
                    rv[i][0] = r[0]; rv[i][1] = r[HW]; rv[i][2] = r[2 * HW];
                }
            }
        if (p) __syncthreads();                                     // the previous pass's reads of the tile
        if (fits) {
            for (int r = wv; r < nrows; r += 4) {
                const f32x4* src = img + (size_t)(yf + r) * w + xf;
                for (int c = lane; c < ncols; c += 64) tile[r * ncols + c] = src[c];
            }
        }
        __syncthreads();
        if (slow) {
            // rare: tap entries wider than 4 pixels or a window beyond the LDS tile -> the per-pixel path (rolled: its code is large)
#pragma unroll 1
            for (int i = 0; i < rpw; ++i) {
                const int ph = ph0 + i;
                if (ph >= PH || !colok) break;
                float v[6];
                roi_pixel_nhwc4<false>(img, h, w, bx[0], bx[1], bin_h, fmaxf(bx[2] - bx[0], 1.f) / (float)PW, ph, pw, v);
                const float* r = renders + (size_t)b * 3 * HW + (size_t)ph * PW + pw;
                v[3] = r[0]; v[4] = r[HW]; v[5] = r[2 * HW];
                store_px8<T>(x + ((size_t)b * HW + (size_t)ph * PW + pw) * 8, v);
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ph = ph0 + i;
            if (i >= rpw || ph >= PH) break;                        // wave-uniform
            const CropTap ty = tys[i];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            if (any && ty.span >= 0 && xvalid) {
                int ox[4], oy[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) { ox[d] = min(tx.first + d, xl) - xf; oy[d] = (min(ty.first + d, yl) - yf) * ncols; }
                const int rows = ty.span + 1;                       // wave-uniform; positions past an entry's span carry weight 0
                const int cols = __ballot(tx.span > 2 && xvalid) ? 4 : __ballot(tx.span > 1 && xvalid) ? 3 : 2;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < rows) {
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            if (d < cols) {
                                const f32x4 q = tile[oy[r] + ox[d]];
                                const float wgt = ty.w[r] * tx.w[d];
                                a0 += wgt * q[0]; a1 += wgt * q[1]; a2 += wgt * q[2];
                            }
                    }
            }
            if (colok) {
                const float v[6] = {a0 / 16.f, a1 / 16.f, a2 / 16.f, rv[i][0], rv[i][1], rv[i][2]};
                store_px8<T>(x + ((size_t)b * HW + (size_t)ph * PW + pw) * 8, v);
            }
        }
    }
}

// render + crop + pack: the rasteriser's resolve pass and the crop in one kernel (one 16-byte store per pixel)
template <typename T>
__global__ __launch_bounds__(256) void render_crop_pack_kernel(T* __restrict__ x, const float* __restrict__ frames4,
                                                               const int* __restrict__ im_id, const float* __restrict__ boxes,
                                                               const unsigned long long* __restrict__ zbuf, const float* __restrict__ uvz,
                                                               MeshView m, const int* __restrict__ obj, const float* __restrict__ TCO,
                                                               ShadeParams sp, const CropTap* __restrict__ taps, int B, int h, int w, int PH,
                                                               int PW) {
    const int id = blockIdx.x, xcd = id & 7, bpc = (PH * PW + 255) / 256;
    const int j = id >> 3, b = (j / bpc) * 8 + xcd;
    if (b >= B) return;
    const int pix = (j % bpc) * 256 + threadIdx.x;
    if (pix >= PH * PW) return;
    const int ph = pix / PW, pw = pix % PW;
    const f32x4* img = (const f32x4*)frames4 + (size_t)(im_id ? im_id[b] : b) * h * w;
    float v[6], zo;
    crop_pixel(img, boxes + (size_t)b * 4, taps, b, h, w, PH, PW, ph, pw, v);
    resolve_pixel(zbuf[(size_t)b * PH * PW + pix], uvz + (size_t)b * m.V * 3, m, obj[b], TCO + (size_t)b * 16, pw, ph, sp, v + 3, zo);
    store_px8<T>(x + ((size_t)b * PH * PW + pix) * 8, v);
}

int launch_render_crop_pack(void* x, int dtype, const float* frames4, const int* im_id, const float* boxes, const void* scratch,
                            const MeshView& m, const int* obj, const float* TCO, const ShadeParams& sp, int B, int h, int w, int H, int W,
                            hipStream_t s) {
    if (B == 0) return COSY_OK;
    const unsigned long long* zbuf = (const unsigned long long*)scratch;
    const float* uvz = (const float*)(zbuf + (size_t)B * H * W);
    CropTap* taps = (CropTap*)(uvz + (((size_t)B * m.V * 3 + 7) & ~(size_t)7));      // behind [zbuf | uvz], 32-byte aligned
    int rc;
    if ((rc = launch_crop_taps(boxes, B, h, w, H, W, taps, s))) return rc;
    dim3 grid((unsigned)(cdiv(H * W, 256) * cdiv(B, 8) * 8));
    COSY_DISPATCH_STMT(dtype, hipLaunchKernelGGL(render_crop_pack_kernel<T>, grid, dim3(256), 0, s, (T*)x, frames4, im_id, boxes, zbuf, uvz, m,
                                                 obj, TCO, sp, (const CropTap*)taps, B, h, w, H, W));
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// taps_ws: crop_taps_bytes(B, H, W) bytes of scratch for the tap tables, or null (taps are then evaluated per pixel)
int launch_crop_pack(void* x, int dtype, const float* frames4, const int* im_id, const float* boxes, const float* renders,
                     int B, int N, int h, int w, int H, int W, void* taps_ws, hipStream_t s) {
    (void)N;
    if (B == 0) return COSY_OK;
    int rc;
    if (taps_ws && (rc = launch_crop_taps(boxes, B, h, w, H, W, taps_ws, s))) return rc;
    dim3 grid((unsigned)(cdiv(H * W, 256) * cdiv(B, 8) * 8));
    if (taps_ws && tune_int("COSY_CROP_TILED", 1)) {
        dim3 tgrid((unsigned)(cdiv(W, CROP_TW) * cdiv(H, CROP_TH) * cdiv(B, 8) * 8));
        COSY_DISPATCH_STMT(dtype, hipLaunchKernelGGL(crop_pack_tile_kernel<T>, tgrid, dim3(256), 0, s, (T*)x, frames4, im_id, boxes, renders,
                                                     (const CropTap*)taps_ws, B, h, w, H, W));
    } else
        COSY_DISPATCH_STMT(dtype, hipLaunchKernelGGL(crop_pack_kernel<T>, grid, dim3(256), 0, s, (T*)x, frames4, im_id, boxes, renders,
                                                     (const CropTap*)taps_ws, B, h, w, H, W, tune_int("COSY_CROP_DBG", 0)));
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_nchw_kernel(T* __restrict__ x, const float* __restrict__ src, int HW) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    float v[6];
    for (int c = 0; c < 6; ++c) v[c] = src[((size_t)b * 6 + c) * HW + pix];
    store_px8<T>(x + ((size_t)b * HW + pix) * 8, v);
}

int launch_pack_nchw(void* x, int dtype, const float* x_nchw6, int B, int H, int W, hipStream_t s) {
    if (B == 0) return COSY_OK;
    dim3 grid(cdiv(H * W, 256), B);
    COSY_DISPATCH_STMT(dtype, hipLaunchKernelGGL(pack_nchw_kernel<T>, grid, dim3(256), 0, s, (T*)x, x_nchw6, H * W));
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// ----------------------------------------------------------------------------------------
// pose update: ortho6d -> dR (rotations.py:6-21), apply_imagespace_predictions (cosypose_ops.py:10-31)
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pose_update_kernel(const float* __restrict__ TCO, const float* __restrict__ Kc,
                                                         const float* __restrict__ pose9, int B, float* __restrict__ out) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* T = TCO + (size_t)b * 16; const float* k = Kc + (size_t)b * 9; const float* p = pose9 + (size_t)b * 9;
    float* o = out + (size_t)b * 16;
    const float a0 = p[0], a1 = p[1], a2 = p[2], c0 = p[3], c1 = p[4], c2 = p[5];
    const float na = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    const float x0 = a0 / na, x1 = a1 / na, x2 = a2 / na;
    float z0 = x1 * c2 - x2 * c1, z1 = x2 * c0 - x0 * c2, z2 = x0 * c1 - x1 * c0;
    const float nz = sqrtf(z0 * z0 + z1 * z1 + z2 * z2);
    z0 /= nz; z1 /= nz; z2 /= nz;
    const float y0 = z1 * x2 - z2 * x1, y1 = z2 * x0 - z0 * x2, y2 = z0 * x1 - z1 * x0;
    const float d[9] = {x0, y0, z0, x1, y1, z1, x2, y2, z2};
    float Tl[16];
    for (int i = 0; i < 16; ++i) Tl[i] = T[i];
    float ol[16];
    for (int i = 0; i < 16; ++i) ol[i] = Tl[i];
    const float zsrc = Tl[11], ztgt = p[8] * zsrc;
    ol[11] = ztgt;
    ol[3] = (p[6] / k[0] + Tl[3] / zsrc) * ztgt;
    ol[7] = (p[7] / k[4] + Tl[7] / zsrc) * ztgt;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = 0.f;
            for (int q = 0; q < 3; ++q) acc += d[i * 3 + q] * Tl[q * 4 + j];
            ol[i * 4 + j] = acc;
        }
    for (int i = 0; i < 16; ++i) o[i] = ol[i];
}

int launch_pose_update(const float* TCO_in, const float* K_crop, const float* pose9, int B, float* TCO_out, hipStream_t s) {
    if (B == 0) return COSY_OK;
    hipLaunchKernelGGL(pose_update_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, TCO_in, K_crop, pose9, B, TCO_out);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// ----------------------------------------------------------------------------------------
// pose initialisation (cosypose_ops.py:121-173)
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tco_init_boxes_kernel(const float* __restrict__ boxes, const float* __restrict__ K,
                                                            const int* __restrict__ im_id, int B, float z, float* __restrict__ TCO) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* bx = boxes + (size_t)b * 4; const float* k = K + (size_t)(im_id ? im_id[b] : b) * 9;
    float* T = TCO + (size_t)b * 16;
    const float uc = (bx[0] + bx[2]) / 2.f, vc = (bx[1] + bx[3]) / 2.f;
    float o[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    o[3] = ((uc - k[2]) * z) / k[0];
    o[7] = ((vc - k[5]) * z) / k[4];
    o[11] = z;
    for (int i = 0; i < 16; ++i) T[i] = o[i];
}

int launch_tco_init_from_boxes(const float* boxes, const float* K, const int* im_id, int B, float z, float* TCO, hipStream_t s) {
    if (B == 0) return COSY_OK;
    hipLaunchKernelGGL(tco_init_boxes_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, boxes, K, im_id, B, z, TCO);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

__global__ __launch_bounds__(64) void tco_init_zup_kernel(const float* __restrict__ boxes, const float* __restrict__ pts_table,
                                                          const int* __restrict__ obj_id, const float* __restrict__ K,
                                                          const int* __restrict__ im_id, int P, float* __restrict__ TCO) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* bx = boxes + (size_t)b * 4; const float* k = K + (size_t)(im_id ? im_id[b] : b) * 9;
    const float uc = (bx[0] + bx[2]) / 2.f, vc = (bx[1] + bx[3]) / 2.f;
    const float zg = 1.0f;
    float T[16] = {0, 1, 0, 0, 0, 0, -1, 0, -1, 0, 0, zg, 0, 0, 0, 1};
    T[3] = ((uc - k[2]) * zg) / k[0];
    T[7] = ((vc - k[5]) * zg) / k[4];
    const float* pts = pts_table + (size_t)obj_id[b] * P * 3;
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int p = lane; p < P; p += 64) {
        const float q0 = pts[p * 3], q1 = pts[p * 3 + 1], q2 = pts[p * 3 + 2];
        const float cx = (T[0] * q0 + T[1] * q1 + T[2] * q2) + T[3];
        const float cy = (T[4] * q0 + T[5] * q1 + T[6] * q2) + T[7];
        xmin = fminf(xmin, cx); xmax = fmaxf(xmax, cx);
        ymin = fminf(ymin, cy); ymax = fmaxf(ymax, cy);
    }
    xmin = wave_min(xmin); ymin = wave_min(ymin); xmax = wave_max(xmax); ymax = wave_max(ymax);
    if (lane != 0) return;
    const float dx3 = xmax - xmin, dy3 = ymax - ymin;
    const float bdx = (bx[2] - bx[0]) + 1.f, bdy = (bx[3] - bx[1]) + 1.f;
    const float zdx = k[0] * dx3 / bdx, zdy = k[4] * dy3 / bdy;
    const float z = (zdy + zdx) / 2.f;
    T[3] = ((uc - k[2]) * z) / k[0];
    T[7] = ((vc - k[5]) * z) / k[4];
    T[11] = z;
    float* o = TCO + (size_t)b * 16;
    for (int i = 0; i < 16; ++i) o[i] = T[i];
}

int launch_tco_init_zup(const float* boxes, const float* pts_table, const int* obj_id, const float* K, const int* im_id,
                        int B, int P, float* TCO, hipStream_t s) {
    if (B == 0) return COSY_OK;
    hipLaunchKernelGGL(tco_init_zup_kernel, dim3(B), dim3(64), 0, s, boxes, pts_table, obj_id, K, im_id, P, TCO);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// ----------------------------------------------------------------------------------------
// scatter_argmin (cosypose_cext.cpp:218-245): segmented argmin, first index wins.
// Single-launch, deterministic: one wave per segment scans the (short) id list.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void scatter_argmin_kernel(const float* __restrict__ dists, const int* __restrict__ ids, int M,
                                                            int* __restrict__ out) {
    const int seg = blockIdx.x, lane = threadIdx.x;
    float best = INFINITY; int bi = -1;
    for (int m = lane; m < M; m += 64) {
        if (ids[m] != seg) continue;
        const float d = dists[m];
        if (bi < 0 || d < best) { best = d; bi = m; }   // strict <: first index wins within a lane
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        const bool take = oi >= 0 && (bi < 0 || ob < best || (ob == best && oi < bi));
        if (take) { best = ob; bi = oi; }
    }
    if (lane == 0) out[seg] = bi;
}

int launch_scatter_argmin(const float* dists, const int* ids, int M, int n_seg, int* out, hipStream_t s) {
    if (n_seg == 0) return COSY_OK;
    hipLaunchKernelGGL(scatter_argmin_kernel, dim3(n_seg), dim3(64), 0, s, dists, ids, M, out);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

}  // namespace cosy
