// Training-step kernels of the refiner network (reference: cosypose/models/efficientnet.py:71-98,
// efficientnet_utils.py:37-48 (Swish backward) / :83-92 (drop_connect), training/pose_forward_loss.py:17-84,
// lib3d/cosypose_ops.py:49-82, training/train_pose.py:317-331).
//
// fp32 throughout (the reference trains in fp32), activations NHWC = row-major (rows = B*H*W pixels, C channels).
// The 1x1 convolutions / linear layers and their two gradients are PLAIN GEMMs and go to rocBLAS from the host side;
// everything else of the step is here: batch-statistics BatchNorm (+Swish) forward/backward, depthwise convolution
// forward / data gradient / weight gradient, squeeze-excite scale and its gradients, pooling, stem im2col,
// the disentangled loss gradient, gradient norm + clip + Adam on the flat parameter buffer.
//
// Per-channel reductions over rows are bandwidth-bound: lanes map to channels (coalesced rows), waves and slabs to
// rows; every thread accumulates in double, partial sums go to a workspace and are combined in a fixed order
// (deterministic, no atomics).
#include "cosy_common.h"
#include "kernels_net.h"
#include <algorithm>

namespace cosy {
namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// d/dy of y*sigmoid(y) as the reference writes it (efficientnet_utils.py:44-48)
__device__ __forceinline__ float swish_grad(float y) {
    const float sg = sigmoidf_(y);
    return sg * (1.f + y * (1.f - sg));
}

struct RedGeom { int cgroups, nslab, rows_per_slab; };
RedGeom red_geom(long M, int C) {
    RedGeom g;
    g.cgroups = cdiv(C, 64);
    static const int red_cap = tune_int("COSY_RED_CAP", 1024);
    long cap = red_cap / g.cgroups;            // up to ~1024 workgroups per reduction: more slabs only lengthen the combine pass (4096: 12.9 us per combine, 1024: 6.7 us, first stage unchanged)
    if (cap < 8) cap = 8;
    long nslab = cdiv(M, 128);
    if (nslab > cap) nslab = cap;
    if (nslab < 1) nslab = 1;
    g.rows_per_slab = (int)cdiv(M, nslab);
    g.nslab = cdiv(M, g.rows_per_slab);
    return g;
}

// ---- cross-wave combine of NQ per-thread double accumulators (lane = channel), result written by wave 0
template <int NQ>
__device__ __forceinline__ void slab_write(const double* acc, double* lds /*[4][64][NQ]*/, double* partial, int slab, int C, int c) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) lds[(wave * 64 + lane) * NQ + q] = acc[q];
    __syncthreads();
    if (wave == 0 && c < C) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double v = ((lds[(0 * 64 + lane) * NQ + q] + lds[(1 * 64 + lane) * NQ + q]) + lds[(2 * 64 + lane) * NQ + q]) +
                             lds[(3 * 64 + lane) * NQ + q];
            partial[((size_t)slab * NQ + q) * C + c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm, training mode
// ------------------------------------------------------------------------------------------
// Row-slab reductions of the BatchNorm kernels.  A workgroup covers 64 channels x one slab of rows as 16 channel quads x 16 row
// lanes: every lane issues 16-byte loads (a row's 64 channels = 256 contiguous bytes over 16 lanes), four independent rows in
// flight per lane, double accumulators per channel; the 16 row lanes of a channel are combined through LDS in a fixed order
// (deterministic).  (Round 2's form -- lane = channel, 4-byte loads, one row per iteration -- ran at 2.5-3.1 TB/s.)
template <int NQ>
__device__ __forceinline__ void slab_write_quads(const double (*acc)[NQ] /*[4 channels][NQ]*/, double* lds /*[16][16][4][NQ]*/, double* partial,
                                                 int slab, int C, int cbase) {
    const int tid = threadIdx.x, cq = tid & 15, rs = tid >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < NQ; ++q) lds[((rs * 16 + cq) * 4 + k) * NQ + q] = acc[k][q];
    __syncthreads();
    if (tid < 64 && cbase + tid < C) {
        const int cq2 = tid >> 2, k = tid & 3;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double v = 0.;
            for (int r = 0; r < 16; ++r) v += lds[((r * 16 + cq2) * 4 + k) * NQ + q];
            partial[((size_t)slab * NQ + q) * C + cbase + tid] = v;
        }
    }
}
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long M, int C, int rows_per_slab,
                                                       double* __restrict__ partial) {
    __shared__ double lds[16 * 16 * 4 * 2];
    const int tid = threadIdx.x, cq = tid & 15, rs = tid >> 4;
    const int cbase = blockIdx.x * 64, c0 = cbase + cq * 4, slab = blockIdx.y;
    const long r0 = (long)slab * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    double acc[4][2] = {{0., 0.}, {0., 0.}, {0., 0.}, {0., 0.}};
    if (c0 < C)
        for (long r = r0 + rs; r < r1; r += 64) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = r + 16 * u < r1 ? *(const f32x4*)(x + (r + 16 * u) * C + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double d = v[u][k]; acc[k][0] += d; acc[k][1] += d * d; }
        }
    slab_write_quads<2>(acc, lds, partial, slab, C, cbase);
}

// out = act(bn(x)) [* rowscale[b]] [+ res];  act: 0 none, 1 swish.  Vectorised over 4 channels.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, long n4, int C4, int act,
                                                       const float* __restrict__ rowscale, int HW, const float* __restrict__ res,
                                                       float* __restrict__ out, const float* __restrict__ cgate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % C4);
    const long row = i / C4;
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)x + i), m = ((const f32x4*)mean)[c4], r = ((const f32x4*)rstd)[c4];
    const f32x4 g = ((const f32x4*)gamma)[c4], b = ((const f32x4*)beta)[c4];
    const float rs = rowscale ? rowscale[row / HW] : 1.f;
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float y = (v[k] - m[k]) * r[k] * g[k] + b[k];
        if (act == 1) y = y * sigmoidf_(y);
        o[k] = y * rs;
    }
    if (cgate) {        // squeeze-excite: out = act(bn(x)) * gate[sample][c] -- the unscaled activation is never stored
        const f32x4 gq = ((const f32x4*)cgate)[(size_t)((unsigned)row / (unsigned)HW) * C4 + c4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] *= gq[k];
    }
    if (res) {
        const f32x4 q = ((const f32x4*)res)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += q[k];
    }
    ((f32x4*)out)[i] = o;
}

// gradient wrt the BN output for one element: upstream * rowscale * act'(y)
__device__ __forceinline__ float bn_dy(float dout, float xhat, float g, float b, int act, float rs) {
    float d = dout * rs;
    if (act == 1) d *= swish_grad(xhat * g + b);
    return d;
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, long M,
                                                            int C, int act, const float* __restrict__ rowscale, int HW,
                                                            int rows_per_slab, double* __restrict__ partial, const float* __restrict__ cgate,
                                                            const float* __restrict__ cadd, float cadd_scale) {
    __shared__ double lds[16 * 16 * 4 * 2];
    const int tid = threadIdx.x, cq = tid & 15, rs = tid >> 4;
    const int cbase = blockIdx.x * 64, c0 = cbase + cq * 4, slab = blockIdx.y;
    const long r0 = (long)slab * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    double acc[4][2] = {{0., 0.}, {0., 0.}, {0., 0.}, {0., 0.}};
    if (c0 < C) {
        const f32x4 m = *(const f32x4*)(mean + c0), rs_ = *(const f32x4*)(rstd + c0), g = *(const f32x4*)(gamma + c0), b = *(const f32x4*)(beta + c0);
        for (long r = r0 + rs; r < r1; r += 64) {
            f32x4 xv[4], dv[4];
            float sc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long rr = r + 16 * u;
                const bool ok = rr < r1;
                xv[u] = ok ? *(const f32x4*)(x + rr * C + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
                dv[u] = ok ? *(const f32x4*)(dout + rr * C + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
                sc[u] = !ok ? 0.f : (rowscale ? rowscale[rr / HW] : 1.f);
                if (cgate && ok) {        // incoming gradient = dout * gate[sample][c] + add[sample][c] * scale, formed on the fly (squeeze-excite backward)
                    const size_t bo = (size_t)((unsigned)rr / (unsigned)HW) * C + c0;
                    const f32x4 gq = *(const f32x4*)(cgate + bo), aq = *(const f32x4*)(cadd + bo);
#pragma unroll
                    for (int k = 0; k < 4; ++k) dv[u][k] = dv[u][k] * gq[k] + aq[k] * cadd_scale;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xhat = (xv[u][k] - m[k]) * rs_[k];
                    const float d = bn_dy(dv[u][k], xhat, g[k], b[k], act, sc[u]);
                    acc[k][0] += d; acc[k][1] += (double)d * xhat;
                }
        }
    }
    slab_write_quads<2>(acc, lds, partial, slab, C, cbase);
}
// dx = gamma * rstd * (dy - sum(dy)/M - xhat * sum(dy*xhat)/M); sum_dy / sum_dyx are THIS call's sums (not accumulated grads)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx, long n4,
                                                           int C4, float inv_M, int act, const float* __restrict__ rowscale, int HW,
                                                           float* __restrict__ dx, const float* __restrict__ cgate, const float* __restrict__ cadd,
                                                           float cadd_scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % C4);
    const long row = i / C4;
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)x + i), d0 = __builtin_nontemporal_load((const f32x4*)dout + i), m = ((const f32x4*)mean)[c4], r = ((const f32x4*)rstd)[c4];
    const f32x4 g = ((const f32x4*)gamma)[c4], b = ((const f32x4*)beta)[c4], s1 = ((const f32x4*)sum_dy)[c4], s2 = ((const f32x4*)sum_dyx)[c4];
    const float rs = rowscale ? rowscale[row / HW] : 1.f;
    f32x4 dq = d0;
    if (cgate) {
        const size_t bo = (size_t)((unsigned)row / (unsigned)HW) * C4 + c4;
        const f32x4 gq = ((const f32x4*)cgate)[bo], aq = ((const f32x4*)cadd)[bo];
#pragma unroll
        for (int k = 0; k < 4; ++k) dq[k] = d0[k] * gq[k] + aq[k] * cadd_scale;
    }
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float xhat = (v[k] - m[k]) * r[k];
        const float d = bn_dy(dq[k], xhat, g[k], b[k], act, rs);
        o[k] = g[k] * r[k] * (d - s1[k] * inv_M - xhat * s2[k] * inv_M);
    }
    ((f32x4*)dx)[i] = o;
}

// ------------------------------------------------------------------------------------------
// depthwise convolution (static "same" padding: lo = pad before), weights as (k*k, C)
// ------------------------------------------------------------------------------------------
// Depthwise correlation with a register sliding window (forward at both strides; data gradient at stride 1: 22 of the 26 blocks).  A thread owns 4 channels x one column x
// R consecutive rows: per tap column it walks the R + K - 1 input rows ONCE and feeds up to K accumulators, so it issues
// K * (R + K - 1) 16-byte loads instead of R * K * K (K = 5: 40 instead of 100), with compile-time taps and no per-tap bounds
// arithmetic.  FLIP = the data gradient: dx = dy (*) flipped taps with padding K - 1 - lo (for the symmetric "same" padding of
// stride 1 that is lo again).  Sums run in a fixed order (kx outer, rows inner): deterministic.
template <int K, int S, bool FLIP>
__global__ __launch_bounds__(256) void dw_rows_kernel(const float* __restrict__ x, const float* __restrict__ wt, int H, int W, int Ho, int Wo, int C4,
                                                      long n, float* __restrict__ out, const float* __restrict__ add = nullptr) {
    static_assert(!FLIP || S == 1, "the data-gradient form is stride 1 only");
    constexpr int R = 4, LO = S == 1 ? (K - 1) / 2 : (K - 2) / 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c4 = (int)(i % C4);
    long p = i / C4;
    const int ox = (int)(p % Wo); p /= Wo;
    const int nyq = (Ho + R - 1) / R;
    const int oy0 = (int)(p % nyq) * R;
    const long b = p / nyq;
    const f32x4* xb = (const f32x4*)x + (size_t)b * H * W * C4 + c4;
    f32x4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
        const int ix = ox * S - LO + kx;
        if (ix < 0 || ix >= W) continue;
        f32x4 w[K];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int tap = FLIP ? (K - 1 - ky) * K + (K - 1 - kx) : ky * K + kx;
            w[ky] = ((const f32x4*)wt)[(size_t)tap * C4 + c4];
        }
#pragma unroll
        for (int rr = 0; rr < (R - 1) * S + K; ++rr) {
            const int iy = oy0 * S - LO + rr;
            if (iy < 0 || iy >= H) continue;
            const f32x4 v = xb[((size_t)iy * W + ix) * C4];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int ky = rr - r * S;
                if (ky >= 0 && ky < K) acc[r] += w[ky] * v;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (oy0 + r < Ho) {
            const size_t o = (((size_t)b * Ho + oy0 + r) * Wo + ox) * C4 + c4;
            // add: the skip connection's gradient of a block without expansion (dx = conv-transpose(dy) + dout) rides on the store
            ((f32x4*)out)[o] = add ? acc[r] + ((const f32x4*)add)[o] : acc[r];
        }
}
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ wt, int H, int W, int C4,
                                                          int Ho, int Wo, int k, int s, int lo, long n4, float* __restrict__ dx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % C4);
    long p = i / C4;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const long b = p / H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // only the taps whose parity matches contribute (ty = iy + lo - ky must be a multiple of s): step through those directly
    for (int ky = (iy + lo) % s; ky < k; ky += s) {
        const int ty = iy + lo - ky;
        if (ty < 0) break;
        const int oy = ty / s;
        if (oy >= Ho) continue;
        for (int kx = (ix + lo) % s; kx < k; kx += s) {
            const int tx = ix + lo - kx;
            if (tx < 0) break;
            const int ox = tx / s;
            if (ox >= Wo) continue;
            const f32x4 v = ((const f32x4*)dy)[((b * Ho + oy) * Wo + ox) * C4 + c4];
            const f32x4 w = ((const f32x4*)wt)[(size_t)(ky * k + kx) * C4 + c4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += v[q] * w[q];
        }
    }
    ((f32x4*)dx)[i] = acc;
}
// dw[tap][c] = sum over output pixels of dy[p][c] * x[p_in(tap)][c].  Lanes = channels; a unit of work is a run of 8
// consecutive output pixels of one row: the K input rows under it are loaded once each (sliding window along x) and
// feed all K*K tap accumulators (registers, static indices).  Waves / slabs stride over the units.
template <int K, int S>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy, int H, int W, int C,
                                                            int Ho, int Wo, int lo, long nunits, int units_per_slab,
                                                            double* __restrict__ partial) {
    // A workgroup covers 64 channels x one slab of units (unit = a run of RUN output pixels of one row) as 16 channel quads x 16
    // unit lanes: 16-byte loads (a pixel's 64 channels = 256 contiguous bytes over 16 lanes), K*K x 4 register accumulators per
    // lane, sliding window along x so each input row is loaded once per tap row.  (Round 2: lane = channel, 4-byte loads.)
    // The 16 unit lanes of a quad are combined with two xor-shuffles inside the wave and a fixed-order LDS pass over the 4 waves.
    constexpr int KK = K * K, RUN = 4, XS = (RUN - 1) * S + K;
    __shared__ float lds[4 * 16 * 4 * KK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cq = lane & 15, us = wave * 4 + (lane >> 4);
    const int cbase = blockIdx.x * 64, c0 = cbase + cq * 4, slab = blockIdx.y;
    const long u0 = (long)slab * units_per_slab, u1 = min(nunits, u0 + units_per_slab);
    const int runs = (Wo + RUN - 1) / RUN;
    f32x4 acc[KK];
#pragma unroll
    for (int q = 0; q < KK; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c0 < C)
        for (long u = u0 + us; u < u1; u += 16) {
            const int run = (int)(u % runs);
            const long t = u / runs;
            const int oy = (int)(t % Ho);
            const long b = t / Ho;
            const int ox0 = run * RUN;
            f32x4 d[RUN];
#pragma unroll
            for (int j = 0; j < RUN; ++j) d[j] = ox0 + j < Wo ? *(const f32x4*)(dy + ((b * Ho + oy) * Wo + ox0 + j) * C + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
            const int ixb = ox0 * S - lo;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = oy * S - lo + ky;
                if (iy < 0 || iy >= H) continue;
                const float* xr = x + ((b * H + iy) * W) * C + c0;
                f32x4 xs[XS];
#pragma unroll
                for (int j = 0; j < XS; ++j) {
                    const int ix = ixb + j;
                    xs[j] = (ix >= 0 && ix < W) ? *(const f32x4*)(xr + (long)ix * C) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int j = 0; j < RUN; ++j) acc[ky * K + kx] += d[j] * xs[j * S + kx];
            }
        }
    // lanes l, l^16, l^32, l^48 hold the same channel quad: fold them, then the 4 waves through LDS
#pragma unroll
    for (int q = 0; q < KK; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = acc[q][k];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) lds[((wave * 16 + cq) * 4 + k) * KK + q] = v;
        }
    __syncthreads();
    for (int i = tid; i < 64 * KK; i += 256) {
        const int cl = i / KK, q = i - cl * KK;              // channel within the group, tap
        if (cbase + cl < C) {
            const int idx = ((cl >> 2) * 4 + (cl & 3)) * KK + q;
            const double v = (((double)lds[idx] + (double)lds[16 * 4 * KK + idx]) + (double)lds[2 * 16 * 4 * KK + idx]) + (double)lds[3 * 16 * 4 * KK + idx];
            partial[((size_t)slab * KK + q) * C + cbase + cl] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient of a 1x1 convolution over MANY rows and few channels (the high-resolution blocks):
// dW[n][k] = sum_m dY[m][n] X[m][k], M ~ 10^5..10^6, N,K <= 192.  rocBLAS runs these tall-skinny products at a few
// percent of the HBM rate; here every wave streams its rows once through v_mfma_f32_16x16x4_f32 (A = 4 rows of dY,
// B = the same 4 rows of X) into TN x TK register tiles, workgroups write partial tiles, a combine pass adds them.
// ------------------------------------------------------------------------------------------
template <int TN, int TK>
__global__ __launch_bounds__(256) void wgrad_tall_kernel(const float* __restrict__ dY, const float* __restrict__ X, long M, int N, int K,
                                                         int rows_per_wave, float* __restrict__ partial) {
    __shared__ float lds[4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int n_base = blockIdx.y * TN * 16, k_base = blockIdx.z * TK * 16;
    const long r0 = ((long)blockIdx.x * 4 + wave) * rows_per_wave, r1 = min(M, r0 + rows_per_wave);
    f32x4 acc[TN][TK];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long r = r0; r < r1; r += 8) {     // two 4-row MFMA steps per iteration: 2 x (TN + TK) loads in flight
        float av[2][TN], bv[2][TK];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long row = r + h * 4 + kq;
            const bool ok = row < r1;
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int n = n_base + a * 16 + i;
                av[h][a] = (ok && n < N) ? dY[row * N + n] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < TK; ++b) {
                const int k = k_base + b * 16 + i;
                bv[h][b] = (ok && k < K) ? X[row * K + k] : 0.f;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][a], bv[h][b], acc[a][b], 0, 0, 0);
    }
    // the 4 waves' tiles are added through LDS one tile at a time; wave 0 writes the workgroup's partial
    float* out = partial + (size_t)blockIdx.x * N * K;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) {
            __syncthreads();
            *(f32x4*)(lds + wave * 256 + lane * 4) = acc[a][b];
            __syncthreads();
            if (wave == 0) {
                const f32x4 s0 = *(const f32x4*)(lds + lane * 4), s1 = *(const f32x4*)(lds + 256 + lane * 4);
                const f32x4 s2 = *(const f32x4*)(lds + 512 + lane * 4), s3 = *(const f32x4*)(lds + 768 + lane * 4);
                const int k = k_base + b * 16 + i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n_base + a * 16 + 4 * kq + q;
                    if (n < N && k < K) out[(size_t)n * K + k] = ((s0[q] + s1[q]) + s2[q]) + s3[q];
                }
            }
        }
}

// out[j] = sum over slabs of partial[slab*n + j]; lanes = 64 consecutive j, 16 waves stride over the slabs
template <typename PT>
__global__ __launch_bounds__(1024) void combine_partials_kernel(const PT* __restrict__ partial, int nslab, long n, double* __restrict__ out_d,
                                                                float* __restrict__ out_f, int tC = 0) {
    __shared__ double lds[16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 64 + lane;
    double acc = 0.;
    if (j < n) {
        int k = wave;
        // 16 loads in flight per lane.  (Measured: no faster than 4 in flight, 11.3 us per launch either way -- up to 64 MB of partial tiles per
        // weight gradient make this pass HBM traffic, not latency; fewer slabs (COSY_WG_CAP 128 / 64) cost the first stage more: 32.2 / 33.8 ms per step
        // against 31.6.)
        for (; k + 240 < nslab; k += 256) {
            PT a[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = partial[(size_t)(k + 16 * u) * n + j];
            double t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = ((double)a[4 * q] + (double)a[4 * q + 1]) + ((double)a[4 * q + 2] + (double)a[4 * q + 3]);
            acc += (t[0] + t[1]) + (t[2] + t[3]);
        }
        for (; k + 48 < nslab; k += 64) {   // 4 loads in flight
            const double a0 = partial[(size_t)k * n + j], a1 = partial[(size_t)(k + 16) * n + j];
            const double a2 = partial[(size_t)(k + 32) * n + j], a3 = partial[(size_t)(k + 48) * n + j];
            acc += (a0 + a1) + (a2 + a3);
        }
        for (; k < nslab; k += 16) acc += partial[(size_t)k * n + j];
    }
    lds[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && j < n) {
        double t = 0.;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += lds[w * 64 + lane];
        if (out_d) out_d[j] = t;
        // tC > 0: the n values are a (n / tC, tC) matrix that is written transposed, (tC, n / tC) -- the depthwise weight gradient [tap][channel] straight
        // into the module's (C, 1, k, k) layout
        if (out_f) out_f[tC > 0 ? (j % tC) * (n / tC) + j / tC : j] = (float)t;
    }
}
// BatchNorm: the combine of the per-slab partials and the per-channel finish in ONE launch (they were two: a tiny dependent
// kernel costs ~4 us here, 156 of them per step).  A workgroup owns 64 channels: 16 waves stride the slabs for both sums,
// fixed-order LDS combine, then wave 0 finishes.  partial = [slab][2][C] doubles.
__device__ __forceinline__ void combine2_(const double* __restrict__ partial, int nslab, int C, int c, double* lds /*[16][64][2]*/, double* out2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a0 = 0., a1 = 0.;
    if (c < C) {
        int k = wave;
        for (; k + 240 < nslab; k += 256) {        // 32 loads in flight (measured: 6.9 us per launch as with 8 -- one workgroup pulls up to 655 KB of partials: a CU's bandwidth; COSY_RED_CAP 512 / 256 / 2048: 31.8 / 34.1 / 32.0 ms per step against 31.6)
            double p0[16], p1[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { p0[u] = partial[((size_t)(k + 16 * u) * 2) * C + c]; p1[u] = partial[((size_t)(k + 16 * u) * 2 + 1) * C + c]; }
            double t0[4], t1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                t0[q] = (p0[4 * q] + p0[4 * q + 1]) + (p0[4 * q + 2] + p0[4 * q + 3]);
                t1[q] = (p1[4 * q] + p1[4 * q + 1]) + (p1[4 * q + 2] + p1[4 * q + 3]);
            }
            a0 += (t0[0] + t0[1]) + (t0[2] + t0[3]); a1 += (t1[0] + t1[1]) + (t1[2] + t1[3]);
        }
        for (; k + 48 < nslab; k += 64) {          // 8 loads in flight
            double p0[4], p1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { p0[u] = partial[((size_t)(k + 16 * u) * 2) * C + c]; p1[u] = partial[((size_t)(k + 16 * u) * 2 + 1) * C + c]; }
            a0 += (p0[0] + p0[1]) + (p0[2] + p0[3]); a1 += (p1[0] + p1[1]) + (p1[2] + p1[3]);
        }
        for (; k < nslab; k += 16) { a0 += partial[((size_t)k * 2) * C + c]; a1 += partial[((size_t)k * 2 + 1) * C + c]; }
    }
    lds[(wave * 64 + lane) * 2] = a0; lds[(wave * 64 + lane) * 2 + 1] = a1;
    __syncthreads();
    double t0 = 0., t1 = 0.;
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 16; ++w) { t0 += lds[(w * 64 + lane) * 2]; t1 += lds[(w * 64 + lane) * 2 + 1]; }
    }
    out2[0] = t0; out2[1] = t1;
}
__global__ __launch_bounds__(1024) void bn_stats_combine_final_kernel(const double* __restrict__ partial, int nslab, long M, int C, float eps,
                                                                      float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                                      float* __restrict__ running_mean, float* __restrict__ running_var) {
    __shared__ double lds[16 * 64 * 2];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double t[2];
    combine2_(partial, nslab, C, c, lds, t);
    if ((threadIdx.x >> 6) != 0 || c >= C) return;
    const double m = t[0] / (double)M;
    double var = t[1] / (double)M - m * m;
    if (var < 0.) var = 0.;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {   // nn.BatchNorm2d: running <- (1-mom) running + mom batch; the variance unbiased
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
}
__global__ __launch_bounds__(1024) void bn_bwd_combine_final_kernel(const double* __restrict__ partial, int nslab, int C, float* __restrict__ sums /*[2][C]*/,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    __shared__ double lds[16 * 64 * 2];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double t[2];
    combine2_(partial, nslab, C, c, lds, t);
    if ((threadIdx.x >> 6) != 0 || c >= C) return;
    const float s0 = (float)t[0], s1 = (float)t[1];
    sums[c] = s0; sums[C + c] = s1;        // this call's sums (needed by dx)
    if (accumulate) { dbeta[c] += s0; dgamma[c] += s1; } else { dbeta[c] = s0; dgamma[c] = s1; }
}

// ------------------------------------------------------------------------------------------
// squeeze-excite pieces, pooling, activations
// ------------------------------------------------------------------------------------------
// partial[(b*nchunk + chunk)][c] = sum over the chunk's pixels of f(a[b][p][c] (, a2[b][p][c]));  MODE 0: a;  MODE 1: a * a2.
// One workgroup per (64 channels, sample, chunk of pixels); rows_reduce_final sums the chunks and scales.
// BN: the activation operand (MODE 0: a; MODE 1: a2) is given as the BatchNorm INPUT and swish(bn(.)) is recomputed per element (same
// arithmetic as bn_apply_kernel), so that the activated tensor need not exist in memory
struct BnParams { const float *mean, *rstd, *gamma, *beta; };
template <int MODE, bool BN = false>
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ a, const float* __restrict__ a2, int HW, int C,
                                                          int rows_per_chunk, double* __restrict__ partial, BnParams bn = BnParams{}) {
    // 16 channel quads x 16 row lanes per workgroup, 16-byte loads, four independent rows in flight per lane (see bn_stats_kernel)
    __shared__ double lds[16 * 16 * 4];
    const int tid = threadIdx.x, cq = tid & 15, rs = tid >> 4;
    const int cbase = blockIdx.x * 64, c0 = cbase + cq * 4, b = blockIdx.y, chunk = blockIdx.z, nchunk = gridDim.z;
    const int p0 = chunk * rows_per_chunk, p1 = min(HW, p0 + rows_per_chunk);
    double acc[4] = {0., 0., 0., 0.};
    f32x4 bm = {0.f, 0.f, 0.f, 0.f}, br = bm, bg = bm, bb = bm;
    if constexpr (BN) {
        if (c0 < C) { bm = *(const f32x4*)(bn.mean + c0); br = *(const f32x4*)(bn.rstd + c0); bg = *(const f32x4*)(bn.gamma + c0); bb = *(const f32x4*)(bn.beta + c0); }
    }
    auto actv = [&](float x, int k) { float y = (x - bm[k]) * br[k] * bg[k] + bb[k]; return y * sigmoidf_(y); };
    if (c0 < C)
        for (int p = p0 + rs; p < p1; p += 64) {
            f32x4 v[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = p + 16 * u < p1;
                const size_t o = ((size_t)b * HW + p + 16 * u) * C + c0;
                v[u] = ok ? *(const f32x4*)(a + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (MODE != 0) w[u] = ok ? *(const f32x4*)(a2 + o) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (BN) {
                        if (p + 16 * u < p1) acc[k] += MODE == 0 ? (double)actv(v[u][k], k) : (double)v[u][k] * actv(w[u][k], k);
                    } else {
                        acc[k] += MODE == 0 ? (double)v[u][k] : (double)v[u][k] * w[u][k];
                    }
                }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) lds[(rs * 16 + cq) * 4 + k] = acc[k];
    __syncthreads();
    if (tid < 64 && cbase + tid < C) {
        double v = 0.;
        for (int r = 0; r < 16; ++r) v += lds[(r * 16 + (tid >> 2)) * 4 + (tid & 3)];
        partial[((size_t)b * nchunk + chunk) * C + cbase + tid] = v;
    }
}
__global__ __launch_bounds__(256) void rows_reduce_final_kernel(const double* __restrict__ partial, int nchunk, int C, float scale,
                                                                float* __restrict__ out) {
    __shared__ double lds[4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, b = blockIdx.y;
    double acc = 0.;
    if (c < C)
        for (int k = wave; k < nchunk; k += 4) acc += partial[((size_t)b * nchunk + k) * C + c];
    lds[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && c < C)
        out[(size_t)b * C + c] = (float)((((lds[lane] + lds[64 + lane]) + lds[128 + lane]) + lds[192 + lane]) * (double)scale);
}
// out = a * g[b][c] (+ add[b][c] * add_scale)
__global__ __launch_bounds__(256) void rows_scale_kernel(const float* __restrict__ a, const float* __restrict__ g, const float* __restrict__ add,
                                                         float add_scale, long n4, int C4, int HW, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % C4);
    const long b = i / C4 / HW;
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)a + i), gg = ((const f32x4*)g)[b * C4 + c4];   // a streams through once
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = v[k] * gg[k];
    if (add) {
        const f32x4 ad = ((const f32x4*)add)[b * C4 + c4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += ad[k] * add_scale;
    }
    ((f32x4*)out)[i] = o;
}
// out[b][p][c] = v[b][c] * scale   (gradient of the mean over pixels)
__global__ __launch_bounds__(256) void rows_broadcast_kernel(const float* __restrict__ v, float scale, long n4, int C4, int HW,
                                                             float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % C4);
    const long b = i / C4 / HW;
    f32x4 o = ((const f32x4*)v)[b * C4 + c4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] *= scale;
    ((f32x4*)out)[i] = o;
}
// kind 0 swish, 1 sigmoid; bwd: dx = dy * f'(x)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, long n, int kind, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i], sg = sigmoidf_(v);
    out[i] = kind == 0 ? v * sg : sg;
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, long n, int kind,
                                                      float* __restrict__ dx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    if (kind == 0) dx[i] = dy[i] * swish_grad(v);
    else { const float sg = sigmoidf_(v); dx[i] = dy[i] * (sg * (1.f - sg)); }
}

// ------------------------------------------------------------------------------------------
// Squeeze-excite FCs and the pose head's Linear layer, forward and backward, as FUSED kernels (round 4).  These are products with
// 64 rows (the batch): on rocBLAS + elementwise launches they were ~14 launches per MBConv block (addmm, swish, addmm, sigmoid;
// sigmoid', column sums, three matmuls, swish', column sum, two matmuls) -- 1.3 ms of rocBLAS and ~1.5 ms of tiny launches per step.
// Weight layouts are the module's own (se_reduce (Cse, C), se_expand (C, Cse), Linear (J, C)): the weights change every step, so
// nothing is re-packed.  Every sum runs in a fixed order (deterministic).
//   se_fc1 / se_fc2 (batched MFMA GEMMs):  h_pre = W1 pooled + b1;  g = sigmoid(W2 swish(h_pre) + b2)
//   se_bwd_x (one workgroup per sample):  dg_pre = dg g (1 - g);  dh_pre = (W2^T dg_pre) swish'(h_pre);  dpooled = W1^T dh_pre
//   se_bwd_w (one wave per 16 channels, MFMA over the batch): dW2 = dg_pre^T swish(h_pre), db2, dW1 = dh_pre^T pooled, db1
// ------------------------------------------------------------------------------------------
// y[j] = sum_c W[j][c] x[c] for j < J (W (J, C) row-major, C % 4 == 0, x in LDS): 8 lanes per output, float4 loads, shuffle combine;
// calls f(j, y) on one lane per output.  All `nthreads` threads of the workgroup must call it.
template <typename F>
__device__ __forceinline__ void fc_rows_reduce(const float* __restrict__ W, const float* xs, int J, int C, int tid, int nthreads, F&& f) {
    const int C4 = C >> 2, slots = nthreads >> 3, prt = tid & 7;
    for (int j0 = 0; j0 < J; j0 += slots) {
        const int j = j0 + (tid >> 3);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const f32x4* wr = (const f32x4*)(W + (size_t)min(j, J - 1) * C);
        for (int c0 = prt; c0 < C4; c0 += 64) {           // 8 independent 16-byte loads in flight, then the FMAs
            f32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = wr[min(c0 + 8 * u, C4 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + 8 * u < C4) acc += w[u] * ((const f32x4*)xs)[c0 + 8 * u];
        }
        float sv = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        sv += __shfl_xor(sv, 1, 64); sv += __shfl_xor(sv, 2, 64); sv += __shfl_xor(sv, 4, 64);
        if (prt == 0 && j < J) f(j, sv);
    }
}
// y[c] = sum_j W[j][c] x[j] (W (J, C) row-major, x in LDS): a thread owns 4 consecutive channels; calls f(c4, y4)
template <typename F>
__device__ __forceinline__ void fc_cols_apply(const float* __restrict__ W, const float* xs, int J, int C, int tid, int nthreads, F&& f) {
    const int C4 = C >> 2;
    for (int c = tid; c < C4; c += nthreads) {
        const f32x4* wc = (const f32x4*)W + c;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        for (int j0 = 0; j0 < J; j0 += 8) {
            f32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = wc[(size_t)min(j0 + u, J - 1) * C4];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                if (j0 + u < J) s0 += w[u] * xs[j0 + u];
                if (j0 + u + 1 < J) s1 += w[u + 1] * xs[j0 + u + 1];
            }
        }
        f(c, s0 + s1);
    }
}
// Forward as two small GEMMs over the BATCH on the fp32 matrix instruction (v_mfma_f32_16x16x4_f32: an exact fp32 FMA chain per output,
// columns = samples are independent of each other): a 16-sample tile shares one read of the weights -- one workgroup per sample re-reads
// both FC matrices (1.77 MB for block 25) per sample at the ~100 GB/s a single CU pulls: 40 us per block.
//   fc1: grid (sample tiles, 16-row tiles of Cse); the WAVES waves split the C range, fixed-order LDS combine;  h_pre = W1 pooled + b1
//   fc2: grid (sample tiles, groups of WAVES 16-channel tiles); k = Cse;  gate = sigmoid(W2 swish(h_pre) + b2)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void se_train_fc1_kernel(const float* __restrict__ pooled, const float* __restrict__ W1, const float* __restrict__ b1,
                                                                  int B, int C, int Cse, float* __restrict__ h_pre) {
    __shared__ f32x4 comb[WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    const int b = min((int)blockIdx.x * 16 + i, B - 1), j0 = blockIdx.y * 16;
    const float* prow = pooled + (size_t)b * C + kq * 4;
    const float* wrow = W1 + (size_t)min(j0 + i, Cse - 1) * C + kq * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nsteps = C >> 4, tail = C & 15;          // C % 4 == 0: a tail step has 1-3 valid k-quads
    for (int st0 = wave; st0 < nsteps + (tail ? 1 : 0); st0 += 4 * WAVES) {
        f32x4 a[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int st = st0 + u * WAVES;
            a[u] = f32x4{0.f, 0.f, 0.f, 0.f}; w[u] = a[u];
            if (st < nsteps || (st == nsteps && kq * 4 < tail)) { w[u] = *(const f32x4*)(wrow + st * 16); a[u] = *(const f32x4*)(prow + st * 16); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][e], a[u][e], acc, 0, 0, 0);
    }
    comb[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        f32x4 sv = comb[0][lane];
#pragma unroll
        for (int q = 1; q < WAVES; ++q) sv += comb[q][lane];
        if ((int)blockIdx.x * 16 + i < B) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + kq * 4 + e;
                if (j < Cse) h_pre[(size_t)b * Cse + j] = sv[e] + b1[j];
            }
        }
    }
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void se_train_fc2_kernel(const float* __restrict__ h_pre, const float* __restrict__ W2, const float* __restrict__ b2,
                                                                  int B, int C, int Cse, float* __restrict__ gate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    const int ct = blockIdx.y * WAVES + wave;
    if (ct * 16 >= C) return;
    const int b = min((int)blockIdx.x * 16 + i, B - 1), c0 = ct * 16;
    const float* hrow = h_pre + (size_t)b * Cse;
    const float* wrow = W2 + (size_t)min(c0 + i, C - 1) * Cse;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < Cse; j0 += 32) {             // 8 k-steps of 4: 16 independent loads in flight
        float hv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * 4 + kq;
            const bool ok = j < Cse;
            const float hp = ok ? hrow[j] : 0.f;
            hv[u] = ok ? hp * sigmoidf_(hp) : 0.f;
            wv[u] = ok ? wrow[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u], hv[u], acc, 0, 0, 0);
    }
    if ((int)blockIdx.x * 16 + i < B && c0 + kq * 4 < C) {
        const f32x4 bias = *(const f32x4*)(b2 + c0 + kq * 4);
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = sigmoidf_(acc[e] + bias[e]);
        *(f32x4*)(gate + (size_t)b * C + c0 + kq * 4) = g;
    }
}
__global__ __launch_bounds__(512) void se_train_bwd_x_kernel(const float* __restrict__ dg, const float* __restrict__ gate, const float* __restrict__ h_pre,
                                                             const float* __restrict__ W1, const float* __restrict__ W2, int C, int Cse,
                                                             float* __restrict__ dg_pre, float* __restrict__ dh_pre, float* __restrict__ dpooled) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ds = sm;                    // C: dg_pre of this sample
    float* dhs = sm + C;               // Cse (padded to 128): dh_pre
    float* red = dhs + 128;            // 32 partial rows x 128
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < (C >> 2); c += 512) {
        const f32x4 d = ((const f32x4*)(dg + (size_t)b * C))[c], g = ((const f32x4*)(gate + (size_t)b * C))[c];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = d[e] * (g[e] * (1.f - g[e]));
        ((f32x4*)ds)[c] = o;
        ((f32x4*)(dg_pre + (size_t)b * C))[c] = o;
    }
    __syncthreads();
    // dh[j] = sum_c dg_pre[c] W2[c][j]: lanes along j (rows of W2 are contiguous), 32 (wave, quarter) groups split the channels
    {
        const int jl = tid & 15, grp = tid >> 4;          // 32 groups
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        for (int c0 = grp; c0 < C; c0 += 4 * 32) {          // 4 rows of W2 per step: their loads are independent of each other
            float w[4][8], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 32 * u;
                d[u] = c < C ? ds[c] : 0.f;
                const float* wr = W2 + (size_t)min(c, C - 1) * Cse;
#pragma unroll
                for (int q = 0; q < 8; ++q) w[u][q] = 16 * q < Cse ? wr[min(jl + 16 * q, Cse - 1)] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += d[u] * w[u][q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) red[grp * 128 + jl + 16 * q] = acc[q];
    }
    __syncthreads();
    if (tid < Cse) {
        float v = 0.f;
        for (int r = 0; r < 32; ++r) v += red[r * 128 + tid];
        v *= swish_grad(h_pre[(size_t)b * Cse + tid]);
        dhs[tid] = v;
        dh_pre[(size_t)b * Cse + tid] = v;
    }
    __syncthreads();
    fc_cols_apply(W1, dhs, Cse, C, tid, 512, [&](int c4, f32x4 v) { ((f32x4*)(dpooled + (size_t)b * C))[c4] = v; });
}
// weight gradients on the fp32 matrix instruction, k = the batch: a wave owns 16 channels and walks the samples four at a time;
//   dW2[c][j] = sum_b dg_pre[b][c] swish(h_pre[b][j])   (A = dg_pre^T, B = h),   db2[c] = sum_b dg_pre[b][c]   (B = ones)
//   dW1[j][c] = sum_b dh_pre[b][j] pooled[b][c]          (A = dh_pre^T, B = pooled),   db1[j] (the wave that owns channels 0-15)
// Each output is one fp32 FMA chain over the samples in order (deterministic).  Cse <= 128 = 8 column tiles of 16.
__global__ __launch_bounds__(256) void se_train_bwd_w_kernel(const float* __restrict__ dg_pre, const float* __restrict__ dh_pre, const float* __restrict__ h_pre,
                                                             const float* __restrict__ pooled, int B, int C, int Cse, float* __restrict__ dW2,
                                                             float* __restrict__ db2, float* __restrict__ dW1, float* __restrict__ db1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    const int c0 = (blockIdx.x * 4 + wave) * 16;
    if (c0 >= C) return;
    const int njt = (Cse + 15) >> 4;                   // <= 8
    f32x4 a2[8], a1[8], sb2 = {0.f, 0.f, 0.f, 0.f}, sb1[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { a2[t] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[t] = a2[t]; sb1[t] = a2[t]; }
    const int cc = min(c0 + i, C - 1);
    const bool first = c0 == 0;
    // UB = 2 MFMA k-steps (8 samples) per iteration with all their loads in flight first: the loop is a chain of dependent memory latencies
    constexpr int UB = 2;
    for (int bb = 0; bb < B; bb += 4 * UB) {
        float d[UB], pv[UB], hv[UB][8], dv[UB][8];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int b = bb + 4 * u + kq;
            const bool okb = b < B;
            d[u] = okb ? dg_pre[(size_t)b * C + cc] : 0.f;
            pv[u] = okb ? pooled[(size_t)b * C + cc] : 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = t * 16 + i;
                const bool ok = okb && t < njt && j < Cse;
                hv[u][t] = ok ? h_pre[(size_t)b * Cse + j] : 0.f;
                dv[u][t] = ok ? dh_pre[(size_t)b * Cse + j] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            sb2 = __builtin_amdgcn_mfma_f32_16x16x4f32(d[u], 1.f, sb2, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < njt) {                                 // wave-uniform
                    const float hp = hv[u][t], h = hp * sigmoidf_(hp);       // swish(0) = 0: masked elements stay 0
                    a2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[u], h, a2[t], 0, 0, 0);              // rows = channels, columns = j
                    a1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][t], pv[u], a1[t], 0, 0, 0);      // rows = j, columns = channels
                    if (first) sb1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][t], 1.f, sb1[t], 0, 0, 0);
                }
        }
    }
    // accumulator element q of lane (i, kq) = output[row 4 kq + q][column i]
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < njt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + 4 * kq + q, j = t * 16 + i;                 // dW2 tile: row = channel, column = j
                if (c < C && j < Cse) dW2[(size_t)c * Cse + j] = a2[t][q];
                const int j1 = t * 16 + 4 * kq + q, c1 = c0 + i;               // dW1 tile: row = j, column = channel
                if (j1 < Cse && c1 < C) dW1[(size_t)j1 * C + c1] = a1[t][q];
                if (first && i == 0 && j1 < Cse) db1[j1] = sb1[t][q];
            }
        }
    if (i == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (c0 + 4 * kq + q < C) db2[c0 + 4 * kq + q] = sb2[q];
    }
}
// Linear layer with few outputs (the pose head: J = 9): y = x W^T + b;  dx = dy W;  dW = dy^T x, db = column sums of dy
__global__ __launch_bounds__(256) void fc_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                           int C, int J, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < (C >> 2); c += 256) ((f32x4*)sm)[c] = ((const f32x4*)(x + (size_t)b * C))[c];
    __syncthreads();
    fc_rows_reduce(W, sm, J, C, tid, 256, [&](int j, float v) { y[(size_t)b * J + j] = v + bias[j]; });
}
__global__ __launch_bounds__(256) void fc_small_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ W, int C, int J, float* __restrict__ dx) {
    __shared__ float dys[128];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < J) dys[tid] = dy[(size_t)b * J + tid];
    __syncthreads();
    fc_cols_apply(W, dys, J, C, tid, 256, [&](int c4, f32x4 v) { ((f32x4*)(dx + (size_t)b * C))[c4] = v; });
}
__global__ __launch_bounds__(256) void fc_small_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B, int C, int J,
                                                             float* __restrict__ dW, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // dy: B x J
    const int tid = threadIdx.x, c = blockIdx.x * 256 + tid;
    for (int i = tid; i < B * J; i += 256) sm[i] = dy[i];
    __syncthreads();
    constexpr int NJ = 16;
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
    if (c < C)
        for (int r = 0; r < B; ++r) {
            const float xv = x[(size_t)r * C + c];
#pragma unroll
            for (int j = 0; j < NJ; ++j) if (j < J) acc[j] += sm[r * J + j] * xv;
        }
    if (c < C) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (j < J) dW[(size_t)j * C + c] = acc[j];
    }
    if (blockIdx.x == 0 && tid < J) {
        float sv = 0.f;
        for (int r = 0; r < B; ++r) sv += sm[r * J + tid];
        db[tid] = sv;
    }
}

// stem: 3x3 stride-2 patches of the 8-channel NHWC input (6 used) as GEMM rows: cols[p][(ky*3+kx)*6 + c]
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x8, int H, int W, int Ho, int Wo, int lo, long n,
                                                          int ld, float* __restrict__ cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per (pixel, tap)
    if (i >= n) return;
    const int tap = (int)(i % 9);
    long p = i / 9;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const long b = p / Ho;
    const int iy = oy * 2 - lo + tap / 3, ix = ox * 2 - lo + tap % 3;
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const float* s = x8 + ((b * H + iy) * W + ix) * 8;
        const f32x4 a = *(const f32x4*)s;
        const f32x2 c = *(const f32x2*)(s + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = c[0]; v[5] = c[1];
    }
    float* o = cols + (i / 9) * ld + tap * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = v[k];
    if (tap == 8)
        for (int k = 54; k < ld; ++k) cols[(i / 9) * ld + k] = 0.f;      // padding columns of a 16-byte aligned row
}

// ------------------------------------------------------------------------------------------
// gradient of loss_refiner_CO_disentangled wrt the network outputs (B,9)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
}
__device__ __forceinline__ void xform3(const float* T, float x, float y, float z, float* q) {
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = T[i * 4 + 0] * x + T[i * 4 + 1] * y + T[i * 4 + 2] * z + T[i * 4 + 3];
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ void cross3(const float* u, const float* v, float* c) {
    c[0] = u[1] * v[2] - u[2] * v[1]; c[1] = u[2] * v[0] - u[0] * v[2]; c[2] = u[0] * v[1] - u[1] * v[0];
}

__global__ __launch_bounds__(256) void loss_disentangled_bwd_kernel(const float* __restrict__ gt, const float* __restrict__ TCO_in,
                                                                    const float* __restrict__ out9, const float* __restrict__ K_crop,
                                                                    const float* __restrict__ pts, const int* __restrict__ obj, int S,
                                                                    int P, const float* __restrict__ dloss, float* __restrict__ dout9) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = pts + (size_t)(obj ? obj[b] : b) * P * 3;
    const float* g0 = gt + (size_t)b * S * 16;
    float Ti[16], o[9], G[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { Ti[i] = TCO_in[(size_t)b * 16 + i]; G[i] = g0[i]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = out9[(size_t)b * 9 + i];
    const float fx = K_crop[(size_t)b * 9], fy = K_crop[(size_t)b * 9 + 4];
    // forward pieces of ortho6d
    const float a[3] = {o[0], o[1], o[2]}, bb[3] = {o[3], o[4], o[5]};
    const float na = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const float x[3] = {a[0] / na, a[1] / na, a[2] / na};
    float w[3]; cross3(x, bb, w);
    const float nw = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const float z[3] = {w[0] / nw, w[1] / nw, w[2] / nw};
    float y[3]; cross3(z, x, y);
    const float dR[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]};

    float grad[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) grad[i] = 0.f;
    float GdR[9];   // dL/d dR
#pragma unroll
    for (int i = 0; i < 9; ++i) GdR[i] = 0.f;
    const float inv = 1.f / (float)(3 * P);
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {
        float pr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pr[i] = G[i];
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) pr[i * 4 + j] = dR[i * 3 + 0] * Ti[0 * 4 + j] + dR[i * 3 + 1] * Ti[1 * 4 + j] + dR[i * 3 + 2] * Ti[2 * 4 + j];
        } else if (t == 1) {
            pr[3] = (o[6] / fx + Ti[3] / Ti[11]) * G[11];
            pr[7] = (o[7] / fy + Ti[7] / Ti[11]) * G[11];
        } else {
            pr[11] = o[8] * Ti[11];
        }
        // the assigned ground truth of this term: first minimum of the mean L1 (as the forward does)
        int arg = 0; float best = 0.f;
        for (int s = 0; s < S; ++s) {
            float gsm[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) gsm[i] = g0[(size_t)s * 16 + i];
            float acc = 0.f;
            for (int i = tid; i < P; i += 256) {
                float q1[3], q2[3];
                xform3(pr, p[i * 3], p[i * 3 + 1], p[i * 3 + 2], q1);
                xform3(gsm, p[i * 3], p[i * 3 + 1], p[i * 3 + 2], q2);
                acc += fabsf(q1[0] - q2[0]) + fabsf(q1[1] - q2[1]) + fabsf(q1[2] - q2[2]);
            }
            const float l = block_sum256(acc, red);
            if (s == 0 || l < best) { best = l; arg = s; }
        }
        float gsm[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gsm[i] = g0[(size_t)arg * 16 + i];
        // sums over the points of sign(pred - gt) (x point coordinates for the rotation term)
        float s9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s3[3] = {0.f, 0.f, 0.f};
        for (int i = tid; i < P; i += 256) {
            const float px = p[i * 3], py = p[i * 3 + 1], pz = p[i * 3 + 2];
            float q1[3], q2[3];
            xform3(pr, px, py, pz, q1);
            xform3(gsm, px, py, pz, q2);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float sg = sgn(q1[r] - q2[r]);
                s3[r] += sg;
                if (t == 0) { s9[r * 3 + 0] += sg * px; s9[r * 3 + 1] += sg * py; s9[r * 3 + 2] += sg * pz; }
            }
        }
        if (t == 0) {
            float GRp[9];   // dL/d(dR Rin)
#pragma unroll
            for (int q = 0; q < 9; ++q) GRp[q] = block_sum256(s9[q], red) * inv;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) GdR[i * 3 + k] = GRp[i * 3 + 0] * Ti[k * 4 + 0] + GRp[i * 3 + 1] * Ti[k * 4 + 1] + GRp[i * 3 + 2] * Ti[k * 4 + 2];
        } else if (t == 1) {
            const float gx = block_sum256(s3[0], red) * inv, gy = block_sum256(s3[1], red) * inv;
            grad[6] = gx * G[11] / fx;
            grad[7] = gy * G[11] / fy;
        } else {
            grad[8] = block_sum256(s3[2], red) * inv * Ti[11];
        }
    }
    // ortho6d backward: R = [x y z] columns; x = a/|a|, w = x x b, z = w/|w|, y = z x x
    const float gx[3] = {GdR[0], GdR[3], GdR[6]}, gy[3] = {GdR[1], GdR[4], GdR[7]}, gz[3] = {GdR[2], GdR[5], GdR[8]};
    float t1[3], t2[3];
    cross3(x, gy, t1);                       // dz from y = z x x
    cross3(gy, z, t2);                       // dx from y = z x x
    const float gz_t[3] = {gz[0] + t1[0], gz[1] + t1[1], gz[2] + t1[2]};
    const float zd = z[0] * gz_t[0] + z[1] * gz_t[1] + z[2] * gz_t[2];
    const float dw[3] = {(gz_t[0] - z[0] * zd) / nw, (gz_t[1] - z[1] * zd) / nw, (gz_t[2] - z[2] * zd) / nw};
    float t3[3], db[3];
    cross3(bb, dw, t3);                      // dx from w = x x b
    cross3(dw, x, db);                       // db
    const float gx_t[3] = {gx[0] + t2[0] + t3[0], gx[1] + t2[1] + t3[1], gx[2] + t2[2] + t3[2]};
    const float xd = x[0] * gx_t[0] + x[1] * gx_t[1] + x[2] * gx_t[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { grad[i] = (gx_t[i] - x[i] * xd) / na; grad[3 + i] = db[i]; }
    if (tid == 0) {
        const float up = dloss[b];
#pragma unroll
        for (int i = 0; i < 9; ++i) dout9[(size_t)b * 9 + i] = grad[i] * up;
    }
}

// ------------------------------------------------------------------------------------------
// optimizer on the flat buffers: squared gradient norm, clip + Adam
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ partial) {
    __shared__ double lds[256];
    double acc = 0.;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += (double)g[i] * g[i];
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) lds[threadIdx.x] += lds[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = lds[0];
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const double* __restrict__ partial, int nparts, float max_norm, float* __restrict__ out /*[norm, clip_coef]*/) {
    __shared__ double lds[256];
    double acc = 0.;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) lds[threadIdx.x] += lds[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(lds[0]);
        out[0] = norm;
        const float coef = max_norm / (norm + 1e-6f);     // torch.nn.utils.clip_grad_norm_
        out[1] = max_norm > 0.f ? (coef < 1.f ? coef : 1.f) : 1.f;
    }
}
// torch.optim.Adam (amsgrad off): g <- clip*g (+ wd*p); m, v moments; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2_sqrt, const float* __restrict__ clip /*[norm, coef]*/) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * (clip ? clip[1] : 1.f);
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}


// ------------------------------------------------------------------------------------------
// 1x1 convolutions of the training step on this library's own fp32 MFMA GEMM (pw_gemm, v_mfma_f32_16x16x4_f32): the
// weights change every step, so they are re-packed into the kernel's fragment order on the device (a few microseconds:
// the largest matrix is 2304 x 384), together with the identity epilogue (scale 1, bias 0) the inference kernel expects.
// ------------------------------------------------------------------------------------------
// element idx of the packed form of one weight: lane-major fragment blocks of 4 k-values (see pw_pack_weights, the host-side packer of the inference path)
__device__ __forceinline__ float pw_pack_f32_elem(const float* __restrict__ w, int K, int N, int w_is_kn, int NI, int WN, int nkb, long idx) {
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    long r = idx >> 8;
    const int kbi = (int)(r % nkb); r /= nkb;
    const int NW = NI * WN, nb = (int)(r % NW), nt = (int)(r / NW);
    const int i = lane & 15, kg = lane >> 4, wn = nb / NI, ni = nb % NI;
    const int n = nt * 16 * NW + wn * 16 * NI + (i >> 2) * 4 * NI + ni * 4 + (i & 3);
    const int k = kbi * 16 + kg * 4 + e;
    return (n < N && k < K) ? (w_is_kn ? w[(size_t)k * N + n] : w[(size_t)n * K + k]) : 0.f;
}
__global__ __launch_bounds__(256) void pw_pack_f32_kernel(const float* __restrict__ w, int K, int N, int w_is_kn, int NI, int WN, int nkb,
                                                          long total, int n_pad, float* __restrict__ dst, float* __restrict__ ones,
                                                          float* __restrict__ zeros) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n_pad + 4) {
        if (idx < n_pad) ones[idx] = 1.f;
        zeros[idx] = 0.f;                     // n_pad zeros for the bias + 16 zero bytes for the DMA's padding source
    }
    if (idx >= total) return;
    dst[idx] = pw_pack_f32_elem(w, K, N, w_is_kn, NI, WN, nkb, idx);
}
// Every 1x1-convolution weight of the network (both orientations: forward and data gradient) in ONE launch at the top of a step instead of a
// ~4.6 us launch in front of each of the ~100 GEMMs.  plan[e] = {source pointer, destination offset in the pool (floats), K, N,
// w_is_kn | NI << 8 | WN << 16, k-blocks, packed elements, first workgroup}; the pool starts with the identity epilogue the GEMM wants
// (PACK_HEAD_N ones, then PACK_HEAD_N + 16 zeros: scale, bias and the DMA's padding source).
constexpr int PACK_HEAD_N = 4096;
constexpr int PACK_HEAD_FLOATS = 2 * PACK_HEAD_N + 16;
constexpr int PACK_HEAD_BLOCKS = (PACK_HEAD_FLOATS + 255) / 256;
__global__ __launch_bounds__(256) void pw_pack_all_kernel(const long long* __restrict__ plan, int n, float* __restrict__ pool) {
    const long blk = blockIdx.x;
    if (blk < PACK_HEAD_BLOCKS) {
        const int idx = (int)blk * 256 + threadIdx.x;
        if (idx < PACK_HEAD_FLOATS) pool[idx] = idx < PACK_HEAD_N ? 1.f : 0.f;
        return;
    }
    int lo = 0, hi = n - 1;
    while (lo < hi) {                         // the last entry whose first workgroup is <= blk (uniform over the workgroup)
        const int mid = (lo + hi + 1) >> 1;
        if (plan[(size_t)mid * 8 + 7] <= blk) lo = mid; else hi = mid - 1;
    }
    const long long* e = plan + (size_t)lo * 8;
    const long idx = (blk - e[7]) * 256 + threadIdx.x;
    if (idx >= e[6]) return;
    const int flags = (int)e[4];
    pool[e[1] + idx] = pw_pack_f32_elem((const float*)e[0], (int)e[2], (int)e[3], flags & 1, (flags >> 8) & 255, (flags >> 16) & 255, (int)e[5], idx);
}
}  // namespace
}  // namespace cosy

using namespace cosy;

#define LAUNCH1D(kernel, n, s, ...)                                                            \
    do {                                                                                       \
        hipLaunchKernelGGL(kernel, dim3((unsigned)cdiv((n), 256)), dim3(256), 0, s, __VA_ARGS__); \
        COSY_CHECK_HIP(hipGetLastError());                                                     \
    } while (0)

extern "C" {

size_t cosy_train_workspace_bytes(void) { return (size_t)64 << 20; }   // >= (4096 + 64) slabs*groups x 64 channels x 25 taps doubles

int cosy_crop_pack_to(void* x_nhwc8, int dtype, const float* frames_nhwc4, const int* im_id, const float* boxes_crop,
                      const float* renders, int B, int N, int h, int w, int H, int W, cosy_stream_t stream) {
    COSY_REQUIRE(x_nhwc8 && frames_nhwc4 && boxes_crop && renders, "crop_pack_to: null argument");
    COSY_REQUIRE(dtype == COSY_F32 || dtype == COSY_BF16 || dtype == COSY_F16, "crop_pack_to: dtype %d", dtype);
    return launch_crop_pack(x_nhwc8, dtype, frames_nhwc4, im_id, boxes_crop, renders, B, N, h, w, H, W, nullptr, (hipStream_t)stream);
}
size_t cosy_crop_pack_workspace_bytes(int B, int H, int W) { return crop_taps_bytes(B > 0 ? B : 0, H, W); }
int cosy_crop_pack_to_ws(void* x_nhwc8, int dtype, const float* frames_nhwc4, const int* im_id, const float* boxes_crop,
                         const float* renders, int B, int N, int h, int w, int H, int W, void* workspace, cosy_stream_t stream) {
    COSY_REQUIRE(x_nhwc8 && frames_nhwc4 && boxes_crop && renders && workspace, "crop_pack_to_ws: null argument");
    COSY_REQUIRE(dtype == COSY_F32 || dtype == COSY_BF16 || dtype == COSY_F16, "crop_pack_to_ws: dtype %d", dtype);
    return launch_crop_pack(x_nhwc8, dtype, frames_nhwc4, im_id, boxes_crop, renders, B, N, h, w, H, W, workspace, (hipStream_t)stream);
}

int cosy_bn_train_stats(const float* x, long M, int C, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                        float* running_var, void* workspace, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(x && mean && rstd && workspace && M > 0 && C > 0, "bn_train_stats: bad argument");
    const RedGeom g = red_geom(M, C);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(g.cgroups, g.nslab), dim3(256), 0, s, x, M, C, g.rows_per_slab, (double*)workspace);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(bn_stats_combine_final_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, s, (const double*)workspace, g.nslab, M, C, eps, momentum,
                       mean, rstd, running_mean, running_var);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

int cosy_bn_train_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, long M, int C,
                        int act, const float* rowscale, int HW, const float* res, float* out, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(x && mean && rstd && gamma && beta && out && C % 4 == 0 && HW > 0, "bn_train_apply: bad argument (C=%d)", C);
    if (M == 0) return COSY_OK;
    LAUNCH1D(bn_apply_kernel, M * (C / 4), s, x, mean, rstd, gamma, beta, M * (C / 4), C / 4, act, rowscale, HW, res, out, (const float*)nullptr);
    return COSY_OK;
}
int cosy_bn_train_apply_gated(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, long M, int C,
                              int act, const float* cgate, int HW, float* out, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(x && mean && rstd && gamma && beta && out && cgate && C % 4 == 0 && HW > 0 && M % HW == 0 && M < (1l << 31),
                 "bn_train_apply_gated: bad argument (C=%d, M=%ld, HW=%d)", C, M, HW);
    if (M == 0) return COSY_OK;
    LAUNCH1D(bn_apply_kernel, M * (C / 4), s, x, mean, rstd, gamma, beta, M * (C / 4), C / 4, act, (const float*)nullptr, HW, (const float*)nullptr, out,
             cgate);
    return COSY_OK;
}

int cosy_bn_train_backward(const float* dout, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, long M, int C, int act, const float* rowscale, int HW, float* dgamma, float* dbeta,
                           int accumulate, float* dx, float* sums /*2*C scratch*/, void* workspace, cosy_stream_t stream) {
    return cosy_bn_train_backward_gated(dout, nullptr, nullptr, 0.f, x, mean, rstd, gamma, beta, M, C, act, rowscale, HW, dgamma, dbeta, accumulate, dx,
                                        sums, workspace, stream);
}
int cosy_bn_train_backward_gated(const float* dout, const float* cgate, const float* cadd, float cadd_scale, const float* x, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, long M, int C, int act, const float* rowscale, int HW,
                                 float* dgamma, float* dbeta, int accumulate, float* dx, float* sums /*2*C scratch*/, void* workspace,
                                 cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(!cgate || (cadd && M % HW == 0 && M < (1l << 31)), "bn_train_backward: the per-sample gate / add rows need M=%ld = samples x HW=%d", M, HW);
    COSY_REQUIRE(dout && x && mean && rstd && gamma && beta && dgamma && dbeta && dx && sums && workspace && C % 4 == 0 && M > 0 && HW > 0,
                 "bn_train_backward: bad argument");
    const RedGeom g = red_geom(M, C);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(g.cgroups, g.nslab), dim3(256), 0, s, dout, x, mean, rstd, gamma, beta, M, C, act, rowscale,
                       HW, g.rows_per_slab, (double*)workspace, cgate, cadd, cadd_scale);
    COSY_CHECK_HIP(hipGetLastError());
    // this call's sums (needed by dx) into `sums`; the parameter gradients accumulate on request
    hipLaunchKernelGGL(bn_bwd_combine_final_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, s, (const double*)workspace, g.nslab, C, sums, dgamma, dbeta,
                       accumulate);
    COSY_CHECK_HIP(hipGetLastError());
    LAUNCH1D(bn_bwd_apply_kernel, M * (C / 4), s, dout, x, mean, rstd, gamma, beta, sums, sums + C, M * (C / 4), C / 4, 1.f / (float)M, act,
             rowscale, HW, dx, cgate, cadd, cadd_scale);
    return COSY_OK;
}

int cosy_dw_train_forward(const float* x, const float* wt, int B, int H, int W, int C, int k, int stride, float* out,
                          cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(x && wt && out && C % 4 == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2), "dw_train_forward: bad argument");
    const int lo = stride == 1 ? (k - 1) / 2 : (k - 2) / 2;
    const int Ho = stride == 1 ? H : (H + (k - 2) - k) / 2 + 1, Wo = stride == 1 ? W : (W + (k - 2) - k) / 2 + 1;
    const long n4 = (long)B * Ho * Wo * (C / 4);
    if (n4 == 0) return COSY_OK;
    (void)lo;
    const long n = (long)B * cdiv(Ho, 4) * Wo * (C / 4);
    if (k == 3 && stride == 1) LAUNCH1D((dw_rows_kernel<3, 1, false>), n, s, x, wt, H, W, Ho, Wo, C / 4, n, out);
    else if (k == 5 && stride == 1) LAUNCH1D((dw_rows_kernel<5, 1, false>), n, s, x, wt, H, W, Ho, Wo, C / 4, n, out);
    else if (k == 3) LAUNCH1D((dw_rows_kernel<3, 2, false>), n, s, x, wt, H, W, Ho, Wo, C / 4, n, out);
    else LAUNCH1D((dw_rows_kernel<5, 2, false>), n, s, x, wt, H, W, Ho, Wo, C / 4, n, out);
    return COSY_OK;
}

int cosy_dw_train_backward_data(const float* dy, const float* wt, int B, int H, int W, int C, int k, int stride, float* dx,
                                cosy_stream_t stream) {
    return cosy_dw_train_backward_data_add(dy, wt, nullptr, B, H, W, C, k, stride, dx, stream);
}
int cosy_dw_train_backward_data_add(const float* dy, const float* wt, const float* add, int B, int H, int W, int C, int k, int stride, float* dx,
                                    cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(dy && wt && dx && C % 4 == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2), "dw_train_backward_data: bad argument");
    COSY_REQUIRE(!add || stride == 1, "dw_train_backward_data: the fused skip gradient exists for stride 1 (blocks with a skip connection)%s", "");
    const int lo = stride == 1 ? (k - 1) / 2 : (k - 2) / 2;
    const int Ho = stride == 1 ? H : (H + (k - 2) - k) / 2 + 1, Wo = stride == 1 ? W : (W + (k - 2) - k) / 2 + 1;
    const long n4 = (long)B * H * W * (C / 4);
    if (n4 == 0) return COSY_OK;
    if (stride == 1) {
        const long n = (long)B * cdiv(H, 4) * W * (C / 4);
        if (k == 3) LAUNCH1D((dw_rows_kernel<3, 1, true>), n, s, dy, wt, H, W, H, W, C / 4, n, dx, add);
        else LAUNCH1D((dw_rows_kernel<5, 1, true>), n, s, dy, wt, H, W, H, W, C / 4, n, dx, add);
        return COSY_OK;
    }
    LAUNCH1D(dw_bwd_data_kernel, n4, s, dy, wt, H, W, C / 4, Ho, Wo, k, stride, lo, n4, dx);
    return COSY_OK;
}

int cosy_dw_train_backward_weight(const float* x, const float* dy, int B, int H, int W, int C, int k, int stride, float* dwt,
                                  void* workspace, cosy_stream_t stream) {
    return cosy_dw_train_backward_weight_ex(x, dy, B, H, W, C, k, stride, dwt, 0, workspace, stream);
}
int cosy_dw_train_backward_weight_ex(const float* x, const float* dy, int B, int H, int W, int C, int k, int stride, float* dw, int module_layout,
                                     void* workspace, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    float* dwt = dw;
    COSY_REQUIRE(x && dy && dwt && workspace && (k == 3 || k == 5) && (stride == 1 || stride == 2), "dw_train_backward_weight: bad argument");
    const int lo = stride == 1 ? (k - 1) / 2 : (k - 2) / 2;
    const int Ho = stride == 1 ? H : (H + (k - 2) - k) / 2 + 1, Wo = stride == 1 ? W : (W + (k - 2) - k) / 2 + 1;
    const long nunits = (long)B * Ho * cdiv(Wo, 4);     // unit = a run of 4 output pixels of one row (RUN in the kernel)
    COSY_REQUIRE(nunits > 0 && C % 4 == 0, "dw_train_backward_weight: empty batch or C %% 4 != 0");
    RedGeom g = red_geom(nunits * 2, C);     // ask for slabs as if there were nunits*2 rows
    const int ups = (int)cdiv(nunits, g.nslab);
    g.nslab = cdiv(nunits, ups);
    const dim3 grid(g.cgroups, g.nslab);
#define DW_BW(KS, ST) hipLaunchKernelGGL((dw_bwd_weight_kernel<KS, ST>), grid, dim3(256), 0, s, x, dy, H, W, C, Ho, Wo, lo, nunits, ups, (double*)workspace)
    if (k == 3 && stride == 1) DW_BW(3, 1);
    else if (k == 3) DW_BW(3, 2);
    else if (stride == 1) DW_BW(5, 1);
    else DW_BW(5, 2);
#undef DW_BW
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(combine_partials_kernel<double>, dim3(cdiv((long)k * k * C, 64)), dim3(1024), 0, s, (const double*)workspace, g.nslab,
                       (long)k * k * C, (double*)nullptr, dwt, module_layout ? C : 0);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// dW (N,K) = dY^T (N,M) . X (M,K) for tall-skinny shapes; returns COSY_EINVAL (nothing launched) when the shape is not
// one this kernel is built for -- the host then uses the library GEMM.
int cosy_wgrad_tall_supported(long M, int N, int K) { (void)M; return N > 0 && K > 0; }
int cosy_wgrad_tall(const float* dY, const float* X, long M, int N, int K, float* dW, void* workspace, cosy_stream_t stream) {
    return cosy_wgrad(dY, X, M, N, K, dW, workspace, stream);
}
// dW (N,K) = dY^T (N,M) . X (M,K), any shape: grid = (row slabs, n tiles, k tiles); every workgroup streams its rows once through
// the MFMA into TN x TK register tiles and writes a partial tile; a fixed-order combine pass adds the slabs (deterministic).
int cosy_wgrad(const float* dY, const float* X, long M, int N, int K, float* dW, void* workspace, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(dY && X && dW && workspace && M > 0 && N > 0 && K > 0, "wgrad: bad argument M=%ld N=%d K=%d", M, N, K);
    const int tk = cdiv(K, 16), tn = cdiv(N, 16);
    const int TN = (tn % 3 == 0 && tk <= 4) ? 3 : 2;      // n-tiles per workgroup; grid.y covers the rest
    const int TK = tk <= 4 ? tk : (tk == 9 || tk == 12) && (size_t)N * K <= 192 * 192 ? tk : 4;
    const int gy = cdiv(tn, TN), gz = cdiv(tk, TK);
    const size_t cap = cosy_train_workspace_bytes() / ((size_t)N * K * sizeof(float));
    COSY_REQUIRE(cap >= 1, "wgrad: N x K = %d x %d exceeds the workspace", N, K);
    static const int wg_cap = tune_int("COSY_WG_CAP", 256);
    int nwg = (int)std::min<size_t>((size_t)wg_cap, cap);
    if (gy * gz >= 64) nwg = std::min(nwg, 64);          // wide outputs: enough workgroups already, keep the combine short
    int rpw = (int)cdiv(M, (long)nwg * 4);
    rpw = cdiv(rpw, 8) * 8;
    nwg = (int)cdiv(M, (long)rpw * 4);
    const dim3 grid(nwg, gy, gz);
    float* partial = (float*)workspace;
#define WG(A, B_) hipLaunchKernelGGL((wgrad_tall_kernel<A, B_>), grid, dim3(256), 0, s, dY, X, M, N, K, rpw, partial)
    if (TN == 3) { if (TK == 1) WG(3, 1); else if (TK == 2) WG(3, 2); else if (TK == 3) WG(3, 3); else WG(3, 4); }
    else { if (TK == 1) WG(2, 1); else if (TK == 2) WG(2, 2); else if (TK == 3) WG(2, 3); else if (TK == 4) WG(2, 4); else if (TK == 9) WG(2, 9); else WG(2, 12); }
#undef WG
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(combine_partials_kernel<float>, dim3(cdiv((long)N * K, 64)), dim3(1024), 0, s, (const float*)partial, nwg, (long)N * K,
                       (double*)nullptr, dW);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// out (M,N) = A (M,K) . op(W) (+ add): op(W) = W^T for W stored (N,K) (a 1x1 convolution's forward), W for W stored (K,N)
// (its data gradient: dX = dY . W).  fp32 MFMA through pw_gemm with an identity epilogue; K % 4 == 0 (16-byte rows).
int cosy_train_gemm(const float* A, const float* W, int w_is_kn, long M, int K, int N, const float* add, float* out, void* workspace,
                    cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(A && W && out && workspace && M > 0 && K > 0 && N > 0, "train_gemm: bad argument M=%ld K=%d N=%d", M, K, N);
    COSY_REQUIRE(K % 4 == 0 && N % 4 == 0 && M < (1l << 31), "train_gemm: K=%d and N=%d must be multiples of 4", K, N);
    const PwCfg cfg = pw_choose_cfg(N);
    const size_t packed = pw_packed_elems(K, N, cfg, COSY_F32);
    const int n_pad = cdiv(N, pw_bn(cfg)) * pw_bn(cfg);
    COSY_REQUIRE((packed + 2 * (size_t)n_pad + 8) * sizeof(float) <= cosy_train_workspace_bytes(), "train_gemm: weights %d x %d exceed the workspace", N, K);
    float* wp = (float*)workspace;
    float* ones = wp + packed;
    float* zeros = ones + n_pad;
    const int nkb = (int)(packed / ((size_t)cdiv(N, pw_bn(cfg)) * cfg.NI * cfg.WN * 256));
    const long total = (long)packed;
    hipLaunchKernelGGL(pw_pack_f32_kernel, dim3(cdiv(std::max<long>(total, n_pad + 4), 256)), dim3(256), 0, s, W, K, N, w_is_kn, cfg.NI, cfg.WN, nkb,
                       total, n_pad, wp, ones, zeros);
    COSY_CHECK_HIP(hipGetLastError());
    PwArgs a{};
    a.A = A; a.Wp = wp; a.out = out; a.scale = ones; a.bias = zeros; a.res = add; a.gate = nullptr;
    a.M = (int)M; a.K = K; a.N = N; a.HW = (int)M; a.silu = 0; a.zeros = zeros;
    return launch_pw_gemm(a, cfg, COSY_F32, s);
}

// ---- the same GEMM on weights packed ahead of time, all of them in one launch (cosy_train_pack_all) ----
int cosy_train_pack_plan(int n, const float* const* W, const int* K, const int* N, const int* w_is_kn, long long* plan, long long* pool_floats,
                         long long* n_blocks) {
    COSY_REQUIRE(n > 0 && W && K && N && w_is_kn && plan && pool_floats && n_blocks, "train_pack_plan: null argument");
    long long off = PACK_HEAD_FLOATS, blk = PACK_HEAD_BLOCKS;
    for (int e = 0; e < n; ++e) {
        COSY_REQUIRE(W[e] && K[e] > 0 && N[e] > 0 && K[e] % 4 == 0 && N[e] % 4 == 0, "train_pack_plan: entry %d: K=%d N=%d", e, K[e], N[e]);
        const PwCfg cfg = pw_choose_cfg(N[e]);
        const size_t packed = pw_packed_elems(K[e], N[e], cfg, COSY_F32);
        COSY_REQUIRE(cdiv(N[e], pw_bn(cfg)) * pw_bn(cfg) <= PACK_HEAD_N, "train_pack_plan: N=%d exceeds the shared epilogue arrays", N[e]);
        long long* p = plan + (size_t)e * 8;
        p[0] = (long long)(uintptr_t)W[e]; p[1] = off; p[2] = K[e]; p[3] = N[e];
        p[4] = (w_is_kn[e] ? 1 : 0) | (cfg.NI << 8) | (cfg.WN << 16);
        p[5] = (long long)(packed / ((size_t)cdiv(N[e], pw_bn(cfg)) * cfg.NI * cfg.WN * 256));
        p[6] = (long long)packed; p[7] = blk;
        off += (long long)((packed + 63) / 64 * 64);                  // 256-byte aligned starts
        blk += (long long)cdiv((long)packed, 256l);
    }
    *pool_floats = off; *n_blocks = blk;
    return COSY_OK;
}
int cosy_train_pack_all(const long long* plan_dev, int n, long long n_blocks, float* pool, cosy_stream_t stream) {
    COSY_REQUIRE(plan_dev && pool && n > 0 && n_blocks > PACK_HEAD_BLOCKS && n_blocks < (1ll << 31), "train_pack_all: bad argument");
    hipLaunchKernelGGL(pw_pack_all_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, plan_dev, n, pool);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int cosy_train_gemm_packed(const float* A, const float* pool, long long packed_offset, long M, int K, int N, const float* add, float* out,
                           cosy_stream_t stream) {
    COSY_REQUIRE(A && pool && out && packed_offset >= PACK_HEAD_FLOATS && M > 0 && K > 0 && N > 0, "train_gemm_packed: bad argument M=%ld K=%d N=%d", M, K, N);
    COSY_REQUIRE(K % 4 == 0 && N % 4 == 0 && M < (1l << 31), "train_gemm_packed: K=%d and N=%d must be multiples of 4", K, N);
    const PwCfg cfg = pw_choose_cfg(N);
    PwArgs a{};
    a.A = A; a.Wp = pool + packed_offset; a.out = out; a.scale = pool; a.bias = pool + PACK_HEAD_N; a.res = add; a.gate = nullptr;
    a.M = (int)M; a.K = K; a.N = N; a.HW = (int)M; a.silu = 0; a.zeros = pool + PACK_HEAD_N;
    return launch_pw_gemm(a, cfg, COSY_F32, (hipStream_t)stream);
}

static int rows_chunks(int B, int HW, int C, int* rows_per_chunk) {
    int cap = 1024 / (B * cdiv(C, 64));
    if (cap < 1) cap = 1;
    int n = cdiv(HW, 128);
    if (n > cap) n = cap;
    *rows_per_chunk = cdiv(HW, n);
    return cdiv(HW, *rows_per_chunk);
}
int cosy_rows_mean(const float* a, int B, int HW, int C, float* out, void* workspace, cosy_stream_t stream) {
    COSY_REQUIRE(a && out && workspace && B > 0 && HW > 0 && C > 0, "rows_mean: bad argument");
    int rpc;
    const int nchunk = rows_chunks(B, HW, C, &rpc);
    hipLaunchKernelGGL(rows_reduce_kernel<0>, dim3(cdiv(C, 64), B, nchunk), dim3(256), 0, (hipStream_t)stream, a, (const float*)nullptr, HW, C,
                       rpc, (double*)workspace);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(rows_reduce_final_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nchunk, C,
                       1.f / (float)HW, out);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int cosy_rows_dot(const float* a, const float* a2, int B, int HW, int C, float* out, void* workspace, cosy_stream_t stream) {
    COSY_REQUIRE(a && a2 && out && workspace && B > 0 && HW > 0 && C > 0, "rows_dot: bad argument");
    int rpc;
    const int nchunk = rows_chunks(B, HW, C, &rpc);
    hipLaunchKernelGGL(rows_reduce_kernel<1>, dim3(cdiv(C, 64), B, nchunk), dim3(256), 0, (hipStream_t)stream, a, a2, HW, C, rpc,
                       (double*)workspace);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(rows_reduce_final_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nchunk, C,
                       1.f, out);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int cosy_rows_mean_bn(const float* raw, const float* mean, const float* rstd, const float* gamma, const float* beta, int B, int HW, int C, float* out,
                      void* workspace, cosy_stream_t stream) {
    COSY_REQUIRE(raw && mean && rstd && gamma && beta && out && workspace && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "rows_mean_bn: bad argument");
    int rpc;
    const int nchunk = rows_chunks(B, HW, C, &rpc);
    hipLaunchKernelGGL((rows_reduce_kernel<0, true>), dim3(cdiv(C, 64), B, nchunk), dim3(256), 0, (hipStream_t)stream, raw, (const float*)nullptr, HW, C,
                       rpc, (double*)workspace, BnParams{mean, rstd, gamma, beta});
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(rows_reduce_final_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nchunk, C,
                       1.f / (float)HW, out);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int cosy_rows_dot_bn(const float* a, const float* raw, const float* mean, const float* rstd, const float* gamma, const float* beta, int B, int HW, int C,
                     float* out, void* workspace, cosy_stream_t stream) {
    COSY_REQUIRE(a && raw && mean && rstd && gamma && beta && out && workspace && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "rows_dot_bn: bad argument");
    int rpc;
    const int nchunk = rows_chunks(B, HW, C, &rpc);
    hipLaunchKernelGGL((rows_reduce_kernel<1, true>), dim3(cdiv(C, 64), B, nchunk), dim3(256), 0, (hipStream_t)stream, a, raw, HW, C, rpc,
                       (double*)workspace, BnParams{mean, rstd, gamma, beta});
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(rows_reduce_final_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nchunk, C,
                       1.f, out);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int cosy_rows_scale(const float* a, const float* g, const float* add, float add_scale, int B, int HW, int C, float* out,
                    cosy_stream_t stream) {
    COSY_REQUIRE(a && g && out && C % 4 == 0 && B > 0 && HW > 0, "rows_scale: bad argument");
    const long n4 = (long)B * HW * (C / 4);
    LAUNCH1D(rows_scale_kernel, n4, (hipStream_t)stream, a, g, add, add_scale, n4, C / 4, HW, out);
    return COSY_OK;
}
int cosy_rows_broadcast(const float* v, float scale, int B, int HW, int C, float* out, cosy_stream_t stream) {
    COSY_REQUIRE(v && out && C % 4 == 0 && B > 0 && HW > 0, "rows_broadcast: bad argument");
    const long n4 = (long)B * HW * (C / 4);
    LAUNCH1D(rows_broadcast_kernel, n4, (hipStream_t)stream, v, scale, n4, C / 4, HW, out);
    return COSY_OK;
}
int cosy_act_forward(const float* x, long n, int kind, float* out, cosy_stream_t stream) {
    COSY_REQUIRE(x && out && (kind == 0 || kind == 1), "act_forward: bad argument");
    if (n == 0) return COSY_OK;
    LAUNCH1D(act_fwd_kernel, n, (hipStream_t)stream, x, n, kind, out);
    return COSY_OK;
}
int cosy_act_backward(const float* x, const float* dy, long n, int kind, float* dx, cosy_stream_t stream) {
    COSY_REQUIRE(x && dy && dx && (kind == 0 || kind == 1), "act_backward: bad argument");
    if (n == 0) return COSY_OK;
    LAUNCH1D(act_bwd_kernel, n, (hipStream_t)stream, x, dy, n, kind, dx);
    return COSY_OK;
}
int cosy_se_train_forward(const float* pooled, const float* w_reduce, const float* b_reduce, const float* w_expand, const float* b_expand, int B, int C,
                          int Cse, float* h_pre, float* gate, cosy_stream_t stream) {
    COSY_REQUIRE(pooled && w_reduce && b_reduce && w_expand && b_expand && h_pre && gate && B > 0 && C > 0 && C % 4 == 0 && Cse > 0 && Cse <= 128,
                 "se_train_forward: bad argument (C=%d must be a multiple of 4, Cse=%d <= 128)", C, Cse);
    constexpr int W1 = 8, W2 = 4;
    hipLaunchKernelGGL(se_train_fc1_kernel<W1>, dim3(cdiv(B, 16), cdiv(Cse, 16)), dim3(W1 * 64), 0, (hipStream_t)stream, pooled, w_reduce, b_reduce, B, C, Cse, h_pre);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(se_train_fc2_kernel<W2>, dim3(cdiv(B, 16), cdiv(cdiv(C, 16), W2)), dim3(W2 * 64), 0, (hipStream_t)stream, (const float*)h_pre, w_expand,
                       b_expand, B, C, Cse, gate);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int cosy_se_train_backward(const float* dgate, const float* gate, const float* h_pre, const float* pooled, const float* w_reduce, const float* w_expand,
                           int B, int C, int Cse, float* dpooled, float* dw_reduce, float* db_reduce, float* dw_expand, float* db_expand, void* workspace,
                           cosy_stream_t stream) {
    COSY_REQUIRE(dgate && gate && h_pre && pooled && w_reduce && w_expand && dpooled && dw_reduce && db_reduce && dw_expand && db_expand && workspace &&
                 B > 0 && C > 0 && C % 4 == 0 && Cse > 0 && Cse <= 128, "se_train_backward: bad argument (C=%d, Cse=%d)", C, Cse);
    COSY_REQUIRE((size_t)B * (C + Cse) * sizeof(float) <= cosy_train_workspace_bytes(), "se_train_backward: batch %d too large for the workspace", B);
    float* dg_pre = (float*)workspace;                 // (B, C)
    float* dh_pre = dg_pre + (size_t)B * C;            // (B, Cse)
    hipLaunchKernelGGL(se_train_bwd_x_kernel, dim3(B), dim3(512), (size_t)(C + 128 + 32 * 128) * sizeof(float), (hipStream_t)stream, dgate, gate, h_pre,
                       w_reduce, w_expand, C, Cse, dg_pre, dh_pre, dpooled);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(se_train_bwd_w_kernel, dim3(cdiv(cdiv(C, 16), 4)), dim3(256), 0, (hipStream_t)stream, (const float*)dg_pre, (const float*)dh_pre, h_pre,
                       pooled, B, C, Cse, dw_expand, db_expand, dw_reduce, db_reduce);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int cosy_fc_small_forward(const float* x, const float* w, const float* bias, int B, int C, int J, float* y, cosy_stream_t stream) {
    COSY_REQUIRE(x && w && bias && y && B > 0 && C > 0 && C % 4 == 0 && J > 0 && J <= 16, "fc_small_forward: bad argument (C=%d, J=%d <= 16)", C, J);
    hipLaunchKernelGGL(fc_small_fwd_kernel, dim3(B), dim3(256), (size_t)C * sizeof(float), (hipStream_t)stream, x, w, bias, C, J, y);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int cosy_fc_small_backward(const float* dy, const float* x, const float* w, int B, int C, int J, float* dx, float* dw, float* db, cosy_stream_t stream) {
    COSY_REQUIRE(dy && x && w && dx && dw && db && B > 0 && B * J <= 12288 && C > 0 && C % 4 == 0 && J > 0 && J <= 16,
                 "fc_small_backward: bad argument (B=%d, C=%d, J=%d)", B, C, J);
    hipLaunchKernelGGL(fc_small_bwd_x_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dy, w, C, J, dx);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(fc_small_bwd_w_kernel, dim3(cdiv(C, 256)), dim3(256), (size_t)B * J * sizeof(float), (hipStream_t)stream, dy, x, B, C, J, dw, db);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int cosy_stem_im2col_ld(const float* x_nhwc8, int B, int H, int W, int ld, float* cols, cosy_stream_t stream) {
    COSY_REQUIRE(x_nhwc8 && cols && B > 0 && ld >= 54, "stem_im2col: bad argument");
    const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;   // static same padding of k=3,s=2: pad total 1, all after
    const long n = (long)B * Ho * Wo * 9;
    LAUNCH1D(stem_im2col_kernel, n, (hipStream_t)stream, x_nhwc8, H, W, Ho, Wo, 0, n, ld, cols);
    return COSY_OK;
}
int cosy_stem_im2col(const float* x_nhwc8, int B, int H, int W, float* cols, cosy_stream_t stream) {
    return cosy_stem_im2col_ld(x_nhwc8, B, H, W, 54, cols, stream);
}

int cosy_loss_refiner_disentangled_backward(const float* TCO_possible_gt, const float* TCO_input, const float* refiner_outputs,
                                            const float* K_crop, const float* pts_table, const int* obj_id, int B, int S, int P,
                                            const float* dloss, float* d_refiner_outputs, cosy_stream_t stream) {
    COSY_REQUIRE(B >= 0 && P > 0 && S > 0, "loss_refiner_disentangled_backward: B=%d S=%d P=%d", B, S, P);
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(TCO_possible_gt && TCO_input && refiner_outputs && K_crop && pts_table && dloss && d_refiner_outputs,
                 "loss_refiner_disentangled_backward: null pointer");
    hipLaunchKernelGGL(loss_disentangled_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, TCO_possible_gt, TCO_input, refiner_outputs,
                       K_crop, pts_table, obj_id, S, P, dloss, d_refiner_outputs);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

int cosy_grad_norm_clip(const float* grads, long n, float max_norm, float* norm_and_coef, void* workspace, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(grads && norm_and_coef && workspace && n > 0, "grad_norm_clip: bad argument");
    const int parts = 1024;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(parts), dim3(256), 0, s, grads, n, (double*)workspace);
    COSY_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, parts, max_norm, norm_and_coef);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

int cosy_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, const float* norm_and_coef, cosy_stream_t stream) {
    COSY_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    LAUNCH1D(adam_kernel, n, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2),
             norm_and_coef);
    return COSY_OK;
}

}  // extern "C"
