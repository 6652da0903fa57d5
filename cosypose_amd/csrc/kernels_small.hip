// The SMALL-WAVE kernels of the backbone: squeeze-excite (per-sample and batched forms), global pooling + Linear(1536, 9), the test probes and the pixel
// re-ordering copy.  A file of their own because of ONE target feature: they run as small waves on SIMDs they share with another HIP stream's MFMA kernels, and
// on gfx950 a wave's packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, which the SLP vectoriser likes) return wrong results while another wave
// of the SIMD issues 16-bit MFMAs with VGPR accumulators (profiles/r04_raster_streams.txt) -- this file is built without that feature (build.NO_PACKED_FP32);
// kernels_net.hip (GEMMs, stem, small-map / tiled fronts: MFMA kernels themselves) keeps it for its register budgets.
#include "net_device.h"
#include "kernels_net.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace cosy {

// ==========================================================================================
// squeeze-excite gate (efficientnet.py:85-88): pooled = sum(partials)/HW -> FC(C->Cse)+b -> swish ->
// FC(Cse->C)+b -> sigmoid.  One 512-thread workgroup per sample; the pooling is spread over
// (channel, tile-group) threads and finished by a fixed-order LDS reduction (deterministic).
// ==========================================================================================
// One 512-thread workgroup per sample (measured: batching 4 samples per workgroup to share the weight reads is
// slower -- the kernel is bound by the length of its dependent chains, not by L2 bandwidth).
__global__ __launch_bounds__(512) void se_kernel(SeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, C4 = C >> 2;       // C is a multiple of 8: everything moves as float4
    float* pooled = sm;                   // C
    float* redv = sm + C;                 // Cse (padded to 4)
    float* scratch = redv + ((a.Cse + 3) & ~3);  // 512 * 4
    const int b = blockIdx.x, tid = threadIdx.x;
    const float inv = 1.f / (float)a.HW;
    const f32x4* part = (const f32x4*)(a.partial + (size_t)b * a.n_tiles * C);
    // ---- pooling: sum the per-tile partial sums (fixed order -> deterministic)
    if (C4 >= 512 || a.n_tiles == 1) {
        for (int c = tid; c < C4; c += 512) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            int t = 0;
            for (; t + 1 < a.n_tiles; t += 2) { s0 += part[(size_t)t * C4 + c]; s1 += part[(size_t)(t + 1) * C4 + c]; }
            if (t < a.n_tiles) s0 += part[(size_t)t * C4 + c];
            ((f32x4*)pooled)[c] = (s0 + s1) * inv;
        }
    } else {
        const int G = 512 / C4;  // tile groups
        const int c = tid % C4, g = tid / C4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (g < G)
            for (int t = g; t < a.n_tiles; t += 8 * G) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = t + u * G < a.n_tiles ? part[(size_t)(t + u * G) * C4 + c] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
        ((f32x4*)scratch)[tid] = s;
        __syncthreads();
        if (tid < C4) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            for (int gg = 0; gg < G; ++gg) r += ((const f32x4*)scratch)[gg * C4 + tid];
            ((f32x4*)pooled)[tid] = r * inv;
        }
    }
    __syncthreads();
    // ---- reduce FC + swish: thread (j = t/8, part = t%8) accumulates the float4 chunks part, part+8, ... of row j:
    // every load is independent (deep memory pipeline), 8 lanes then combine with 3 shuffles.
    for (int j0 = 0; j0 < a.Cse; j0 += 64) {
        const int j = j0 + (tid >> 3), prt = tid & 7;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (j < a.Cse) {
            const f32x4* wr = (const f32x4*)(a.w_red + (size_t)j * C);
            for (int c = prt; c < C4; c += 64) {       // 8 independent 16-byte loads in flight, then the FMAs
                f32x4 w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = c + 8 * u < C4 ? wr[c + 8 * u] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c + 8 * u < C4) acc += w[u] * ((const f32x4*)pooled)[c + 8 * u];
            }
        }
        float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (prt == 0 && j < a.Cse) {
            s += a.b_red[j];
            redv[j] = s * (1.f / (1.f + expf(-s)));
        }
    }
    __syncthreads();
    // ---- expand FC + sigmoid: a thread owns 4 consecutive channels (w_exp stored (Cse, C))
    for (int c = tid; c < C4; c += 512) {
        const f32x4* we = (const f32x4*)a.w_exp + c;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        for (int j = 0; j < a.Cse; j += 8) {           // 8 independent loads per batch
            f32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = j + u < a.Cse ? we[(size_t)(j + u) * C4] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                if (j + u < a.Cse) s0 += w[u] * redv[j + u];
                if (j + u + 1 < a.Cse) s1 += w[u + 1] * redv[j + u + 1];
            }
        }
        const f32x4 be = ((const f32x4*)a.b_exp)[c];
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = 1.f / (1.f + expf(-(s0[e] + s1[e] + be[e])));
        ((f32x4*)(a.gate + (size_t)b * C))[c] = g;
    }
}
int launch_se(const SeArgs& a, hipStream_t s) {
    if (a.B == 0) return COSY_OK;
    hipLaunchKernelGGL(se_kernel, dim3(a.B), dim3(512), (a.C + ((a.Cse + 3) & ~3) + 2048) * sizeof(float), s, a);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// ------------------------------------------------------------------------------------------
// Batched squeeze-excite for the late blocks (Cmid 1392 / 2304, Cse 58 / 96).  se_kernel runs one workgroup per sample and
// every workgroup reads the whole of both FC matrices (0.65 / 1.77 MB): 256 samples = 166 / 453 MB through L2, 20 / 48 us
// per block.  The two FCs are small GEMMs over the BATCH: z = pooled (B x C) * Wr^T (C x Cse), gate = sigmoid(swish(z + br)
// (B x Cse) * We^T (Cse x C) + be), so a 16-sample tile shares one read of the weights.  Both run on the fp32 matrix
// instruction (v_mfma_f32_16x16x4_f32: an exact fp32 FMA chain per output, rows independent of each other -> a sample's
// gate does not depend on its tile mates: batch-invariant and deterministic).  Weights = MFMA A operand (rows = outputs),
// samples = B operand (columns), so a lane ends up with 4 consecutive outputs of one sample: 16-byte stores.
//   se_fc1: grid (sample tiles, Cse tiles of 16); the WAVES waves of a workgroup split the C range (k), fixed-order LDS combine.
//   se_fc2: grid (sample tiles, groups of WAVES channel tiles); k = padded Cse (4-6 steps: every load in flight at once).
// wr_p (CseP, C) and we_p (C, CseP) are zero-padded copies (CseP = Cse rounded up to 16) made at create time.
// ------------------------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void se_fc1_kernel(const float* __restrict__ partial, int n_tiles, const float* __restrict__ wr_p,
                                                            const float* __restrict__ br_p, float* __restrict__ redv, int B, int C, int CseP,
                                                            float inv_hw) {
    __shared__ f32x4 comb[WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kg = lane >> 4;
    const int b = min((int)blockIdx.x * 16 + i, B - 1), j0 = blockIdx.y * 16;
    const float* prow = partial + (size_t)b * n_tiles * C + kg * 4;
    const float* wrow = wr_p + (size_t)(j0 + i) * C + kg * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nsteps = C >> 4;
    for (int st0 = wave; st0 < nsteps; st0 += 4 * WAVES) {        // 4 independent steps (8+ loads) in flight per wave
        f32x4 a[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int st = st0 + u * WAVES;
            a[u] = f32x4{0.f, 0.f, 0.f, 0.f}; w[u] = a[u];
            if (st < nsteps) {
                w[u] = *(const f32x4*)(wrow + st * 16);
                for (int t = 0; t < n_tiles; ++t) a[u] += *(const f32x4*)(prow + (size_t)t * C + st * 16);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) mma(acc, w[u], a[u]);
    }
    comb[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        f32x4 s = comb[0][lane];
#pragma unroll
        for (int q = 1; q < WAVES; ++q) s += comb[q][lane];
        const f32x4 bias = *(const f32x4*)(br_p + j0 + kg * 4);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float z = s[e] * inv_hw + bias[e]; r[e] = z * (1.f / (1.f + expf(-z))); }
        if ((int)blockIdx.x * 16 + i < B) *(f32x4*)(redv + (size_t)b * CseP + j0 + kg * 4) = r;
    }
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void se_fc2_kernel(const float* __restrict__ redv, const float* __restrict__ we_p,
                                                            const float* __restrict__ be, float* __restrict__ gate, int B, int C, int CseP) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kg = lane >> 4;
    const int ct = blockIdx.y * WAVES + wave;
    if (ct * 16 >= C) return;
    const int b = min((int)blockIdx.x * 16 + i, B - 1), c0 = ct * 16;
    const float* rrow = redv + (size_t)b * CseP + kg * 4;
    const float* wrow = we_p + (size_t)(c0 + i) * CseP + kg * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nsteps = CseP >> 4;       // <= 8
    f32x4 a[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a[u] = f32x4{0.f, 0.f, 0.f, 0.f}; w[u] = a[u];
        if (u < nsteps) { w[u] = *(const f32x4*)(wrow + u * 16); a[u] = *(const f32x4*)(rrow + u * 16); }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) mma(acc, w[u], a[u]);
    const f32x4 bias = *(const f32x4*)(be + c0 + kg * 4);
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = 1.f / (1.f + expf(-(acc[e] + bias[e])));
    if ((int)blockIdx.x * 16 + i < B) *(f32x4*)(gate + (size_t)b * C + c0 + kg * 4) = g;
}
bool se_batched_supported(int C, int Cse) { return C % 16 == 0 && (Cse + 15) / 16 <= 8; }
int launch_se_batched(const SeArgs& a, const float* wr_p, const float* br_p, const float* we_p, float* redv, hipStream_t s) {
    if (a.B == 0) return COSY_OK;
    const int CseP = (a.Cse + 15) & ~15;
    COSY_REQUIRE(se_batched_supported(a.C, a.Cse), "se_batched: C=%d Cse=%d not supported", a.C, a.Cse);
    constexpr int W1 = 8, W2 = 4;
    hipLaunchKernelGGL(se_fc1_kernel<W1>, dim3(cdiv(a.B, 16), CseP / 16), dim3(W1 * 64), 0, s, a.partial, a.n_tiles, wr_p, br_p, redv, a.B, a.C,
                       CseP, 1.f / (float)a.HW);
    hipLaunchKernelGGL(se_fc2_kernel<W2>, dim3(cdiv(a.B, 16), cdiv(a.C / 16, W2)), dim3(W2 * 64), 0, s, redv, we_p, a.b_exp, a.gate, a.B, a.C, CseP);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}


// ==========================================================================================
// global average pool + Linear(1536, 9)   (pose.py:83-86): pool over (sample, 256-channel chunk)
// workgroups, then one small workgroup per sample for the 9 dot products.
// ==========================================================================================
template <typename T>
__global__ __launch_bounds__(256) void pool_kernel(const T* __restrict__ head, float* __restrict__ feat, int HW) {
    constexpr int C = 1536;
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    const T* h = head + (size_t)b * HW * C + c;
    float s0 = 0.f, s1 = 0.f;
    int p = 0;
    for (; p + 1 < HW; p += 2) { s0 += (float)h[(size_t)p * C]; s1 += (float)h[(size_t)(p + 1) * C]; }
    if (p < HW) s0 += (float)h[(size_t)p * C];
    feat[(size_t)b * C + c] = (s0 + s1) / (float)HW;
}
__global__ __launch_bounds__(576) void fc9_kernel(const float* __restrict__ feat, const float* __restrict__ fw,
                                                  const float* __restrict__ fb, float* __restrict__ pose) {
    constexpr int C = 1536;
    const int b = blockIdx.x, lane = threadIdx.x & 63, j = threadIdx.x >> 6;  // 9 waves, one output each
    const float* f = feat + (size_t)b * C;
    float s0 = 0.f, s1 = 0.f;
    for (int c = lane; c < C; c += 128) { s0 += fw[j * C + c] * f[c]; s1 += fw[j * C + c + 64] * f[c + 64]; }
    float s = s0 + s1;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) pose[b * 9 + j] = s + fb[j];
}
int launch_pool_fc(const void* head, const float* fc_w, const float* fc_b, float* feat, float* feat_scratch, float* pose, int B,
                   int HW, int dtype, hipStream_t s) {
    if (B == 0) return COSY_OK;
    float* f = feat ? feat : feat_scratch;
    COSY_REQUIRE(f, "pool_fc: no feature buffer");
    dim3 grid(1536 / 256, B);
    COSY_DISPATCH_STMT(dtype, hipLaunchKernelGGL(pool_kernel<T>, grid, dim3(256), 0, s, (const T*)head, f, HW));
    hipLaunchKernelGGL(fc9_kernel, dim3(B), dim3(576), 0, s, f, fc_w, fc_b, pose);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// NHWC (T) -> NCHW fp32 export of an activation (API parity for backbone(x) and the test probes; not on the hot path).
// chunked = 1: the source is in the fused fronts' D layout [sample][C/16][HW][16].
__device__ __forceinline__ size_t perm_pixel(size_t p, int lw, int lp) {      // PwArgs::out_perm_lw / out_perm_lp
    if (!lw) return p;
    const size_t x = p & (((size_t)1 << lw) - 1);
    return (p - x) + ((x & (((size_t)1 << lp) - 1)) << 4) + (x >> lp);
}
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ act, int HW, int C, float* __restrict__ out, int chunked, int colH, int lw, int lp) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // index in (C,HW)
    if (i >= (size_t)HW * C) return;
    const size_t c = i / HW;
    size_t p = i % HW;
    if (colH > 0) { const size_t Wm = (size_t)HW / colH; p = (p % Wm) * colH + p / Wm; }      // row-major pixel (y, x) lives at x * H + y
    const size_t src = chunked ? ((size_t)b * ((C + 15) >> 4) + (c >> 4)) * HW * 16 + perm_pixel(p, lw, lp) * 16 + (c & 15) : ((size_t)b * HW + p) * C + c;
    out[(size_t)b * HW * C + i] = (float)act[src];
}
int launch_nhwc_to_nchw(const void* act, int B, int HW, int C, int dtype, float* out, hipStream_t s, int chunked, int colH, int perm_lw, int perm_lp) {
    if (B == 0) return COSY_OK;
    dim3 grid(cdiv((long)HW * C, 256), B);
    COSY_DISPATCH_STMT(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, grid, dim3(256), 0, s, (const T*)act, HW, C, out, chunked, colH, perm_lw, perm_lp));
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// ==========================================================================================
// test probe: [mean, mean|x|, 14 strided samples] of one NHWC activation, indexed as if NCHW-flattened
// ==========================================================================================
template <typename T>
__global__ __launch_bounds__(256) void taps_kernel(const T* __restrict__ act, int HW, int C, float* __restrict__ taps, int tap_index, int colH, int chunked, int lw, int lp) {
    __shared__ double s1[256], s2[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    // chunked = 1: [sample][ceil(C/16)][HW][16] (the input layout of the matrix-pipe wave fronts; the pad channels of the last chunk are not part of the tensor)
    const T* a = act + (size_t)b * HW * (chunked ? (size_t)((C + 15) & ~15) : (size_t)C);
    const size_t n = (size_t)HW * C;
    auto at = [&](size_t p, size_t c) -> float { return (float)(chunked ? a[((c >> 4) * HW + perm_pixel(p, lw, lp)) * 16 + (c & 15)] : a[p * C + c]); };
    double x1 = 0, x2 = 0;
    for (size_t i = tid; i < n; i += 256) { const float v = at(i / C, i % C); x1 += v; x2 += fabsf(v); }
    s1[tid] = x1; s2[tid] = x2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s1[tid] += s1[tid + o]; s2[tid] += s2[tid + o]; }
        __syncthreads();
    }
    float* t = taps + ((size_t)b * 9 + tap_index) * 16;
    if (tid == 0) { t[0] = (float)(s1[0] / (double)n); t[1] = (float)(s2[0] / (double)n); }
    if (tid < 14) {
        const size_t idx = (size_t)(tid * 2 + 1) * n / 29;  // index in (C,H,W) order
        const size_t c = idx / HW;
        size_t p = idx % HW;
        if (colH > 0) { const size_t Wm = (size_t)HW / colH; p = (p % Wm) * colH + p / Wm; }
        t[2 + tid] = at(p, c);
    }
}
int launch_taps(const void* act, int B, int HW, int C, int dtype, float* taps, int tap_index, hipStream_t s, int colH, int chunked, int perm_lw, int perm_lp) {
    if (B == 0) return COSY_OK;
    COSY_DISPATCH_STMT(dtype, hipLaunchKernelGGL(taps_kernel<T>, dim3(B), dim3(256), 0, s, (const T*)act, HW, C, taps, tap_index, colH, chunked, perm_lw, perm_lp));
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

// ==========================================================================================
// exit of a resolution stage stored column-major (240x320 crops: the 30x40 and 15x20 maps, whose columns fill the wave kernel's lanes): one
// copy puts the last block's output back into row-major order for the kernels behind it.  16 bytes per thread, destination-contiguous.
// ==========================================================================================
__global__ __launch_bounds__(256) void pixels_to_rowmajor_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int H, int W, int C16, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // destination index in (b, y, x, c16)
    if (i >= n) return;
    const int c = (int)(i % C16);
    long r = i / C16;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const long b = r / H;
    out[i] = in[((b * W + x) * H + y) * C16 + c];
}
int launch_pixels_to_rowmajor(const void* in, void* out, int B, int H, int W, int C, int dtype, hipStream_t s) {
    if (B == 0) return COSY_OK;
    const int esz = dtype == COSY_F32 ? 4 : 2;
    if ((C * esz) % 16 != 0) { set_error("pixels_to_rowmajor: %d channels of %d bytes are not whole 16-byte pieces", C, esz); return COSY_EINVAL; }
    const int C16 = C * esz / 16;
    const long n = (long)B * H * W * C16;
    hipLaunchKernelGGL(pixels_to_rowmajor_kernel, dim3((unsigned)cdiv(n, 256l)), dim3(256), 0, s, (const uint4*)in, (uint4*)out, H, W, C16, n);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}


}  // namespace cosy
