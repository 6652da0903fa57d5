// Depthwise conv + BN + SiLU + squeeze partial sums for the blocks that run unfused (blocks 0, 1: no expansion; block 18).
// A file of its own because of ONE compiler flag: hipcc's SLP vectoriser pairs this kernel's tap FMAs into v_pk_fma_f32 --
// which issues no faster than two v_fma_f32 on gfx950 -- and pays for the pairing with register shuffles; built with
// -fno-slp-vectorize (cosypose_amd/build.py FILE_FLAGS) block 0 runs in 178 us instead of 207, block 1 in 110 instead of 125
// (256 crops of 256x256, fp16).  The GEMM kernels of kernels_net.hip want the vectoriser ON (their packed conversions:
// 41.7 -> 61 us without it on the N=136 project layers), hence the split.
#include "net_device.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace cosy {

// ==========================================================================================
// depthwise conv + BN + SiLU + squeeze partial sums
//
// LDS-staged tiles.  A workgroup owns TH x TW output pixels x CGB channel groups (8 channels = one 16-byte
// bf16 vector each) of one sample:
//   1. stage: the (TH*s+k-s) x (TW*s+k-s) x CGB input tile is pulled into LDS with wide, mutually independent
//      16-byte loads (several per thread in flight -> memory-level parallelism; zero fill implements the static
//      "same" padding of image_size=300 incl. the 15x20 -> 7x10 quirk), the k*k taps of the channel chunk too;
//   2. compute: a thread owns (channel group, output column, 4 output rows): for every tap column it walks the
//      input rows once from LDS (lanes are contiguous in LDS: conflict-free ds_read_b128) and feeds up to k
//      accumulators (sliding window over rows), then BN + SiLU, one 16-byte NHWC store per output;
//   3. the activated outputs are reduced per workgroup in a fixed order (deterministic) into
//      partial[b][spatial tile][C] for the squeeze-excite pooling.
// ==========================================================================================
static constexpr int DW_R = 4;  // output rows per thread
struct DwPlan { int CGB, TH, TW, THin, TWin, threads, n_chunks, ntx, nty; size_t lds; };
static DwPlan dw_plan(int C, int Ho, int Wo, int k, int s, int esz) {
    DwPlan p;
    p.TW = Wo < 32 ? Wo : 32;
    p.TH = Ho <= 4 ? 4 : 8;
    p.THin = (p.TH - 1) * s + k; p.TWin = (p.TW - 1) * s + k;
    const int cg = C / 8;
    static const int lim_kb = tune_int("COSY_DW_LDS_KB", 40);
    const size_t lim = esz == 2 ? (size_t)lim_kb * 1024 : 60 * 1024;
    p.CGB = 1;
    for (int d = 1; d <= 16 && d <= cg; ++d) {
        if (cg % d) continue;
        const int units = p.TW * (p.TH / DW_R) * d;
        const int thr = ((units < 256 ? units : 256) + 63) / 64 * 64;
        const size_t lds = (size_t)p.THin * p.TWin * d * 8 * esz + (size_t)k * k * d * 32 + (size_t)thr * 32;
        if (lds <= lim) p.CGB = d;
    }
    const int units = p.TW * (p.TH / DW_R) * p.CGB;
    p.threads = ((units < 256 ? units : 256) + 63) / 64 * 64;
    p.lds = (size_t)p.THin * p.TWin * p.CGB * 8 * esz + (size_t)k * k * p.CGB * 32 + (size_t)p.threads * 32;
    p.n_chunks = cg / p.CGB;
    p.ntx = cdiv(Wo, p.TW); p.nty = cdiv(Ho, p.TH);
    return p;
}
int dw_num_tiles(int C, int Ho, int Wo, int k) { DwPlan p = dw_plan(C, Ho, Wo, k, 1, 2); return p.ntx * p.nty; }

struct DwKArgs {
    const void* in; const float* w; const float* scale; const float* bias; void* out; float* partial;
    int H, W, C, Ho, Wo, lo, CGB, TH, TW, THin, TWin, ntx, n_tiles, n_chunks, n_jobs, dbg;
    const void* zeros;  // >= 16 zero bytes in global memory
};


template <typename T, int KS, int S>
__global__ __launch_bounds__(256) void dwconv_kernel(DwKArgs a) {
    constexpr int R = DW_R;
    constexpr int NROW = (R - 1) * S + KS;  // input rows feeding R output rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int CGB = a.CGB, TWin = a.TWin, THin = a.THin;
    T* tile = (T*)smem;                                                   // [THin][TWin][CGB][8]
    float* wl = (float*)(smem + (size_t)THin * TWin * CGB * 8 * sizeof(T));  // [KS*KS][CGB][8]
    float* red = wl + KS * KS * CGB * 8;                                  // [threads][8]
    const int tid = threadIdx.x, nthr = blockDim.x;
    // XCD-aware decode: all channel chunks of one (sample, spatial tile) job run on the same XCD (id % 8), so the
    // 128-byte lines they share are fetched into / written back from ONE L2 (the per-XCD L2s are not coherent).
    const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
    const int job = (jj / a.n_chunks) * 8 + xcd, chunk = jj % a.n_chunks;
    if (job >= a.n_jobs) return;
    const int tile_id = job % a.n_tiles, b = job / a.n_tiles;
    const int tx = tile_id % a.ntx, ty = tile_id / a.ntx;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int iy0 = oy0 * S - a.lo, ix0 = ox0 * S - a.lo;
    const int c0 = chunk * CGB * 8;
    {   // ---- stage the input tile (zero padded) with asynchronous global->LDS DMA (global_load_lds, 16 B per
        // lane): every 16-byte unit of the tile is in flight at once, no staging VGPRs, no ds_write.  The tile rows
        // are lane-linear in LDS (unit index = (x, channel) with channels fastest), which is exactly the DMA's
        // destination rule (wave-uniform base + lane*16); out-of-image lanes fetch from a 16-byte zero page
        // (= the static "same" padding).  Wave w stages tile rows w, w+nwaves, ...
        constexpr int UPV = 8 * sizeof(T) / 16;      // 16-byte units per 8-channel vector (1 bf16, 2 fp32)
        constexpr int JN = 5 * UPV;                  // max units per lane per tile row (TWin*CGB <= 320)
        const T* __restrict__ in = (const T*)a.in + (size_t)b * a.H * a.W * a.C + c0;
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthr >> 6;
        const int rowunits = TWin * CGB * UPV, upp = CGB * UPV;  // units per row / per pixel
        int goff[JN]; bool ok[JN];
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int idx = lane + 64 * j;
            const int xx = idx / upp, q = idx - xx * upp, ix = ix0 + xx;
            ok[j] = ix >= 0 && ix < a.W;
            goff[j] = ix * a.C + q * (16 / (int)sizeof(T));
        }
        for (int yy = wave; yy < (COSY_DBG(a.dbg & 2) ? 0 : THin); yy += nwaves) {
            const int iy = iy0 + yy;
            const bool yok = iy >= 0 && iy < a.H;
            const T* rowp = in + (size_t)iy * a.W * a.C;
            char* dst = smem + (size_t)yy * rowunits * 16;
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                if (64 * j < rowunits) {
                    const void* src = (yok && ok[j]) ? (const void*)(rowp + goff[j]) : (const void*)a.zeros;
                    if (lane + 64 * j < rowunits)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(dst + 64 * j * 16), 16, 0, 0);
                }
            }
        }
        const int nw = KS * KS * CGB * 2;  // float4 pieces
        for (int i = tid; i < nw; i += nthr) {
            const int h = i & 1, cg = (i >> 1) % CGB, tap = (i >> 1) / CGB;
            *(f32x4*)(wl + (tap * CGB + cg) * 8 + h * 4) = *(const f32x4*)(a.w + (size_t)tap * a.C + c0 + cg * 8 + h * 4);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA has landed; the barrier publishes it
    __syncthreads();
    // ---- compute
    const int nyq = a.TH / R;
    const int units = a.TW * nyq * CGB;
    const int stride = (nthr / CGB) * CGB;  // keeps a thread's channel group fixed across its units
    float sum[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) sum[c] = 0.f;
    if (tid < stride && !COSY_DBG(a.dbg & 1)) {
        const int cg = tid % CGB;
        float sc[8], bi[8];
        load8(a.scale + c0 + cg * 8, sc);
        load8(a.bias + c0 + cg * 8, bi);
        T* __restrict__ out = (T*)a.out + (size_t)b * a.Ho * a.Wo * a.C + c0 + cg * 8;
#pragma unroll 1
        for (int u = tid; u < units; u += stride) {
            const int q = u / CGB, x = q % a.TW, yq = q / a.TW;
            const int ox = ox0 + x, oyb = oy0 + yq * R;
            if (ox >= a.Wo || oyb >= a.Ho) continue;
            float acc[R][8];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[r][c] = 0.f;
#pragma unroll 1
            for (int kx = 0; kx < KS; ++kx) {   // rolled on purpose: bounds live registers to one tap column
                float wc[KS][8];
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) load8(wl + ((ky * KS + kx) * CGB + cg) * 8, wc[ky]);
                const T* col = tile + ((size_t)(yq * R * S) * TWin + x * S + kx) * CGB * 8 + cg * 8;
#pragma unroll
                for (int rr = 0; rr < NROW; ++rr) {
                    float v[8];
                    lds_ld8(col + (size_t)rr * TWin * CGB * 8, v);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int ky = rr - r * S;
                        if (ky >= 0 && ky < KS) {
#pragma unroll
                            for (int c = 0; c < 8; ++c) acc[r][c] += wc[ky][c] * v[c];
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int oy = oyb + r;
                if (oy < a.Ho) {
                    float y[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float v = acc[r][c] * sc[c] + bi[c];
                        v = v * sigmoid_t<T>(v);
                        y[c] = v;
                        sum[c] += v;
                    }
                    store8(out + ((size_t)oy * a.Wo + ox) * a.C, y);
                }
            }
        }
    }
    // ---- deterministic reduction of the squeeze sums over the threads that share a channel group
#pragma unroll
    for (int c = 0; c < 8; ++c) red[tid * 8 + c] = sum[c];
    __syncthreads();
    reduce_squeeze_sums(red, stride, CGB, 8, tid, nthr, a.partial + ((size_t)b * a.n_tiles + tile_id) * a.C + c0);
}

template <typename T>
static int launch_dw_t(const DwArgs& a, hipStream_t s) {
    const DwPlan p = dw_plan(a.C, a.Ho, a.Wo, a.k, a.s, sizeof(T));
    DwKArgs k;
    k.in = a.in; k.w = a.w; k.scale = a.scale; k.bias = a.bias; k.out = a.out; k.partial = a.partial;
    k.H = a.H; k.W = a.W; k.C = a.C; k.Ho = a.Ho; k.Wo = a.Wo; k.lo = a.pad_lo;
    k.CGB = p.CGB; k.TH = p.TH; k.TW = p.TW; k.THin = p.THin; k.TWin = p.TWin; k.ntx = p.ntx; k.n_tiles = p.ntx * p.nty;
    k.n_chunks = p.n_chunks; k.n_jobs = k.n_tiles * a.B; k.zeros = a.zeros;
    static const int dbg = tune_int("COSY_DW_DBG", 0);   // phase knock-out, timing experiments only
    k.dbg = dbg;
    dim3 grid((unsigned)cdiv(k.n_jobs, 8) * 8 * p.n_chunks), block(p.threads);
    if (a.k == 3 && a.s == 1) hipLaunchKernelGGL((dwconv_kernel<T, 3, 1>), grid, block, p.lds, s, k);
    else if (a.k == 3 && a.s == 2) hipLaunchKernelGGL((dwconv_kernel<T, 3, 2>), grid, block, p.lds, s, k);
    else if (a.k == 5 && a.s == 1) hipLaunchKernelGGL((dwconv_kernel<T, 5, 1>), grid, block, p.lds, s, k);
    else if (a.k == 5 && a.s == 2) hipLaunchKernelGGL((dwconv_kernel<T, 5, 2>), grid, block, p.lds, s, k);
    else { set_error("dwconv: unsupported k=%d s=%d", a.k, a.s); return COSY_EINVAL; }
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int launch_dwconv(const DwArgs& a, int dtype, hipStream_t s) {
    if (a.B == 0) return COSY_OK;
    COSY_REQUIRE(a.C % 8 == 0, "dwconv: C=%d must be a multiple of 8", a.C);
    return COSY_DISPATCH_T(dtype, launch_dw_t<T>(a, s));
}

}  // namespace cosy
