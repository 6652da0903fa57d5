// Row-streaming fused MBConv front for the high-resolution blocks (2-byte storage types):
//     expand 1x1 (MFMA) -> BN -> SiLU -> [LDS ring of expanded rows] -> depthwise kxk -> BN -> SiLU -> D, squeeze sums
// Reference: MBConvBlock.forward, cosypose/models/efficientnet.py:71-90 (expand_conv/bn0/swish, depthwise_conv/bn1/swish,
// adaptive_avg_pool2d of the squeeze-excite branch).
//
// Why another front kernel.  mbconv_front_kernel (kernels_net.hip) owns a small (sample, TH x TW) tile: per tile it pays the
// input DMA round trip, per-chunk weight / parameter loads, a x1.2-1.4 halo recompute of the expansion and a reduction
// of the squeeze sums -- its "skeleton" was 30 % of its time and it left the VALU at a few per cent of its rate.  Here a
// workgroup owns (sample, 48-channel chunk of the expanded tensor) and WALKS DOWN THE ROWS of the whole map:
//   * the chunk's expand weights (3 x KBN MFMA fragments) stay in registers and its BN / tap parameters in LDS for the
//     whole image: nothing is re-fetched per tile;
//   * the block input arrives as MFMA B fragments straight from global memory (16 pixels x 64 B are contiguous in NHWC
//     for Cin <= 32), one batch of rows AHEAD of its use (software prefetch into registers): no LDS staging, no DMA wait;
//   * expanded rows live in an LDS ring of RING = (RS-1)*S + KS rows: every expanded pixel is computed exactly once
//     (no halo recompute); rows outside the image are zero rows, the left/right padding columns are zeroed once;
//   * per step: expand NIN = RS*S new rows (matrix cores + BN + SiLU) | barrier | depthwise for RS output rows (thread =
//     4/8 channels x one column, sliding window over the ring) + BN + SiLU + NHWC store | barrier;
//   * squeeze sums accumulate in registers over the whole image and are reduced ONCE per workgroup in a fixed order
//     (deterministic); partial has one tile per sample.
// The chunk workgroups of one sample re-read the (small) block input; they are placed on the same XCD so that one L2
// serves them.
#include "net_device.h"

namespace cosy {

constexpr int rows_pitch(int elem_size) { return 48 * elem_size + 16; }   // bytes per pixel of the ring: conflict-free 16-byte accesses

struct RowsKArgs {
    const void* X; const void* Wp; const float* s0; const float* b0; const float* dww; const float* s1; const float* b1;
    void* D; float* partial; const void* zeros;
    int B, H, W, Cin, Cmid, Ho, Wo, lo, TWin, MBW, nkb_total, nchunks, nbatch, npre, base, mps;
};

// Opaque use + redefinition of a 128-bit register value: the compiler has to have the producing load finished HERE (it
// places its s_waitcnt in front of this statement) and cannot move memory operations across it.
template <typename V> __device__ __forceinline__ void retire_here(V& v) {
    f32x4 t = __builtin_bit_cast(f32x4, v);
    asm volatile("" : "+v"(t) : : "memory");
    v = __builtin_bit_cast(V, t);
}

// T = storage type of X / D (MFMA operand type), ET = element type of the LDS ring, RS = output rows per step,
// MPS = upper bound of the 16-pixel blocks a wave expands per step, UPT = upper bound of the depthwise (channel group,
// column) units per thread and step (register arrays).
//
// Ordering of the global memory operations inside a step matters more than anything else here: vmcnt retires loads and
// stores IN ORDER and hipcc cannot count stores issued under divergent control flow, so every wait for a prefetched
// fragment becomes s_waitcnt vmcnt(0) -- which also waits for the acknowledgement of every store issued before it.  A step
// therefore (1) computes its depthwise outputs into registers, (2) retires the fragments fetched a whole step ago (the only
// older stores are the previous step's: long done), (3) issues the loads for two batches ahead and only then (4) its own
// stores, which stay in flight across the barrier and the next step's expansion.  (With the wait behind the stores a step
// took ~12k cycles, of which ~10k were store acknowledgements.)
template <typename T, typename ET, int KS, int S, int KBN, int RS, int MPS, int UPT>
__global__ __launch_bounds__(384) void mbconv_rows_kernel(RowsKArgs a) {
    using raw_t = typename DT<T>::raw_t;
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int NI = 3, CC = 48;
    constexpr int PITCH = rows_pitch((int)sizeof(ET));
    constexpr int CPT = 16 / (int)sizeof(ET), NG = CC / CPT;
    constexpr int NIN = RS * S, RING = (RS - 1) * S + KS, NROW = RING;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TWin = a.TWin, rowbytes = TWin * PITCH;
    char* Et = smem;
    float* P = (float*)(Et + (size_t)RING * rowbytes);   // [s0 48][b0 48][s1 48][b1 48][taps KS*KS x 48]
    float* wl = P + 4 * CC;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthr >> 6;
    // XCD-aware decode: the chunk workgroups of one sample run on one XCD (id % 8)
    const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
    const int b = (jj / a.nchunks) * 8 + xcd, ch = jj % a.nchunks;
    if (b >= a.B) return;
    const int prow = lane & 15, kg = lane >> 4;
    const T* __restrict__ X = (const T*)a.X + (size_t)b * a.H * a.W * a.Cin;

    // ---- prologue: weights -> registers, parameters -> LDS, ring zeroed
    raw_t wf[NI][KBN];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb)
            wf[ni][kb] = *(const raw_t*)((const T*)a.Wp + ((size_t)(ch * NI + ni) * a.nkb_total + kb) * 64 * EPL + lane * EPL);
    for (int i = tid; i < (4 + KS * KS) * (CC / 4); i += nthr) {
        const int arr = i / (CC / 4), q = i - arr * (CC / 4);
        const float* src = arr == 0 ? a.s0 : arr == 1 ? a.b0 : arr == 2 ? a.s1 : arr == 3 ? a.b1 : a.dww + (size_t)(arr - 4) * a.Cmid;
        *(f32x4*)(P + arr * CC + q * 4) = *(const f32x4*)(src + ch * CC + q * 4);
    }
    for (int i = tid; i < RING * rowbytes / 16; i += nthr) *(f32x4*)(Et + (size_t)i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nmb = NIN * a.MBW;                 // 16-pixel blocks per batch of rows
    raw_t xf[MPS][KBN], xc[MPS][KBN];
    auto load_x = [&](int j) {                   // B fragments of batch j: wave w owns blocks w, w + nwaves, ...
#pragma unroll
        for (int q = 0; q < MPS; ++q) {
            const int idx = wave + q * nwaves;
            const int i = idx / a.MBW, xb = idx - i * a.MBW;
            const int r = a.base + j * NIN + i, x = xb * 16 + prow;
#pragma unroll
            for (int kb = 0; kb < KBN; ++kb) {
                const int k = kb * KB + kg * EPL;
                const bool ok = idx < nmb && r >= 0 && r < a.H && x < a.W && k < a.Cin;
                xf[q][kb] = *(const raw_t*)(ok ? (const void*)(X + ((size_t)r * a.W + x) * a.Cin + k) : a.zeros);
            }
        }
    };
    auto take_x = [&]() {                        // xc <- the fragments fetched last, retired at this program point
#pragma unroll
        for (int q = 0; q < MPS; ++q)
#pragma unroll
            for (int kb = 0; kb < KBN; ++kb) { retire_here(xf[q][kb]); xc[q][kb] = xf[q][kb]; }
    };
    load_x(0);
    take_x();
    if (a.nbatch > 1) load_x(1);

    // depthwise roles: thread = (channel group cq, column); the group is fixed for the whole image
    const int units = NG * a.Wo;
    const int stride = (nthr / NG) * NG;
    const int cq = tid % NG;
    float sum[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) sum[c] = 0.f;
    T* __restrict__ Dout = (T*)a.D + (size_t)b * a.Ho * a.Wo * a.Cmid + ch * CC + cq * CPT;
    typedef T out_t __attribute__((ext_vector_type(CPT)));
    __syncthreads();

    for (int j = 0; j < a.nbatch; ++j) {
        // ---- expand the rows of batch j
        {
            const int n0 = kg * 4 * NI;
            float sc[NI * 4], bi[NI * 4];
#pragma unroll
            for (int q = 0; q < NI; ++q) { load4(P + n0 + q * 4, sc + q * 4); load4(P + CC + n0 + q * 4, bi + q * 4); }
#pragma unroll
            for (int q = 0; q < MPS; ++q) {
                const int idx = wave + q * nwaves;
                const int i = idx / a.MBW, xb = idx - i * a.MBW;
                const int r = a.base + j * NIN + i, x = xb * 16 + prow;
                if (idx < nmb && r >= 0 && r < a.H) {      // wave-uniform
                    f32x4 acc[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < KBN; ++kb)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) mma(acc[ni], wf[ni][kb], xc[q][kb]);
                    float y[NI * 4];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[ni][e] * sc[ni * 4 + e] + bi[ni * 4 + e];
                            y[ni * 4 + e] = v * sigmoid_t<T>(v);
                        }
                    if (x < a.W) {
                        const int slot = (r + a.lo) % RING;
                        ET* dst = (ET*)(Et + (size_t)slot * rowbytes + (size_t)(x + a.lo) * PITCH) + kg * 4 * NI;
                        if constexpr (sizeof(ET) == 2) { store8(dst, y); store4(dst + 8, y + 8); }
                        else { store4(dst, y); store4(dst + 4, y + 4); store4(dst + 8, y + 8); }
                    }
                }
            }
            // rows below the image enter the window as zeros (their slot held a real row before)
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int r = a.base + j * NIN + i;
                if (r >= a.H) {
                    char* row = Et + (size_t)((r + a.lo) % RING) * rowbytes;
                    for (int u = tid; u < rowbytes / 16; u += nthr) *(f32x4*)(row + (size_t)u * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        __syncthreads();
        // ---- depthwise over the ring: output rows [oy0, oy0 + RS), results parked in registers
        const bool dw_step = j >= a.npre;
        const int oy0 = (j - a.npre) * RS;
        out_t yv[UPT][RS];
        if (dw_step) {
            const int slot0 = (oy0 * S) % RING;
            int rowofs[NROW];
#pragma unroll
            for (int rr = 0; rr < NROW; ++rr) { int s = slot0 + rr; s -= s >= RING ? RING : 0; rowofs[rr] = s * rowbytes; }
            float sc[CPT], bi[CPT];
#pragma unroll
            for (int c = 0; c < CPT; c += 4) { load4(P + 2 * CC + cq * CPT + c, sc + c); load4(P + 3 * CC + cq * CPT + c, bi + c); }
#pragma unroll
            for (int ui = 0; ui < UPT; ++ui) {
                const int u = tid + ui * stride;
                if (tid < stride && u < units) {
                    const int x = u / NG;
                    float acc[RS][CPT];
#pragma unroll
                    for (int r = 0; r < RS; ++r)
#pragma unroll
                        for (int c = 0; c < CPT; ++c) acc[r][c] = 0.f;
#pragma unroll 1
                    for (int kx = 0; kx < KS; ++kx) {
                        float wc[KS][CPT];
#pragma unroll
                        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                            for (int c = 0; c < CPT; c += 4) load4(wl + (ky * KS + kx) * CC + cq * CPT + c, wc[ky] + c);
                        const char* col = Et + (size_t)(x * S + kx) * PITCH + cq * 16;
#pragma unroll
                        for (int rr = 0; rr < NROW; ++rr) {
                            float v[CPT];
                            if constexpr (sizeof(ET) == 2) lds_ld8((const ET*)(col + rowofs[rr]), v);
                            else load4((const float*)(col + rowofs[rr]), v);
#pragma unroll
                            for (int r = 0; r < RS; ++r) {
                                const int ky = rr - r * S;
                                if (ky >= 0 && ky < KS) {
#pragma unroll
                                    for (int c = 0; c < CPT; ++c) acc[r][c] += wc[ky][c] * v[c];
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < RS; ++r) {
                        float y[CPT];
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            float v = acc[r][c] * sc[c] + bi[c];
                            v = v * sigmoid_t<T>(v);
                            y[c] = v;
                            if (oy0 + r < a.Ho) sum[c] += v;
                        }
                        if constexpr (sizeof(T) == 2 && __is_same(T, f16_t)) {
#pragma unroll
                            for (int c = 0; c < CPT; ++c) yv[ui][r][c] = to_f16_sat(y[c]);
                        } else {
#pragma unroll
                            for (int c = 0; c < CPT; ++c) yv[ui][r][c] = (T)y[c];
                        }
                    }
                }
            }
        }
        // ---- retire the fragments fetched a step ago, fetch two batches ahead, THEN store this step's outputs
        if (j + 1 < a.nbatch) take_x();
        if (j + 2 < a.nbatch) load_x(j + 2);
        if (dw_step) {
#pragma unroll
            for (int ui = 0; ui < UPT; ++ui) {
                const int u = tid + ui * stride;
                if (tid < stride && u < units) {
                    const int x = u / NG;
#pragma unroll
                    for (int r = 0; r < RS; ++r)
                        if (oy0 + r < a.Ho) *(out_t*)(Dout + ((size_t)(oy0 + r) * a.Wo + x) * a.Cmid) = yv[ui][r];
                }
            }
        }
        __syncthreads();   // the next batch overwrites ring rows this step has read
    }
    // ---- squeeze sums of the chunk: one fixed-order reduction per workgroup
    float* red = (float*)Et;
#pragma unroll
    for (int c = 0; c < CPT; ++c) red[tid * CPT + c] = sum[c];
    __syncthreads();
    reduce_squeeze_sums(red, stride, NG, CPT, tid, nthr, a.partial + (size_t)b * a.Cmid + ch * CC);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct RowsPlan { int kbn, RS, et32, mps, upt, TWin, MBW, nbatch, npre, base, threads; size_t lds; bool ok; };

// the instantiations that exist (KS, S, KBN, RS, 4-byte ring?, MPS bound, UPT bound)
#define COSY_ROWS_VARIANTS(X)                                                                         \
    X(3, 2, 1, 1, 1, 4, 2) X(3, 2, 1, 1, 1, 4, 3) X(3, 2, 1, 1, 0, 4, 2) X(3, 2, 1, 1, 0, 4, 3)         /* block 2 */       \
    X(3, 1, 1, 2, 1, 2, 2) X(3, 1, 1, 2, 1, 2, 3) X(3, 1, 1, 4, 1, 2, 2)                                /* blocks 3, 4 */   \
    X(5, 2, 1, 1, 1, 2, 1) X(5, 2, 1, 1, 1, 2, 2) X(5, 2, 1, 1, 0, 2, 1) X(5, 2, 1, 2, 0, 2, 1)         /* block 5 */       \
    X(5, 1, 2, 2, 1, 2, 1) X(5, 1, 2, 2, 1, 2, 2) X(5, 1, 2, 4, 1, 2, 1)                                /* blocks 6, 7 */   \
    X(3, 2, 2, 1, 1, 2, 1) X(3, 2, 2, 2, 1, 2, 1)                                                      /* block 8 */
static bool rows_built(int k, int s, int kbn, int rs, int et32, int mpsb, int upt) {
#define X(KS, S, KBN, RS, E32, MPS, UPT) if (k == KS && s == S && kbn == KBN && rs == RS && et32 == E32 && mpsb == MPS && upt == UPT) return true;
    COSY_ROWS_VARIANTS(X)
#undef X
    return false;
}

static RowsPlan rows_plan(int Cin, int Cmid, int H, int W, int Ho, int Wo, int k, int s) {
    RowsPlan p{};
    p.threads = 384;
    const int nwaves = p.threads / 64;
    p.kbn = cdiv(Cin, 32);
    p.MBW = cdiv(W, 16);
    p.TWin = s == 1 ? W + k - 1 : W + k - 2;      // static "same" padding: lo + hi = k-1 (s=1) / k-2 (s=2)
    if (p.TWin < (Wo - 1) * s + k) p.TWin = (Wo - 1) * s + k;
    // rows per step and ring element type: the ring should leave >= 2 workgroups per CU (160 KB of LDS)
    const size_t budget = (size_t)tune_int("COSY_ROWS_LDS_KB", 64) * 1024;
    const size_t par = (size_t)(4 + k * k) * 48 * 4;
    auto lds_of = [&](int rs, int et32) { return (size_t)((rs - 1) * s + k) * p.TWin * rows_pitch(et32 ? 4 : 2) + par; };
    p.RS = s == 1 ? 2 : 1; p.et32 = 1;
    if (lds_of(p.RS, 1) > budget && s == 2 && p.kbn == 1) p.et32 = 0;
    p.RS = tune_int("COSY_ROWS_RS", p.RS);
    p.et32 = tune_int("COSY_ROWS_ET32", p.et32);
    const int nin = p.RS * s;
    p.mps = cdiv((long)nin * p.MBW, nwaves);
    p.upt = cdiv((long)(p.et32 ? 12 : 6) * Wo, p.threads);
    p.npre = cdiv(k - s, nin);
    const int lo = s == 1 ? (k - 1) / 2 : (k - 2) / 2;
    p.base = -lo + (k - s) - p.npre * nin;
    p.nbatch = p.npre + cdiv(Ho, p.RS);
    p.lds = lds_of(p.RS, p.et32);
    if (p.lds < (size_t)p.threads * 32) p.lds = (size_t)p.threads * 32;   // the final reduction parks 8 floats per thread in the ring
    int upt_b = p.upt;    // smallest built bound >= what the shape needs
    while (upt_b <= 3 && !rows_built(k, s, p.kbn, p.RS, p.et32, p.mps <= 2 ? 2 : 4, upt_b)) ++upt_b;
    p.ok = Cmid % 48 == 0 && p.mps <= 4 && upt_b <= 3 && p.lds <= 150 * 1024;
    p.upt = upt_b;
    return p;
}
bool rows_supported(int Cin, int Cmid, int k, int s, int dtype, int H, int W, int Ho, int Wo) {
    if (dtype == COSY_F32 || H <= 0) return false;
    return rows_plan(Cin, Cmid, H, W, Ho, Wo, k, s).ok;
}
static const char* tname2(int dtype) { return dtype == COSY_BF16 ? "__bf16" : "_Float16"; }
void rows_kernel_name(int Cin, int Cmid, int k, int s, int dtype, int H, int W, int Ho, int Wo, char* buf, size_t n) {
    const RowsPlan p = rows_plan(Cin, Cmid, H, W, Ho, Wo, k, s);
    snprintf(buf, n, "mbconv_rows_kernel<%s, %s, %d, %d, %d, %d, %d, %d>", tname2(dtype), p.et32 ? "float" : tname2(dtype), k, s, p.kbn, p.RS,
             p.mps <= 2 ? 2 : 4, p.upt);
}

template <typename T, typename ET, int KS, int S, int KBN, int RS, int MPS, int UPT>
static int launch_rows_k(const RowsPlan& p, const RowsKArgs& k, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        COSY_CHECK_HIP(hipFuncSetAttribute((const void*)mbconv_rows_kernel<T, ET, KS, S, KBN, RS, MPS, UPT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const dim3 grid((unsigned)(cdiv(k.B, 8) * 8 * k.nchunks)), block(p.threads);
    hipLaunchKernelGGL((mbconv_rows_kernel<T, ET, KS, S, KBN, RS, MPS, UPT>), grid, block, p.lds, s, k);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
template <typename T>
static int launch_rows_t(const FuseArgs& a, hipStream_t s) {
    const RowsPlan p = rows_plan(a.Cin, a.Cmid, a.H, a.W, a.Ho, a.Wo, a.k, a.s);
    COSY_REQUIRE(p.ok, "mbconv_rows: unsupported shape Cin=%d Cmid=%d %dx%d k=%d s=%d", a.Cin, a.Cmid, a.H, a.W, a.k, a.s);
    RowsKArgs k;
    k.X = a.X; k.Wp = a.Wp; k.s0 = a.s0; k.b0 = a.b0; k.dww = a.dww; k.s1 = a.s1; k.b1 = a.b1; k.D = a.D; k.partial = a.partial;
    k.zeros = a.zeros; k.B = a.B; k.H = a.H; k.W = a.W; k.Cin = a.Cin; k.Cmid = a.Cmid; k.Ho = a.Ho; k.Wo = a.Wo; k.lo = a.pad_lo;
    k.TWin = p.TWin; k.MBW = p.MBW; k.nkb_total = (p.kbn + 1) & ~1; k.nchunks = a.Cmid / 48; k.nbatch = p.nbatch; k.npre = p.npre;
    k.base = p.base; k.mps = p.mps;
    const int mpsb = p.mps <= 2 ? 2 : 4, ks_ = a.k, st_ = a.s, kbn_ = p.kbn, rs_ = p.RS, e32_ = p.et32, upt_ = p.upt;
#define X(KS, S, KBN, RS, E32, MPS, UPT)                                                                              \
    if (ks_ == KS && st_ == S && kbn_ == KBN && rs_ == RS && e32_ == E32 && mpsb == MPS && upt_ == UPT) {              \
        if constexpr (E32) return launch_rows_k<T, float, KS, S, KBN, RS, MPS, UPT>(p, k, s);                          \
        else return launch_rows_k<T, T, KS, S, KBN, RS, MPS, UPT>(p, k, s);                                           \
    }
    COSY_ROWS_VARIANTS(X)
#undef X
    set_error("mbconv_rows: variant k=%d s=%d kbn=%d RS=%d et32=%d mps=%d upt=%d not built", a.k, a.s, p.kbn, p.RS, p.et32, p.mps, p.upt);
    return COSY_EINVAL;
}
int launch_mbconv_rows(const FuseArgs& a, int dtype, hipStream_t s) {
    if (a.B == 0) return COSY_OK;
    COSY_REQUIRE(dtype != COSY_F32, "mbconv_rows: 2-byte storage types only");
    if (dtype == COSY_BF16) return launch_rows_t<bf16_t>(a, s);
    return launch_rows_t<f16_t>(a, s);
}

}  // namespace cosy
