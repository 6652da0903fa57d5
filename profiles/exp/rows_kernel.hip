// NOT BUILT -- record of a measured dead end of round 5 (profiles/r05_dead_ends.txt): the stride-1 MBConv front rewritten in the style of kernels_stem.hip.
// It lived in cosypose_amd/csrc/ as kernels_rows.hip, dispatched from effnet.hip in place of mbconv_wave_kernel (Block::rows, COSY_ROWS_MASK); parity was green.

// Stride-1 MBConv front in the style of kernels_stem.hip (round 5): expand 1x1 (MFMA) -> BN -> SiLU -> depthwise kxk (stride 1) -> BN -> SiLU -> D,
// squeeze sums, with the expanded rows in registers -- the same fusion and the same arithmetic as mbconv_wave_kernel (kernels_wave.hip; reference:
// MBConvBlock.forward, cosypose/models/efficientnet.py:71-84), organised differently:
//   * lane (p = lane & 15, kg = lane >> 4) owns the pixels x = 16 q + p (q < PPL = row width / 16) and the channel quad kg: the 16 lanes of a fragment are
//     16 CONSECUTIVE pixels, so a fragment load is one contiguous run of 16 pixels x Cin channels and an output store 512 contiguous bytes (the wave
//     kernel gives a lane PPL consecutive pixels: its loads / stores touch 16 lines 4 * PPL pixels apart).  Every x-neighbour comes from the adjacent
//     lane by DPP; at the ends of a 16-lane run from lane 15 / 0 of the neighbouring fragment (rotate into the `old` operand of the row shift);
//   * the row loop is ONE basic block: rows outside the map or the band run on clamped addresses and are switched off by a factor / stored to a dump row,
//     every select is between ready values.  hipcc therefore counts its own vmcnt waits (next row's fragment loads issued behind this row's MFMAs,
//     waited for with the row's output stores still in flight) -- no inline-asm loads into registers the compiler does not know (kernels_wave.hip needs
//     an ISA check for that), one fragment set instead of two: the k = 5 / 16-pixel-row shape of blocks 14-17 fits 5 waves per SIMD instead of 4;
//   * depthwise accumulation is input-stationary over KS open output rows (compile-time slots: the loop is unrolled by KS); taps, BatchNorm rows and the
//     expand-weight fragments live in a wave-private LDS block and are re-read where they are used (held in registers they cost the extra waves).
// Parameters, weights and the chunked D layout are exactly the wave kernel's (wave_pack_params, PwCfg{1,1} fragments, [sample][Cmid/16][HW][16]).
#include "net_device.h"
#include <algorithm>
#include <type_traits>
#include <utility>

namespace cosy {

template <int CTRL, bool ZERO> __device__ __forceinline__ float rk_dpp(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, ZERO));
}
// value of pixel (16 q + p + D) for every lane p: lane p + D of `self` where it exists, else lane p + D -+ 16 of the neighbouring fragment `other`
// (HAS = false: the image border -> 0)
template <int D, bool HAS> __device__ __forceinline__ float rk_neighbour(float self, float other) {
    static_assert(D >= -2 && D <= 2 && D != 0, "taps reach two pixels");
    constexpr int SH = D < 0 ? 0x110 - D : 0x100 + D;            // row_shr:-D / row_shl:D
    constexpr int RO = D < 0 ? 0x120 - D : 0x130 - D;            // row_ror:-D / row_ror:16-D (= rol D)
    if constexpr (!HAS) return rk_dpp<SH, true>(0.f, self);
    else return rk_dpp<SH, false>(rk_dpp<RO, false>(0.f, other), self);
}
template <typename F, int... Us>
__device__ __forceinline__ void rk_unroll(F&& f, std::integer_sequence<int, Us...>) { (f(std::integral_constant<int, Us>{}), ...); }
template <bool SCALED> __device__ __forceinline__ void rk_silu4(float* v) {     // see kernels_wave.hip: silu4
    float t0, t1, t2, t3;
    if constexpr (SCALED) {
        asm volatile(
            "v_exp_f32 %4, -%0\n v_exp_f32 %5, -%1\n v_exp_f32 %6, -%2\n v_exp_f32 %7, -%3\n"
            "v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n"
            "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
            "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7\n s_nop 0"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
    } else {
        asm volatile(
            "v_mul_f32 %4, 0xbfb8aa3b, %0\n v_mul_f32 %5, 0xbfb8aa3b, %1\n v_mul_f32 %6, 0xbfb8aa3b, %2\n v_mul_f32 %7, 0xbfb8aa3b, %3\n"
            "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
            "v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n"
            "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
            "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7\n s_nop 0"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
    }
}

struct RowsKArgs {
    const void* X; const void* Wp; const float* wparams;
    void* D; float* partial; void* dump;
    int B, H, W, Cin, Cmid, nkb_total, nchunks, rsplit, rows_per;
};

template <typename T, int KS, int KBN, int PPL, int MINW>
__global__ __launch_bounds__(256, MINW) void mbconv_rows_kernel(RowsKArgs a) {
    using raw_t = typename DT<T>::raw_t;
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int LO = (KS - 1) / 2;
    constexpr int PF = (4 + KS * KS) * 16;                    // floats of the chunk's parameter block (wave_pack_params)
    constexpr int PFW = PF + KBN * 256;                       // + the expand-weight fragments
    typedef T out_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float rk_smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, kg = lane >> 4;
    // XCD-aware job order as in the wave kernel: the jobs of one sample stay on one XCD (block id % 8); the four waves of a workgroup are four
    // consecutive chunks of one (sample, band): they read the same input rows at the same time
    const int id = blockIdx.x, xcd = id & 7, sidx = (id >> 3) * 4 + wave;
    const int jps = a.nchunks * a.rsplit;
    const int b = (sidx / jps) * 8 + xcd, jrem = sidx % jps;
    const int band = jrem / a.nchunks, ch = jrem - band * a.nchunks;
    if (b >= a.B) return;

    float* P = rk_smem + wave * PFW;
    const float* Pl = P + kg * 4;
    char* Wl = (char*)(P + PF);
    {
        const f32x4* PP = (const f32x4*)(a.wparams + (size_t)ch * PF);
#pragma unroll
        for (int j = 0; j < (PF / 4 + 63) / 64; ++j)
            if (lane + 64 * j < PF / 4) *(f32x4*)(P + (lane + 64 * j) * 4) = PP[lane + 64 * j];
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb)
            *(raw_t*)(Wl + kb * 1024 + lane * 16) = *(const raw_t*)((const T*)a.Wp + ((size_t)ch * a.nkb_total + kb) * 64 * EPL + lane * EPL);
    }
    // fragment (q, kb) of lane (p, kg): EPL channels from k = KB kb + EPL kg of pixel 16 q + p (the channel tail of the last k-block reads a valid
    // neighbour quad: it meets the zero padding of the packed weights)
    const int pixb = a.Cin * (int)sizeof(T);
    unsigned xo[KBN];
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) xo[kb] = (unsigned)(p * pixb + min(kb * KB + kg * EPL, a.Cin - EPL) * (int)sizeof(T));
    const char* Xs = (const char*)a.X + (size_t)b * a.H * a.W * pixb;
    const int rowb = a.W * pixb;
    raw_t x[PPL][KBN];
    auto load_row = [&](int iy) {
        const char* rowp = Xs + (size_t)min(iy, a.H - 1) * rowb;
#pragma unroll
        for (int q = 0; q < PPL; ++q)
#pragma unroll
            for (int kb = 0; kb < KBN; ++kb) x[q][kb] = *(const raw_t*)(rowp + xo[kb] + q * 16 * pixb);
    };
    const int oy_a = band * a.rows_per, oy_b = min(a.H, oy_a + a.rows_per);
    const int iy0 = (max(oy_a - LO, 0) / KS) * KS;              // first input row of the walk: a multiple of KS, so a row's accumulator slots are compile-time
    const int nit = (oy_b - 1 + LO - iy0) / KS + 1;             // rows iy0 .. iy0 + KS nit - 1 include row oy_b - 1 + LO (which finishes output row oy_b - 1)
    load_row(iy0);

    float acc[KS][PPL][4];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < PPL; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[s][t][c] = 0.f;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    T* __restrict__ Dch = (T*)a.D + (size_t)(b * a.nchunks + ch) * a.H * a.W * 16;
    const int dlane = p * 16 + kg * 4;                           // element offset of the lane's pixel q = 0 inside a row of the chunk; q adds 256
    const int drow = a.W * 16;
    const long d_dump_rel = (T*)a.dump - Dch;
    // PPL stores to the dump row: the loop is entered with the memory queue in the state its back edge leaves ([fragment loads][row stores]), so the wait
    // in front of a row's first MFMA is a counted vmcnt on both paths (kernels_stem.hip)
#pragma unroll
    for (int t = 0; t < PPL; ++t) *(out_t*)((T*)a.dump + dlane + t * 256) = out_t{(T)0.f, (T)0.f, (T)0.f, (T)0.f};

    auto row = [&](auto uc, const int base, const long dbase_off) {
        constexpr int u = decltype(uc)::value;
        const int iy = base + u;
        asm volatile("" ::: "memory");          // parameter / tap / weight reads from LDS stay inside the row
        // ---- A. expanded row iy: MFMAs, then the next row's loads, then BN + SiLU
        f32x4 m[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            m[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KBN; ++kb) mma(m[q], *(const raw_t*)(Wl + kb * 1024 + lane * 16), x[q][kb]);
        }
        __builtin_amdgcn_sched_barrier(0);      // the next row's fragments re-use this row's registers
        load_row(iy + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float rv = iy < a.H ? 1.f : 0.f;  // rows below the map are the depthwise conv's zero padding
        float sc0[4], bi0[4];
        load4(Pl + 0 * 16, sc0); load4(Pl + 1 * 16, bi0);
        float E[PPL][4];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            float y4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y4[e] = m[q][e] * sc0[e] + bi0[e];     // = log2(e) * BN0(expand)
            rk_silu4<true>(y4);
#pragma unroll
            for (int e = 0; e < 4; ++e) E[q][e] = y4[e] * rv;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- B. scatter into the KS open output rows: input row iy is tap row ky of output row iy + LO - ky; pixel by pixel
        rk_unroll([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            asm volatile("" ::: "memory");      // (keeps hipcc from merging the passes' tap reads into KS * KS * 4 live registers)
            float N[KS][4];                     // the pixel and its x-neighbours: N[kx] = pixel 16 t + p + kx - LO
            constexpr bool HL = t > 0, HR = t < PPL - 1;
            constexpr int tl = HL ? t - 1 : t, tr = HR ? t + 1 : t;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                N[LO][c] = E[t][c];
                N[LO - 1][c] = rk_neighbour<-1, HL>(E[t][c], E[tl][c]);
                N[LO + 1][c] = rk_neighbour<1, HR>(E[t][c], E[tr][c]);
                if constexpr (KS == 5) {
                    N[0][c] = rk_neighbour<-2, HL>(E[t][c], E[tl][c]);
                    N[4][c] = rk_neighbour<2, HR>(E[t][c], E[tr][c]);
                }
            }
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int os = (u + LO - ky + KS) % KS;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float w[4];
                    load4(Pl + (4 + ky * KS + kx) * 16, w);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[os][t][c] += w[c] * N[kx][c];
                }
            }
        }, std::make_integer_sequence<int, PPL>{});
        __builtin_amdgcn_sched_barrier(0);
        // ---- C. output row iy - LO is complete (its slot took its last tap row just now)
        {
            const long drow_off = dbase_off + u * drow;
            const int os = (u - LO + KS) % KS;
            const int oy = iy - LO;
            const bool valid = oy >= oy_a && oy < oy_b;          // wave-uniform
            const float fv = valid ? 1.f : 0.f;
            float sc1[4], bi1[4];
            load4(Pl + 2 * 16, sc1); load4(Pl + 3 * 16, bi1);
            T* o = Dch + (valid ? drow_off : d_dump_rel) + dlane;      // a select between two ready offsets: no branch in the row body
#pragma unroll
            for (int t = 0; t < PPL; ++t) {
                float y4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y4[e] = acc[os][t][e] * sc1[e] + bi1[e];
                rk_silu4<false>(y4);
                out_t yo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sum[e] = __builtin_fmaf(y4[e], fv, sum[e]);
                    if constexpr (__is_same(T, f16_t)) yo[e] = to_f16_sat(y4[e]); else yo[e] = (T)y4[e];
                    acc[os][t][e] = 0.f;
                }
                *(out_t*)(o + t * 256) = yo;
            }
        }
    };
    for (int it = 0; it < nit; ++it) {
        const int base = iy0 + KS * it;
        const long db = (long)(base - LO) * drow;      // (formed outside the row: a 64-bit multiply in a select arm makes hipcc branch)
        rk_unroll([&](auto uc) { row(uc, base, db); }, std::make_integer_sequence<int, KS>{});
    }
    // ---- squeeze sums: fixed-order tree over the 16 lanes of a row (one channel quad per row), lane 15 writes
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = sum[c];
        v += rk_dpp<0x111, true>(0.f, v); v += rk_dpp<0x112, true>(0.f, v); v += rk_dpp<0x114, true>(0.f, v); v += rk_dpp<0x118, true>(0.f, v);
        sum[c] = v;
    }
    if (p == 15)
        *(f32x4*)(a.partial + ((size_t)b * a.rsplit + band) * a.Cmid + ch * 16 + kg * 4) = f32x4{sum[0], sum[1], sum[2], sum[3]};
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// built shapes (KS, KBN, PPL, minimum waves per SIMD, row bands): the stride-1 MBConv blocks of EfficientNet-B3 at 256x256 crops whose rows fill
// 16 * PPL lanes exactly
#define COSY_ROWS_VARIANTS(X) X(3, 1, 4, 4, 2) X(5, 2, 2, 4, 2) X(3, 3, 1, 5, 1) X(5, 3, 1, 5, 1) X(5, 5, 1, 5, 1)

static bool rows_shape(int Cin, int Cmid, int H, int W, int k, int s, int dtype, int* kbn, int* ppl) {
    *kbn = cdiv(Cin, 32); *ppl = W / 16;
    if (dtype == COSY_F32 || s != 1 || W % 16 || H < k || Cmid % 16 || Cin % 8) return false;
    bool ok = false;
#define X(KS, KBN, PPL, MW, RSP) if (k == KS && *kbn == KBN && *ppl == PPL) ok = true;
    COSY_ROWS_VARIANTS(X)
#undef X
    return ok;
}
bool rows_supported(int Cin, int Cmid, int k, int s, int dtype, int H, int W) {
    int kbn, ppl;
    return H > 0 && rows_shape(Cin, Cmid, H, W, k, s, dtype, &kbn, &ppl);
}
void rows_kernel_name(int Cin, int k, int dtype, int W, char* buf, size_t n) {
    int mw = 0;
    const int kbn = cdiv(Cin, 32), ppl = W / 16;
#define X(KS, KBN, PPL, MW, RSP) if (k == KS && kbn == KBN && ppl == PPL) mw = MW;
    COSY_ROWS_VARIANTS(X)
#undef X
    snprintf(buf, n, "mbconv_rows_kernel<%s, %d, %d, %d, %d>", dtype == COSY_BF16 ? "__bf16" : "_Float16", k, kbn, ppl, mw);
}

template <typename T, int KS, int KBN, int PPL, int MW, int RSP>
static int launch_rows_k(RowsKArgs k, int* n_tiles_out, hipStream_t s) {
    k.rsplit = std::min(std::max(tune_int("COSY_ROWS_RSPLIT", RSP), 1), 4);
    while (k.rsplit > 1 && cdiv(k.H, k.rsplit) < 8) --k.rsplit;
    k.rows_per = cdiv(k.H, k.rsplit);
    *n_tiles_out = k.rsplit;
    const size_t lds = (size_t)4 * ((4 + KS * KS) * 16 + KBN * 256) * sizeof(float);
    const long jobs_per_xcd = (long)cdiv(k.B, 8) * k.nchunks * k.rsplit;
    hipLaunchKernelGGL((mbconv_rows_kernel<T, KS, KBN, PPL, MW>), dim3((unsigned)(cdiv(jobs_per_xcd, 4) * 8)), dim3(256), lds, s, k);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
int launch_mbconv_rows(const FuseArgs& a, int dtype, void* dump, int* n_tiles_out, hipStream_t s) {
    *n_tiles_out = 1;
    if (a.B == 0) return COSY_OK;
    int kbn, ppl;
    COSY_REQUIRE(rows_shape(a.Cin, a.Cmid, a.H, a.W, a.k, a.s, dtype, &kbn, &ppl) && !a.x_colmajor && !a.d_colmajor && a.wparams && dump,
                 "mbconv_rows: unsupported shape Cin=%d Cmid=%d %dx%d k=%d s=%d", a.Cin, a.Cmid, a.H, a.W, a.k, a.s);
    RowsKArgs k;
    k.X = a.X; k.Wp = a.Wp; k.wparams = a.wparams; k.D = a.D; k.partial = a.partial; k.dump = dump;
    k.B = a.B; k.H = a.H; k.W = a.W; k.Cin = a.Cin; k.Cmid = a.Cmid; k.nkb_total = (kbn + 1) & ~1; k.nchunks = a.Cmid / 16; k.rsplit = 1; k.rows_per = a.H;
    const int ks_ = a.k;
#define X(KS, KBN, PPL, MW, RSP)                                                                                                   \
    if (ks_ == KS && kbn == KBN && ppl == PPL)                                                                                     \
        return dtype == COSY_BF16 ? launch_rows_k<bf16_t, KS, KBN, PPL, MW, RSP>(k, n_tiles_out, s) : launch_rows_k<f16_t, KS, KBN, PPL, MW, RSP>(k, n_tiles_out, s);
    COSY_ROWS_VARIANTS(X)
#undef X
    set_error("mbconv_rows: variant not built");
    return COSY_EINVAL;
}

}  // namespace cosy
