// Stem + block 0 front as ONE wave-autonomous kernel (16-bit storage types, 256- and 320-pixel-wide crops):
//     stem conv 3x3 s2 6->40 (MFMA, implicit GEMM) -> BN -> SiLU -> block 0's depthwise 3x3 -> BN -> SiLU -> D, squeeze sums
// The stem tensor (B x 128 x 128 x 40: 335 MB per 256 crops in a 16-bit type) is never written to or read from memory: it lives as fp32
// rows in registers exactly as the expanded tensor of the MBConv fronts does (kernels_wave.hip).  Block 0 has no expansion
// (expand_ratio 1, efficientnet.py:58-63), so the stem convolution takes the expansion's place in the wave design:
// Reference: EfficientNet.extract_features' stem (efficientnet.py:174-176: _swish(_bn0(_conv_stem(x)))) + MBConvBlock.forward of block 0
// (:71-84, depthwise + BN + swish; the squeeze-excite gate and the project conv follow in their own kernels).
//
//   * job = one wavefront = (sample, 16-channel chunk of the 40 -> 48 stem channels, half of a stem row -- 64 of 128 or 80 of 160 pixels --, band of output rows);
//   * lane (p = lane & 15, kg = lane >> 4) owns the PPL = 4 (5) stem pixels x = 16 PPL * seg + 16 q + p (q < PPL) and the channel quad kg: the 16 lanes
//     of a fragment are 16 CONSECUTIVE pixels, so a fragment load touches 16 input pixels 32 bytes apart (the stride-2 window) -- one 512-byte
//     span per tap -- and an output store covers 512 contiguous bytes.  (First version: x = 4 p + q, neighbours in the lane's own registers as
//     in kernels_wave.hip; its fragment loads touched 64 different 128-byte lines each and the kernel ran 427 us, bound by the texture
//     addresser, against 362 us for the two kernels it replaces.)  Every x-neighbour therefore comes from the adjacent lane by DPP, and the
//     run ends (p = 0 / 15) from the neighbouring fragment's lane 15 / 0: row_ror / row_rol into the `old` operand of the row shift;
//   * the stem conv is an implicit GEMM with K = 9 taps x 8 NHWC8 input channels = 72 -> 3 k-blocks of 32: the B fragment of lane
//     (pixel, kg) in k-block kb is exactly ONE 16-byte input pixel -- tap 4 kb + kg of the 3x3 stride-2 window -- loaded straight from the
//     network input (as stem_kernel does); taps 9..11 meet zero weights.  The window's one out-of-image column / row (static "same" padding:
//     lo 0, hi 1) is fetched from the zero page by swapping the lane's byte offset, not by predicating the data;
//   * at the SEAM between the two halves of a row the neighbour is another job, so every job computes the one stem pixel beyond its seam itself
//     (a fifth 16-pixel fragment of which one pixel is used: +25 % of the stem's BN + SiLU work, the price of keeping 4 pixels per lane = 60
//     instead of 96 fragment registers);
//   * depthwise accumulation is input-stationary over three open output rows (compile-time slots: the row loop is unrolled by 3); the loop
//     body is ONE basic block -- rows outside the map or the band are computed on clamped addresses and switched off by a factor / a store
//     to a dump row -- so hipcc counts its own vmcnt waits (the next row's 15 fragment loads are issued behind this row's MFMAs and waited for
//     with the 4 output stores still in flight) and no register is hidden from it: no ISA check needed for this file.
// D is written in the chunked layout [sample][3][Hs * Ws][16] (channels 40..47 are exact zeros: zero weights, zero BatchNorm rows), which the
// project GEMM reads directly (PwArgs::a_chunked with K = 40).
#include "net_device.h"
#include <algorithm>
#include <type_traits>
#include <vector>

namespace cosy {

template <int CTRL> __device__ __forceinline__ float sf_dpp_mov0(float v) {   // lanes without a source read 0 (bound_ctrl:0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// rotate within the 16-lane row: ror1 -> lane i takes lane i - 1 (lane 0 takes lane 15); rol1 = ror15 -> lane i takes lane i + 1 (lane 15 takes lane 0)
__device__ __forceinline__ float sf_dpp_ror1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)); }
__device__ __forceinline__ float sf_dpp_rol1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x12f, 0xf, 0xf, false)); }
// row shift by one that KEEPS `old` in the lane without a source (lane 0 of shr, lane 15 of shl): bound_ctrl off
__device__ __forceinline__ float sf_dpp_shr1_keep(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float sf_dpp_shl1_keep(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}
// SiLU of four values, hand-scheduled (see kernels_wave.hip: silu4).  SCALED: x arrives times log2(e), the result is log2(e) * silu.
template <bool SCALED> __device__ __forceinline__ void sf_silu4(float* v) {
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

struct StemFrontKArgs {
    const void* X;            // (B, H, W, 8) network input
    const void* Wp;           // stem weights as MFMA A fragments: [chunk 3][k-block 4 (3 used)][lane 64][8]
    const float* params;      // [chunk 3][4 + 9][16]: s0 * log2 e, b0 * log2 e (stem BatchNorm), s1, b1 (block 0's BatchNorm 1), taps * ln 2
    void* D;                  // [sample][3][Hs * Ws][16]
    float* partial;           // (B, n_tiles, 40) squeeze partial sums, tile = band * 2 + seg
    void* dump;               // >= 128 * 32 bytes: where the finished rows outside a job's band go
    long zrel;                // zero page - X, in bytes (0 <= zrel < 2^32 - 2^24: checked by the launcher)
    int B, H, W, Hs, Ws, rsplit, rows_per, n_tiles;
};

enum { SF_KBN = 3, SF_PF = (4 + 9) * 16 };

// PPL = pixels per lane = half-row width / 16: 4 for 256-pixel-wide inputs (128-pixel stem rows), 5 for 320-pixel-wide ones (160-pixel rows)
template <typename T, int PPL>
__global__ __launch_bounds__(192, 2) void stem_front_kernel(StemFrontKArgs a) {
    using raw_t = typename DT<T>::raw_t;
    constexpr int KBN = SF_KBN, NQ = PPL + 1, PF = SF_PF, SEGW = 16 * PPL;      // SEGW: stem pixels of a half row
    typedef T out_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float sf_smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, kg = lane >> 4;
    // A workgroup = the THREE chunk jobs of one (sample, half, band): they read the same input rows at the same time, so the CU's L1 serves two
    // of the three (as four unrelated jobs per workgroup the kernel moved 12x the input through L2 -> L1: 427 us against 362 for the two kernels
    // it replaces).  XCD-aware order (block id % 8 = XCD): the workgroups of one sample stay on one XCD.
    const int id = blockIdx.x, xcd = id & 7, g = id >> 3;
    const int wps = 2 * a.rsplit;               // workgroups per sample
    const int b = (g / wps) * 8 + xcd, jrem = g % wps;
    const int ch = wave, seg = jrem & 1, band = jrem >> 1;
    if (b >= a.B) return;                       // nothing below synchronises across waves

    float* P = sf_smem + wave * (PF + KBN * 256);
    const float* Pl = P + kg * 4;               // this lane's channel quad inside every 16-float group
    // ---- the chunk's parameters -> wave-private LDS; weight fragments -> registers
    {
        const f32x4* PP = (const f32x4*)(a.params + (size_t)ch * PF);
        if (lane < PF / 4) *(f32x4*)(P + lane * 4) = PP[lane];
    }
    // (the three weight fragments are parked in the wave's LDS block too and re-read in front of every row's MFMAs: 12 registers that are
    // only needed while nothing else of a row is live)
    char* Wl = (char*)(P + PF);
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb)
        *(raw_t*)(Wl + kb * 1024 + lane * 16) = *(const raw_t*)((const T*)a.Wp + ((size_t)(ch * 4 + kb) * 64 + lane) * 8);

    // ---- per-lane byte offsets of the fragments inside a window of input rows (row base = 2 * sy, uniform)
    const int rowb = a.W * 16;                  // bytes per input row (NHWC8, 16-bit)
    // fragment (q, kb) of the lane's own pixels sits at xo[kb] + 512 q (an immediate of the load); so[kb]: the pixel beyond the seam
    unsigned xo[KBN], so[KBN];
    bool ky2[KBN], xoob[KBN];                   // this lane's tap lies in window row 2 / (q = 3 only) beyond the last input column
    const int xs = seg == 0 ? SEGW : SEGW - 1;  // stem pixel beyond the seam: the right neighbour of the left half, the left neighbour of the right half
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
        const int t = 4 * kb + kg, tt = t < 9 ? t : 0;      // taps 9..11: zero weights, any valid pixel will do
        const int ky = tt / 3, kx = tt - ky * 3;
        ky2[kb] = ky == 2;
        xo[kb] = (unsigned)(ky * rowb + (2 * (seg * SEGW + p) + kx) * 16);
        xoob[kb] = 2 * (seg * SEGW + 16 * (PPL - 1) + p) + kx >= a.W;
        so[kb] = (unsigned)(ky * rowb + (2 * xs + kx) * 16);
    }
    const char* Xs = (const char*)a.X + (size_t)b * a.H * rowb;
    const unsigned zrel_s = (unsigned)(a.zrel - (long)b * a.H * rowb);       // zero page relative to this sample's first byte

    raw_t x[NQ][KBN];
    auto load_row = [&](int sy) {               // fragments of stem row sy (clamped into the map: rows beyond it are switched off by `rv`)
        const int syc = min(sy, a.Hs - 1);
        const unsigned rb = (unsigned)(2 * syc) * (unsigned)rowb;
        const char* rowp = Xs + rb;
        const unsigned zr = zrel_s - rb;                        // the zero page seen from this row base
        const bool ylast = 2 * syc + 2 >= a.H;                  // window row 2 is the padding row (wave-uniform)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int kb = 0; kb < KBN; ++kb) {
                unsigned off = q < PPL ? xo[kb] : so[kb];
                const int imm = q < PPL ? 512 * q : 0;
                bool redirect = false;
                if (kb > 0) redirect = ylast && ky2[kb];                        // (k-block 0 holds taps 0..3: window rows 0 and 1 only)
                if (q == PPL - 1) redirect = redirect || xoob[kb];
                if (kb > 0 || q == PPL - 1) off = redirect ? zr - (unsigned)imm : off;
                x[q][kb] = *(const raw_t*)(rowp + off + imm);
            }
    };

    const int oy_a = band * a.rows_per, oy_b = min(a.Hs, oy_a + a.rows_per);
    const int sy0 = (max(oy_a - 1, 0) / 3) * 3;                 // first stem row of the walk: a multiple of 3, so a row's accumulator slots are compile-time
    const int n3 = (oy_b - sy0) / 3 + 1;                        // rows sy0 .. sy0 + 3 n3 - 1 include row oy_b (which finishes output row oy_b - 1)
    load_row(sy0);

    float acc[3][PPL][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < PPL; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[s][t][c] = 0.f;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    // Four stores to the dump row behind the first row's loads: the loop is entered with the memory queue in the state its back edge leaves
    // ([15 fragment loads][4 row stores]), so hipcc's wait in front of a row's first MFMA is the counted vmcnt(4) on both paths -- without them
    // it has to assume the worst of the two orders and drains the queue (vmcnt(0): the acknowledgement of the row just stored) every third row.
#pragma unroll
    for (int t = 0; t < PPL; ++t) *(out_t*)((T*)a.dump + (seg * SEGW + 16 * t + p) * 16 + kg * 4) = out_t{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    const float sl = seg == 1 ? 1.f : 0.f, sr = seg == 0 ? 1.f : 0.f;      // which side of this half is the seam (the other is the image border)
    T* __restrict__ Dch = (T*)a.D + (size_t)(b * 3 + ch) * a.Hs * a.Ws * 16;
    const int dlane = (seg * SEGW + p) * 16 + kg * 4;            // element offset of the lane's first pixel (q = 0) inside a row of the chunk; q adds 256
    const int drow = a.Ws * 16;

    auto row = [&](auto uc, const int base) {
        constexpr int u = decltype(uc)::value;
        const int sy = base + u;
        asm volatile("" ::: "memory");          // the parameter / tap reads from LDS stay inside the row (hoisted out of the loop they cost 52 registers)
        // ---- A. stem row sy: MFMAs, then the next row's loads, then BN + SiLU
        f32x4 m[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            m[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KBN; ++kb) mma(m[q], *(const raw_t*)(Wl + kb * 1024 + lane * 16), x[q][kb]);
        }
        __builtin_amdgcn_sched_barrier(0);      // the next row's fragments re-use this row's registers: not before the last MFMA has been issued
        load_row(sy + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float rv = sy < a.Hs ? 1.f : 0.f;                 // the row below the map is the depthwise conv's zero padding
        float sc0[4], bi0[4];
        load4(Pl + 0 * 16, sc0); load4(Pl + 1 * 16, bi0);
        float E[PPL][4];
        float seam[4];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float y4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y4[e] = m[q][e] * sc0[e] + bi0[e];     // = log2(e) * BN(stem conv)
            sf_silu4<true>(y4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (q < PPL) E[q][e] = y4[e] * rv; else seam[e] = y4[e] * rv;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- B. scatter into the three open output rows: stem row sy is tap row ky of output row sy + 1 - ky.  Pixel by pixel (q): its
        // x-neighbours are lane p -+ 1 of the same fragment, and for p = 0 / 15 lane 15 / 0 of fragment q -+ 1 (the seam pixel or the zero
        // padding at the ends of the half row): one rotate puts the other fragment's end lane into place, the row shift keeps it where it has no
        // source.  The nine tap quads are re-read from LDS per pixel: holding them (36 registers) or all twelve neighbour quads (48) costs the
        // third wave per SIMD.
#pragma unroll
        for (int t = 0; t < PPL; ++t) {
            asm volatile("" ::: "memory");      // (keeps hipcc from merging the four passes' tap reads into 36 live registers)
            float X3[3][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float lo = t > 0 ? sf_dpp_ror1(E[t - 1][c]) : seam[c] * sl;          // lane 0: pixel 16 t - 1
                const float hi = t < PPL - 1 ? sf_dpp_rol1(E[t + 1][c]) : seam[c] * sr;    // lane 15: pixel 16 t + 16
                X3[0][c] = sf_dpp_shr1_keep(lo, E[t][c]);
                X3[1][c] = E[t][c];
                X3[2][c] = sf_dpp_shl1_keep(hi, E[t][c]);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int os = (u + 1 - ky + 3) % 3;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float w[4];
                    load4(Pl + (4 + ky * 3 + kx) * 16, w);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[os][t][c] += w[c] * X3[kx][c];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- C. output row sy - 1 is complete (its slot took tap row 2 just now)
        {
            const long drow_off = (long)(sy - 1) * drow;        // (negative for the discarded row above the map: never dereferenced)
            const int os = (u + 2) % 3;
            const int oy = sy - 1;
            const bool valid = oy >= oy_a && oy < oy_b;          // wave-uniform
            const float fv = valid ? 1.f : 0.f;
            float sc1[4], bi1[4];
            load4(Pl + 2 * 16, sc1); load4(Pl + 3 * 16, bi1);
            // (a select between two ready pointers: a branch here would break the body into blocks and make every vmcnt wait a full drain)
            T* o = (valid ? Dch + drow_off : (T*)a.dump) + dlane;
#pragma unroll
            for (int t = 0; t < PPL; ++t) {
                float y4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y4[e] = acc[os][t][e] * sc1[e] + bi1[e];
                sf_silu4<false>(y4);
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
    for (int it = 0; it < n3; ++it) {
        const int base = sy0 + 3 * it;
        row(std::integral_constant<int, 0>{}, base);
        row(std::integral_constant<int, 1>{}, base);
        row(std::integral_constant<int, 2>{}, base);
    }
    // ---- squeeze sums: fixed-order tree over the 16 lanes of a row (one channel quad per row), lane 15 writes
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = sum[c];
        v += sf_dpp_mov0<0x111>(v); v += sf_dpp_mov0<0x112>(v); v += sf_dpp_mov0<0x114>(v); v += sf_dpp_mov0<0x118>(v);
        sum[c] = v;
    }
    if (p == 15 && ch * 16 + kg * 4 < 40)
        *(f32x4*)(a.partial + ((size_t)b * a.n_tiles + band * 2 + seg) * 40 + ch * 16 + kg * 4) = f32x4{sum[0], sum[1], sum[2], sum[3]};
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// row bands per (sample, half).  Measured at 256 crops of 256x256, fp16 (gpurun_out/r05g): 2 / 3 / 4 / 6 / 8 bands -> 302 / 267 / 285 / 280 / 287 us
// (3 bands = 1536 workgroups of 3 chunk waves; two recomputed stem rows per 43)
enum { SF_RSPLIT = 3, SF_RSPLIT_MAX = 8 };

bool stem_front_supported(int dtype, int H, int W) {
    static const int on = tune_int("COSY_STEM_FRONT", 1);
    return on && dtype != COSY_F32 && (W == 256 || W == 320) && H >= 64 && H % 2 == 0;
}
int stem_front_tiles(int H) { (void)H; return 2 * SF_RSPLIT_MAX; }      // upper bound (sizing); a launch reports what it wrote
size_t stem_front_weight_elems() { return (size_t)3 * 4 * 64 * 8; }
size_t stem_front_param_floats() { return (size_t)3 * SF_PF; }
size_t stem_front_dump_bytes() { return 16384; }     // one row of a chunk: <= 160 pixels x 32 bytes (the lanes keep their in-row offsets)

// w: reference layout (40, 6, 3, 3) -> A fragments per 16-channel chunk: row i of chunk c = channel 16 c + i, k = tap * 8 + ci (tap = 3 ky + kx)
void stem_front_pack_weights(const float* w, int dtype, void* dst) {
    std::vector<float> m((size_t)48 * 72, 0.f);
    for (int n = 0; n < 40; ++n)
        for (int ci = 0; ci < 6; ++ci)
            for (int tap = 0; tap < 9; ++tap) m[(size_t)n * 72 + tap * 8 + ci] = w[((size_t)n * 6 + ci) * 9 + tap];
    pw_pack_weights(m.data(), 72, 48, PwCfg{1, 1}, dtype, dst, 1);        // [n-tile 3][k-block 4 (72 -> 3, padded to an even count)][lane][8]
}
// s0 / b0: folded stem BatchNorm (40); dww: block 0's depthwise taps [tap][40]; s1 / b1: its folded BatchNorm 1 (40)
void stem_front_pack_params(const float* s0, const float* b0, const float* dww, const float* s1, const float* b1, float* dst) {
    const float L2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    for (int ch = 0; ch < 3; ++ch)
        for (int c = 0; c < 16; ++c) {
            float* d = dst + (size_t)ch * SF_PF + c;
            const int cc = ch * 16 + c;
            const bool real = cc < 40;
            d[0 * 16] = real ? s0[cc] * L2E : 0.f; d[1 * 16] = real ? b0[cc] * L2E : 0.f;
            d[2 * 16] = real ? s1[cc] : 0.f; d[3 * 16] = real ? b1[cc] : 0.f;
            for (int t = 0; t < 9; ++t) d[(4 + t) * 16] = real ? dww[(size_t)t * 40 + cc] * LN2 : 0.f;
        }
}

int launch_stem_front(const StemFrontArgs& f, int dtype, int* n_tiles_out, hipStream_t s) {
    static const int rsplit = std::min(std::max(tune_int("COSY_STEM_RSPLIT", SF_RSPLIT), 1), (int)SF_RSPLIT_MAX);
    *n_tiles_out = 2 * rsplit;
    if (f.B == 0) return COSY_OK;
    COSY_REQUIRE(stem_front_supported(dtype, f.H, f.W), "stem_front: unsupported input %dx%d / dtype %d", f.H, f.W, dtype);
    StemFrontKArgs k;
    k.X = f.X; k.Wp = f.Wp; k.params = f.params; k.D = f.D; k.partial = f.partial; k.dump = f.dump;
    k.B = f.B; k.H = f.H; k.W = f.W; k.Hs = f.H / 2; k.Ws = f.W / 2;
    k.rsplit = rsplit; k.rows_per = cdiv(k.Hs, k.rsplit); k.n_tiles = 2 * rsplit;
    k.zrel = (long)((const char*)f.zeros - (const char*)f.X);
    COSY_REQUIRE(k.zrel >= 0 && k.zrel < ((long)1 << 32) - ((long)1 << 24) && (long)f.B * f.H * f.W * 16 <= k.zrel,
                 "stem_front: the zero page must lie behind the input, within 4 GB of its start (zrel %ld)", k.zrel);
    const long wgs_per_xcd = (long)cdiv(f.B, 8) * 2 * k.rsplit;
    const dim3 grid((unsigned)(wgs_per_xcd * 8)), block(192);
    const size_t lds = (size_t)3 * (SF_PF + SF_KBN * 256) * sizeof(float);
    if (f.W == 256) {
        if (dtype == COSY_BF16) hipLaunchKernelGGL((stem_front_kernel<bf16_t, 4>), grid, block, lds, s, k);
        else hipLaunchKernelGGL((stem_front_kernel<f16_t, 4>), grid, block, lds, s, k);
    } else {
        if (dtype == COSY_BF16) hipLaunchKernelGGL((stem_front_kernel<bf16_t, 5>), grid, block, lds, s, k);
        else hipLaunchKernelGGL((stem_front_kernel<f16_t, 5>), grid, block, lds, s, k);
    }
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

}  // namespace cosy
