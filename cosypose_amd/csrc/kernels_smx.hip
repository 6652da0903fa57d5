// Fused MBConv front of the 8x8 maps (blocks 19-25 at 256x256 crops) with the depthwise taps on the matrix pipe:
//     expand 1x1 (MFMA) -> BN -> SiLU -> depthwise kxk (v_mfma_f32_4x4x4) -> BN -> SiLU -> D, squeeze sums.
// Reference: MBConvBlock.forward, cosypose/models/efficientnet.py:71-90.
//
// Same organisation as mbconv_small_kernel (kernels_net.hip): a workgroup of 4 waves owns (sample, a run of 48-channel chunks), the block input stays in
// REGISTERS for the whole kernel (wave w holds the MFMA fragments of the map's rows 2w, 2w+1 = one 16-pixel segment, all k-blocks), the chunk's expand
// weights and parameters are DMA'd into LDS one chunk ahead.  What differs is the depthwise phase, which there is 25 (9) fp32 FMAs per output value
// on 3 of the 4 waves from an fp32 tile in LDS -- half of the kernel's time by knock-out (profiles/r05_small_knockouts.txt):
//   * the expansion runs with its operands swapped (pixels as the MFMA's rows; see kernels_wave.hip, "depthwise taps on the matrix pipe"): a lane
//     leaves it with the 4 pixels of one quad of one channel; quads of a segment: j = 2 * (row in the segment) + (half of the row);
//   * the expanded values are rounded to the storage type, permuted to the small MFMA's lanes (lane = 4 * channel + quad) and parked in LDS as
//     8-byte operands: [segment -1 .. 4][16-channel tile][lane] (segments -1 and 4 stay zero: the padding above and below the map) -- 9 KB instead
//     of the 30 KB fp32 tile;
//   * wave w then produces OUTPUT segment w for the chunk's three channel tiles: tap row ky needs the input rows y + ky - LO of both of its rows.
//     For an even row offset that is a whole neighbouring segment; for an odd one the two rows come from two segments with the row halves of the
//     lanes swapped (quad_perm [2,3,0,1] and a select).  Per operand the halo along x is the other half of the same row (quad_perm move, zero at
//     the row ends).  2 KS small MFMAs per tile against Toeplitz fragments packed on the host (the wave kernel's);
//   * BatchNorm 1 + SiLU + squeeze sums where the outputs land (lane = (channel, quad)), rounding, lane permutation back and the transposing
//     v_mfma_f32_16x16x16 against the identity -> lane = (pixel, 4 channels): 8-byte stores into the chunked D layout [sample][Cmid/16][64][16].
// Numerics: E and the taps are rounded to the tap MFMAs' operand type -- fp16 in both 16-bit modes, as in the wave kernel (block_info kind 6; the oracle's emulation
// follows), accumulation fp32.  bf16 (round 6): the expand weights are hi + lo pairs (kernels_net.hip: pw_hl), two MFMAs per k-block against the same input fragment.
#include "net_device.h"

namespace cosy {

typedef short smx_s16x4 __attribute__((ext_vector_type(4)));
typedef int smx_i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 smx_mma4(f16x4 a, f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 smx_mma4(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(smx_s16x4, a), __builtin_bit_cast(smx_s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 smx_mma16(f16x4 a, f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 smx_mma16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(smx_s16x4, a), __builtin_bit_cast(smx_s16x4, b), c, 0, 0, 0);
}
template <int CTRL> __device__ __forceinline__ int smx_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float smx_dppf(float v) { return __builtin_bit_cast(float, smx_dpp<CTRL>(__builtin_bit_cast(int, v))); }

struct SmxKArgs {
    const void* X; const void* Wp; const char* params; void* D; float* partial; const void* zeros;
    int Cin, Cmid, nkb_total, ncg, cpw, dbg;
    // the map as the kernel walks it: NSEG segments of two 8-pixel walk rows; walk pixel (r, c) = map pixel r * Wm + c, or c * Wm + r for a transposed walk (7x10 maps:
    // the 10 columns are the walk rows); cv = valid pixels of a walk row (8, or 7: the eighth is expanded from a clamped address, zeroed as a tap operand, left out of
    // the squeeze sums and not stored); HW = pixels of the map
    int HW, Wm, tr, cv;
};
enum { SMX_HDR = 1024 };       // bytes of a chunk's parameter header [b0 * log2 e 48][s1 48][b1 48] fp32 (padded to one DMA instruction)
constexpr int smx_pbytes(int ks) { return SMX_HDR + 3 * ks * 2 * 512; }   // + [tile 3][ky][operand 2][lane 64] 8-byte Toeplitz fragments

// NSEG = 4: the 8x8 maps, wave w = segment w, three work units per wave and chunk (its segment x the chunk's three 16-channel tiles).
// NSEG = 5 (round 6: the 7x10 / 10x7 maps of 240x320 / 320x240 crops): still FOUR waves -- a fifth wave would share a SIMD with another wave of its workgroup and
// double that SIMD's share of every barrier-separated phase (measured: 95 us per block against 56 us on the 8x8 maps) -- wave w < 3 also owns the unit (segment 4, tile w):
// 4 + 4 + 4 + 3 units.  Unit u < 3 = (segment wave, tile u); unit 3 = (segment 4, tile wave).
template <typename T, int KS, int KBN, int NSEG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void mbconv_small_mx_kernel(SmxKArgs a) {
    using raw_t = typename DT<T>::raw_t;
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int NI = 3, CC = 48, LO = (KS - 1) / 2, PBYTES = smx_pbytes(KS), PJ = PBYTES / 1024;
    constexpr int HL = __is_same(T, bf16_t) ? 2 : 1, NF = KBN * HL;      // weight fragments per 16-channel tile: [k-block][hi | lo]
    constexpr int NU = NSEG == 5 ? 4 : 3;                                // work units per wave (the fourth: waves 0-2 only)
    typedef T t4 __attribute__((ext_vector_type(4)));
    typedef f16_t tt4 __attribute__((ext_vector_type(4)));          // operands of the tap MFMAs: fp16 in both 16-bit modes (a register / LDS format between two MFMAs, not storage)
    typedef T out_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Eh = smem;                               // [NSEG + 2 segments][NI][64 lanes] 8 bytes
    char* Wl = Eh + (NSEG + 2) * NI * 512;
    char* Pl = Wl + NI * NF * 1024;
    float* red = (float*)(Pl + 2 * PBYTES);        // [NSEG segments][48]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.ncg, cg = blockIdx.x - b * a.ncg;
    const int nchunks = a.Cmid / CC;
    const int ch0 = cg * a.cpw, ch1 = min(nchunks, ch0 + a.cpw);
    const int prow = lane & 15, kg = lane >> 4;
    const int cb = lane >> 2, jq = lane & 3;       // small-MFMA roles: channel of the tile, quad (row jq >> 1 of the segment, half jq & 1)
    const bool has4 = NSEG == 5 && wave < 3;       // this wave owns (segment 4, tile wave) as well
    auto useg = [&](int u) -> int { return u < 3 ? wave : 4; };
    auto utile = [&](int u) -> int { return u < 3 ? u : wave; };

    auto issue_w = [&](int ch) {
        for (int blk = wave; blk < NI * NF; blk += 4) {
            const int ni = blk / NF, f = blk - ni * NF;
            const T* src = (const T*)a.Wp + (((size_t)(ch * NI + ni) * a.nkb_total + f / HL) * HL + f % HL) * 64 * EPL + lane * EPL;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Wl + (size_t)blk * 1024), 16, 0, 0);
        }
        for (int j = wave; j < PJ; j += 4) {
            const char* src = a.params + (size_t)ch * PBYTES + (size_t)j * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Pl + (size_t)(ch & 1) * PBYTES + (size_t)j * 1024), 16, 0, 0);
        }
    };
    issue_w(ch0);
    // the segments of the block input this wave expands (walk rows 2 seg, 2 seg + 1) -> registers: the A operand (rows = pixels) of the expansion
    auto walk_pix = [&](int seg, int c) -> int { const int r = 2 * seg + (prow >> 3); return a.tr ? c * a.Wm + r : r * a.Wm + c; };
    raw_t xf[NU == 4 ? 2 : 1][KBN];
#pragma unroll
    for (int q = 0; q < (NU == 4 ? 2 : 1); ++q) {
        const int seg = q == 0 ? wave : 4;
        const T* __restrict__ X = (const T*)a.X + ((size_t)b * a.HW + walk_pix(seg, min(prow & 7, a.cv - 1))) * a.Cin;      // (beyond the walk row: a clamped, valid address)
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb) {
            const int k = kb * KB + kg * EPL;
            xf[q][kb] = *(const raw_t*)(k < a.Cin ? (const void*)(X + k) : a.zeros);
        }
    }
    for (int i = tid; i < (NSEG + 2) * NI * 512 / 16; i += 256) *(f32x4*)(Eh + (size_t)i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};   // segments -1 and NSEG are never written
    const int bp_in = (cb + 16 * jq) * 4, bp_out = (4 * prow + kg) * 4;
    t4 ident;
#pragma unroll
    for (int e = 0; e < 4; ++e) ident[e] = (T)(4 * kg + e == prow ? 1.f : 0.f);
    auto cvt = [](float v) -> T { if constexpr (__is_same(T, f16_t)) return (T)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); else return (T)v; };
    auto cvt_e = [](float v) -> f16_t { return (f16_t)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); };      // E as a tap operand: fp16, saturating
    const bool row0 = jq < 2, half0 = (jq & 1) == 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // weights / parameters of the first chunk and the input fragments have landed
    __syncthreads();
    for (int ch = ch0; ch < ch1; ++ch) {
        const char* P = Pl + (size_t)(ch & 1) * PBYTES;
        const float* hdr = (const float*)P;
        // ---- expansion (operands swapped) -> E as fp16, in the small MFMA's lanes -> Eh[segment + 1].  The units' MFMA chains advance together
        // (k-block outer): with 2 waves per SIMD the chains' own latency is what a phase costs
        if (!COSY_DBG(a.dbg & 2)) {
            f32x4 acc[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float b0 = hdr[utile(u) * 16 + prow];                // log2(e) * BN0 bias of this lane's channel: the C operand of the first MFMA
                acc[u] = f32x4{b0, b0, b0, b0};
            }
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (u < 3 || has4) mma(acc[u], xf[u < 3 ? 0 : (NU == 4 ? 1 : 0)][f / HL], *(const raw_t*)(Wl + (size_t)(utile(u) * NF + f) * 1024 + lane * 16));
            smx_i32x2 hh[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                tt4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[u][e];                              // = log2(e) * BN0(expand)
                    hv[e] = cvt_e(t * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-t)) * 0.6931471805599453f);      // silu
                }
                if (a.cv < 8 && (kg & 1)) hv[3] = (f16_t)0.f;         // this lane's pixels 4 kg .. 4 kg + 3 of the segment: the eighth pixel of a walk row pads its neighbours
                hh[u] = __builtin_bit_cast(smx_i32x2, hv);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const smx_i32x2 w1 = smx_i32x2{__builtin_amdgcn_ds_bpermute(bp_in, hh[u][0]), __builtin_amdgcn_ds_bpermute(bp_in, hh[u][1])};      // (every lane takes part in the permutation)
                if (u < 3 || has4) *(smx_i32x2*)(Eh + (size_t)((useg(u) + 1) * NI + utile(u)) * 512 + lane * 8) = w1;
            }
        }
        __syncthreads();                       // Eh complete; the weight buffer is free
        if (ch + 1 < ch1 && !COSY_DBG(a.dbg & 4)) issue_w(ch + 1);     // lands while the tap phase computes
        // ---- taps: a unit's output segment
        out_t yv[NU];
        if (!COSY_DBG(a.dbg & 1)) {
            smx_i32x2 S[NU][3], Sw[NU][3];
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q) S[u][q] = *(const smx_i32x2*)(Eh + (size_t)((useg(u) + q) * NI + utile(u)) * 512 + lane * 8);      // input segments seg - 1, seg, seg + 1
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q) Sw[u][q] = smx_i32x2{smx_dpp<0x4E>(S[u][q][0]), smx_dpp<0x4E>(S[u][q][1])};             // rows of the segment swapped (quad_perm [2,3,0,1])
            // every LDS read of the phase is issued up front (the Toeplitz fragments: 2 KS x 3 register pairs): a read's round trip is several hundred
            // cycles while the next chunk's DMA is writing into the LDS, and 2 waves per SIMD do not hide one per tap row
            tt4 A0[KS][NI], A1[KS][NI];
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    A0[ky][ni] = *(const tt4*)(P + SMX_HDR + (size_t)((ni * KS + ky) * 2 + 0) * 512 + lane * 8);
                    A1[ky][ni] = *(const tt4*)(P + SMX_HDR + (size_t)((ni * KS + ky) * 2 + 1) * 512 + lane * 8);
                }
            // unit 3's tile is this wave's index: its fragments selected once (wave-uniform)
            tt4 A0x[NU == 4 ? KS : 1], A1x[NU == 4 ? KS : 1];
            if constexpr (NU == 4) {
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    A0x[ky] = wave == 0 ? A0[ky][0] : wave == 1 ? A0[ky][1] : A0[ky][2];
                    A1x[ky] = wave == 0 ? A1[ky][0] : wave == 1 ? A1[ky][1] : A1[ky][2];
                }
            }
            float s1v[NU], b1v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) { s1v[u] = hdr[CC + utile(u) * 16 + cb]; b1v[u] = hdr[2 * CC + utile(u) * 16 + cb]; }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 accx[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) accx[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {                              // tap row outer: the units' accumulation chains advance together
                const int d = ky - LO;                                      // input row = output row + d
                smx_i32x2 op[NU], w2[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (d == -2) op[u] = S[u][0];
                    else if (d == 0) op[u] = S[u][1];
                    else if (d == 2) op[u] = S[u][2];
                    else if (d == -1) op[u] = smx_i32x2{row0 ? Sw[u][0][0] : Sw[u][1][0], row0 ? Sw[u][0][1] : Sw[u][1][1]};
                    else op[u] = smx_i32x2{row0 ? Sw[u][1][0] : Sw[u][2][0], row0 ? Sw[u][1][1] : Sw[u][2][1]};
                    // halo along x: the other half of the same row (pixels 2, 3 of the left half / 0, 1 of the right half), zero at the row ends
                    // (the moves run in ALL lanes, the selects follow: a DPP move inside a divergent branch reads nothing from the lanes the branch switched off)
                    const int hpm = smx_dpp<0xA0>(op[u][1]), lnm = smx_dpp<0xF5>(op[u][0]);       // quad_perm [0,0,2,2] / [1,1,3,3]
                    w2[u] = smx_i32x2{half0 ? 0 : hpm, half0 ? lnm : 0};
                }
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (u < 3) accx[u] = smx_mma4(A0[ky][u < 3 ? u : 0], __builtin_bit_cast(tt4, op[u]), accx[u]);
                    else if (has4) accx[u] = smx_mma4(A0x[NU == 4 ? ky : 0], __builtin_bit_cast(tt4, op[u]), accx[u]);
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (u < 3) accx[u] = smx_mma4(A1[ky][u < 3 ? u : 0], __builtin_bit_cast(tt4, w2[u]), accx[u]);
                    else if (has4) accx[u] = smx_mma4(A1x[NU == 4 ? ky : 0], __builtin_bit_cast(tt4, w2[u]), accx[u]);
            }
            smx_i32x2 hh[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float s1 = s1v[u], b1 = b1v[u];
                float sum = 0.f;
                t4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = accx[u][e] * s1 + b1;
                    v = v * sigmoid_t<T>(v);
                    sum += (e == 3 && a.cv < 8 && (jq & 1)) ? 0.f : v;          // (this lane's pixels 4 jq .. 4 jq + 3: the eighth pixel of a walk row is not part of the map)
                    hv[e] = cvt(v);
                }
                sum += smx_dppf<0xB1>(sum);                                 // the 4 quads of the segment (fixed order)
                sum += smx_dppf<0x4E>(sum);
                if (jq == 0 && (u < 3 || has4)) red[useg(u) * CC + utile(u) * 16 + cb] = sum;
                hh[u] = __builtin_bit_cast(smx_i32x2, hv);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const smx_i32x2 back = smx_i32x2{__builtin_amdgcn_ds_bpermute(bp_out, hh[u][0]), __builtin_amdgcn_ds_bpermute(bp_out, hh[u][1])};
                const f32x4 tr = smx_mma16(__builtin_bit_cast(t4, back), ident, f32x4{0.f, 0.f, 0.f, 0.f});     // -> lane (pixel, 4 channels), exact
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[u][e] = (T)tr[e];
            }
        }
        // Global stores count in vmcnt and retire in order with the loads: the wait for the next chunk's DMA comes BEFORE this chunk's stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!COSY_DBG(a.dbg & 1) && (prow & 7) < a.cv) {
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (u < 3 || has4)
                    *(out_t*)((T*)a.D + (((size_t)b * (a.Cmid >> 4) + ch * NI + utile(u)) * a.HW + walk_pix(useg(u), prow & 7)) * 16 + kg * 4) = yv[u];
        }
        __syncthreads();   // red complete; everybody's DMA(ch+1) landed; all Eh / parameter reads of this chunk are done
        if (tid < CC && !COSY_DBG(a.dbg & 1)) {
            float sm = ((red[tid] + red[CC + tid]) + red[2 * CC + tid]) + red[3 * CC + tid];
            if constexpr (NSEG == 5) sm += red[4 * CC + tid];
            a.partial[(size_t)b * a.Cmid + ch * CC + tid] = sm;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline uint16_t smx_f16_bits(float f) { _Float16 h = (_Float16)(f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f)); return __builtin_bit_cast(uint16_t, h); }
static inline uint16_t smx_bf16_bits(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
bool small_mx_transposed(int H, int W) { return H == 7 && W == 10; }      // the map's columns are the walk rows
static int smx_enabled() { static const int v = tune_int("COSY_SMALL_MX", 1); return v; }
bool small_mx_supported(int Cin, int Cmid, int k, int s, int dtype, int H, int W) {
    const int kbn = cdiv(Cin, 32);
    // 8x8 maps (256x256 crops); round 6: 7x10 (240x320: the ten columns walked as rows of 7 + 1 masked pixel) and 10x7 (320x240) maps: five segments, five waves
    const bool map_ok = (H == 8 && W == 8) || (H == 7 && W == 10) || (H == 10 && W == 7);
    return smx_enabled() && dtype != COSY_F32 && map_ok && s == 1 && (k == 3 || k == 5) && Cmid % 48 == 0 && (kbn == 8 || kbn == 12) && Cin % 8 == 0;
}
size_t small_mx_param_bytes(int Cmid, int k) { return (size_t)(Cmid / 48) * smx_pbytes(k); }
// b0l2e: log2(e) * BN0 bias (its scale is folded into the expand weights); dww: fp32 taps [k*k][Cmid]; s1 / b1: folded BatchNorm 1
void small_mx_pack_params(const float* b0l2e, const float* dww, const float* s1, const float* b1, int Cmid, int k, int dtype, void* dst, int transposed) {
    const int pb = smx_pbytes(k), lo = (k - 1) / 2;
    static const int off[2][4] = {{0, 1, 2, 3}, {-2, -1, 4, 5}};
    for (size_t i = 0; i < small_mx_param_bytes(Cmid, k); ++i) ((char*)dst)[i] = 0;
    for (int ch = 0; ch < Cmid / 48; ++ch) {
        char* d = (char*)dst + (size_t)ch * pb;
        float* h = (float*)d;
        for (int c = 0; c < 48; ++c) { h[c] = b0l2e[ch * 48 + c]; h[48 + c] = s1[ch * 48 + c]; h[96 + c] = b1[ch * 48 + c]; }
        uint16_t* fr = (uint16_t*)(d + SMX_HDR);
        for (int ni = 0; ni < 3; ++ni)
            for (int ky = 0; ky < k; ++ky)
                for (int m = 0; m < 2; ++m)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int q = 0; q < 4; ++q) {
                            const int i = lane & 3, cc = ch * 48 + ni * 16 + (lane >> 2), kx = off[m][q] - i + lo;
                            const float w = (kx >= 0 && kx < k) ? dww[(size_t)(transposed ? kx * k + ky : ky * k + kx) * Cmid + cc] : 0.f;      // transposed walk: tap row = the map's kx
                            fr[((((size_t)ni * k + ky) * 2 + m) * 64 + lane) * 4 + q] = smx_f16_bits(w);      // fp16 in both 16-bit modes (the tap MFMAs' operand type)
                        }
    }
}
void small_mx_kernel_name(int Cin, int k, int dtype, char* buf, size_t n) {
    snprintf(buf, n, "mbconv_small_mx_kernel<%s, %d, %d>", dtype == COSY_BF16 ? "__bf16" : "_Float16", k, cdiv(Cin, 32));
}

template <typename T, int KS, int KBN, int NSEG>
static int launch_smx_ks(const SmxKArgs& k, int B, hipStream_t s) {
    constexpr int HL = __is_same(T, bf16_t) ? 2 : 1;      // bf16: hi + lo weight fragments
    const size_t lds = (size_t)(NSEG + 2) * 3 * 512 + (size_t)3 * KBN * HL * 1024 + (size_t)2 * smx_pbytes(KS) + NSEG * 48 * sizeof(float);
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)mbconv_small_mx_kernel<T, KS, KBN, NSEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    COSY_CHECK_HIP(attr_rc);
    hipLaunchKernelGGL((mbconv_small_mx_kernel<T, KS, KBN, NSEG>), dim3((unsigned)(B * k.ncg)), dim3(256), lds, s, k);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
template <typename T, int KS, int KBN>
static int launch_smx_k(const SmxKArgs& k, int B, hipStream_t s) {
    return k.HW == 64 ? launch_smx_ks<T, KS, KBN, 4>(k, B, s) : launch_smx_ks<T, KS, KBN, 5>(k, B, s);
}
template <typename T>
static int launch_smx_t(const FuseArgs& a, hipStream_t s) {
    SmxKArgs k;
    k.X = a.X; k.Wp = a.Wp; k.params = (const char*)a.wparams; k.D = a.D; k.partial = a.partial; k.zeros = a.zeros;
    k.Cin = a.Cin; k.Cmid = a.Cmid; k.nkb_total = (cdiv(a.Cin, 32) + 1) & ~1;
    k.HW = a.H * a.W; k.Wm = a.W; k.tr = small_mx_transposed(a.H, a.W); k.cv = a.H * a.W == 64 ? 8 : 7;
    const int nchunks = a.Cmid / 48;
    static const int cpw_target = tune_int("COSY_SMALL_CPW", 15);
    k.cpw = nchunks <= cpw_target ? nchunks : cdiv(nchunks, cdiv(nchunks, cpw_target));
    k.ncg = cdiv(nchunks, k.cpw);
    k.dbg = tune_int("COSY_SMALL_DBG", 0);
    const int kbn = cdiv(a.Cin, 32);
    if (a.k == 3 && kbn == 8) return launch_smx_k<T, 3, 8>(k, a.B, s);
    if (a.k == 3 && kbn == 12) return launch_smx_k<T, 3, 12>(k, a.B, s);
    if (a.k == 5 && kbn == 8) return launch_smx_k<T, 5, 8>(k, a.B, s);
    if (a.k == 5 && kbn == 12) return launch_smx_k<T, 5, 12>(k, a.B, s);
    set_error("mbconv_small_mx: unsupported k=%d k-blocks=%d", a.k, kbn);
    return COSY_EINVAL;
}
// X (B,8,8,Cin); Wp: expand weights * s0 * log2(e) packed in 16-channel tiles (PwCfg{1,1}); wparams: small_mx_pack_params; D chunked; partial (B, 1, Cmid)
int launch_mbconv_small_mx(const FuseArgs& a, int dtype, hipStream_t s) {
    if (a.B == 0) return COSY_OK;
    COSY_REQUIRE(small_mx_supported(a.Cin, a.Cmid, a.k, a.s, dtype, a.H, a.W), "mbconv_small_mx: unsupported shape Cin=%d Cmid=%d %dx%d k=%d s=%d", a.Cin, a.Cmid, a.H, a.W, a.k, a.s);
    COSY_REQUIRE(a.wparams != nullptr, "mbconv_small_mx: packed parameters missing (small_mx_pack_params)%s", "");
    return dtype == COSY_BF16 ? launch_smx_t<bf16_t>(a, s) : launch_smx_t<f16_t>(a, s);
}

}  // namespace cosy
