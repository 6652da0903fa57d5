// Wave-autonomous fused MBConv front (bf16 / fp16 storage; fp32 for the parity mode):
//     expand 1x1 (MFMA) -> BN -> SiLU -> depthwise kxk -> BN -> SiLU -> D, squeeze sums
// with the expanded rows held in REGISTERS -- no LDS ring, no workgroup barrier.
// Reference: MBConvBlock.forward, cosypose/models/efficientnet.py:71-90.
//
// Why.  VALU micro-benchmark on MI355X (profiles/exp/valu_bench.hip): plain fp32 op 3.1, v_pk_fma_f32 5.9 (no gain),
// v_exp_f32 / v_rcp_f32 8.7 cycles per wave instruction -> BN + SiLU costs ~32 cycles per 64 elements and the fused fronts
// are VALU-bound; the LDS-ring kernels of rounds 1-2 (a tiled and a row-streaming front, removed in round 3) reached only ~35 % of that bound because every
// workgroup alternates barrier-separated expand / depthwise phases at 2-3 waves per SIMD.  Here every wave is independent:
//   * a wave owns (sample, 16*NI expanded channels) and walks down the rows of the whole map;
//   * pixel mapping of the MFMA B operand: lane (p = lane & 15) owns the PPL CONSECUTIVE pixels x = p*PPL .. p*PPL+PPL-1
//     of a row (one 16-pixel fragment per q = x - p*PPL), so after the expansion a lane holds 4 channels (kg = lane >> 4)
//     of PPL neighbouring pixels: horizontal taps are plain FMAs on the lane's own registers, only the run ends come
//     from the neighbouring lane -- one DPP row_shr / row_shl move per halo pixel (bound_ctrl:0 yields the zero padding
//     at the image border for free);
//   * BN parameters and depthwise taps sit in a wave-private LDS block (broadcast ds_read_b128, 4 distinct addresses);
//   * depthwise accumulation is INPUT-stationary: an expanded row lives in registers only while it is scattered into the
//     ceil(KS/S) output rows it feeds (accumulators of the rows in flight), so no ring of expanded rows is kept; the row
//     loop is unrolled S*ceil(KS/S) times so that every accumulator slot is a compile-time constant;
//   * the block input is read as MFMA fragments straight from global memory one row ahead; per row the global memory
//     operations are issued right after the expansion, ordered [previous output row's stores] -> [next input row's loads]
//     (vmcnt retires in order and the compiler waits with vmcnt(0)), so the wait in front of the
//     next row's MFMAs only sees operations that are a whole depthwise phase old;
//   * squeeze sums: per-lane registers over the whole image, one DPP tree per 16-lane row at the end (fixed order).
// The 16*NI-channel chunks of one sample re-read the block input (x Cmid/16/NI through L2; it is the small tensor).
#include "net_device.h"
#include <type_traits>
#include <utility>

#ifdef COSY_KO_TAPS          // knock-out build (timing only): every tap of a row reads the same quad, i.e. one LDS read per row
#define COSY_KO_TAP_INDEX(i) 0
#else
#define COSY_KO_TAP_INDEX(i) (i)
#endif

namespace cosy {

struct WaveKArgs {
    const void* X; const void* Wp; const float* wparams;     // wparams: wave_pack_params()
    void* D; float* partial; const void* zeros;
    int B, H, W, Cin, Cmid, Ho, Wo, nkb_total, nchunks, rsplit, rows_per, dbg;   // rsplit row bands per (sample, chunk), rows_per output rows each
    // H / W / Ho / Wo are the WALKED axes: the wave walks H "rows" of W pixels.  For a transposed job (host: wave_plan) the rows are the
    // map's columns; only the four strides below and the order of the taps know: the map is addressed through them.
    int xs_pix, xs_row;      // bytes between two neighbouring pixels of a row / between two rows of the block input
    // channel addressing of the block input: channel k of a pixel lives (k >> 4) * xs_chunk + (k & 15) * sizeof(T) bytes behind the pixel's first byte.
    // NHWC rows: xs_chunk = 16 * sizeof(T) (i.e. k * sizeof(T)); chunked input [sample][ceil(Cin/16)][H*W][16] (FuseArgs::x_chunked): xs_chunk = H * W * 16 * sizeof(T).
    int xs_chunk;
    long xs_sample;          // elements between two samples of the block input
    int x_perm;              // fp32-FMA form: the pixels of an input row are stored permuted (FuseArgs::x_perm): the lanes' pixels p * PPL + q of fragment q are 16 neighbours
    int ds_pix, ds_row;      // elements between two neighbouring pixels of a row / between two rows inside a D chunk
    // -DCOSY_TUNE only (null in the shipping library): s_memtime stamps of one job in every `stamp_stride`-th workgroup (wave 0), see WAVE_STAMP
    unsigned long long* stamps; int stamp_stride, stamp_slots;
    int wpb;                 // jobs (waves) per workgroup
};
// Timeline of a job (-DCOSY_WAVE_STAMPS build only): the shader clock (s_memtime) at the job's start, behind its prologue, at the start of every
// input row (scalar bookkeeping: sum / min / max of the row-to-row intervals) and at its end -> 8 words per recorded job.  A first version with
// five LDS-parked stamps per row pinned hipcc's schedule so hard that the kernels ran 3-5x slower and spilled: a timeline of a different kernel.

template <int CTRL> __device__ __forceinline__ float dpp_mov0(float v) {   // lanes without a source read 0 (bound_ctrl:0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int N> __device__ __forceinline__ float row_shr0(float v) { return dpp_mov0<0x110 + N>(v); }   // lane i <- lane i-N of its 16-lane row
template <int N> __device__ __forceinline__ float row_shl0(float v) { return dpp_mov0<0x100 + N>(v); }   // lane i <- lane i+N

// SiLU of four values as ONE hand-scheduled block: y = x / (1 + 2^-(x * log2 e)); SCALED: x arrives pre-multiplied by log2(e)
// and y = x / (1 + 2^-x) (= log2(e) * silu).  The four dependency chains are interleaved so that every transcendental
// result is consumed three instructions after it was issued.  Hand-written because hipcc's own schedule of the same
// arithmetic was observed to be NOT run-to-run deterministic under load on gfx950 (blocks 6/7 variant, 256 crops: a few
// outputs per launch one bf16 ulp off, gone with any change of the schedule or with wait states behind v_exp / v_rcp --
// the signature of a transcendental result read before it has landed); inside an asm block nothing is re-scheduled.
template <bool SCALED> __device__ __forceinline__ void silu4(float* v) {
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
// the same block WITHOUT volatile (a pure function of its operands): the software-pipelined rows let hipcc place it between the ordered fragment MFMAs
template <bool SCALED> __device__ __forceinline__ void silu4p(float* v) {
    float t0, t1, t2, t3;
    if constexpr (SCALED) {
        asm(
            "v_exp_f32 %4, -%0\n v_exp_f32 %5, -%1\n v_exp_f32 %6, -%2\n v_exp_f32 %7, -%3\n"
            "v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n"
            "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
            "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7\n s_nop 0"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
    } else {
        asm(
            "v_mul_f32 %4, 0xbfb8aa3b, %0\n v_mul_f32 %5, 0xbfb8aa3b, %1\n v_mul_f32 %6, 0xbfb8aa3b, %2\n v_mul_f32 %7, 0xbfb8aa3b, %3\n"
            "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
            "v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n"
            "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
            "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7\n s_nop 0"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
    }
}


// Input fragment i of the wave kernel is loaded into a RESERVED register quad at the top of the kernel's VGPR budget,
// v[TOP-4(i+1) : TOP-4i-1], and copied into ordinary registers behind the wait.  The compiler never sees a value whose load
// is still in flight (given one, it copies it at the merge points of the row loop -- garbage).  The reservation is not
// something hipcc can be told: the kernel only forces the allocation up to TOP with a clobber, and relies on the register
// allocator filling from v0 upwards; profiles/check_wave_isa.py verifies on the built ISA that no compiler-generated
// instruction touches the reserved range (tests/test_build_isa.py).  (AGPRs would be the natural home, but naming one in
// inline asm makes hipcc split the budget 128 + 128 and spill through v_accvgpr_write.)
template <int TOP, int I, int IMM = 0> __device__ __forceinline__ void xfrag_load(int voff, const char* sbase) {
    asm volatile("global_load_dwordx4 v[%2:%3], %0, %1 offset:%4 ; XLOAD" :: "v"(voff), "s"(sbase), "n"(TOP - 4 * (I + 1)), "n"(TOP - 4 * I - 1), "n"(IMM) : "memory");
}
template <int TOP, int I> __device__ __forceinline__ f32x4 xfrag_read() {
    float x0, x1, x2, x3;
    asm volatile("v_mov_b32 %0, v[%4]\n v_mov_b32 %1, v[%5]\n v_mov_b32 %2, v[%6]\n v_mov_b32 %3, v[%7] ; XREAD"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3)
                 : "n"(TOP - 4 * (I + 1)), "n"(TOP - 4 * (I + 1) + 1), "n"(TOP - 4 * (I + 1) + 2), "n"(TOP - 4 * (I + 1) + 3));
    return f32x4{x0, x1, x2, x3};
}
// 16-bit types: the expansion's MFMAs read the fragments WHERE THEY LANDED -- inline-asm v_mfma with the reserved quad as one operand -- so no copy
// (4 v_mov_b32 and 4 registers per fragment and row) is made at all.  XA: the fragment is the A operand (pixels as the MFMA's rows: the form with
// the taps on the matrix pipe), else the B operand.  FIRST: SrcC = 0.  hipcc does not look inside the asm: a chain of these is issued back to back
// (dependent MFMAs on one accumulator need no wait states) and closed by xfrag_mma_done(), the wait states a VALU read of the last result needs
// (hipcc's own code puts s_nop 7 there; two more for margin).  fp32 (parity mode) keeps the copies: its 16x16x4 MFMAs take single registers.
#ifndef COSY_WAVE_XASM
#define COSY_WAVE_XASM 1
#endif
#ifndef COSY_WAVE_PIPE
#define COSY_WAVE_PIPE 1
#endif
template <typename T, int TOP, int I, bool FIRST, bool XA, typename W> __device__ __forceinline__ void xfrag_mma(f32x4& m, const W& w) {
    constexpr int lo = TOP - 4 * (I + 1), hi = TOP - 4 * I - 1;
    if constexpr (__is_same(T, f16_t)) {
        if constexpr (FIRST && XA) asm volatile("s_nop 1\n v_mfma_f32_16x16x32_f16 %0, v[%2:%3], %1, 0 ; XMMA" : "=&v"(m) : "v"(w), "n"(lo), "n"(hi));
        else if constexpr (FIRST) asm volatile("s_nop 1\n v_mfma_f32_16x16x32_f16 %0, %1, v[%2:%3], 0 ; XMMA" : "=&v"(m) : "v"(w), "n"(lo), "n"(hi));
        else if constexpr (XA) asm volatile("v_mfma_f32_16x16x32_f16 %0, v[%2:%3], %1, %0 ; XMMA" : "+v"(m) : "v"(w), "n"(lo), "n"(hi));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, v[%2:%3], %0 ; XMMA" : "+v"(m) : "v"(w), "n"(lo), "n"(hi));
    } else {
        if constexpr (FIRST && XA) asm volatile("s_nop 1\n v_mfma_f32_16x16x32_bf16 %0, v[%2:%3], %1, 0 ; XMMA" : "=&v"(m) : "v"(w), "n"(lo), "n"(hi));
        else if constexpr (FIRST) asm volatile("s_nop 1\n v_mfma_f32_16x16x32_bf16 %0, %1, v[%2:%3], 0 ; XMMA" : "=&v"(m) : "v"(w), "n"(lo), "n"(hi));
        else if constexpr (XA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, v[%2:%3], %1, %0 ; XMMA" : "+v"(m) : "v"(w), "n"(lo), "n"(hi));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, v[%2:%3], %0 ; XMMA" : "+v"(m) : "v"(w), "n"(lo), "n"(hi));
    }
}
__device__ __forceinline__ void xfrag_mma_done(f32x4& m) { asm volatile("s_nop 7\n s_nop 1 ; XMMA" : "+v"(m)); }
// Fence: an empty asm that CLOBBERS the reserved fragment registers.  No value that is live across it can be allocated there, so
// placing one at every point of the row loop where fragments are (or are about to be) in flight -- after the loads are issued,
// in front of the wait -- keeps every long-lived value of the kernel out of the range by construction; what is left to luck
// (and to profiles/check_wave_isa.py) are temporaries that live entirely between two fences.
template <int TOP, int NFRAG> __device__ __forceinline__ void xfrag_fence() {
#include "wave_fence.inc"      // generated from the variant tables below (cosypose_amd/wave_isa.py): one clobber list per (TOP, NFRAG) in use
}
template <int MINW> __device__ __forceinline__ void xfrag_reserve() {
    if constexpr (MINW == 2) asm volatile("; XRESERVE" ::: "v255");
    else if constexpr (MINW == 3) asm volatile("; XRESERVE" ::: "v167");
    else if constexpr (MINW == 5) asm volatile("; XRESERVE" ::: "v95");
    else asm volatile("; XRESERVE" ::: "v127");
}

// ---- depthwise taps on the matrix pipe (16-bit types, stride 1, 16-pixel rows: blocks 9-17 at 256x256 crops).
// v_mfma_f32_4x4x4_16B computes 16 independent 4x4x4 products, D_b[i][j] = sum_k A_b[i][k] * B_b[k][j], block b in lanes 4b .. 4b+3 (lane 4b+i holds
// row i of A, lane 4b+j holds column j of B and of D; checked on the device by profiles/exp/mfma_depthwise.hip).  With block = channel, j = a quad of
// 4 neighbouring pixels of the row, A = the Toeplitz matrix of one tap row (A[i][k] = w[ky][x_k - x_i + LO]) and B = the expanded values of the
// quad (W1) or of the two pixels either side of it (W2), one instruction applies a tap row to 16 pixels x 16 channels: 2 KS instructions per input row
// instead of KS * KS * 4 fp32 FMAs + 4 (KS - 1) DPP moves + KS * KS LDS reads (micro-benchmark of the tap phase alone, same file: 1.5x with its
// operands staged through LDS).  What it takes:
//   * the expansion runs with its operands SWAPPED (pixels as the MFMA's rows), so a lane leaves it with 4 neighbouring pixels of ONE channel
//     (lane = channel + 16 * quad) -- the operand format of the small MFMA up to a lane permutation (lane -> 4 * channel + quad: one ds_bpermute per
//     packed register); BatchNorm parameters are one scalar per lane;
//   * the expanded values and the taps are rounded to the storage type (what the unfused kernels store for E; the oracle's emulation follows:
//     block_info kind 5), products are exact in fp32 and accumulate in fp32;
//   * the halo operand W2 = (upper half of the previous quad, lower half of the next one): two quad_perm DPP moves, zeroed at the row ends;
//   * a finished output row (lane = (channel, quad): 4 pixels) goes through BatchNorm + SiLU + the squeeze sums where it is, is rounded, permuted
//     back and TRANSPOSED by one v_mfma_f32_16x16x16 against the identity (A lane (row r, 4 k) -> C lane (column k, 4 rows): exact), which leaves
//     lane = (pixel, 4 channels): the 8-byte D store of the other form.
// Rows wider than 16 pixels (PPL > 1) are PPL SEGMENTS of 16 pixels: fragment q holds the pixels 16 q .. 16 q + 15 (the other form gives a lane PPL
// neighbouring pixels instead), every segment goes through the steps above on its own, and the halo operand of a segment's first / last quad comes
// from the neighbouring segment's last / first quad (one more quad_perm move and a select) instead of the zero padding.
// The chunk's parameters arrive as [s0][b0][s1][b1] (16 floats each) + 2 KS ready-made A fragments (wave_pack_params); no LDS parameter block.
#ifndef COSY_WAVE_MX
#define COSY_WAVE_MX 1
#endif
// Rows that do not fill their last 16-pixel segment (round 6; 240x320 crops: the 15- and 30-pixel columns of the 15x20 / 30x40 maps): the pixels beyond the row end
// are expanded from a clamped address, forced to zero before they become tap operands (= the zero padding their neighbours need), left out of the squeeze sums and
// not stored.  -DCOSY_WAVE_MXP=0: such rows keep the fp32-FMA form.
#ifndef COSY_WAVE_MXP
#define COSY_WAVE_MXP 1
#endif
constexpr bool wave_mx(int esz, int ks, int s, int ppl, bool fullw) { return COSY_WAVE_MX && esz == 2 && s == 1 && ppl >= 1 && (fullw || COSY_WAVE_MXP) && (ks == 3 || ks == 5); }
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mma4(f16x4 a, f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma4(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(f16x4 a, f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma4(f32x4, f32x4, f32x4 c) { return c; }      // never instantiated for fp32 (wave_mx), only parsed
__device__ __forceinline__ f32x4 mma16(f32x4, f32x4, f32x4 c) { return c; }


// ---- the fused interior row of the matrix-pipe form (round 6): every instruction of the row is issued from `asm volatile` statements in ONE hand-chosen order
// (volatile statements keep their program order; hipcc only allocates the registers and resolves the sub-registers of the accumulator tuples), so nothing of the row
// is left to hipcc's scheduler: it kept the two dependent chains of a software-pipelined row apart (profiles/r05_wave_occupancy.txt).  Wait states and counted waits
// are written out by hand (hipcc pads nothing inside or between asm statements): see the hazard notes at row_px below.
template <typename T> struct PxT;
template <> struct PxT<f16_t> { static constexpr bool F16 = true; };
template <> struct PxT<bf16_t> { static constexpr bool F16 = false; };
template <> struct PxT<float> { static constexpr bool F16 = false; };
template <typename T, typename V4> __device__ __forceinline__ void px_tap(f32x4& acc, const V4& A, const V4& W) {     // acc += A * W (16 channel blocks of 4x4x4)
    if constexpr (PxT<T>::F16) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0 ; PX" : "+v"(acc) : "v"(A), "v"(W));
    else asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0 ; PX" : "+v"(acc) : "v"(A), "v"(W));
}
template <typename T, typename V4> __device__ __forceinline__ void px_tap0(f32x4& acc, const V4& A, const V4& W) {    // acc = A * W: the first tap row of a new output row
    if constexpr (PxT<T>::F16) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, 0 ; PX" : "=&v"(acc) : "v"(A), "v"(W));
    else asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, 0 ; PX" : "=&v"(acc) : "v"(A), "v"(W));
}
// expansion MFMA on the reserved fragment quad I (pixels as the MFMA's rows), no wait states of its own
template <typename T, int TOP, int I, bool FIRST, typename W> __device__ __forceinline__ void px_mma(f32x4& m, const W& w) {
    constexpr int lo = TOP - 4 * (I + 1), hi = TOP - 4 * I - 1;
    if constexpr (PxT<T>::F16) {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, v[%2:%3], %1, 0 ; XMMA" : "=&v"(m) : "v"(w), "n"(lo), "n"(hi));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, v[%2:%3], %1, %0 ; XMMA" : "+v"(m) : "v"(w), "n"(lo), "n"(hi));
    } else {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, v[%2:%3], %1, 0 ; XMMA" : "=&v"(m) : "v"(w), "n"(lo), "n"(hi));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, v[%2:%3], %1, %0 ; XMMA" : "+v"(m) : "v"(w), "n"(lo), "n"(hi));
    }
}
template <int OFF, typename R> __device__ __forceinline__ void px_wread(R& w, unsigned addr) {      // parked weight fragment -> registers (in flight behind the statement)
    asm volatile("ds_read_b128 %0, %1 offset:%2 ; PX" : "=v"(w) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void px_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0) ; PX" :: "n"(N)); }
template <int N> __device__ __forceinline__ void px_nop() { asm volatile("s_nop %0 ; PX" :: "n"(N)); }
// y[e] = s * m[e] + b   (the BatchNorm of an MFMA result: reads the tuple's sub-registers)
__device__ __forceinline__ void px_bn(float* y, const f32x4& m, float s, float b) {
    asm volatile("v_fma_f32 %0, %8, %4, %9\n v_fma_f32 %1, %8, %5, %9\n v_fma_f32 %2, %8, %6, %9\n v_fma_f32 %3, %8, %7, %9 ; PX"
                 : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(s), "v"(b));
}
// SiLU in five statements (the arithmetic of silu4<false>, value for value): t = -log2(e) * y | t = 2^t | t = 1 + t | t = 1 / t | y = y * t
__device__ __forceinline__ void px_silu_a(const float* y, float* t) {
    asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %4\n v_mul_f32 %1, 0xbfb8aa3b, %5\n v_mul_f32 %2, 0xbfb8aa3b, %6\n v_mul_f32 %3, 0xbfb8aa3b, %7 ; PX"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
}
__device__ __forceinline__ void px_silu_b(float* t) {
    asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3 ; PX" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
}
__device__ __forceinline__ void px_silu_c(float* t) {
    asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3 ; PX" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
}
__device__ __forceinline__ void px_silu_d(float* t) {
    asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3 ; PX" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
}
__device__ __forceinline__ void px_silu_e(float* y, const float* t) {
    asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7 ; PX"
                 : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]) : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
}
__device__ __forceinline__ void px_sum(float& s, const float* y) {      // ((((s + y0) + y1) + y2) + y3): the order of the unfused form
    asm volatile("v_add_f32 %0, %1, %0\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %4, %0 ; PX" : "+v"(s) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
}
// round four values to the storage type: (y0, y1) -> lo, (y2, y3) -> hi; fp16 saturates (v_med3 against +-65504, as cvt())
template <typename T> __device__ __forceinline__ void px_cvt(int& lo, int& hi, float* y, float nclamp, float pclamp) {
    if constexpr (PxT<T>::F16)
        asm volatile("v_med3_f32 %2, %2, %6, %7\n v_med3_f32 %3, %3, %6, %7\n v_med3_f32 %4, %4, %6, %7\n v_med3_f32 %5, %5, %6, %7\n"
                     "v_cvt_pk_f16_f32 %1, %4, %5\n v_cvt_pk_f16_f32 %0, %2, %3 ; PX"
                     : "=&v"(lo), "=&v"(hi), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]) : "s"(nclamp), "v"(pclamp));
    else
        asm volatile("v_cvt_pk_bf16_f32 %1, %4, %5\n v_cvt_pk_bf16_f32 %0, %2, %3 ; PX" : "=&v"(lo), "=&v"(hi) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
}
__device__ __forceinline__ void px_bperm2(int& lo, int& hi, int addr, int a, int b) {      // both results in flight behind the statement
    asm volatile("ds_bpermute_b32 %0, %2, %3\n ds_bpermute_b32 %1, %2, %4 ; PX" : "=&v"(lo), "=&v"(hi) : "v"(addr), "v"(a), "v"(b));
}
// the halo operand of a 16-pixel row's tap MFMAs: (previous quad's pixels 2, 3 | next quad's pixels 0, 1), zero at the row ends
__device__ __forceinline__ void px_halo(int& hp, int& ln, int lo, int hi, unsigned long long first_quad, unsigned long long last_quad) {
    asm volatile("v_mov_b32_dpp %0, %3 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_e64 %0, %0, 0, %4\n v_cndmask_b32_e64 %1, %1, 0, %5 ; PX"
                 : "=&v"(hp), "=&v"(ln) : "v"(lo), "v"(hi), "s"(first_quad), "s"(last_quad));
}
template <typename T, typename V4> __device__ __forceinline__ void px_transpose(f32x4& tr, const V4& a, const V4& ident) {
    if constexpr (PxT<T>::F16) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, 0 ; PX" : "=&v"(tr) : "v"(a), "v"(ident));
    else asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, 0 ; PX" : "=&v"(tr) : "v"(a), "v"(ident));
}
template <typename T> __device__ __forceinline__ void px_cvt_store(const f32x4& tr, unsigned voff, const void* srow) {      // exact values: plain converts, 8-byte store
    int o0, o1;
    if constexpr (PxT<T>::F16)
        asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n v_cvt_pk_f16_f32 %1, %4, %5\n" : "=&v"(o0), "=&v"(o1) : "v"(tr[0]), "v"(tr[1]), "v"(tr[2]), "v"(tr[3]));
    else
        asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n v_cvt_pk_bf16_f32 %1, %4, %5\n" : "=&v"(o0), "=&v"(o1) : "v"(tr[0]), "v"(tr[1]), "v"(tr[2]), "v"(tr[3]));
    typedef int i2_t __attribute__((ext_vector_type(2)));
    const i2_t d = i2_t{o0, o1};
    asm volatile("global_store_dwordx2 %0, %1, %2 ; PX" :: "v"(voff), "v"(d), "s"(srow) : "memory");
}

template <typename F, int... Us>
__device__ __forceinline__ void unroll_seq(F&& f, std::integer_sequence<int, Us...>) { (f(std::integral_constant<int, Us>{}), ...); }

constexpr int wave_lcm(int a, int b) { int x = a; while (x % b) x += a; return x; }
// shapes whose straight-line interior row does not fit the register budget (the scheduler hoists every tap read: spills, and
// temporaries in the reserved fragment range -- profiles/check_wave_isa.py) keep the branchy boundary form for every row
constexpr bool wave_interior_form(int ks, int s, int kbn, int ppl, int minw) { return !(ks == 3 && s == 1 && kbn == 1 && ppl == 4 && minw == 3); }
#ifndef COSY_WAVE_WLDS
#define COSY_WAVE_WLDS 1
#endif
// (the argument counts weight FRAGMENTS per pixel fragment: k-blocks, x 2 for bf16's hi + lo pairs; 4 fragments at three waves per SIMD only exist there --
// blocks 6 / 7 in bf16 spill with 16 weight registers)
constexpr bool wave_wlds(int kbn, int minw) { return (COSY_WAVE_WLDS && kbn >= 5 && minw >= 4) || kbn >= 9 || (kbn == 4 && minw == 3); }

template <typename T, int KS, int S, int KBN, int PPL, int NI, bool FULLW, int MINW>
__global__ __launch_bounds__(256, MINW) void mbconv_wave_kernel(WaveKArgs a) {
    using raw_t = typename DT<T>::raw_t;
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int NCH = 4 * NI;                               // channels per lane
    constexpr int LO = S == 1 ? (KS - 1) / 2 : (KS - 2) / 2;  // static "same" padding, both axes
    constexpr int HI = KS - S - LO;                           // right halo pixels a lane needs
    constexpr int TO = PPL / S;                               // output pixels per lane and row
    constexpr int RP = LO + PPL + HI;                         // pixels of an expanded row a lane sees: [left halo | own | right halo]
    constexpr int NOPEN = (KS + S - 1) / S;                   // output rows that are accumulating at the same time
    constexpr int U = S * NOPEN;                              // input rows per unrolled super-iteration
    constexpr int PF = (4 + KS * KS) * 16 * NI;               // floats of the parameter block
    constexpr int HL = __is_same(T, bf16_t) ? 2 : 1;          // bf16: the expand weights are hi + lo pairs (kernels_net.hip: pw_hl), two fragments per k-block
    constexpr int NF = KBN * HL;                              // weight fragments (= MFMAs of the expansion chain) per pixel fragment
    constexpr bool WLDS = wave_wlds(NF, MINW);                // expand-weight fragments parked in LDS instead of registers
    constexpr bool MX = wave_mx(sizeof(T), KS, S, PPL, FULLW);  // depthwise taps on the matrix pipe
    constexpr int PFX = 64 + 2 * KS * 128;                    // floats of the parameter block of that form
    constexpr int PFW = (MX ? 0 : PF) + (WLDS ? NI * NF * 256 : 0);
    static_assert(PPL % S == 0, "a lane's pixel run must hold whole output pixels");
    typedef T out_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, kg = lane >> 4;
    // XCD-aware job order: the chunks of one sample run on one XCD (block id % 8), 4 jobs (waves) per block
    const int id = blockIdx.x, xcd = id & 7, sidx = (id >> 3) * a.wpb + wave;
    // a job = (sample, chunk, row band); the jobs of one sample stay on one XCD
    const int jps = a.nchunks * a.rsplit;
    const int b = (sidx / jps) * 8 + xcd, jrem = sidx % jps;
    int ch = jrem / a.rsplit, band = jrem - ch * a.rsplit;
    if (COSY_DBG(a.dbg & 32)) { band = jrem / a.nchunks; ch = jrem - band * a.nchunks; }      // dbg 32: chunk-fastest order (the 4 waves of a workgroup share a band)
    const bool active = b < a.B;
    const int c0 = ch * 16 * NI;

    float* P = smem_w + wave * PFW;
    char* Wl = (char*)(P + (MX ? 0 : PF));
#ifdef COSY_WAVE_STAMPS
    // timeline build (python -m cosypose_amd.build --tune with COSY_WAVE_STAMPS=1 in the environment): scalar-only bookkeeping, no LDS, no VGPRs
    const bool stamping = a.stamps != nullptr && wave == 0 && blockIdx.x % a.stamp_stride == 0 && (int)(blockIdx.x / a.stamp_stride) < a.stamp_slots;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long st_t1 = 0, st_first = 0;
    unsigned st_prev = 0, st_sum = 0, st_min = 0xffffffffu, st_max = 0, st_n = 0;
#define WAVE_STAMP_ROW()                                                                  \
    do {                                                                                  \
        const unsigned long long t64_ = __builtin_amdgcn_s_memtime();                     \
        const unsigned t_ = (unsigned)t64_;                                               \
        const unsigned d_ = t_ - st_prev;                                                 \
        if (st_n == 0) st_first = t64_;                                                   \
        else { st_sum += d_; st_min = min(st_min, d_); st_max = max(st_max, d_); }        \
        st_prev = t_; ++st_n;                                                             \
    } while (0)
#else
#define WAVE_STAMP_ROW() do { } while (0)
#endif

    const T* __restrict__ X = (const T*)a.X + (size_t)min(b, a.B - 1) * (size_t)a.xs_sample;
    const float* Pl = P + kg * 4;            // this lane's channel quad inside every 16-float group
    const float* taps = Pl + 4 * 16 * NI;

    // Global addressing without per-row multiplies: a row's address = uniform row base + a per-lane byte offset that never
    // changes.  Lanes outside the row / channel range read a valid neighbour instead of a zero page: pixels beyond the row
    // end are forced to zero after the expansion anyway, and the channel tail of the last k-block meets the zero padding of the
    // packed weights (finite * 0).
    const size_t xrow_bytes = (size_t)a.xs_row;
    // Full rows (FULLW: no pixel is clamped): fragment q's lane offsets are fragment 0's plus a wave-uniform step -- added to the row's scalar base --
    // and k-block kb's are k-block 0's plus 64 kb bytes (the instruction's immediate) except for the last k-block, whose channel tail is clamped:
    // two offset registers instead of PPL * KBN.
    constexpr bool XUNI = FULLW;
    const bool seg_addr = MX || a.x_perm;      // fragment q = 16 neighbouring stored pixels, 16 pixels behind fragment q - 1
    const int qstep = (seg_addr ? 16 : 1) * a.xs_pix;
    int xoff[XUNI ? 1 : PPL][XUNI ? 2 : KBN];
    auto koff = [&](int k) -> int { return (k >> 4) * a.xs_chunk + (k & 15) * (int)sizeof(T); };      // byte offset of channel k inside a pixel (NHWC: k * sizeof(T))
    const int kbstep = (KB >> 4) * a.xs_chunk;     // bytes from k-block to k-block (wave-uniform: added to the row's scalar base; a k-block is whole 16-channel chunks)
    static_assert(KB % 16 == 0 && (EPL == 4 || EPL == 8), "a lane's EPL channels lie inside one 16-channel chunk");
    if constexpr (XUNI) {
        const int px = (seg_addr ? p : p * PPL) * a.xs_pix;
        xoff[0][0] = px + koff(kg * EPL);
        xoff[0][1] = px + koff(min((KBN - 1) * KB + kg * EPL, a.Cin - EPL));
    } else {
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb) {
            const int k = kb * KB + kg * EPL;
#pragma unroll
            for (int q = 0; q < PPL; ++q) xoff[q][kb] = min(MX ? 16 * q + p : p * PPL + q, a.W - 1) * a.xs_pix + koff(min(k, a.Cin - EPL));      // (MX: fragment q = the segment's 16 pixels)
        }
    }
    // The input fragments are loaded by inline asm and waited for with a COUNTED s_waitcnt: vmcnt retires loads and stores
    // in order, and hipcc, which cannot count stores across this loop's control flow, waits with vmcnt(0) -- i.e. for the
    // acknowledgement of the output row stored a moment ago as well (knock-out timing: 30 % of block 3's time).  Here the
    // next row's loads are issued FIRST and the previous output row's stores behind them, so vmcnt(#stores) in front of
    // the next row's MFMAs covers the loads and leaves the stores in flight.
    constexpr int XTOP = MINW == 2 ? 256 : MINW == 3 ? 168 : MINW == 4 ? 128 : 96;
    static_assert(MINW >= 2 && MINW <= 5);
    xfrag_reserve<MINW>();
    constexpr bool XASM = COSY_WAVE_XASM && sizeof(T) == 2;   // the MFMAs read the fragments in place (xfrag_mma)
    constexpr bool PIPE = COSY_WAVE_PIPE && MX && XASM && PPL == 1 && FULLW;       // matrix-pipe form: row iy's expansion side and row iy - 1's tap / output side in one iteration (row_mx)
    raw_t wf[NI][NF];                        // the chunk's expand-weight fragments, [k-block][hi | lo] (WLDS: parked in LDS instead)
    f32x4 xc[XASM ? 1 : PPL][XASM ? 1 : KBN];
    int st_in_flight = 0;                    // stores issued behind the newest loads (wave-uniform)
    auto load_row = [&](int iy) {            // B fragments of input row iy: fragment q holds the lanes' pixels p*PPL + q
        const char* rowp = (const char*)X + (size_t)min(iy, a.H - 1) * xrow_bytes;    // rows below the map are never used
        unroll_seq([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (XUNI) {
                constexpr int q = i / KBN, kb = i % KBN;
                if constexpr (kb == KBN - 1) xfrag_load<XTOP, i>(xoff[0][1], rowp + (size_t)q * qstep);
                else xfrag_load<XTOP, i>(xoff[0][0], rowp + (size_t)q * qstep + (size_t)kb * kbstep);
            } else xfrag_load<XTOP, i>(xoff[i / KBN][i % KBN], rowp);
        }, std::make_integer_sequence<int, PPL * KBN>{});
        xfrag_fence<XTOP, PPL * KBN>();
        st_in_flight = 0;
    };
    // XASM: the expansion's MFMA chains are issued HERE, in the basic block of the wait (rows below the map expand a valid row for nothing: load_row
    // clamps) -- one chain per pixel fragment into mq[], all chains first (asm blocks keep their order), then one set of wait states
    auto wait_row = [&](f32x4* mq) {
        constexpr int NST = (PPL / S) * NI;
        xfrag_fence<XTOP, PPL * KBN>();
        if (st_in_flight) asm volatile("s_waitcnt vmcnt(%0) ; XWAIT" :: "n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) ; XWAIT" ::: "memory");
        if constexpr (XASM) {
            unroll_seq([&](auto ic) {
                constexpr int i = decltype(ic)::value / HL, h = decltype(ic)::value % HL, q = i / KBN, kb = i % KBN, f = kb * HL + h;
                if constexpr (WLDS) xfrag_mma<T, XTOP, i, f == 0, MX>(mq[q], *(const raw_t*)(Wl + f * 1024 + lane * 16));
                else xfrag_mma<T, XTOP, i, f == 0, MX>(mq[q], wf[0][f]);
            }, std::make_integer_sequence<int, PPL * KBN * HL>{});
            xfrag_mma_done(mq[PPL - 1]);
        }
        if constexpr (!XASM) {
            unroll_seq([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                xc[i / KBN][i % KBN] = xfrag_read<XTOP, i>();
            }, std::make_integer_sequence<int, PPL * KBN>{});
            asm volatile("s_nop 1");             // VALU write -> MFMA read
        }
    };
    // this job's output rows [oy_a, oy_b) and the input rows they need (rows above a band are recomputed, KS-S of them)
    const int oy_a = band * a.rows_per, oy_b = min(a.Ho, oy_a + a.rows_per);
    const int iy_first = max(0, oy_a * S - LO), iy_last = (oy_b - 1) * S - LO + KS - 1;

    // ---- everything a job needs from memory before its first row, issued back to back under ONE latency: the first input
    // row (asm loads, oldest), the chunk's expand-weight fragments, the parameter block (staged to LDS: its wait is a vmcnt(0)
    // that covers all three).  Issued one after the other they cost three memory latencies per job -- 15 % of a 16-row job.
    // The chunk's parameters arrive PACKED (host: wave_pack_params): one contiguous block [s0*log2e][b0*log2e][s1][b1][taps*ln2 in
    // walked order] of PF floats per chunk, so a lane fetches its one or two 16-byte pieces without address arithmetic.  (The
    // expansion's SiLU runs on t = log2(e) * v: t / (1 + 2^-t) = log2(e) * silu(v) -- one multiply fewer per expanded element,
    // v_exp_f32 takes the negation as a source modifier; the inverse factor rides in the taps, the only consumers of the
    // expanded values.)  Round 3: these loads used to be issued BEHIND the wait for the weight fragments, one pass of a rolled
    // loop at a time -- three serial memory latencies per job where the comment above promised one.
    constexpr int NPL = MX ? 1 : (PF / 4 + 63) / 64;
    f32x4 pv[NPL];
    typedef T t4 __attribute__((ext_vector_type(4)));
    // operands of the tap MFMAs (matrix-pipe form): fp16 in BOTH 16-bit modes -- a register format between two MFMAs, not storage; bf16's 8 bits on the expanded
    // values and on the taps were part of what kept it outside the pose bound (round 6; the oracle's emulation follows: front kind 5)
    using TT = typename std::conditional<sizeof(T) == 2, f16_t, T>::type;
    typedef TT tt4 __attribute__((ext_vector_type(4)));
    tt4 Af[MX ? KS : 1][2];                   // MX: Toeplitz fragments of the tap rows, [ky][quad | halo operand]
    float mxp[4] = {0.f, 0.f, 0.f, 0.f};     // MX: s0, b0 of this lane's expansion channel (lane & 15), s1, b1 of its depthwise channel (lane >> 2)
    if (active) {
        if constexpr (MX) {
            const float* PP = a.wparams + (size_t)ch * PFX;
            mxp[0] = PP[lane & 15]; mxp[1] = PP[16 + (lane & 15)]; mxp[2] = PP[32 + (lane >> 2)]; mxp[3] = PP[48 + (lane >> 2)];
#pragma unroll
            for (int f = 0; f < 2 * KS; ++f) Af[f >> 1][f & 1] = *(const tt4*)(PP + 64 + (f * 64 + lane) * 2);
        } else {
            const f32x4* PP = (const f32x4*)(a.wparams + (size_t)ch * PF);
#pragma unroll
            for (int j = 0; j < NPL; ++j) pv[j] = PP[min(lane + 64 * j, PF / 4 - 1)];
        }
        load_row(iy_first);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int f = 0; f < NF; ++f)
                wf[ni][f] = *(const raw_t*)((const T*)a.Wp + (((size_t)(ch * NI + ni) * a.nkb_total + f / HL) * HL + f % HL) * 64 * EPL + lane * EPL);
        // -> the wave-private LDS block (no workgroup barrier: a wave reads only what it wrote itself, and LDS operations of one
        // wave execute in order)
        if constexpr (!MX) {
#pragma unroll
            for (int j = 0; j < NPL; ++j)
                if (lane + 64 * j < PF / 4) *(f32x4*)(P + (lane + 64 * j) * 4) = pv[j];
        }
        if constexpr (WLDS) {
            // 4 * KBN registers that a 4-waves-per-SIMD budget does not have: parked in the wave's LDS block and re-read in front
            // of every row's MFMAs (one conflict-free ds_read_b128 per fragment and row)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int f = 0; f < NF; ++f) *(raw_t*)(Wl + (ni * NF + f) * 1024 + lane * 16) = wf[ni][f];
        }
    }
    if (!active) return;
#ifdef COSY_WAVE_STAMPS
    st_t1 = __builtin_amdgcn_s_memtime();
#endif

    float acc[NOPEN][TO][NCH];               // output rows in flight (input-stationary accumulation)
#pragma unroll
    for (int s = 0; s < NOPEN; ++s)
#pragma unroll
        for (int t = 0; t < TO; ++t)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[s][t][c] = 0.f;
    float sum[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) sum[c] = 0.f;
    // MX: the accumulators are the small MFMA's C operands (lane = (channel lane >> 2, quad lane & 3): 4 pixels); lane permutations to and from
    // the expansion's layout (lane = channel + 16 * quad); the identity fragment of the transposing MFMA
    f32x4 accx[NOPEN][PPL];
#pragma unroll
    for (int s = 0; s < NOPEN; ++s)
#pragma unroll
        for (int q = 0; q < PPL; ++q) accx[s][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bp_in = ((lane >> 2) + 16 * (lane & 3)) * 4, bp_out = (4 * (lane & 15) + (lane >> 4)) * 4;
    const int jq = lane & 3;
    // fused interior iteration (row_px): lane masks of a row's first / last quad, the parked weight fragments' LDS address, the output row's store as
    // (uniform row pointer, per-lane byte offset)
    const unsigned long long first_quad = __builtin_amdgcn_ballot_w64(jq == 0), last_quad = __builtin_amdgcn_ballot_w64(jq == 3);
    const unsigned wl_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)(Wl + lane * 16);
    t4 ident;
#pragma unroll
    for (int e = 0; e < 4; ++e) ident[e] = (T)(4 * kg + e == p ? 1.f : 0.f);
    auto cvt = [](float v) -> T {           // fp16 saturates (as to_f16_sat; one v_med3_f32)
        if constexpr (__is_same(T, f16_t)) return (T)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); else return (T)v;
    };
    auto cvt_e = [](float v) -> TT {        // the expanded values as tap operands: fp16, saturating
        if constexpr (sizeof(T) == 2) return (TT)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); else return (TT)v;
    };
    out_t yv[TO][NI];                        // finished output row waiting for its store (issued one row later)
    int oy_pending = -1;
    // D is written in the CHUNKED layout [sample][Cmid/16][Ho*Wo][16]: this wave's output row is one contiguous run of
    // Wo * 32 bytes, so every 128-byte line is completed by one wave within one row.  (In plain NHWC the four 32-byte quarters
    // of a line belong to four different jobs that run at different times; with ~30 MB of half-written rows in flight per XCD
    // the 4 MB L2 evicted them quarter by quarter -- 25 % of block 3's time by knock-out timing.)  The project GEMM reads
    // this layout directly (PwArgs::a_chunked).
    static_assert(NI == 1, "chunked D: one 16-channel chunk per job");
    T* __restrict__ Dlane = (T*)a.D + (size_t)(b * a.nchunks + ch) * a.Ho * a.Wo * 16 + (size_t)(MX ? p : p * TO) * a.ds_pix + kg * 4;   // + uniform row offset (MX: element t of a row = segment t)
    const size_t drow = (size_t)a.ds_row;
    const T* D_chunk = (const T*)a.D + (size_t)(b * a.nchunks + ch) * a.Ho * a.Wo * 16;
    const unsigned d_voff = (unsigned)(((size_t)(MX ? p : p * TO) * a.ds_pix + kg * 4) * sizeof(T));
    const int dpix = MX ? 16 * a.ds_pix : a.ds_pix;
    auto flush = [&]() {                     // store the pending output row
        if (oy_pending >= 0 && !COSY_DBG(a.dbg & 1)) {      // dbg 1: no output stores (timing experiments)
            T* o = Dlane + (size_t)(COSY_DBG(a.dbg & 8) ? 0 : oy_pending) * drow;      // dbg 8: every row lands on row 0
#pragma unroll
            for (int t = 0; t < TO; ++t) {
                if (COSY_DBG(a.dbg & 4) && t > 0) break;                                 // dbg 4: one store per row
                if (FULLW || (MX ? 16 * t + p : p * TO + t) < a.Wo) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) *(out_t*)(o + t * dpix) = yv[t][ni];
                }
            }
            st_in_flight = 1;
        }
    };

    // One input row.  IN = true is the INTERIOR form: the row lies inside the map, a next row exists, and every output row it feeds
    // belongs to this job -- all of that is known at compile time there, so the body is straight-line code: without it every tap
    // row sits behind a wave-uniform branch, the accumulators are copied at each merge (24 v_mov_b64 per row of the k=5 shape)
    // and the tap reads cannot be scheduled across the blocks (one exposed LDS latency per tap row).
    auto row = [&](auto uc, auto inc, const int base) {
            constexpr int u = decltype(uc)::value;
            constexpr bool IN = decltype(inc)::value;
            const int iy = base + u;
            if constexpr (!IN) { if (iy < iy_first || iy > iy_last) return; }
            // ---- A. expanded row iy (transient registers); its global loads were issued one row ago
            float Er[RP][NCH];
            WAVE_STAMP_ROW();
            f32x4 mq[XASM ? PPL : 1];
            wait_row(mq);
            tt4 W1[MX ? PPL : 1], W2[MX ? PPL : 1];      // MX: the row's operands of the tap MFMAs, per 16-pixel segment
            if constexpr (MX) {
                if (IN || iy < a.H) {
                    int lo[PPL], hi[PPL];
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        f32x4 m = f32x4{0.f, 0.f, 0.f, 0.f};
                        // operands swapped: rows = pixels -> lane (channel lane & 15, quad lane >> 4) holds 4 pixels
                        if constexpr (XASM) m = mq[q];
                        else {
#pragma unroll
                            for (int f = 0; f < NF; ++f) {
                                raw_t xv = __builtin_bit_cast(raw_t, xc[q][f / HL]);
                                if constexpr (WLDS) mma(m, xv, *(const raw_t*)(Wl + f * 1024 + lane * 16));
                                else mma(m, xv, wf[0][f]);
                            }
                        }
                        float y4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) y4[e] = m[e] * mxp[0] + mxp[1];
                        if (!COSY_DBG(a.dbg & 512)) silu4<false>(y4);      // dbg 512: no SiLU (timing)
                        if constexpr (!FULLW) {                           // pixels beyond the row end pad their neighbours (this lane: pixels 16 q + 4 (lane >> 4) + e)
#pragma unroll
                            for (int e = 0; e < 4; ++e) y4[e] = 16 * q + 4 * kg + e < a.W ? y4[e] : 0.f;
                        }
                        const tt4 hv = tt4{cvt_e(y4[0]), cvt_e(y4[1]), cvt_e(y4[2]), cvt_e(y4[3])};      // E as the tap MFMAs' operand (fp16)
                        const i32x2 hh = __builtin_bit_cast(i32x2, hv);
                        lo[q] = hh[0]; hi[q] = hh[1];
                        if (!COSY_DBG(a.dbg & 128)) { lo[q] = __builtin_amdgcn_ds_bpermute(bp_in, hh[0]); hi[q] = __builtin_amdgcn_ds_bpermute(bp_in, hh[1]); }   // dbg 128: no lane permutation (timing)
                    }
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        int hp = __builtin_amdgcn_update_dpp(0, hi[q], 0x90, 0xf, 0xf, false);     // quad_perm [0,0,1,2]: the previous quad's pixels 2, 3
                        int ln = __builtin_amdgcn_update_dpp(0, lo[q], 0xF9, 0xf, 0xf, false);     // quad_perm [1,2,3,3]: the next quad's pixels 0, 1
                        // a segment's first / last quad: the neighbouring segment's last / first quad, zero padding at the row ends
                        int hp0 = 0, ln3 = 0;
                        if constexpr (PPL > 1) {
                            if (q > 0) hp0 = __builtin_amdgcn_update_dpp(0, hi[q > 0 ? q - 1 : 0], 0x93, 0xf, 0xf, false);             // quad_perm [3,0,1,2]
                            if (q < PPL - 1) ln3 = __builtin_amdgcn_update_dpp(0, lo[q < PPL - 1 ? q + 1 : 0], 0x39, 0xf, 0xf, false);   // quad_perm [1,2,3,0]
                        }
                        hp = jq == 0 ? hp0 : hp; ln = jq == 3 ? ln3 : ln;
                        W1[q] = __builtin_bit_cast(tt4, i32x2{lo[q], hi[q]}); W2[q] = __builtin_bit_cast(tt4, i32x2{hp, ln});
                    }
                }
            } else
            if (IN || iy < a.H) {
                float sc0[NCH], bi0[NCH];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) { load4(Pl + 0 * 16 * NI + ni * 16, sc0 + ni * 4); load4(Pl + 1 * 16 * NI + ni * 16, bi0 + ni * 4); }
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        f32x4 m = f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (XASM) m = mq[q];
                        else {
#pragma unroll
                            for (int f = 0; f < NF; ++f) {
                                raw_t xv = __builtin_bit_cast(raw_t, xc[q][f / HL]);
                                if constexpr (WLDS) mma(m, *(const raw_t*)(Wl + (ni * NF + f) * 1024 + lane * 16), xv);
                                else mma(m, wf[ni][f], xv);
                            }
                        }
                        float y4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) y4[e] = m[e] * sc0[ni * 4 + e] + bi0[ni * 4 + e];     // = log2(e) * BN0(expand)
                        silu4<true>(y4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (!FULLW) { if (p * PPL + q >= a.W) y4[e] = 0.f; }   // pixels beyond the row end pad their neighbours
                            Er[LO + q][ni * 4 + e] = y4[e];
                        }
                    }
                }
                // halo pixels of the lane's run come from the neighbouring lanes of its 16-lane row (0 at the image border)
#pragma unroll
                for (int m = 0; m < LO; ++m) {            // ring pixel m = local pixel m - LO (< 0)
                    const int d = m - LO;
                    const int off = (d - (PPL - 1)) / PPL;   // floor(d / PPL) for d < 0
                    const int q = d - off * PPL;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const float src = Er[LO + q][c];
                        Er[m][c] = off == -1 ? row_shr0<1>(src) : off == -2 ? row_shr0<2>(src) : row_shr0<3>(src);
                    }
                }
#pragma unroll
                for (int m = 0; m < HI; ++m) {            // ring pixel LO + PPL + m = local pixel PPL + m
                    const int off = (PPL + m) / PPL, q = (PPL + m) % PPL;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const float src = Er[LO + q][c];
                        Er[LO + PPL + m][c] = off == 1 ? row_shl0<1>(src) : off == 2 ? row_shl0<2>(src) : row_shl0<3>(src);
                    }
                }
            }
            // ---- global memory, issued HERE (between the expansion and the depthwise work): the next input row's loads, then
            // the previous output row's stores; both have this row's whole depthwise phase to complete, and the stores longer.
            if ((IN || iy + 1 <= iy_last) && !COSY_DBG(a.dbg & 2)) load_row(iy + 1);     // dbg 2: the first row's fragments are reused
            flush();
            oy_pending = -1;
            // ---- B. scatter the row into the output rows it feeds: input row iy is tap row ky of output row (iy + LO - ky) / S
            bool done = false;
            int oy_done = -1;
            out_t ynew[TO][NI];
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                constexpr int KSc = KS;
                const int num = u + LO - ky;                          // compile-time after unrolling
                if (((num % S) + S) % S != 0) continue;
                const int os = (((num - (((num % S) + S) % S)) / S) % NOPEN + NOPEN) % NOPEN;   // accumulator slot of that output row
                const int oy = (iy + LO - ky) / S;
                if constexpr (!IN) { if (iy + LO - ky < 0 || oy < oy_a || oy >= oy_b) continue; }   // wave-uniform
                if constexpr (MX) {
                    if ((IN || iy < a.H) && !COSY_DBG(a.dbg & 64)) {       // dbg 64: no tap MFMAs (timing)
#pragma unroll
                        for (int q = 0; q < PPL; ++q) accx[os][q] = mma4(Af[ky][0], W1[q], accx[os][q]);
#pragma unroll
                        for (int q = 0; q < PPL; ++q) accx[os][q] = mma4(Af[ky][1], W2[q], accx[os][q]);
                    }
                } else
                if (IN || iy < a.H) {
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        float w[NCH];
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) load4(taps + COSY_KO_TAP_INDEX((ky * KSc + kx) * 16 * NI) + ni * 16, w + ni * 4);
#pragma unroll
                        for (int t = 0; t < TO; ++t)
#pragma unroll
                            for (int c = 0; c < NCH; ++c) acc[os][t][c] += w[c] * Er[t * S + kx][c];
                    }
                }
                if constexpr (MX) {
                    if (ky == KS - 1) {
#pragma unroll
                        for (int q = 0; q < PPL; ++q) {
                            float y4[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) y4[e] = accx[os][q][e] * mxp[2] + mxp[3];
                            if (!COSY_DBG(a.dbg & 512)) silu4<false>(y4);      // dbg 512: no SiLU (timing)
#pragma unroll
                            for (int e = 0; e < 4; ++e) { if constexpr (FULLW) sum[0] += y4[e]; else sum[0] += 16 * q + 4 * jq + e < a.Wo ? y4[e] : 0.f; }      // (this lane: pixels 16 q + 4 jq + e)
                            const t4 hv = t4{cvt(y4[0]), cvt(y4[1]), cvt(y4[2]), cvt(y4[3])};
                            const i32x2 hh = __builtin_bit_cast(i32x2, hv);
                            if (COSY_DBG(a.dbg & 256)) {                      // dbg 256: no permutation / transposition of the output row (timing)
                                ynew[q][0] = hv;
                            } else {
                                const int lo = __builtin_amdgcn_ds_bpermute(bp_out, hh[0]), hi = __builtin_amdgcn_ds_bpermute(bp_out, hh[1]);
                                const f32x4 tr = mma16(__builtin_bit_cast(t4, i32x2{lo, hi}), ident, f32x4{0.f, 0.f, 0.f, 0.f});   // -> lane (pixel, 4 channels), exact
#pragma unroll
                                for (int e = 0; e < 4; ++e) ynew[q][0][e] = (T)tr[e];
                            }
                            accx[os][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        done = true; oy_done = oy;
                    }
                } else
                if (ky == KS - 1) {                                   // last tap row: the output row is complete
                    float sc1[NCH], bi1[NCH];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) { load4(Pl + 2 * 16 * NI + ni * 16, sc1 + ni * 4); load4(Pl + 3 * 16 * NI + ni * 16, bi1 + ni * 4); }
#pragma unroll
                    for (int t = 0; t < TO; ++t) {
                        const bool okx = FULLW || p * TO + t < a.Wo;
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            float y4[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) y4[e] = acc[os][t][ni * 4 + e] * sc1[ni * 4 + e] + bi1[ni * 4 + e];
                            silu4<false>(y4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int c = ni * 4 + e;
                                if constexpr (FULLW) sum[c] += y4[e]; else sum[c] += okx ? y4[e] : 0.f;
                                if constexpr (__is_same(T, f16_t)) ynew[t][ni][e] = to_f16_sat(y4[e]);
                                else ynew[t][ni][e] = (T)y4[e];
                                acc[os][t][c] = 0.f;
                            }
                        }
                    }
                    done = true; oy_done = oy;
                }
            }
            if (done) {                      // parked until the next row's memory block
#pragma unroll
                for (int t = 0; t < TO; ++t)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) yv[t][ni] = ynew[t][ni];
                oy_pending = oy_done;
                // MX: stored right away -- the row finishes behind the next input row's loads in any case (the queue order the counted wait needs),
                // and PPL register pairs are not held across the next expansion.  (The fp32-FMA form keeps the parking: with the stores inside its
                // straight-line tap rows hipcc's schedule spills 70-150 bytes in six variants.)
                if constexpr (MX) { flush(); oy_pending = -1; }
            }
    };
    // ---- matrix-pipe form of the 16-pixel-row shapes (PPL = 1: blocks 9-17 at 256x256), SOFTWARE-PIPELINED over the rows (round 6).  A wave's row is a chain of
    // dependent steps -- wait, expansion MFMAs, BN, SiLU, convert, lane permutation, tap MFMAs, BN, SiLU, convert, permutation, transposition, store: 1300-1600 cycles
    // for ~130 instructions when a wave has its SIMD to itself (profiles/r05_wave_occupancy.txt).  One ITERATION here does the tap / output side of row iy - 1
    // (operands W1p / W2p kept from the previous iteration) and the expansion side of row iy: two independent chains.  row_mx is the plain form of an iteration
    // (hipcc's schedule; rows at the job's and the map's edges: every part behind a wave-uniform condition); row_px below is the interior form, every instruction
    // of it placed by hand.  Measured (profiles/r06_wave_rows.txt): row to row alone on a SIMD 1581 -> 1102 cycles (blocks 14-17) / 1333 -> 1004 (block 13), at four
    // waves per SIMD 2323 -> 1931; results bit-identical to the unpipelined form (profiles/exp/ab_bits.py).
    tt4 W1p[PIPE ? PPL : 1], W2p[PIPE ? PPL : 1];
    bool prev_fused = false;      // the previous iteration was the fused form (row_px)
    auto row_mx = [&](auto uc, auto inc, const int base) {
        if constexpr (PIPE) {
            constexpr int u = decltype(uc)::value, up = (u + U - 1) % U;      // accumulator-slot arithmetic of row iy - 1
            constexpr bool IN = decltype(inc)::value;
            const int iy = base + u, ip = iy - 1;
            const bool doA = IN || (iy >= iy_first && iy <= iy_last), doB = IN || (ip >= iy_first && ip <= iy_last);
            if constexpr (!IN) { if (!doA && !doB) return; }
            tt4 W1n[PPL], W2n[PPL];
            if constexpr (!IN) {
                // behind a fused iteration (row_px, which starts a new output row with SrcC = 0 and never resets): the accumulator slot this iteration's first tap row
                // adds into still holds the output row the fused iteration finished
                if (prev_fused) { constexpr int os0 = ((up + LO) % NOPEN + NOPEN) % NOPEN; accx[os0][0] = f32x4{0.f, 0.f, 0.f, 0.f}; prev_fused = false; }
            }
            // ---- A1: row iy's expansion chain issued, the next row's loads behind it
            f32x4 mq[PPL];
            if (doA) {
                WAVE_STAMP_ROW();
                wait_row(mq);
                if ((IN || iy + 1 <= iy_last) && !COSY_DBG(a.dbg & 2)) load_row(iy + 1);
            }
            // ---- B: row iy - 1 scattered into the output rows it feeds, the finished output row stored
            if (doB) {
                bool done = false;
                int oy_done = -1;
                out_t ynew[TO][NI];
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    constexpr int os_dummy = 0; (void)os_dummy;
                    const int os = ((up + LO - ky) % NOPEN + NOPEN) % NOPEN;     // compile-time after unrolling (S = 1)
                    const int oy = ip + LO - ky;
                    if constexpr (!IN) { if (oy < 0 || oy < oy_a || oy >= oy_b) continue; }
                    if ((IN || ip < a.H) && !COSY_DBG(a.dbg & 64)) {
#pragma unroll
                        for (int q = 0; q < PPL; ++q) accx[os][q] = mma4(Af[ky][0], W1p[q], accx[os][q]);
#pragma unroll
                        for (int q = 0; q < PPL; ++q) accx[os][q] = mma4(Af[ky][1], W2p[q], accx[os][q]);
                    }
                    if (ky == KS - 1) {
#pragma unroll
                        for (int q = 0; q < PPL; ++q) {
                            float y4[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) y4[e] = accx[os][q][e] * mxp[2] + mxp[3];
                            if (!COSY_DBG(a.dbg & 512)) silu4p<false>(y4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) sum[0] += y4[e];
                            const t4 hv = t4{cvt(y4[0]), cvt(y4[1]), cvt(y4[2]), cvt(y4[3])};
                            const i32x2 hh = __builtin_bit_cast(i32x2, hv);
                            if (COSY_DBG(a.dbg & 256)) {
                                ynew[q][0] = hv;
                            } else {
                                const int lo = __builtin_amdgcn_ds_bpermute(bp_out, hh[0]), hi = __builtin_amdgcn_ds_bpermute(bp_out, hh[1]);
                                const f32x4 tr = mma16(__builtin_bit_cast(t4, i32x2{lo, hi}), ident, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                                for (int e = 0; e < 4; ++e) ynew[q][0][e] = (T)tr[e];
                            }
                            accx[os][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        done = true; oy_done = oy;
                    }
                }
                if (done) {
#pragma unroll
                    for (int t = 0; t < TO; ++t) yv[t][0] = ynew[t][0];
                    oy_pending = oy_done;
                    flush(); oy_pending = -1;
                }
            }
            // ---- A2: row iy's expanded values -> its tap operands (independent of B: hipcc interleaves the two)
            if (doA) {
                if (IN || iy < a.H) {
                    int lo[PPL], hi[PPL];
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        float y4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) y4[e] = mq[q][e] * mxp[0] + mxp[1];
                        if (!COSY_DBG(a.dbg & 512)) silu4p<false>(y4);
                        const tt4 hv = tt4{cvt_e(y4[0]), cvt_e(y4[1]), cvt_e(y4[2]), cvt_e(y4[3])};
                        const i32x2 hh = __builtin_bit_cast(i32x2, hv);
                        lo[q] = hh[0]; hi[q] = hh[1];
                        if (!COSY_DBG(a.dbg & 128)) { lo[q] = __builtin_amdgcn_ds_bpermute(bp_in, hh[0]); hi[q] = __builtin_amdgcn_ds_bpermute(bp_in, hh[1]); }
                    }
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        int hp = __builtin_amdgcn_update_dpp(0, hi[q], 0x90, 0xf, 0xf, false);     // quad_perm [0,0,1,2]
                        int ln = __builtin_amdgcn_update_dpp(0, lo[q], 0xF9, 0xf, 0xf, false);     // quad_perm [1,2,3,3]
                        int hp0 = 0, ln3 = 0;
                        if constexpr (PPL > 1) {
                            if (q > 0) hp0 = __builtin_amdgcn_update_dpp(0, hi[q > 0 ? q - 1 : 0], 0x93, 0xf, 0xf, false);             // quad_perm [3,0,1,2]
                            if (q < PPL - 1) ln3 = __builtin_amdgcn_update_dpp(0, lo[q < PPL - 1 ? q + 1 : 0], 0x39, 0xf, 0xf, false);   // quad_perm [1,2,3,0]
                        }
                        hp = jq == 0 ? hp0 : hp; ln = jq == 3 ? ln3 : ln;
                        W1n[q] = __builtin_bit_cast(tt4, i32x2{lo[q], hi[q]}); W2n[q] = __builtin_bit_cast(tt4, i32x2{hp, ln});
                    }
                }
            }
            if (doA) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) { W1p[q] = W1n[q]; W2p[q] = W2n[q]; }
            }
        }
    };

    // ---- the FUSED interior iteration (round 6): the tap / output side of row iy - 1 and the expansion side of row iy as ONE hand-ordered instruction stream.
    // Order (every line an asm volatile statement, nothing for hipcc to schedule):
    //   [parked weight fragments 0..2 -> registers]  taps of row iy - 1 (first operands of all KS output rows, then the halo operands: a dependent pair is KS - 1 MFMAs apart)
    //   wait for row iy's fragments (counted: the previous output row's store stays in flight)
    //   expansion chain m1 .. mKBN, one group of four output-side VALU instructions (BatchNorm 1, SiLU's five steps) behind each link: the chain's 16-cycle links cover them
    //   row iy + 1's loads, BatchNorm 0 of the expanded row, then the two sides interleaved: squeeze adds | SiLU steps | converts | both lane permutations
    //   transposing MFMA of the output row; under its latency the halo operands of the next taps; convert, store.
    // Hazards written out by hand (hipcc pads nothing here; padding it would insert: 4x4x4 result -> VALU / VMEM read 5 wait states, -> the next MFMA's SrcC 2;
    // 16x16x16 / 16x16x32 result -> VALU read 8; transcendental -> VALU read 1; VALU write -> MFMA A / B read 2 -- read off hipcc's own padding of
    // profiles/exp/mfma_hazards.hip): every such pair below has at least that many instructions of the other chain between its two ends, or an s_nop.
    auto row_px = [&](auto uc, const int base) {
        if constexpr (PIPE) {
            constexpr int u = decltype(uc)::value, up = (u + U - 1) % U;
            constexpr int od = ((up + LO - (KS - 1)) % NOPEN + NOPEN) % NOPEN;       // the output row that row iy - 1 completes
            const int iy = base + u, ip = iy - 1, oy = ip - LO;
            WAVE_STAMP_ROW();
            if (!st_in_flight) asm volatile("s_waitcnt vmcnt(0) ; XWAIT" ::: "memory");       // (first fused iteration of a job only: nothing was stored behind its loads)
            px_lgkm<0>();
            raw_t wb[3];
            if constexpr (WLDS) { px_wread<0>(wb[0], wl_addr); px_wread<1024>(wb[1], wl_addr); px_wread<2048>(wb[2], wl_addr); }
            // ---- taps of row iy - 1: tap row ky feeds output row ip + LO - ky (accumulator slot compile-time); ky = KS - 1 first: that row is complete behind its pair
            unroll_seq([&](auto kc) {
                constexpr int ky = KS - 1 - decltype(kc)::value, os = ((up + LO - ky) % NOPEN + NOPEN) % NOPEN;
                if constexpr (ky == 0) px_tap0<TT>(accx[os][0], Af[ky][0], W1p[0]); else px_tap<TT>(accx[os][0], Af[ky][0], W1p[0]);
            }, std::make_integer_sequence<int, KS>{});
            unroll_seq([&](auto kc) {
                constexpr int ky = KS - 1 - decltype(kc)::value, os = ((up + LO - ky) % NOPEN + NOPEN) % NOPEN;
                px_tap<TT>(accx[os][0], Af[ky][1], W2p[0]);
            }, std::make_integer_sequence<int, KS>{});
            xfrag_fence<XTOP, PPL * KBN>();
            asm volatile("s_waitcnt vmcnt(1) ; XWAIT" ::: "memory");
            // ---- expansion chain with the output side's first steps in its shadow
            f32x4 macc;
            float yo[4], to[4], ye[4], te[4];
            unroll_seq([&](auto fc) {
                constexpr int f = decltype(fc)::value;           // link f of the chain: k-block f / HL (bf16: its hi, then its lo weight fragment)
                if constexpr (WLDS) {
                    // fragments in flight at this point: f .. min(f + 2, NF - 1)
                    px_lgkm<(f + 2 < NF ? 2 : NF - 1 - f)>();
                    px_mma<T, XTOP, f / HL, f == 0>(macc, wb[f % 3]);
                    if constexpr (f + 3 < NF) px_wread<(f + 3) * 1024>(wb[f % 3], wl_addr);
                } else px_mma<T, XTOP, f / HL, f == 0>(macc, wf[0][f]);
                if constexpr (f == 0) { if constexpr (KS == 3) px_nop<1>(); px_bn(yo, accx[od][0], mxp[2], mxp[3]); }
                if constexpr (f == 1) px_silu_a(yo, to);
                if constexpr (f == 2) px_silu_b(to);
                if constexpr (f == 3) px_silu_c(to);
            }, std::make_integer_sequence<int, NF>{});
            if constexpr (NF <= 3) px_silu_c(to);
            px_silu_d(to);
            load_row(iy + 1);                        // row iy's fragments are consumed: the next row's loads (they return long after the last link has read its operands)
            px_silu_e(yo, to);
            // ---- both sides interleaved
            px_bn(ye, macc, mxp[0], mxp[1]);         // >= 8 instructions behind the last link
            px_sum(sum[0], yo);
            px_silu_a(ye, te);
            int po0, po1, qo0, qo1;
            px_cvt<T>(po0, po1, yo, -65504.f, 65504.f);
            px_silu_b(te);
            px_bperm2(qo0, qo1, bp_out, po0, po1);
            px_silu_c(te);
            px_silu_d(te);
            px_silu_e(ye, te);                   // (the first reciprocal is three instructions old)
            int pe0, pe1, w1lo, w1hi;
            px_cvt<TT>(pe0, pe1, ye, -65504.f, 65504.f);
            px_bperm2(w1lo, w1hi, bp_in, pe0, pe1);
            // ---- the output row: transposed back to (pixel, 4 channels), stored; the next taps' operands under the transposition's latency
            px_lgkm<2>();
            f32x4 tr;
            px_transpose<T>(tr, __builtin_bit_cast(t4, i32x2{qo0, qo1}), ident);
            px_lgkm<0>();
            int hp, ln;
            px_halo(hp, ln, w1lo, w1hi, first_quad, last_quad);
            px_nop<3>();
            px_cvt_store<T>(tr, d_voff, (const T*)D_chunk + (size_t)oy * drow);
            st_in_flight = 1;
            W1p[0] = __builtin_bit_cast(tt4, i32x2{w1lo, w1hi}); W2p[0] = __builtin_bit_cast(tt4, i32x2{hp, ln});
            prev_fused = true;
        }
    };
    if constexpr (PIPE) {
        // interior iterations: row iy - 1 feeds only output rows of this job and lies inside the map; row iy lies inside the map and has a successor
        const int in_lo2 = oy_a + KS - LO, in_hi2 = min(min(oy_b - LO, a.H - 1), iy_last - 1);
        for (int base = (iy_first / U) * U; base <= iy_last + 1; base += U) {
            unroll_seq([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                int iy = base + u;
                // Whole groups of U fused iterations as ONE straight-line loop body (entered from the instance whose u is the first fused row's: band 0, i.e. every
                // launch of the shipped row-band counts of these shapes): between two fused iterations there is then no merge with the plain form, so hipcc renames the
                // accumulator slots from iteration to iteration instead of copying them at every merge (8 v_mov_b64 per row of the k = 5 shape otherwise).
                if constexpr (u == (KS - LO) % U) {
                    if (iy == in_lo2 && oy_a == 0 && !COSY_DBG(a.dbg & 16)) {
                        for (int g = (in_hi2 - in_lo2 + 1) / U; g > 0; --g) {
                            unroll_seq([&](auto jc) {
                                constexpr int j = decltype(jc)::value;
                                row_px(std::integral_constant<int, (u + j) % U>{}, base + ((u + j) / U) * U);
                            }, std::make_integer_sequence<int, U>{});
                            base += U;
                        }
                        iy = base + u;
                    }
                }
                if (iy >= in_lo2 && iy <= in_hi2 && !COSY_DBG(a.dbg & 16)) { row_px(uc, base); return; }
                row_mx(uc, std::false_type{}, base);
            }, std::make_integer_sequence<int, U>{});
        }
    } else {
    // interior rows: iy + LO - (KS-1) >= oy_a * S (every tap row's output row is >= oy_a), iy + LO < oy_b * S (... < oy_b),
    // iy < H, iy + 1 <= iy_last
    const int in_lo = oy_a * S + KS - 1 - LO, in_hi = min(min(oy_b * S - 1 - LO, a.H - 1), iy_last - 1);
    for (int base = (iy_first / U) * U; base <= iy_last; base += U) {
        unroll_seq([&](auto uc) {
            const int iy = base + decltype(uc)::value;
            if constexpr (wave_interior_form(KS, S, KBN, PPL, MINW)) {
                if (iy >= in_lo && iy <= in_hi && !COSY_DBG(a.dbg & 16)) { row(uc, std::true_type{}, base); return; }   // dbg 16: boundary form everywhere
            }
            row(uc, std::false_type{}, base);
        }, std::make_integer_sequence<int, U>{});
    }
    }
    flush();
    // ---- squeeze sums: fixed-order tree over the 16 lanes of a row (one channel quad per row), lane 15 writes
    if constexpr (MX) {      // a lane summed the 4 pixels of its quad per row: the channel's total = the 4 lanes of its MFMA block (fixed order)
        float v = sum[0];
        v += dpp_mov0<0xB1>(v);      // quad_perm [1,0,3,2]
        v += dpp_mov0<0x4E>(v);      // quad_perm [2,3,0,1]
        if (jq == 0) a.partial[((size_t)b * a.rsplit + band) * a.Cmid + c0 + (lane >> 2)] = v;
    } else {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float v = sum[c];
        v += row_shr0<1>(v); v += row_shr0<2>(v); v += row_shr0<4>(v); v += row_shr0<8>(v);
        sum[c] = v;
    }
    if (p == 15) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            *(f32x4*)(a.partial + ((size_t)b * a.rsplit + band) * a.Cmid + c0 + ni * 16 + kg * 4) = f32x4{sum[ni * 4], sum[ni * 4 + 1], sum[ni * 4 + 2], sum[ni * 4 + 3]};
    }
    }
#ifdef COSY_WAVE_STAMPS
    if (stamping && lane == 0) {
        unsigned long long* dst = a.stamps + (size_t)(blockIdx.x / a.stamp_stride) * 8;
        dst[0] = st_n; dst[1] = st_t0; dst[2] = st_t1; dst[3] = st_first; dst[4] = __builtin_amdgcn_s_memtime();
        dst[5] = st_sum; dst[6] = st_min; dst[7] = st_max;      // over the row-start-to-row-start intervals (st_n - 1 of them)
    }
#endif
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// built shapes (KS, S, KBN, PPL, NI, row is exactly 16*PPL pixels wide, minimum waves per SIMD the register allocation
// must allow, row bands per (sample, chunk)): the MBConv blocks of EfficientNet-B3 at 256x256 and 240x320 inputs whose maps
// are 16..128 pixels wide.  Row bands: a job = (sample, chunk, band of output rows); bands only balance the number of jobs
// against the resident wave slots of the chip (e.g. block 2 at 256 crops: 2304 jobs on 2048 slots = a second round that is
// 12 % full; 4 bands: 4.5 rounds of quarter-length jobs) at the price of KS-S recomputed rows per band.  The band count is
// a property of the SHAPE, never of the batch size: the squeeze sums are reduced per band, so results stay bit-identical
// across batch sizes.
#define COSY_WAVE_VARIANTS(X)                                                                                      \
    X(3, 2, 1, 8, 1, true, 2, 4) X(3, 1, 1, 4, 1, true, 3, 2) X(5, 2, 1, 4, 1, true, 3, 1) X(5, 1, 2, 2, 1, true, 3, 2)      \
    X(3, 2, 2, 2, 1, true, 4, 2) X(3, 1, 3, 1, 1, true, 4, 1) X(5, 1, 3, 1, 1, true, 4, 1) X(5, 1, 5, 1, 1, true, 4, 1)      \
    X(3, 1, 1, 5, 1, true, 2, 2) X(5, 2, 1, 6, 1, false, 2, 1) X(5, 1, 2, 3, 1, false, 2, 2) X(3, 2, 2, 4, 1, false, 3, 1)   \
    X(3, 1, 3, 2, 1, false, 3, 1) X(5, 1, 3, 2, 1, false, 2, 1) X(5, 1, 5, 2, 1, false, 2, 1)                                \
    /* rows that do not fill their 16 * PPL lanes, reached by walking the map's COLUMNS (wave_plan: transposed): 240x320 crops */ \
    X(3, 2, 1, 8, 1, false, 2, 4) X(5, 2, 1, 4, 1, false, 3, 1) X(5, 1, 2, 2, 1, false, 3, 2) X(3, 2, 2, 2, 1, false, 4, 2)   \
    X(3, 1, 3, 1, 1, false, 4, 1) X(5, 1, 3, 1, 1, false, 4, 1) X(5, 1, 5, 1, 1, false, 4, 1)                                \
    /* block 2 at 240x320 crops: the 160-pixel rows of the 120x160 map fill 16 x 10 lanes-pixels exactly (a column walk of 120 on 128 reads     \
       8 pixels per lane that are 7.7 KB apart: its loads and stores were 57 % of the kernel, knock-out timing) */                           \
    X(3, 2, 1, 10, 1, true, 2, 4)
// fp32 (parity mode): k-blocks are 16 deep (v_mfma_f32_16x16x4_f32 x 4 per fragment), so KBN = ceil(Cin / 16).  Same design, same
// checks; these instantiations are what test_backbone_fp32_vs_reference holds to the reference's own per-stage outputs at <= 1e-4.
// Not built (the fragment registers do not fit 256): 128- and 80-pixel rows with 2 k-blocks, 40-pixel rows with 3, 20-pixel rows with 9
// -- those blocks run unfused in fp32.
#define COSY_WAVE_VARIANTS_F32(X)                                                                                  \
    X(3, 1, 2, 4, 1, true, 2, 2) X(5, 2, 2, 4, 1, true, 2, 1) X(5, 1, 3, 2, 1, true, 2, 2)      \
    X(3, 2, 3, 2, 1, true, 3, 2) X(3, 1, 6, 1, 1, true, 3, 1) X(5, 1, 6, 1, 1, true, 3, 1) X(5, 1, 9, 1, 1, true, 3, 1)      \
    X(5, 2, 2, 6, 1, false, 2, 1) X(3, 2, 3, 4, 1, false, 2, 1) X(3, 1, 6, 2, 1, false, 2, 1) X(5, 1, 6, 2, 1, false, 2, 1)
enum { WAVE_MAX_RSPLIT = 4 };

struct WavePlan { int kbn, ppl, ni; bool fullw, ok, transposed, mx; };
// one orientation: rows of `W` pixels (the lane axis), `H` of them
static WavePlan wave_plan_1(int Cin, int Cmid, int H, int W, int k, int s, int dtype) {
    WavePlan p{};
    p.kbn = cdiv(Cin, dtype == COSY_F32 ? 16 : 32);
    p.ppl = cdiv(W, 16);
    if (p.ppl % s) ++p.ppl;
    p.fullw = W == 16 * p.ppl;
    p.ok = false;
    if (H < k) return p;
    // 16 channels per wave (NI = 1).  Measured at 256 crops on the 16x16 maps: NI = 2 is 3-20 % slower, NI = 3 up to 3.5x slower
    // (register spills into AGPRs); re-reading the small block input Cmid/16 times per sample is not what limits the kernel.
    p.ni = 1;
    if (Cmid % 16) return p;
#define X(KS, S, KBN, PPL, NI, FW, MW, RSP) if (k == KS && s == S && p.kbn == KBN && p.ppl == PPL && p.ni == NI && p.fullw == FW) p.ok = true;
    if (dtype == COSY_F32) { COSY_WAVE_VARIANTS_F32(X) } else { COSY_WAVE_VARIANTS(X) }
#undef X
    p.mx = p.ok && wave_mx(dtype == COSY_F32 ? 4 : 2, k, s, p.ppl, p.fullw);
    return p;
}
// The wave walks the map row by row with 16 * PPL lanes-pixels per row; a row that does not fill them wastes lanes (240x320 crops:
// 20-pixel rows on 32, 40 on 48 / 64).  The depthwise conv does not care which axis is walked, so when the map's COLUMNS fill their
// lanes better (15 on 16, 30 on 32, 120 on 128) the job is transposed: rows := columns, taps read as w[kx][ky], the map addressed through
// strides (WaveKArgs).  Stride 2 only with even H and W (the static "same" padding is then the same on both axes).  Square maps are
// never transposed.
static WavePlan wave_plan(int Cin, int Cmid, int H, int W, int k, int s, int dtype) {
    WavePlan n = wave_plan_1(Cin, Cmid, H, W, k, s, dtype);
    static const int allow = tune_int("COSY_WAVE_TRANSPOSE", 1);
    if (H == W || dtype == COSY_F32 || !allow || (s == 2 && ((H | W) & 1))) return n;
    WavePlan t = wave_plan_1(Cin, Cmid, W, H, k, s, dtype);
    if (!t.ok) return n;
    const long cost_n = (long)16 * n.ppl * H, cost_t = (long)16 * t.ppl * W;      // lane-slots x steps
    if (n.ok && cost_t * 108 >= cost_n * 100) return n;
    t.transposed = true;
    return t;
}
// Per-chunk parameter blocks of the wave kernel: [chunk][4 + k*k][16] fp32 = s0 * log2(e), b0 * log2(e), s1, b1, then the taps
// * ln 2 in the order the job walks them (w[kx][ky] for a transposed job).  The products are single-precision, exactly what the
// kernel used to compute per job.
// Taps on the matrix pipe (WavePlan::mx): [chunk][s0 16][b0 16][s1 16][b1 16] fp32 (no log2 e: the expanded values are rounded to the storage
// type as they are), then 2 k fragments [ky][operand 0 | 1][lane 64][4] of the storage type: the Toeplitz rows of tap row ky for the lane's
// output pixel i = lane & 3 and channel lane >> 2 -- operand 0 against the pixels 4j .. 4j+3 of the lane quad, operand 1 against 4j-2, 4j-1, 4j+4, 4j+5.
static inline uint16_t wave_f16_bits(float f) { _Float16 h = (_Float16)(f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f)); return __builtin_bit_cast(uint16_t, h); }
static inline uint16_t wave_bf16_bits(float f) {       // round to nearest even (finite weights)
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static size_t wave_chunk_floats(const WavePlan& p, int k) { return p.mx ? (size_t)64 + 2 * k * 128 : (size_t)(4 + k * k) * 16; }
size_t wave_params_floats(int Cin, int Cmid, int k, int s, int dtype, int H, int W) {
    return (size_t)(Cmid / 16) * wave_chunk_floats(wave_plan(Cin, Cmid, H, W, k, s, dtype), k);
}
void wave_pack_params(const float* s0, const float* b0, const float* dww, const float* s1, const float* b1, int Cin, int Cmid, int k, int s,
                      int dtype, int H, int W, float* dst) {
    const WavePlan p = wave_plan(Cin, Cmid, H, W, k, s, dtype);
    const float L2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const size_t pf = wave_chunk_floats(p, k);
    if (p.mx) {
        const int lo = (k - 1) / 2;
        static const int off[2][4] = {{0, 1, 2, 3}, {-2, -1, 4, 5}};
        for (int ch = 0; ch < Cmid / 16; ++ch) {
            float* d = dst + (size_t)ch * pf;
            for (int c = 0; c < 16; ++c) { const int cc = ch * 16 + c; d[c] = s0[cc]; d[16 + c] = b0[cc]; d[32 + c] = s1[cc]; d[48 + c] = b1[cc]; }
            uint16_t* a = (uint16_t*)(d + 64);
            for (int ky = 0; ky < k; ++ky)
                for (int m = 0; m < 2; ++m)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int q = 0; q < 4; ++q) {
                            const int i = lane & 3, cc = ch * 16 + (lane >> 2), kx = off[m][q] - i + lo;
                            float w = 0.f;
                            if (kx >= 0 && kx < k) w = dww[(size_t)(p.transposed ? kx * k + ky : ky * k + kx) * Cmid + cc];
                            a[(((size_t)ky * 2 + m) * 64 + lane) * 4 + q] = wave_f16_bits(w);      // fp16 in both 16-bit modes (the tap MFMAs' operand type)
                        }
        }
        return;
    }
    for (int ch = 0; ch < Cmid / 16; ++ch)
        for (int c = 0; c < 16; ++c) {
            float* d = dst + (size_t)ch * pf + c;
            const int cc = ch * 16 + c;
            d[0 * 16] = s0[cc] * L2E; d[1 * 16] = b0[cc] * L2E; d[2 * 16] = s1[cc]; d[3 * 16] = b1[cc];
            for (int t = 0; t < k * k; ++t) {
                const int tsrc = p.transposed ? (t % k) * k + t / k : t;     // walked (ky, kx) -> stored w[ky][kx]
                d[(4 + t) * 16] = dww[(size_t)tsrc * Cmid + cc] * LN2;
            }
        }
}
int wave_input_perm_lp(int Cin, int Cmid, int k, int s, int dtype, int H, int W) {
    if (H <= 0 || dtype == COSY_F32) return 0;
    const WavePlan p = wave_plan(Cin, Cmid, H, W, k, s, dtype);
    if (!p.ok || p.mx || p.transposed || !p.fullw || p.ppl < 2 || (p.ppl & (p.ppl - 1))) return 0;      // full rows of 16 * 2^lp pixels (the uniform-offset loads)
    int lp = 0;
    while ((1 << lp) < p.ppl) ++lp;
    return lp;
}
// does this block's wave front read its input at the full rate of the vector-memory address path from the chunked layout [sample][C/16][H*W][16]?  Yes when the 16
// lanes of a fragment are 16 NEIGHBOURING pixels of the walk (32 bytes apart: 4 lanes per 128-byte line): the matrix-pipe form (fragment = a 16-pixel segment) and the
// fp32-FMA form with one pixel per lane.  (A lane's run of P > 1 pixels puts the lanes 32 P bytes apart -- P = 4 is the 128-byte stride that costs four cycles per quad.)
bool wave_input_chunk_ok(int Cin, int Cmid, int k, int s, int dtype, int H, int W) {
    if (H <= 0 || dtype == COSY_F32) return false;
    const WavePlan p = wave_plan(Cin, Cmid, H, W, k, s, dtype);
    return p.ok && (p.mx || p.ppl == 1);
}
bool wave_taps_on_mfma(int Cin, int Cmid, int k, int s, int dtype, int H, int W) {
    if (H <= 0) return false;
    return wave_plan(Cin, Cmid, H, W, k, s, dtype).mx;
}
bool wave_supported(int Cin, int Cmid, int k, int s, int dtype, int H, int W) {
    if (H <= 0) return false;
    return wave_plan(Cin, Cmid, H, W, k, s, dtype).ok;
}
bool wave_walks_columns(int Cin, int Cmid, int k, int s, int dtype, int H, int W) {
    if (H <= 0) return false;
    const WavePlan p = wave_plan(Cin, Cmid, H, W, k, s, dtype);
    return p.ok && p.transposed;
}
int wave_max_tiles() { return WAVE_MAX_RSPLIT; }
void wave_kernel_name(int Cin, int Cmid, int k, int s, int dtype, int H, int W, char* buf, size_t n) {
    const WavePlan p = wave_plan(Cin, Cmid, H, W, k, s, dtype);
    int mw = 0;
#define X(KS, S, KBN, PPL, NI, FW, MW, RSP) if (k == KS && s == S && p.kbn == KBN && p.ppl == PPL && p.ni == NI && p.fullw == FW) mw = MW;
    if (dtype == COSY_F32) { COSY_WAVE_VARIANTS_F32(X) } else { COSY_WAVE_VARIANTS(X) }
#undef X
    snprintf(buf, n, "mbconv_wave_kernel<%s, %d, %d, %d, %d, %d, %s, %d>", dtype == COSY_F32 ? "float" : dtype == COSY_BF16 ? "__bf16" : "_Float16", k, s, p.kbn, p.ppl,
             p.ni, p.fullw ? "true" : "false", mw);
}

template <typename T, int KS, int S, int KBN, int PPL, int NI, bool FW, int MW, int RSP>
static int launch_wave_k(WaveKArgs k, int* n_tiles_out, hipStream_t s) {
    constexpr int HL = __is_same(T, bf16_t) ? 2 : 1;      // bf16: hi + lo weight fragments
    size_t lds = (size_t)4 * ((wave_mx(sizeof(T), KS, S, PPL, FW) ? 0 : (4 + KS * KS) * 16 * NI * sizeof(float)) + (wave_wlds(KBN * HL, MW) ? NI * KBN * HL * 1024 : 0));
    k.dbg = tune_int("COSY_WAVE_DBG", 0);
    k.stamps = nullptr; k.stamp_stride = 1; k.stamp_slots = 0;
#ifdef COSY_TUNE
    // COSY_WAVE_STAMP_PTR = device address of a (slots, WAVE_STAMPS_MAX + 2) uint64 buffer; only launches whose Cmid equals COSY_WAVE_STAMP_CMID
    // record (profiles/exp/wave_timeline.py sets both around one forward)
    if (const char* sp = getenv("COSY_WAVE_STAMP_PTR")) {
        const char* sc = getenv("COSY_WAVE_STAMP_CMID");
        if (sc && atoi(sc) == k.Cmid) {
            k.stamps = (unsigned long long*)strtoull(sp, nullptr, 0);
            k.stamp_stride = getenv("COSY_WAVE_STAMP_STRIDE") ? atoi(getenv("COSY_WAVE_STAMP_STRIDE")) : 64;
            k.stamp_slots = getenv("COSY_WAVE_STAMP_SLOTS") ? atoi(getenv("COSY_WAVE_STAMP_SLOTS")) : 64;
        }
    }
#endif
    k.rsplit = tune_int("COSY_WAVE_RSPLIT", RSP);
    if (k.rsplit < 1) k.rsplit = 1;
    if (k.rsplit > WAVE_MAX_RSPLIT) k.rsplit = WAVE_MAX_RSPLIT;
    while (k.rsplit > 1 && cdiv(k.Ho, k.rsplit) < 8) --k.rsplit;
    k.rows_per = cdiv(k.Ho, k.rsplit);
    *n_tiles_out = k.rsplit;
    const long jobs_per_xcd = (long)cdiv(k.B, 8) * k.nchunks * k.rsplit;
    k.wpb = tune_int("COSY_WAVE_WPB", 4);
    if (k.wpb != 1 && k.wpb != 2) k.wpb = 4;
    lds = lds / 4 * k.wpb;
#ifdef COSY_TUNE
    lds += (size_t)tune_int("COSY_WAVE_LDS_PAD", 0);      // experiment: fewer resident workgroups per CU
#endif
    const dim3 grid((unsigned)(cdiv(jobs_per_xcd, k.wpb) * 8)), block(64 * k.wpb);
#ifdef COSY_TUNE
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)mbconv_wave_kernel<T, KS, S, KBN, PPL, NI, FW, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
    hipLaunchKernelGGL((mbconv_wave_kernel<T, KS, S, KBN, PPL, NI, FW, MW>), grid, block, lds, s, k);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
template <typename T>
static int launch_wave_t(const FuseArgs& a, int* n_tiles_out, hipStream_t s) {
    const WavePlan p = wave_plan(a.Cin, a.Cmid, a.H, a.W, a.k, a.s, sizeof(T) == 4 ? COSY_F32 : COSY_BF16);
    COSY_REQUIRE(p.ok, "mbconv_wave: unsupported shape Cin=%d Cmid=%d %dx%d k=%d s=%d", a.Cin, a.Cmid, a.H, a.W, a.k, a.s);
    WaveKArgs k;
    k.X = a.X; k.Wp = a.Wp; k.wparams = a.wparams; k.D = a.D; k.partial = a.partial;
    COSY_REQUIRE(a.wparams != nullptr, "mbconv_wave: packed parameters missing (wave_pack_params)%s", "");
    k.zeros = a.zeros; k.B = a.B; k.H = a.H; k.W = a.W; k.Cin = a.Cin; k.Cmid = a.Cmid; k.Ho = a.Ho; k.Wo = a.Wo;
    // strides of the walked axes through the stored map: a pixel step / a row step of the walk is a y or an x step of the map (transposed walk: rows =
    // the map's columns), and the map is stored row-major (y * W + x) or column-major (x * H + y; FuseArgs::x_colmajor / d_colmajor)
    const int px = (a.x_chunked ? 16 : a.Cin) * (int)sizeof(T);
    k.xs_chunk = a.x_chunked ? a.H * a.W * 16 * (int)sizeof(T) : 16 * (int)sizeof(T);
    k.xs_sample = a.x_chunked ? (long)((a.Cin + 15) >> 4) * a.H * a.W * 16 : (long)a.H * a.W * a.Cin;
    k.x_perm = a.x_chunked && a.x_perm;
    COSY_REQUIRE(!k.x_perm || (wave_input_perm_lp(a.Cin, a.Cmid, a.k, a.s, sizeof(T) == 4 ? COSY_F32 : COSY_BF16, a.H, a.W) > 0 && !a.x_colmajor), "mbconv_wave: permuted input rows need full power-of-two rows (%dx%d)", a.H, a.W);
    const int x_dx = a.x_colmajor ? a.H * px : px, x_dy = a.x_colmajor ? px : a.W * px;          // bytes per x step / y step of the block input
    const int d_dx = a.d_colmajor ? a.Ho * 16 : 16, d_dy = a.d_colmajor ? 16 : a.Wo * 16;       // elements per x step / y step inside a D chunk
    k.xs_pix = x_dx; k.xs_row = x_dy; k.ds_pix = d_dx; k.ds_row = d_dy;
    if (p.transposed) {
        k.H = a.W; k.W = a.H; k.Ho = a.Wo; k.Wo = a.Ho;
        k.xs_pix = x_dy; k.xs_row = x_dx; k.ds_pix = d_dy; k.ds_row = d_dx;
    }
    k.nkb_total = (p.kbn + 1) & ~1; k.nchunks = a.Cmid / (16 * p.ni); k.rsplit = 1; k.rows_per = a.Ho;
    const int ks_ = a.k, st_ = a.s, kbn_ = p.kbn, ppl_ = p.ppl, ni_ = p.ni;
    const bool fw_ = p.fullw;
#define X(KS, S, KBN, PPL, NI, FW, MW, RSP) \
    if (ks_ == KS && st_ == S && kbn_ == KBN && ppl_ == PPL && ni_ == NI && fw_ == FW) return launch_wave_k<T, KS, S, KBN, PPL, NI, FW, MW, RSP>(k, n_tiles_out, s);
    if constexpr (sizeof(T) == 4) { COSY_WAVE_VARIANTS_F32(X) } else { COSY_WAVE_VARIANTS(X) }
#undef X
    set_error("mbconv_wave: variant not built");
    return COSY_EINVAL;
}
// n_tiles_out: number of partial-sum tiles per sample this launch wrote (row bands), for the squeeze-excite kernel
int launch_mbconv_wave(const FuseArgs& a, int dtype, int* n_tiles_out, hipStream_t s) {
    *n_tiles_out = 1;
    if (a.B == 0) return COSY_OK;
    if (dtype == COSY_F32) return launch_wave_t<float>(a, n_tiles_out, s);
    if (dtype == COSY_BF16) return launch_wave_t<bf16_t>(a, n_tiles_out, s);
    return launch_wave_t<f16_t>(a, n_tiles_out, s);
}

}  // namespace cosy
