// EfficientNet-B3 backbone kernels for gfx950 (wave64, MFMA).  Activations are NHWC in
// T = bf16 (throughput) or fp32 (parity); accumulation is always fp32.
//
//   pw_gemm   : 1x1 convolutions as GEMM on the matrix cores.  Operands are swapped (weights are the
//               MFMA "A" operand, activations the "B" operand) so that each lane ends up holding
//               contiguous output channels of one pixel -> wide NHWC stores; fragments travel
//               global -> VGPR -> LDS in MFMA-fragment order ("lane-linear" 1 KiB blocks), which makes both
//               the LDS writes and the ds_read_b128 fragment reads conflict-free without padding.
//               Epilogue: folded BN scale/bias, optional SiLU, optional residual; prologue: optional
//               squeeze-excite gate on the activation rows.
//   dwconv    : (kernels_dw.hip) depthwise kxk (k 3/5, stride 1/2) with static "same" padding folded into index math,
//               BN+SiLU epilogue and deterministic per-tile partial sums for the SE squeeze.
//   se        : squeeze -> reduce FC -> swish -> expand FC -> sigmoid gate, one workgroup per sample.
//   stem      : dense 3x3 stride-2 conv 6->40 + BN + SiLU.
//   pool_fc   : global average pool + Linear(1536, 9).
#include "net_device.h"
#include <type_traits>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace cosy {


// ==========================================================================================
// pointwise conv GEMM
// ==========================================================================================
PwCfg pw_choose_cfg(int N) {
    // BN options: 128 (NI4,WN2) 96 (NI3,WN2) 48 (NI3,WN1) 32 (NI2,WN1).  Cost = padded columns divided by a
    // rough efficiency of the tile shape (narrow tiles re-read the activation rows more often).
    const PwCfg opts[4] = {{4, 2}, {3, 2}, {3, 1}, {2, 1}};
    const double eff[4] = {1.0, 0.9, 0.65, 0.5};
    PwCfg best = opts[0];
    double best_cost = -1;
    for (int i = 0; i < 4; ++i) {
        const int bn = pw_bn(opts[i]);
        const double cost = (double)cdiv(N, bn) * bn / eff[i];
        if (best_cost < 0 || cost < best_cost - 1e-9) { best_cost = cost; best = opts[i]; }
    }
    return best;
}
// Tile choice for the late layers (2-byte types).  Phase knock-out timing (COSY_PW_DBG) showed these GEMMs bound by the
// L2 -> LDS DMA stream, not by MFMA or HBM: with everything but the DMA ring and its barriers removed a late GEMM still
// takes 2/3 of its time.  Per launch the ring moves NT * M*K (activation rows, once per n-tile) + MT * Npad*K (weights,
// once per m-tile) elements, so a tile that covers N in ONE pass saves a whole pass over the activations -- as long as it
// keeps two workgroups per CU.  Measured (256 crops): N = 136 with BN = 160 (NI5,WN2) instead of 2 x 96: 72 -> 55 us
// (b14-17 project), 51 -> 41 us (b13).  Wider tiles lose: BN = 256 (NI8,WN2; 84 KB of LDS, one workgroup per CU) 58 -> 86
// us (82 us with a 2-stage ring that restores two workgroups per CU), 256 x 128 (NI8,WN1) 66 -> 107 us, 2 x 192 for N = 384
// 61 -> 70 us.
PwCfg pw_choose_cfg_late(int K, int N, int HW, bool gated, int dtype) {
    // 8-wave tiles (128 x 128 as 2 x 4 waves of 64 x 32): each wave issues half the LDS-DMAs, gate multiplies and MFMAs of a
    // k-step and two waves share a SIMD, so one wave's gate arithmetic / fragment reads overlap the other's MFMAs (the
    // 4-wave tile runs them back to back).  Measured at 256 crops: N = 232 project 44 -> 36 us, N = 816 expand 55 -> 44 us,
    // head (N = 1536) 43 -> 38 us, N = 384 unchanged; N = 96 unchanged and N = 136 slower (63 us vs 45 us with the one-pass
    // 160-column tile) -> only for N >= 192.
    static const int pw8 = tune_int("COSY_PW8", 1);
    // K >= 1024 (blocks 19-25 project): 16 waves in two K-groups (see the kernel); N = 384 as two 192-column tiles, so that 256 crops are
    // 256 workgroups = one per CU (three 128-column tiles are 384 = a second round on half the CUs)
    // Measured at 256 crops, fp16 (profiles/r04_gemm_splitk.txt): blocks 19-23 33.3 -> 28.4 us, block 24 41.5 -> 39.8, block 25 63.6 -> 57.2.
    // A 128 x 256 full-N tile (A read once, 128 workgroups) took 65 us for block 19: these layers run at ~25 GB/s of operands per CU (32 KB per 1.3 us step here,
    // 16 KB per 0.75 us with 8 waves), not at a chip-wide byte count -- fewer, larger tiles lose.  (Round 6: that rate is NOT the LDS-DMA path's, which fills at
    // 137-144 GB/s per CU from the L2 -- profiles/r06_dma_rate.txt; a four-stage ring and an 8-wave 64 x 64-per-wave split-K tile were measured and lose too:
    // profiles/r06_gemm_kloop.txt, r06_dead_ends.txt.)
    static const int pw16 = tune_int("COSY_PW16", 1);
    // Maps whose pixel count is not a multiple of 64 (240x320 crops: 8x10, 15x20, ...) take the row-side gate.  Measured there (256 crops, fp16,
    // profiles/r04_rowgate_tiles.txt): the split-K tile loses (blocks 19-23 51 -> 58 us, 24 / 25 51 / 79 -> 74 / 109); the 8-wave tile wins for
    // N = 384 (blocks 24 / 25: 51 / 79 -> 47 / 74 us); N <= 256 goes to the 4-wave tile with 32 rows per wave (pw_mi below), which wins more.
    static const int pw16_rg = tune_int("COSY_PW16_RG", 0), pw8_rg = tune_int("COSY_PW8_RG", 1);
    // bf16 (round 6): every weight fragment is a hi + lo pair (pw_hl), so a ring stage carries twice the weight blocks: the 16-wave split-K tile runs a TWO-stage ring
    // there (2 x (8 + 16..24) KB x 2 stages + the gate rows <= 158 KB; three stages do not fit)
    // (bf16: 128-column tiles for N = 384 too -- the 192-column one does not fit 128 registers with the weight pairs)
    if (pw16 && gated && K >= 1024 && (HW % 64 == 0 || pw16_rg) && N >= 192) return N > 256 && N <= 384 && dtype != COSY_BF16 ? PwCfg{3, 4, 16, 2} : PwCfg{2, 4, 16, 2};
    if (pw8 && K >= 128 && N >= 192 && (!gated || HW % 64 == 0 || (pw8_rg && N > 256))) return PwCfg{2, 4, 8};
    static const int wide = tune_int("COSY_PW_WIDE", 1);
    if (wide && K >= 96 && HW >= 64 && N > 128 && N <= 160) return PwCfg{5, 2};
    return pw_choose_cfg(N);
}
int pw_kb(int dtype) { return dtype == COSY_F32 ? 16 : 32; }
// where the project GEMM applies the squeeze-excite gate for a map of HW pixels: to the weight fragments, or to the activation rows
bool pw_gate_on_weights(int HW, int dtype) { return HW % 64 == 0 && dtype != COSY_BF16; }
// bf16 weights are ERROR-COMPENSATED PAIRS (round 6): hi = bf16(w), lo = bf16(w - hi), stored as two consecutive fragment blocks per k-block; every MFMA kernel
// multiplies the same activation fragment with both (fp32 accumulation), so what meets the activations is hi + lo = w to 16 significant bits.  bf16's 8-bit
// weights were what kept the type BASELINE configs[1] names outside north_star's 1e-4 pose bound (1.7e-4; with the pairs 3e-5, below fp16's: profiles/r06_bf16.txt).
int pw_hl(int dtype) { return dtype == COSY_BF16 ? 2 : 1; }
static inline uint16_t f32_to_f16_host(float f) { _Float16 h = (_Float16)(f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f)); uint16_t u; memcpy(&u, &h, 2); return u; }
static int pw_nkb_total(int K, int dtype) { int n = cdiv(K, pw_kb(dtype)); return (n + 1) & ~1; }
size_t pw_packed_elems(int K, int N, PwCfg c, int dtype, int hl) {      // hl: fragment blocks per k-block; < 0 = the type's own (pw_hl), 1 = single values (the stem kernels)
    const int epl = dtype == COSY_F32 ? 4 : 8;
    return (size_t)cdiv(N, pw_bn(c)) * c.NI * c.WN * pw_nkb_total(K, dtype) * (hl < 0 ? pw_hl(dtype) : hl) * 64 * epl;
}
static inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_bits_to_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
void pw_pack_weights(const float* w, int K, int N, PwCfg c, int dtype, void* dst, int hl_) {
    const int epl = dtype == COSY_F32 ? 4 : 8, kb = pw_kb(dtype), nkb = pw_nkb_total(K, dtype), hl = hl_ < 0 ? pw_hl(dtype) : hl_;
    const int NW = c.NI * c.WN, BN = pw_bn(c), NT = cdiv(N, BN);
    size_t idx = 0;
    for (int nt = 0; nt < NT; ++nt)
        for (int nb = 0; nb < NW; ++nb)
            for (int kbi = 0; kbi < nkb; ++kbi)
                for (int h = 0; h < hl; ++h)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < epl; ++e, ++idx) {
                            const int i = lane & 15, kg = lane >> 4, wn = nb / c.NI, ni = nb % c.NI;
                            const int n = nt * BN + wn * 16 * c.NI + (i >> 2) * 4 * c.NI + ni * 4 + (i & 3);
                            const int k = kbi * kb + kg * epl + e;
                            const float v = (n < N && k < K) ? w[(size_t)n * K + k] : 0.f;
                            if (dtype == COSY_F32) ((float*)dst)[idx] = v;
                            else if (dtype == COSY_F16) ((uint16_t*)dst)[idx] = f32_to_f16_host(v);
                            else {
                                const uint16_t hi = f32_to_bf16_host(v);
                                ((uint16_t*)dst)[idx] = h == 0 ? hi : f32_to_bf16_host(v - bf16_bits_to_f32(hi));
                            }
                        }
}

// element offsets in the chunked activation layout [sample][ceil(C/16)][HW][16]: row m = (sample, pixel) / channel c
__device__ __forceinline__ size_t chunked_row(int m, int HW, int C, int lw = 0, int lp = 0) {
    const int bs = m / HW;
    int pix = m - bs * HW;
    if (lw) {      // pixel x of a 2^lw-pixel row -> position (x mod 2^lp) * 16 + x / 2^lp
        const int x = pix & ((1 << lw) - 1);
        pix = (pix - x) + ((x & ((1 << lp) - 1)) << 4) + (x >> lp);
    }
    return ((size_t)bs * (size_t)((C + 15) >> 4) * (size_t)HW + (size_t)pix) * 16;
}
__device__ __forceinline__ size_t chunked_col(int c, int HW) { return (size_t)(c >> 4) * (size_t)HW * 16 + (size_t)(c & 15); }

// samples an m-tile of bm rows can touch when a sample has hw rows
static inline int pw_gate_nsamp(int bm, int hw) { return (bm + hw - 2) / hw + 1; }
struct PwKArgs {
    const void* A; const void* Wp; void* out; const float* scale; const float* bias; const void* res; const float* gate;
    int M, K, N, HW, silu, MT, NT, nkb_total, nkb_valid;
    const void* zeros;
    int nsamp, rowgate;   // DMA kernel gate: samples under one m-tile; 1 = gate the activation fragments per row
    int a_chunked;        // A is [sample][K/16][HW][16] (see PwArgs)
    int out_chunked, res_chunked;   // out / res are [sample][ceil(N/16)][HW][16] (see PwArgs)
    int out_perm_lw, out_perm_lp;
    int a_nt;             // A is read exactly once (one n-tile) and is large: its DMAs carry the non-temporal hint
    // squeeze-excite computed in the prologue (PwArgs::se_fused): squeeze partial sums (B, se_tiles, K), reduce FC (Cse, K) + bias,
    // expand FC stored (Cse, K) + bias (K); se_wr == nullptr: the gate rows are read from `gate` (a squeeze-excite kernel wrote them)
    const float* se_partial; const float* se_wr; const float* se_br; const float* se_we; const float* se_be; float* gate_out;
    int se_tiles, Cse; float inv_hw;
};

// ------------------------------------------------------------------------------------------
// Pipelined variant: both operands arrive by asynchronous global->LDS DMA (global_load_lds, 16 B/lane, one
// instruction per 1 KiB fragment block: the blocks are lane-linear, which is exactly the DMA's destination rule)
// into an NS-stage ring of 32-deep k-blocks.  Per k-block: counted s_waitcnt vmcnt (never 0 in steady state, the
// next NS-2 k-blocks stay in flight across the barrier), ONE raw s_barrier, issue the DMA for k-block kb+NS-1, MFMA.
// No staging VGPRs, no ds_write, no ordinary global loads inside the loop (they would make the compiler drain the
// DMA queue).  Every wave issues the same number L of DMA instructions per k-block (surplus ones land in a dummy
// block) so that the vmcnt immediates are compile-time constants.
// SE gate (project convs): out = sum_k D[m,k] g[b,k] W[n,k]; the gate is folded into the WEIGHT fragments
// (W[n,k]*g[b,k]) at fragment-read time from a gate row staged in LDS when the 64 pixel rows of a wave belong to one
// sample (HW % 64 == 0); otherwise (240x320 input) each lane scales its ACTIVATION fragments with its own row's gate.
// ------------------------------------------------------------------------------------------
// KG = 2 (round 4, the K >= 1024 project convs of the 8x8 maps): IN-WORKGROUP SPLIT-K.  Those layers have only M / 128 = 128 m-tiles
// at 256 crops, i.e. ONE 8-wave workgroup per CU (2 waves per SIMD) that spends a k-step waiting -- barrier, DMA issue, fragment reads,
// 8 MFMAs, in sequence: 0.75 us per k-step against 0.11 us of matrix-pipe time.  With KG = 2 the workgroup has 16 waves in two K-groups;
// a ring stage holds TWO k-blocks, group g multiplies k-block 2 * step + g of the same output tile, so a barrier interval carries twice
// the MFMA work at 4 waves per SIMD and the loop has half as many barriers.  After the loop group 1 hands its accumulators to group 0
// through the (now free) ring -- a fixed-order fp32 add, deterministic -- and group 0 runs the epilogue.  No global partials, no combine
// launch, the activations are still read once per n-tile.
// Waves per SIMD every tile shape has to reach (what its LDS ring is sized for).  Several shapes sit one register below a cliff -- <5,2> at
// 176 VGPRs + 80 AGPRs = 256: a single extra register in a prologue halves the occupancy (round 4: blocks 13-17's project GEMMs went
// 54 -> 91 us that way, unnoticed for a few commits).  NOT enforced through the second __launch_bounds__ argument for the 4-wave tiles: with it
// hipcc gives up the VGPR + AGPR split (accumulators in AGPRs) and the <5,2> tile runs 59 instead of 53 us; the bound is held by
// tests/test_build_isa.py on the resource table the build writes (cosypose_amd/build.py: kernel_resources).
constexpr int pw_min_waves(int NI, int MI, int NWV) {
    return NWV >= 8 ? 4 : MI < 4 ? 1 : NI <= 2 ? 4 : NI == 3 ? 3 : 2;
}
constexpr int pw_bound_waves(int NI, int MI, int NWV) { return NWV >= 8 ? 4 : 1; }
// SEF: the squeeze-excite prologue is compiled in (PwArgs::se_fused).  A template parameter, not a run-time branch: carried by every gated
// instantiation it cost the streaming project GEMMs of blocks 0-4, which never use it, 4-10 us per launch in registers and code.
template <typename T, int NI, int WN, int NS, bool GATE, int MI, int NWV = 4, int KG = 1, bool SEF = false>
__global__ __launch_bounds__(NWV * 64, pw_bound_waves(NI, MI, NWV)) void pw_gemm_dma_kernel(PwKArgs a) {
    using D = DT<T>;
    using raw_t = typename D::raw_t;
    constexpr int EPL = D::EPL, KB = D::KB;
    constexpr int GW = NWV / KG;              // waves of one K-group
    constexpr int WM = GW / WN, BM = 16 * MI * WM, BN = 16 * NI * WN;
    constexpr int HL = __is_same(T, bf16_t) ? 2 : 1;        // bf16: weight fragments are hi + lo pairs (pw_hl), two consecutive blocks
    constexpr int NA = BM / 16, NW = NI * WN, NB = NA + NW * HL;
    constexpr int SB = KG * NB;               // 1 KiB blocks of one ring stage (KG k-blocks)
    constexpr int L = (SB + NWV - 1) / NWV;   // DMA instructions per wave per stage
    static_assert(GW * KG == NWV && WM * WN == GW, "wave grid");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dummy = lds + NS * SB * 1024;
    float* sbl = (float*)(dummy + 1024);  // [256] BatchNorm scale + [256] bias of this n-tile's columns (DMA'd in front of the first stage, read by the epilogue)
    float* gl = (float*)(dummy + 3072);   // GATE: [nsamp][Kpad] gate rows of the samples under this m-tile

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgp = KG == 1 ? 0 : wave / GW, wq = KG == 1 ? wave : wave % GW;      // K-group, wave inside the group (constants for the plain tiles)
    const int wm = wq / WN, wn = wq % WN;
    const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
    const int mt = (jj / a.NT) * 8 + xcd, nt = jj % a.NT;
    if (mt >= a.MT) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const T* __restrict__ A = (const T*)a.A;
    const T* __restrict__ Wp = (const T*)a.Wp;
    const int K = a.K, M = a.M, N = a.N;
    const int row = lane & 15, kg = lane >> 4;
    const int Kpad = a.nkb_total * KB;

    constexpr bool HALF_GATE = sizeof(T) == 2 && !__is_same(T, bf16_t);   // fp16: the gate multiplies as packed halves
    int gsel = 0, grow[MI];
    // The operands' DMA sources, one per DMA slot of this wave (slot i = block i * NWV + wave of a stage): a per-lane pointer that ADVANCES by a constant per
    // stage.  (Round 6: the first version recomputed every source in the k-loop -- tile bounds, chunk arithmetic, block kind, ~250 mostly scalar / branch
    // instructions per step and wave for 8-20 MFMAs; knock-out timing of the split-K tiles of blocks 19-25 showed the loop's SHELL at 60 % of a step,
    // profiles/r06_gemm_kloop.txt.)  Rows beyond M read the last row again (finite values into accumulator rows that are never stored), the k tail of the
    // last k-block reads the zero page; W is packed with zero padding.
    const T* src[L];          // this lane's source of slot i in the NEXT stage to be issued
    int sstep[L];             // its advance per stage in elements (wave-uniform: a slot is an A block or a W block for the whole wave)
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int sblk = i * NWV + wave, blk = KG == 1 ? sblk : sblk % NB, kofs = KG == 1 ? 0 : sblk / NB;
        if (blk < NA) {
            const int m = m0 + blk * 16 + row, k0 = kofs * KB + kg * EPL;
            const int mc = min(m, M - 1), bs = mc / a.HW;
            const size_t o = a.a_chunked ? ((size_t)bs * (size_t)((K + 15) >> 4) * (size_t)a.HW + (size_t)(mc - bs * a.HW) + (size_t)(k0 >> 4) * (size_t)a.HW) * 16 + (k0 & 15)
                                         : (size_t)mc * K + k0;
            src[i] = A + o;
            sstep[i] = a.a_chunked ? KG * KB * a.HW : KG * KB;
        } else {
            src[i] = Wp + (((size_t)(nt * NW + (blk - NA) / HL) * a.nkb_total + kofs) * HL + (blk - NA) % HL) * 64 * EPL + lane * EPL;
            sstep[i] = KG * HL * 64 * EPL;
        }
    }
    const bool ktail = K % KB != 0;      // the last k-block is partly beyond K
    // The k-loop of the 8- / 16-wave tiles is bound by SCALAR issue: a CU's four SIMDs share one scalar unit (one SALU instruction per SIMD every four cycles),
    // and the general form below spends ~90 of them per step and wave on stage / validity / destination arithmetic -- 16 waves x 90 = ~1,450 cycles per step against
    // 128 cycles of MFMAs per wave (knock-outs: profiles/r06_gemm_kloop.txt).  Stages whose k-blocks are all inside K (every stage but the last one or two) take
    // issue_fast: no validity, no tail, the destination advances by a constant.
    int wr_off = 0;                      // byte offset of the ring stage the next issue writes
    bool slot_ok[L], slot_nt[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int sblk = i * NWV + wave, blk = KG == 1 ? sblk : sblk % NB;
        slot_ok[i] = sblk < SB; slot_nt[i] = a.a_nt && blk < NA;
    }
    auto issue_fast = [&]() {
#pragma unroll
        for (int i = 0; i < L; ++i) {
            if ((i + 1) * NWV <= SB || slot_ok[i]) {      // (compile-time true for every slot but a partly filled last one)
                char* dst = lds + wr_off + (i * NWV + wave) * 1024;
                if (slot_nt[i])
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a.zeros,
                                                 (__attribute__((address_space(3))) void*)dummy, 16, 0, 0);      // (every wave issues L DMAs per stage: the counted wait)
            }
            src[i] += sstep[i];
        }
        wr_off = wr_off + SB * 1024 == NS * SB * 1024 ? 0 : wr_off + SB * 1024;
    };
    auto issue = [&](int ks) {              // stage ks = the k-blocks ks * KG .. ks * KG + KG - 1; stages are issued in order, once each
        char* st = lds + wr_off;
        wr_off = wr_off + SB * 1024 == NS * SB * 1024 ? 0 : wr_off + SB * 1024;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int sblk = i * NWV + wave, blk = KG == 1 ? sblk : sblk % NB, kb = KG == 1 ? ks : ks * KG + sblk / NB;     // wave-uniform
            const bool v = kb < a.nkb_valid && sblk < SB;
            const T* p = src[i];
            if (ktail && blk < NA && kb == a.nkb_valid - 1) p = kb * KB + kg * EPL < K ? p : (const T*)a.zeros;      // (per lane, last k-block only)
            if (!v) p = (const T*)a.zeros;
            char* dst = v ? st + sblk * 1024 : dummy;
            if (a.a_nt && blk < NA)     // wave-uniform
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 2);      // aux 2 = nt
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            src[i] += sstep[i];
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // BatchNorm scale / bias of the tile's columns -> LDS by two DMAs per wave, OLDER than every operand DMA (so any wait that proves stage 0 proves
    // them; all waves write the same bytes).  The epilogue reads them column group by column group at LDS latency: neither 8 * NI live registers
    // (what made the <5,2> tile need the packed-fp32 forms to stay at 176 VGPRs) nor one dependent L2 round trip per column group.
    {
        static_assert(BN <= 256, "scale / bias staging holds 256 columns");
        const int c4 = min(lane * 4, BN - 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.scale + n0 + c4),
                                         (__attribute__((address_space(3))) void*)sbl, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bias + n0 + c4),
                                         (__attribute__((address_space(3))) void*)(sbl + 256), 16, 0, 0);
    }
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0);
    // Behind the first DMAs, in the same latency shadow: the residual rows this lane will add in the epilogue (2-byte types:
    // 2 registers per 4 channels) and the SE gate rows.  Fetched in the epilogue / before the first DMA (round 1) each of them
    // exposed one more full memory latency per tile -- a third of the time of the 1-2 k-block layers.  They are ordinary loads
    // YOUNGER than the DMAs and are waited for with the vmcnt(0) of the barrier below, so no count spans both kinds.
    typedef T rv4_t __attribute__((ext_vector_type(4)));
    constexpr bool RES_PREFETCH = sizeof(T) == 2;
    constexpr bool CHK = KG == 1 && sizeof(T) == 2;      // tiles that can meet the chunked output / residual layout (the split-K tiles of the 8x8 maps never do)
    rv4_t rpre[RES_PREFETCH ? MI : 1][RES_PREFETCH ? NI : 1];
    auto prefetch_residual = [&]() {
        if constexpr (RES_PREFETCH) {
            if (a.res && kgp == 0) {
                const int nl_ = n0 + wn * 16 * NI + kg * 4 * NI;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int m = min(m0 + (wm * MI + mi) * 16 + row, M - 1);
                    const size_t rbase = CHK && a.res_chunked ? chunked_row(m, a.HW, N) : (size_t)m * N;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int c = min(nl_ + ni * 4, N - 4);
                        rpre[mi][ni] = *(const rv4_t*)((const T*)a.res + rbase + (CHK && a.res_chunked ? chunked_col(c, a.HW) : (size_t)c));
                    }
                }
            }
        }
    };
    if constexpr (KG == 1) prefetch_residual();     // split-K tiles (128 registers per wave): behind the loop, in the shadow of the hand-over
    if constexpr (GATE) {
        const int b_first = m0 / a.HW;
        if constexpr (SEF) {
            // ---- squeeze-excite in the prologue (efficientnet.py:85-88): no se launch between the front kernel and this GEMM.  Every
            // workgroup computes the gate of each sample under its m-tile: pooled = sum of the squeeze partials / HW -> reduce FC + bias
            // -> swish -> expand FC + bias -> sigmoid, fp32, fixed summation order (deterministic; a sample's gate depends on nothing
            // but that sample: batch-invariant).  The FC matrices are tiny next to the tile's operands (blocks 0-18: 3-220 KB, L2 hits
            // shared by all workgroups) and their loads are issued in batches of SEB independent 16-byte loads, so the prologue costs a
            // few L2 latencies in the shadow of the first DMAs -- against a dependent 8-15 us launch (+ its launch gap) per block.
            constexpr int NT = NWV * 64;
            constexpr int SEB = NI * MI >= 20 ? 16 : NI * MI >= 16 ? 12 : 8;     // loads in flight per thread: what the tile's own budget leaves
            float* pooled = gl + a.nsamp * Kpad;
            float* redv = pooled + Kpad;
            const int C4 = K >> 2, Cse = a.Cse;
            const int P = Cse * 8 <= NT ? 8 : Cse * 4 <= NT ? 4 : Cse * 2 <= NT ? 2 : 1;     // lanes per reduce-FC output
            const int n_under = (min(m0 + BM, M) - 1) / a.HW - b_first + 1;      // samples that own a row of this m-tile (<= nsamp)
            for (int i = tid + n_under * (Kpad >> 2); i < a.nsamp * (Kpad >> 2); i += NT)     // rows no pixel selects stay finite (0)
                ((f32x4*)gl)[HALF_GATE ? i >> 1 : i] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int sidx = 0; sidx < n_under; ++sidx) {
                const int b = b_first + sidx;
                constexpr bool live = true;
                if (live) {
                    const f32x4* part = (const f32x4*)(a.se_partial + (size_t)b * a.se_tiles * K);
                    for (int c = tid; c < C4; c += NT) {
                        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
                        int t = 0;
                        for (; t + 1 < a.se_tiles; t += 2) { s0 += part[(size_t)t * C4 + c]; s1 += part[(size_t)(t + 1) * C4 + c]; }
                        if (t < a.se_tiles) s0 += part[(size_t)t * C4 + c];
                        ((f32x4*)pooled)[c] = (s0 + s1) * a.inv_hw;
                    }
                }
                __syncthreads();
                if (live) {
                    const int slots = NT / P, prt = tid & (P - 1);
                    for (int j0 = 0; j0 < Cse; j0 += slots) {
                        const int j = j0 + tid / P;
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        const f32x4* wr = (const f32x4*)(a.se_wr + (size_t)min(j, Cse - 1) * K);
                        for (int c0 = prt; c0 < C4; c0 += P * SEB) {
                            f32x4 w[SEB];
#pragma unroll
                            for (int u = 0; u < SEB; ++u) w[u] = wr[min(c0 + P * u, C4 - 1)];
#pragma unroll
                            for (int u = 0; u < SEB; ++u)
                                if (c0 + P * u < C4) acc += w[u] * ((const f32x4*)pooled)[c0 + P * u];
                        }
                        float sv = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                        if (P >= 2) sv += __shfl_xor(sv, 1, 64);
                        if (P >= 4) sv += __shfl_xor(sv, 2, 64);
                        if (P >= 8) sv += __shfl_xor(sv, 4, 64);
                        if (prt == 0 && j < Cse) {
                            sv += a.se_br[j];
                            redv[j] = sv * (1.f / (1.f + expf(-sv)));
                        }
                    }
                }
                __syncthreads();
                for (int c = tid; c < (Kpad >> 2); c += NT) {
                    f32x4 g = {0.f, 0.f, 0.f, 0.f};
                    if (live && c < C4) {
                        const f32x4* we = (const f32x4*)a.se_we + c;
                        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
                        for (int j0 = 0; j0 < Cse; j0 += SEB) {
                            f32x4 w[SEB];
#pragma unroll
                            for (int u = 0; u < SEB; ++u) w[u] = we[(size_t)min(j0 + u, Cse - 1) * C4];
#pragma unroll
                            for (int u = 0; u < SEB; u += 2) {
                                if (j0 + u < Cse) s0 += w[u] * redv[j0 + u];
                                if (u + 1 < SEB && j0 + u + 1 < Cse) s1 += w[u + 1] * redv[j0 + u + 1];
                            }
                        }
                        const f32x4 be = ((const f32x4*)a.se_be)[c];
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = 1.f / (1.f + expf(-(s0[e] + s1[e] + be[e])));
                        // the workgroup that holds the sample's first row (first n-tile) publishes the gate: test probes read it
                        if (nt == 0 && (long)b * a.HW >= m0) ((f32x4*)(a.gate_out + (size_t)b * K))[c] = g;
                    }
                    if constexpr (HALF_GATE) {
                        typedef f16_t h4_t __attribute__((ext_vector_type(4)));
                        ((h4_t*)gl)[(sidx * Kpad >> 2) + c] = h4_t{(f16_t)g[0], (f16_t)g[1], (f16_t)g[2], (f16_t)g[3]};
                    } else {
                        ((f32x4*)gl)[(sidx * Kpad >> 2) + c] = g;
                    }
                }
                // the next sample's pooled / redv are written behind barriers that every thread reaches after this phase
            }
        } else
        for (int i = tid; i < a.nsamp * Kpad; i += NWV * 64) {       // the gate rows a squeeze-excite kernel wrote
            const int sidx = i / Kpad, k = i - sidx * Kpad;
            const long mrow = (long)(b_first + sidx) * a.HW;
            const float g = (k < K && mrow < M) ? a.gate[(size_t)(b_first + sidx) * K + k] : 0.f;
            if constexpr (HALF_GATE) ((f16_t*)gl)[i] = (f16_t)g; else gl[i] = g;
        }
        gsel = (m0 + wm * 16 * MI) / a.HW - b_first;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) grow[mi] = min((m0 + (wm * MI + mi) * 16 + row) / a.HW - b_first, a.nsamp - 1) * Kpad;
        __syncthreads();
    }

    const int nks = (a.nkb_valid + KG - 1) / KG;
    const int n_clean = (a.nkb_valid - (ktail ? 1 : 0)) / KG;                  // stages 0 .. n_clean - 1: every k-block inside K, no tail block
    const int n_fast = max(0, n_clean - (NS - 1));                             // steps whose issue (stage ks + NS - 1) is a clean stage
    auto kstep = [&](const int ks, auto fastc) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * L) : "memory");  // stage ks has landed (this wave's part)
        __builtin_amdgcn_s_barrier();                                          // ... and everybody's; stage (ks-1)%NS is free
        asm volatile("" ::: "memory");
        if constexpr (decltype(fastc)::value) issue_fast(); else issue(ks + NS - 1);
        const int kb = ks * KG + kgp;                                          // this K-group's k-block of the stage
        if (KG > 1 && kb >= a.nkb_valid) return;                               // odd number of k-blocks: the last stage is half full (wave-uniform)
        const char* st = lds + ((ks % NS) * SB + kgp * NB) * 1024;
        raw_t fw[NI], fa[MI], fwl[HL == 2 ? NI : 1];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            fw[ni] = *(const raw_t*)(st + (NA + (wn * NI + ni) * HL) * 1024 + lane * 16);
            if constexpr (HL == 2) fwl[ni] = *(const raw_t*)(st + (NA + (wn * NI + ni) * HL + 1) * 1024 + lane * 16);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = *(const raw_t*)(st + (wm * MI + mi) * 1024 + lane * 16);
        if (GATE && a.rowgate) {
            // maps whose pixel count is not a multiple of 64 (240x320 input): a wave's rows straddle samples, so the
            // gate multiplies the ACTIVATION fragments, each lane with the gate row of its own pixel's sample
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if constexpr (HALF_GATE) {
                    const f16x8 g8 = *(const f16x8*)((const f16_t*)gl + grow[mi] + kb * KB + kg * EPL);
                    f16x8 a8 = __builtin_bit_cast(f16x8, fa[mi]);
                    a8 = a8 * g8;
                    fa[mi] = __builtin_bit_cast(raw_t, a8);
                } else {
                    const float* gp = gl + grow[mi] + kb * KB + kg * EPL;
#pragma unroll
                    for (int e0 = 0; e0 < EPL; e0 += 4) {      // four elements at a time, repacked before the next four are unpacked (left alone,
                        float g[4];                            // hipcc keeps all MI * EPL products live: one register too many for the <2,1> tile)
                        load4(gp + e0, g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) fa[mi][e0 + e] = (T)((float)fa[mi][e0 + e] * g[e]);
                        asm volatile("" : "+v"(fa[mi]));
                    }
                }
            }
        } else if constexpr (GATE && HALF_GATE) {
            // 4 v_pk_mul_f16 per fragment instead of unpack / multiply / repack
            const f16x8 g8 = *(const f16x8*)((const f16_t*)gl + gsel * Kpad + kb * KB + kg * EPL);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                f16x8 w8 = __builtin_bit_cast(f16x8, fw[ni]);
                w8 = w8 * g8;
                fw[ni] = __builtin_bit_cast(raw_t, w8);
            }
        } else if constexpr (GATE && HL == 1) {      // (fp32; bf16 always takes the row-side gate: a folded weight would have to be re-split into a pair)
            float g[EPL];
            const float* gp = gl + gsel * Kpad + kb * KB + kg * EPL;
#pragma unroll
            for (int e = 0; e < EPL; e += 4) load4(gp + e, g + e);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                float f[EPL];
                to_f32(fw[ni], f);
#pragma unroll
                for (int e = 0; e < EPL; ++e) f[e] *= g[e];
                from_f32(fw[ni], f);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                mma(acc[mi][ni], fw[ni], fa[mi]);
                if constexpr (HL == 2) mma(acc[mi][ni], fwl[ni], fa[mi]);
            }
    };
    {
        int ks = 0;
        for (; ks < n_fast; ++ks) kstep(ks, std::true_type{});
        for (; ks < nks; ++ks) kstep(ks, std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (KG > 1) {
        // K-group 1 -> K-group 0 through the ring (every wave is past its last fragment read behind this barrier): one fixed-order add
        static_assert(KG == 2 && GW * MI * NI * 1024 <= NS * SB * 1024, "split-K hand-over fits the ring");
        prefetch_residual();
        __syncthreads();
        f32x4* red = (f32x4*)lds + (size_t)wq * MI * NI * 64 + lane;
        if (kgp == 1) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) red[(mi * NI + ni) * 64] = acc[mi][ni];
        }
        __syncthreads();
        if (kgp != 0) return;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += red[(mi * NI + ni) * 64];
    }

    // ---- epilogue: lane holds, for pixel row m, the 4*NI consecutive channels starting at nl.  Column-group major (round 5): the BatchNorm
    // scale / bias of ONE store's channels (4 or 8) are live at a time instead of all 8 * NI of them -- that is what lets the <5,2> tile keep its
    // two waves per SIMD (152 VGPRs beside 80 accumulators; 186-191 before) without packed-fp32 arithmetic (build.NO_PACKED_FP32 holds for the
    // whole library).
    const int nl = n0 + wn * 16 * NI + kg * 4 * NI;
    T* __restrict__ out = (T*)a.out;
    const T* __restrict__ res = (const T*)a.res;
    constexpr int CG = (sizeof(T) == 2 && NI % 2 == 0) ? 2 : 1;      // sub-tiles per store (16-byte stores where a lane's run allows)
#pragma unroll
    for (int n1 = 0; n1 < NI; n1 += CG) {
        float sc[CG * 4], bi[CG * 4];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            load4(sbl + (nl - n0) + (n1 + c) * 4, sc + c * 4);
            load4(sbl + 256 + (nl - n0) + (n1 + c) * 4, bi + c * 4);
        }
        if (nl + n1 * 4 < N) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = m0 + (wm * MI + mi) * 16 + row;
                if (m >= M) continue;
                float y[CG * 4];
#pragma unroll
                for (int c = 0; c < CG; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[mi][n1 + c][r] * sc[c * 4 + r] + bi[c * 4 + r];
                        if (a.silu) v = v * sigmoid_t<T>(v);
                        y[c * 4 + r] = v;
                    }
                // a lane's run of 4 CG channels starts at a multiple of 4 CG: it never straddles a 16-channel chunk
                const int c0 = nl + n1 * 4;
                const size_t o = CHK && a.out_chunked ? chunked_row(m, a.HW, N, a.out_perm_lw, a.out_perm_lp) + chunked_col(c0, a.HW) : (size_t)m * N + c0;
                if (res) {
#pragma unroll
                    for (int c = 0; c < CG; ++c) {
                        float rv[4];
                        if constexpr (RES_PREFETCH) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) rv[r] = (float)rpre[mi][n1 + c][r];
                        } else {
                            load4(res + (CHK && a.res_chunked ? chunked_row(m, a.HW, N) + chunked_col(c0 + c * 4, a.HW) : (size_t)m * N + c0 + c * 4), rv);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[c * 4 + r] += rv[r];
                    }
                }
                if constexpr (CG == 2) store8(out + o, y); else store4(out + o, y);
            }
        }
    }
}

template <typename T, int NI, int WN, bool GATE, int NS, int MI, int NWV = 4, int KG = 1, bool SEF = false>
static int launch_pw_dma_mi(PwKArgs k, hipStream_t s) {
    if constexpr (!SEF) COSY_REQUIRE(!k.se_wr, "pw_gemm_dma: this tile shape has no squeeze-excite prologue (NI=%d WN=%d)", NI, WN);
    if constexpr (!(KG == 1 && sizeof(T) == 2)) COSY_REQUIRE(!k.out_chunked && !k.res_chunked, "pw_gemm_dma: this tile shape does not write / read the chunked layout (NI=%d WN=%d)", NI, WN);
    constexpr int WM = NWV / KG / WN, NB = KG * (MI * WM + NI * WN * (__is_same(T, bf16_t) ? 2 : 1));
    k.MT = cdiv(k.M, 16 * MI * WM);
    const int grid = cdiv(k.MT, 8) * 8 * k.NT;
    k.rowgate = GATE && (k.HW % 64 != 0 || __is_same(T, bf16_t));      // bf16: the gate always multiplies the activation rows (pw_gate_on_weights)
    k.nsamp = k.rowgate ? pw_gate_nsamp(16 * MI * WM, k.HW) : 2;
    const size_t lds = (size_t)NS * NB * 1024 + 3072 + (GATE ? (size_t)k.nsamp * k.nkb_total * DT<T>::KB * 4 : 0) +
                       (GATE && k.se_wr ? ((size_t)k.nkb_total * DT<T>::KB + 128) * 4 : 0);       // + pooled[Kpad], redv[<= 128]
    COSY_REQUIRE(lds <= 160 * 1024, "pw_gemm_dma: the gate rows of %d samples x K=%d do not fit the LDS (map of %d pixels too small)", k.nsamp, k.K, k.HW);
    // once per instantiation and process, race-free: function-local statics are initialised exactly once (C++11), also when two
    // nets launch from two threads at the same time (the header promises thread-safety across distinct nets / streams)
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)pw_gemm_dma_kernel<T, NI, WN, NS, GATE, MI, NWV, KG, SEF>,
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    COSY_CHECK_HIP(attr_rc);
    hipLaunchKernelGGL((pw_gemm_dma_kernel<T, NI, WN, NS, GATE, MI, NWV, KG, SEF>), dim3(grid), dim3(NWV * 64), lds, s, k);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
// MI = 16-row blocks per wave: 4 (64 rows) by default; 2 halves the accumulators (-> more resident workgroups)
static int pw_mi_for(bool gated, int HW, int nkb, int N, int elem_bytes) {
    static const int mi = tune_int("COSY_PW_MI", 4), mi_rg = tune_int("COSY_PW_MI_RG", 1);
    // Row-side gate (pixel count of the map not a multiple of 64) with 128 <= N <= 256, 16-bit types: 32 rows per wave.  The tile count doubles and
    // a workgroup holds half the accumulators, so more of them are resident: at 240x320 / 256 crops the 600 m-tiles of blocks 13-17 (two rounds on
    // 512 slots, the second 17 % full) become 1200 small ones, 69 -> 58 us; blocks 18-23 (320 tiles for 256 CUs) 51 -> 42 us.  N = 96 (blocks 8-12)
    // and N = 384 (blocks 24 / 25) lose with it (34 -> 37, 51 / 79 -> 59 / 101 us) and keep 64 rows per wave (profiles/r04_rowgate_tiles.txt).
    if (mi_rg && elem_bytes == 2 && gated && HW % 64 != 0 && nkb > 2 && N >= 128 && N <= 256) return 2;
    return (mi == 2 && nkb > 2 && (!gated || HW % 64 == 0)) ? 2 : 4;
}
template <typename T, int NI, int WN, bool GATE, int NS>
static int launch_pw_dma_ns(const PwKArgs& k, int grid, hipStream_t s) {
    (void)grid;
    if (pw_mi_for(k.gate != nullptr, k.HW, k.nkb_valid, k.N, (int)sizeof(T)) == 2) {
        if constexpr (!GATE) return launch_pw_dma_mi<T, NI, WN, GATE, NS, 2>(k, s);
        else if ((k.HW % 32 == 0 && k.HW >= 64) || k.HW % 64 != 0) return launch_pw_dma_mi<T, NI, WN, GATE, NS, 2>(k, s);
    }
    if constexpr (GATE && NS == 3 && NI >= 3) {       // the fused blocks' project GEMMs: >= 3 k-blocks (Cmid >= 96), tiles of >= 48 columns
        if (k.se_wr) return launch_pw_dma_mi<T, NI, WN, GATE, NS, 4, 4, 1, true>(k, s);
    }
    return launch_pw_dma_mi<T, NI, WN, GATE, NS, 4>(k, s);
}
// Ring depth of the 4-wave tiles.  Short k-loops (<= 2 k-blocks: the streaming 1x1 convs of the high-resolution blocks) need no deep ring: 2 stages keep the
// LDS footprint small so that more workgroups are resident per CU.  bf16 (hi + lo weight blocks, pw_hl): a 3-stage ring of the 160-column tile is 84 KB -- ONE
// workgroup per CU instead of two (blocks 13-17's project GEMMs: 41 -> 85 us, profiles/r06_bf16.txt); two stages keep the second workgroup.
int pw_ring_stages(int K, PwCfg c, int dtype) {
    static const int ns2 = tune_int("COSY_PW_NS2_MAXKB", 2), deep = tune_int("COSY_PW_NS", 3);
    const int nkb = cdiv(K, pw_kb(dtype));
    if (nkb <= ns2) return 2;
    if (c.WV <= 4 && (16 / c.WN + c.NI * c.WN * pw_hl(dtype)) * 3 > 80) return 2;
    return deep >= 4 && nkb >= 8 ? 4 : 3;
}
template <typename T, int NI, int WN, bool GATE>
static int launch_pw_dma_cfg(const PwKArgs& k, int grid, hipStream_t s) {
    const int ns = pw_ring_stages(k.K, PwCfg{NI, WN}, sizeof(T) == 4 ? COSY_F32 : __is_same(T, bf16_t) ? COSY_BF16 : COSY_F16);
    if (ns == 2) return launch_pw_dma_ns<T, NI, WN, GATE, 2>(k, grid, s);
    if (ns == 4) return launch_pw_dma_ns<T, NI, WN, GATE, 4>(k, grid, s);
    return launch_pw_dma_ns<T, NI, WN, GATE, 3>(k, grid, s);
}
template <typename T, bool GATE>
static int launch_pw_dma(const PwKArgs& k, PwCfg c, int grid, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {   // 8-wave tiles of the late layers: 2 waves per SIMD take turns on the matrix pipe
        if (c.WV == 16) {        // in-workgroup split-K (KG = 2): the K >= 1024 project convs of the 8x8 maps
            if constexpr (GATE) {
                constexpr int NS16 = __is_same(T, bf16_t) ? 2 : 3;      // bf16: hi + lo weight blocks, two ring stages (pw_choose_cfg_late)
                if (k.nkb_valid > 4 && (k.HW % 64 == 0 || tune_int("COSY_PW16_RG", 0)) && c.KG == 2 && c.WN == 4) {
                    if (c.NI == 2) return launch_pw_dma_mi<T, 2, 4, GATE, NS16, 4, 16, 2>(k, s);
                    if constexpr (!__is_same(T, bf16_t)) { if (c.NI == 3) return launch_pw_dma_mi<T, 3, 4, GATE, NS16, 4, 16, 2>(k, s); }
                }
            }
            set_error("pw_gemm_dma: 16-wave split-K tile NI=%d WN=%d KG=%d not built for this layer", c.NI, c.WN, c.KG);
            return COSY_EINVAL;
        }
        if (c.WV == 8) {
            const bool wave_gate = true;    // the weight-side gate needs a wave's 64 rows inside one sample (HW % 64 == 0); other maps take the row-side gate (k.rowgate)
            if (k.nkb_valid > 2 && wave_gate) {
                if (c.NI == 2 && c.WN == 4) return launch_pw_dma_mi<T, 2, 4, GATE, 3, 4, 8>(k, s);
            }
            set_error("pw_gemm_dma: 8-wave tile NI=%d WN=%d not built for this layer", c.NI, c.WN);
            return COSY_EINVAL;
        }
    }
    if (c.NI == 4 && c.WN == 2) return launch_pw_dma_cfg<T, 4, 2, GATE>(k, grid, s);
    if (c.NI == 3 && c.WN == 2) return launch_pw_dma_cfg<T, 3, 2, GATE>(k, grid, s);
    if (c.NI == 3 && c.WN == 1) return launch_pw_dma_cfg<T, 3, 1, GATE>(k, grid, s);
    if (c.NI == 2 && c.WN == 1) return launch_pw_dma_cfg<T, 2, 1, GATE>(k, grid, s);
    if constexpr (sizeof(T) == 2) {   // wide tiles of the late layers (pw_choose_cfg_late)
        if (c.NI == 5 && c.WN == 2) return launch_pw_dma_cfg<T, 5, 2, GATE>(k, grid, s);
    }
    set_error("pw_gemm_dma: unsupported tile config NI=%d WN=%d", c.NI, c.WN);
    return COSY_EINVAL;
}

template <typename T>
static int launch_pw_t(const PwArgs& a, PwCfg c, int dtype, hipStream_t s) {
    PwKArgs k;
    k.a_chunked = a.a_chunked; k.out_chunked = a.out_chunked; k.res_chunked = a.res_chunked; k.out_perm_lw = a.out_chunked ? a.out_perm_lw : 0; k.out_perm_lp = a.out_perm_lp;
    k.A = a.A; k.Wp = a.Wp; k.out = a.out; k.scale = a.scale; k.bias = a.bias; k.res = a.res; k.gate = a.gate;
    k.M = a.M; k.K = a.K; k.N = a.N; k.HW = a.HW; k.silu = a.silu;
    k.MT = cdiv(a.M, pw_bm(c)); k.NT = cdiv(a.N, pw_bn(c));
    k.nkb_total = pw_nkb_total(a.K, dtype); k.nkb_valid = cdiv(a.K, pw_kb(dtype));
    const int grid = cdiv(k.MT, 8) * 8 * k.NT;
    k.zeros = a.zeros; k.nsamp = 2; k.rowgate = 0;
    k.se_partial = k.se_wr = k.se_br = k.se_we = k.se_be = nullptr; k.gate_out = nullptr; k.se_tiles = 0; k.Cse = 0; k.inv_hw = 0.f;
    if (a.se_fused) {
        const SeArgs& se = *a.se_fused;
        COSY_REQUIRE(a.gate && se.gate == a.gate && se.C == a.K && se.HW == a.HW && se.Cse >= 1 && se.Cse <= 128 && a.K % 4 == 0,
                     "pw_gemm: fused squeeze-excite arguments do not match the GEMM (C=%d K=%d Cse=%d)", se.C, a.K, se.Cse);
        k.se_partial = se.partial; k.se_tiles = se.n_tiles; k.se_wr = se.w_red; k.se_br = se.b_red; k.se_we = se.w_exp; k.se_be = se.b_exp;
        k.gate_out = se.gate; k.Cse = se.Cse; k.inv_hw = 1.f / (float)se.HW;
    }
    // A = the depthwise output D of a wave-front block (chunked layout), read exactly once (a single n-tile) and far larger than the L2:
    // its DMAs carry the non-temporal hint, so the stream does not evict what the neighbouring kernels keep there.  Measured at 256
    // crops: project GEMMs of blocks 3 / 4 130.5 -> 112.4 us and the wave fronts that follow them 210 -> 196-200 us, block 2 91 -> 87;
    // NOT for the NHWC outputs of the unfused depthwise kernel (blocks 0 / 1: 131.6 -> 154.9, 127.9 -> 133.8 us with the hint).
    k.a_nt = tune_int("COSY_PW_ANT", 1) && a.a_chunked && k.NT == 1 && (size_t)a.M * a.K * sizeof(T) >= ((size_t)64 << 20);
    COSY_REQUIRE(a.zeros, "pw_gemm: the zero page is missing");
    // chunked A: [sample][ceil(K / 16)][HW][16]; a last chunk that is half full (K = 40: stem front) is read up to K only (K % 8 == 0)
    COSY_REQUIRE(!a.a_chunked || (size_t)a.M * (size_t)((a.K + 15) >> 4) < ((size_t)1 << 32), "pw_gemm: %d rows x %d channels exceed the chunked layout's 32-bit row index", a.M, a.K);
    return a.gate ? launch_pw_dma<T, true>(k, c, grid, s) : launch_pw_dma<T, false>(k, c, grid, s);
}

static const char* tname(int dtype) { return dtype == COSY_F32 ? "float" : dtype == COSY_BF16 ? "__bf16" : "_Float16"; }
// the kernel symbol (as rocprofv3 demangles it) that launch_pw_gemm will run for these arguments
void pw_kernel_name(const PwArgs& a, PwCfg c, int dtype, char* buf, size_t n) {
    const int nkb = cdiv(a.K, pw_kb(dtype));
    const int mi = pw_mi_for(a.gate != nullptr, a.HW, nkb, a.N, dtype == COSY_F32 ? 4 : 2);
    if (c.WV == 16) snprintf(buf, n, "pw_gemm_dma_kernel<%s, %d, %d, %d, %s, 4, 16, %d>", tname(dtype), c.NI, c.WN, dtype == COSY_BF16 ? 2 : 3, a.gate ? "true" : "false", c.KG);
    else if (c.WV == 8) snprintf(buf, n, "pw_gemm_dma_kernel<%s, %d, %d, 3, %s, 4, 8>", tname(dtype), c.NI, c.WN, a.gate ? "true" : "false");
    else snprintf(buf, n, "pw_gemm_dma_kernel<%s, %d, %d, %d, %s, %d>", tname(dtype), c.NI, c.WN, pw_ring_stages(a.K, c, dtype), a.gate ? "true" : "false", mi);
}
void small_kernel_name(int Cin, int k, int s, int dtype, int H, int W, char* buf, size_t n) {
    const int mbr = cdiv(H * W, 16), mpw = cdiv(mbr, (mbr <= 4 ? 256 : 512) / 64);
    snprintf(buf, n, "mbconv_small_kernel<%s, %d, %d, %d, %d, %d>", tname(dtype), k, s, s == 1 ? (H == 7 && W == 10 ? 5 : 4) : 2, cdiv(Cin, 32), mpw);
}

int launch_pw_gemm(const PwArgs& a, PwCfg cfg, int dtype, hipStream_t s) {
    if (a.M == 0) return COSY_OK;
    COSY_REQUIRE(a.K % 8 == 0 && a.N % 8 == 0, "pw_gemm: K=%d and N=%d must be multiples of 8", a.K, a.N);

    return COSY_DISPATCH_T(dtype, launch_pw_t<T>(a, cfg, dtype, s));
}


// ==========================================================================================
// Tiled fused MBConv front (round 1's design): expand 1x1 (MFMA) + BN + SiLU -> LDS tile -> depthwise kxk + BN + SiLU + squeeze
// partials, for the high-resolution blocks whose ROW WIDTH the wave kernel is not built for -- block 2 of the reference's native
// 240x320 crops (160-pixel rows) and blocks 2-8 of crop sizes outside the two tuned ones.  The 6x-expanded tensor never touches
// HBM.  A workgroup owns one (sample, TH x TW output tile):
//   1. the halo'd input tile ((TH-1)s+k) x ((TW-1)s+k) pixels x Cin is DMA'd once into LDS in MFMA-fragment order
//      (1 KiB blocks = 16 pixels x 32 k, lane-linear: one global_load_lds per block, zero page for padding);
//   2. per chunk of 48 expanded channels: every wave runs the 1x1 expansion of a few 16-pixel blocks on the matrix
//      cores (weights = MFMA A operand, so a lane ends up with 12 contiguous channels of one pixel), applies BN+SiLU,
//      ZEROES pixels outside the image (the depthwise conv's padding acts on the expanded tensor) and parks the
//      result in an fp32 LDS tile Et[pixel][48] (row pitch +16 B: conflict-free 16-byte writes);
//   3. depthwise from Et (sliding window over rows), BN+SiLU, NHWC store, squeeze sums.
// Cost: the expansion is recomputed on the halo (x1.2-1.9), and the phases alternate behind barriers -- which is why the wave
// kernel replaced it wherever its row mapping fits (block 2 at 256 crops of 256x256: 554 -> 335 us).
// ==========================================================================================
// Bytes per pixel row of the LDS tile of expanded activations: +16 keeps the expand epilogue's 16-byte writes conflict-free.
constexpr int et_pitch(int elem_size, int stride) { (void)stride; return 48 * elem_size + 16; }
struct FusePlan { int TH, TW, THin, TWin, MB, kbn, threads, ntx, nty, et_f32; size_t lds; };
static FusePlan fuse_plan(int Cin, int Ho, int Wo, int k, int s, int esz) {
    FusePlan p;
    const int R = s == 1 ? 4 : 2;
    p.et_f32 = 1;                                  // the LDS tile that holds the expanded activations is fp32
    const int ees = p.et_f32 ? 4 : 2;
    if (s == 1) { p.TH = esz == 2 ? 8 : 4; p.TW = 16; }
    else { p.TH = 4; p.TW = 8; }
    if (p.TW > Wo) p.TW = Wo;
    p.THin = (p.TH - 1) * s + k; p.TWin = (p.TW - 1) * s + k;
    p.MB = (p.THin * p.TWin + 15) / 16;
    p.kbn = cdiv(Cin, esz == 2 ? 32 : 16);
    const int cpt = 16 / ees, units = (48 / cpt) * p.TW * (p.TH / R);
    p.threads = ((units < 384 ? units : 384) + 63) / 64 * 64;
    {   // enough waves for the expand phase too: at most 2 sixteen-pixel blocks per wave (COSY_FUSE_MBW, experiments)
        static const int mbw = tune_int("COSY_FUSE_MBW", 0);
        if (mbw > 0) { int t = cdiv(p.MB, mbw) * 64; if (t > 384) t = 384; if (t > p.threads) p.threads = t; }
    }
    p.lds = (size_t)p.MB * p.kbn * 1024 + (size_t)p.MB * 16 * et_pitch(ees, s) + (size_t)k * k * 48 * 4 + (size_t)p.threads * 8 * 4;
    p.ntx = cdiv(Wo, p.TW); p.nty = cdiv(Ho, p.TH);
    return p;
}
int tile_num_tiles(int Cin, int Ho, int Wo, int k, int s, int dtype) {
    FusePlan p = fuse_plan(Cin, Ho, Wo, k, s, dtype == COSY_F32 ? 4 : 2);
    return p.ntx * p.nty;
}
// shapes the tiled kernel is built for: up to 2 k-blocks of input channels (the high-resolution blocks), 48-channel chunks
bool tile_supported(int Cin, int Cmid, int k, int s, int dtype) {
    const int kbn = cdiv(Cin, dtype == COSY_F32 ? 16 : 32);
    return Cmid % 48 == 0 && kbn <= 2 && (k == 3 || k == 5) && (s == 1 || s == 2);
}

struct FuseKArgs {
    const void* X; const void* Wp; const float* s0; const float* b0; const float* dww; const float* s1; const float* b1;
    void* D; float* partial; const void* zeros;
    int H, W, Cin, Cmid, Ho, Wo, lo, TH, TW, THin, TWin, MB, ntx, n_tiles, nkb_total, dbg;
    unsigned rcp_tw;   // ceil(2^16 / TWin): p / TWin == (p * rcp_tw) >> 16 for the tile's pixel range (checked on the host)
};

// T = storage type of X/D (and of the MFMA operands), ET = element type of the LDS tile of expanded activations
template <typename T, typename ET, int KS, int S, int R, int KBN>
__global__ __launch_bounds__(384) void mbconv_tile_kernel(FuseKArgs a) {
    using raw_t = typename DT<T>::raw_t;
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int NI = 3, CC = 48;
    constexpr int PITCH = et_pitch((int)sizeof(ET), S);  // bytes per Et pixel row: conflict-free depthwise reads
    constexpr int CPT = 16 / (int)sizeof(ET);         // channels per depthwise thread (one 16-byte Et read)
    constexpr int NG = CC / CPT;                      // channel groups per chunk
    constexpr int NROW = (R - 1) * S + KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TWin = a.TWin, Pin = a.THin * a.TWin, MB = a.MB;
    char* Xt = smem;
    char* Et = Xt + (size_t)MB * KBN * 1024;
    float* wl = (float*)(Et + (size_t)MB * 16 * PITCH);
    float* red = wl + KS * KS * CC;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthr >> 6;
    const int job = blockIdx.x, tile_id = job % a.n_tiles, b = job / a.n_tiles;
    const int tx = tile_id % a.ntx, ty = tile_id / a.ntx;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int iy0 = oy0 * S - a.lo, ix0 = ox0 * S - a.lo;
    const int prow = lane & 15, kg = lane >> 4;

    // ---- 1. DMA the input tile into LDS in fragment order
    {
        const T* __restrict__ X = (const T*)a.X + (size_t)b * a.H * a.W * a.Cin;
        for (int blk = wave; blk < MB * KBN; blk += nwaves) {
            const int mb = blk / KBN, kb = blk - mb * KBN;
            const int p = mb * 16 + prow, k = kb * KB + kg * EPL;
            const int yy = (int)(((unsigned)p * a.rcp_tw) >> 16), xx = p - yy * TWin, iy = iy0 + yy, ix = ix0 + xx;
            const bool ok = p < Pin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && k < a.Cin;
            const void* src = ok ? (const void*)(X + ((size_t)iy * a.W + ix) * a.Cin + k) : a.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Xt + (size_t)blk * 1024), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int nchunks = a.Cmid / CC;
    const int nyq = a.TH / R;
    const int units = NG * a.TW * nyq;
    const int stride = (nthr / NG) * NG;
    const int cq = tid % NG;
    for (int ch = 0; ch < nchunks; ++ch) {
        // taps of this chunk -> LDS
        for (int i = tid; i < KS * KS * (CC / 4); i += nthr) {
            const int tap = i / (CC / 4), q = i - tap * (CC / 4);
            *(f32x4*)(wl + tap * CC + q * 4) = *(const f32x4*)(a.dww + (size_t)tap * a.Cmid + ch * CC + q * 4);
        }
        // ---- 2. expansion on the matrix cores -> Et
        {
            raw_t wf[NI][KBN];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int kb = 0; kb < KBN; ++kb)
                    wf[ni][kb] = *(const raw_t*)((const T*)a.Wp + ((size_t)(ch * NI + ni) * a.nkb_total + kb) * 64 * EPL + lane * EPL);
            const int n0 = ch * CC + kg * 4 * NI;   // this lane's 12 consecutive expanded channels
            float sc[NI * 4], bi[NI * 4];
#pragma unroll
            for (int q = 0; q < NI; ++q) { load4(a.s0 + n0 + q * 4, sc + q * 4); load4(a.b0 + n0 + q * 4, bi + q * 4); }
            for (int mb = wave; mb < (COSY_DBG(a.dbg & 2) ? 0 : MB); mb += nwaves) {
                f32x4 acc[NI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KBN; ++kb) {
                    const raw_t xf = *(const raw_t*)(Xt + (size_t)(mb * KBN + kb) * 1024 + lane * 16);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) mma(acc[ni], wf[ni][kb], xf);
                }
                const int p = mb * 16 + prow;
                const int yy = (int)(((unsigned)p * a.rcp_tw) >> 16), xx = p - yy * TWin, iy = iy0 + yy, ix = ix0 + xx;
                const bool inside = p < Pin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                float y[NI * 4];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[ni][r] * sc[ni * 4 + r] + bi[ni * 4 + r];
                        v = v * sigmoid_t<T>(v);
                        y[ni * 4 + r] = inside ? v : 0.f;
                    }
                ET* dst = (ET*)(Et + (size_t)p * PITCH) + kg * 4 * NI;
                if constexpr (sizeof(ET) == 2) { store8(dst, y); store4(dst + 8, y + 8); }
                else { store4(dst, y); store4(dst + 4, y + 4); store4(dst + 8, y + 8); }
            }
        }
        __syncthreads();
        // ---- 3. depthwise from Et
        float sum[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) sum[c] = 0.f;
        if (tid < stride && !COSY_DBG(a.dbg & 1)) {
            const int c0 = ch * CC + cq * CPT;
            float sc[CPT], bi[CPT];
#pragma unroll
            for (int c = 0; c < CPT; c += 4) { load4(a.s1 + c0 + c, sc + c); load4(a.b1 + c0 + c, bi + c); }
            T* __restrict__ out = (T*)a.D + (size_t)b * a.Ho * a.Wo * a.Cmid + c0;
#pragma unroll 1
            for (int u = tid; u < units; u += stride) {
                const int q = u / NG, x = q % a.TW, yq = q / a.TW;
                const int ox = ox0 + x, oyb = oy0 + yq * R;
                if (ox >= a.Wo || oyb >= a.Ho) continue;
                float acc[R][CPT];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < CPT; ++c) acc[r][c] = 0.f;
#pragma unroll 1
                for (int kx = 0; kx < KS; ++kx) {
                    float wc[KS][CPT];
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int c = 0; c < CPT; c += 4) load4(wl + (ky * KS + kx) * CC + cq * CPT + c, wc[ky] + c);
                    const char* col = Et + ((size_t)(yq * R * S) * TWin + x * S + kx) * PITCH + cq * 16;
#pragma unroll
                    for (int rr = 0; rr < NROW; ++rr) {
                        float v[CPT];
                        if constexpr (sizeof(ET) == 2) lds_ld8((const ET*)(col + (size_t)rr * TWin * PITCH), v);
                        else load4((const float*)(col + (size_t)rr * TWin * PITCH), v);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const int ky = rr - r * S;
                            if (ky >= 0 && ky < KS) {
#pragma unroll
                                for (int c = 0; c < CPT; ++c) acc[r][c] += wc[ky][c] * v[c];
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int oy = oyb + r;
                    if (oy < a.Ho) {
                        float y[CPT];
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            float v = acc[r][c] * sc[c] + bi[c];
                            v = v * sigmoid_t<T>(v);
                            y[c] = v;
                            sum[c] += v;
                        }
                        T* o = out + ((size_t)oy * a.Wo + ox) * a.Cmid;
                        if constexpr (CPT == 8) store8(o, y); else store4(o, y);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) red[tid * CPT + c] = sum[c];
        __syncthreads();
        reduce_squeeze_sums(red, stride, NG, CPT, tid, nthr, a.partial + ((size_t)b * a.n_tiles + tile_id) * a.Cmid + ch * CC);
        // next chunk: its Et writes happen after this barrier's readers are done (all Et reads precede the barrier above);
        // `red` is rewritten only after the next chunk's first barrier.
    }
}

template <typename T, typename ET, int KBN>
static int launch_fuse_k(const FuseArgs& a, const FusePlan& p, const FuseKArgs& k, hipStream_t s) {
    dim3 grid((unsigned)(k.n_tiles * a.B)), block(p.threads);
    if (a.k == 3 && a.s == 1) hipLaunchKernelGGL((mbconv_tile_kernel<T, ET, 3, 1, 4, KBN>), grid, block, p.lds, s, k);
    else if (a.k == 3 && a.s == 2) hipLaunchKernelGGL((mbconv_tile_kernel<T, ET, 3, 2, 2, KBN>), grid, block, p.lds, s, k);
    else if (a.k == 5 && a.s == 1) hipLaunchKernelGGL((mbconv_tile_kernel<T, ET, 5, 1, 4, KBN>), grid, block, p.lds, s, k);
    else if (a.k == 5 && a.s == 2) hipLaunchKernelGGL((mbconv_tile_kernel<T, ET, 5, 2, 2, KBN>), grid, block, p.lds, s, k);
    else { set_error("mbconv_tile: unsupported k=%d s=%d", a.k, a.s); return COSY_EINVAL; }
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
template <typename T>
static int launch_tile_t(const FuseArgs& a, hipStream_t s) {
    const FusePlan p = fuse_plan(a.Cin, a.Ho, a.Wo, a.k, a.s, sizeof(T));
    FuseKArgs k;
    k.X = a.X; k.Wp = a.Wp; k.s0 = a.s0; k.b0 = a.b0; k.dww = a.dww; k.s1 = a.s1; k.b1 = a.b1; k.D = a.D; k.partial = a.partial;
    k.zeros = a.zeros; k.H = a.H; k.W = a.W; k.Cin = a.Cin; k.Cmid = a.Cmid; k.Ho = a.Ho; k.Wo = a.Wo; k.lo = a.pad_lo;
    k.TH = p.TH; k.TW = p.TW; k.THin = p.THin; k.TWin = p.TWin; k.MB = p.MB; k.ntx = p.ntx; k.n_tiles = p.ntx * p.nty;
    k.nkb_total = pw_nkb_total(a.Cin, sizeof(T) == 4 ? COSY_F32 : COSY_BF16);   // k-block geometry depends on the element size only
    k.rcp_tw = (65536u + p.TWin - 1) / p.TWin;
    static const int dbg = tune_int("COSY_FUSE_DBG", 0);   // phase knock-out, timing experiments only
    k.dbg = dbg;
    for (int q = 0; q < p.MB * 16; ++q)
        if ((int)(((unsigned)q * k.rcp_tw) >> 16) != q / p.TWin) { set_error("mbconv_tile: reciprocal division inexact"); return COSY_EINVAL; }
    if (p.kbn == 1) return launch_fuse_k<T, float, 1>(a, p, k, s);
    return launch_fuse_k<T, float, 2>(a, p, k, s);
}

void tile_kernel_name(int Cin, int k, int s, int dtype, char* buf, size_t n) {
    snprintf(buf, n, "mbconv_tile_kernel<%s, float, %d, %d, %d, %d>", dtype == COSY_BF16 ? "__bf16" : dtype == COSY_F16 ? "_Float16" : "float", k, s, s == 1 ? 4 : 2,
             cdiv(Cin, dtype == COSY_F32 ? 16 : 32));
}
int launch_mbconv_tile(const FuseArgs& a, int dtype, hipStream_t s) {
    if (a.B == 0) return COSY_OK;
    COSY_REQUIRE(tile_supported(a.Cin, a.Cmid, a.k, a.s, dtype), "mbconv_tile: unsupported shape Cin=%d Cmid=%d k=%d s=%d", a.Cin, a.Cmid, a.k, a.s);
    return dtype == COSY_BF16 ? launch_tile_t<bf16_t>(a, s) : dtype == COSY_F16 ? launch_tile_t<f16_t>(a, s) : launch_tile_t<float>(a, s);
}

// ==========================================================================================
// Fused MBConv front for the SMALL maps of the late blocks (H*W <= 320 pixels, Cin up to 384):
// expand 1x1 (MFMA) + BN + SiLU -> LDS -> depthwise kxk + BN + SiLU + squeeze sums; the 6x-expanded tensor never touches HBM.
// (The larger maps run the wave-autonomous kernel of kernels_wave.hip; shapes neither kernel is built for run unfused:
// pw_gemm_dma -> E -> dwconv.)
// ==========================================================================================
// Bytes per pixel row of the LDS tile of expanded activations: +16 keeps the expand epilogue's 16-byte writes (16 pixels
// per instruction, one pitch apart) conflict-free.
static bool fuse_use_small(int Cin, int Cmid, int H, int W, int Ho, int Wo, int k, int s, int dtype);
static int conv_out(int in, int k, int s) { return s == 1 ? in : (in - 2) / 2 + 1; }   // static same padding (image_size 300)
// H, W = the block's input map
bool small_supported(int Cin, int Cmid, int k, int s, int dtype, int H, int W) {
    if (H <= 0) return false;
    return fuse_use_small(Cin, Cmid, H, W, conv_out(H, k, s), conv_out(W, k, s), k, s, dtype);
}

// ------------------------------------------------------------------------------------------
// Whole-image variant for the SMALL maps of the late blocks (H*W <= 320 pixels, Cin up to 384): the same fusion
// (expand 1x1 -> BN -> SiLU -> depthwise -> BN -> SiLU -> squeeze sums, the 6x expanded tensor never leaves the CU),
// organised differently because here the map is one tile and the input has many channels:
//   * a workgroup owns (sample, a run of 48-channel chunks); there is no halo to recompute;
//   * the block INPUT lives in REGISTERS for the whole kernel: every wave keeps the MFMA B fragments of its MPW
//     16-pixel blocks for all KBN k-blocks (KBN*4 VGPRs each), loaded once from global;
//   * the chunk's expand weights (3*KBN fragment blocks) are DMA'd into one LDS buffer while the previous chunk's
//     depthwise phase runs (they are dead during that phase: a single buffer suffices);
//   * Et holds the zero-padded expanded image in fp32; only real pixels are ever written, so the halo is zeroed once;
//   * squeeze sums are complete per (sample, chunk): partial has n_tiles = 1.
// ------------------------------------------------------------------------------------------
struct FuseSKArgs {
    const void* X; const void* Wp; const float* b0; const float* dww; const float* b1;
    void* D; float* partial; const void* zeros;
    int H, W, NP, Cin, Cmid, Ho, Wo, lo, THin, TWin, nkb_total, MBr, ncg, cpw, dbg;
    unsigned rcp_w;   // ceil(2^16 / W): p / W == (p * rcp_w) >> 16 for p < MBr*16 (checked on the host)
    int xpose;        // the depthwise phase walks the TRANSPOSED map (7x10 maps: 10 rows of 7 pixels): Ho / Wo / THin / TWin are the walked
                      // dimensions, the taps arrive as w[kx][ky]; H / W stay the stored map (the expansion decodes pixels with them)
};

template <typename T, int KS, int S, int R, int KBN, int MPW, bool ROWMAP = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3))) void mbconv_small_kernel(FuseSKArgs a) {
    using raw_t = typename DT<T>::raw_t;
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int NI = 3, CC = 48, PITCH = et_pitch(4, S), CPT = 4, NG = CC / CPT;
    constexpr int NROW = (R - 1) * S + KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TWin = a.TWin, npos = a.THin * a.TWin;
    // per-chunk parameters [b0 48][b1 48][taps KS*KS x 48] fp32, double buffered: they are prefetched by DMA one chunk ahead
    // (fetching them with ordinary loads at the top of a chunk exposes ~2 us of latency per chunk).  As in the wave kernel the
    // BatchNorm scales are folded into the expand weights (x log2 e) and the taps (x ln 2); the biases start the accumulators.
    constexpr int PU = 2 * (CC / 4) + KS * KS * (CC / 4);      // 16-byte units per parameter block
    constexpr int PJ = (PU + 63) / 64, PBYTES = PJ * 1024;
    char* Et = smem;
    char* Wl = Et + (size_t)npos * PITCH;
    char* Pl = Wl + NI * KBN * 1024;
    float* red = (float*)(Pl + 2 * PBYTES);
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthr >> 6;
    const int b = blockIdx.x / a.ncg, cg = blockIdx.x - b * a.ncg;
    const int nchunks = a.Cmid / CC;
    const int ch0 = cg * a.cpw, ch1 = min(nchunks, ch0 + a.cpw);
    const int prow = lane & 15, kg = lane >> 4;

    auto issue_w = [&](int ch) {
        for (int blk = wave; blk < NI * KBN; blk += nwaves) {
            const int ni = blk / KBN, kb = blk - ni * KBN;
            const T* src = (const T*)a.Wp + ((size_t)(ch * NI + ni) * a.nkb_total + kb) * 64 * EPL + lane * EPL;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Wl + (size_t)blk * 1024), 16, 0, 0);
        }
        // parameters of the chunk: PJ more DMA instructions, spread over the waves
        for (int j = wave; j < PJ; j += nwaves) {
            const int u = j * 64 + lane;
            const float* src = (const float*)a.zeros;
            if (u < PU) {
                const int arr = u / (CC / 4), q = u - arr * (CC / 4);
                const float* base = arr == 0 ? a.b0 : arr == 1 ? a.b1 : a.dww + (size_t)(arr - 2) * a.Cmid;
                src = base + ch * CC + q * 4;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Pl + (size_t)(ch & 1) * PBYTES + (size_t)j * 1024), 16, 0, 0);
        }
    };
    issue_w(ch0);
    // the block input of this sample -> registers (fragment layout: lane = (pixel row of the 16-block, k-group))
    raw_t xf[MPW][KBN];
    {
        const T* __restrict__ X = (const T*)a.X + (size_t)b * a.NP * a.Cin;
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
            const int p = (wave + mi * nwaves) * 16 + prow;
#pragma unroll
            for (int kb = 0; kb < KBN; ++kb) {
                const int k = kb * KB + kg * EPL;
                const bool ok = p < a.NP && k < a.Cin;
                xf[mi][kb] = *(const raw_t*)(ok ? (const void*)(X + (size_t)p * a.Cin + k) : a.zeros);
            }
        }
    }
    // zero the padded expanded image once (the halo is never written again)
    for (int i = tid; i < npos * (PITCH / 16); i += nthr) *(f32x4*)(Et + (size_t)i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nyq = (a.Ho + R - 1) / R;
    const int units = NG * a.Wo * nyq;           // <= threads: every thread owns at most ONE (4-channel, column, R rows) unit
    const int stride = (nthr / NG) * NG;
    // Unit mapping.  Default: channel quads fastest (a pixel's 12 quads are 12 neighbouring lanes).  ROWMAP (8x8 maps: exactly 16
    // spatial units per channel quad): the 16 units of a quad are the 16 lanes of one DPP row -- the squeeze sums reduce with
    // 4 row shifts (no LDS pass, no barrier), the depthwise reads of 8 consecutive lanes hit 8 different pixels (pitch 208 B:
    // all 32 banks once, conflict-free; quad-fastest had 2-way conflicts on every second half-row), and with D in the chunked
    // layout a wave's store covers 16 pixels x 32 bytes contiguously.
    const int cq = ROWMAP ? tid >> 4 : tid % NG;
    const int uq = ROWMAP ? (tid & 15) : tid / NG, ux = uq % a.Wo, uyq = uq / a.Wo;
    const bool has_unit = (ROWMAP ? tid < 16 * NG && uq < a.Wo * nyq : tid < units) && !COSY_DBG(a.dbg & 1);   // ROWMAP: up to 16 units per quad
    // Global stores count in vmcnt on this ISA and retire in order with the loads: a wait for the NEXT chunk's DMA issued
    // after this chunk's output stores would also wait for the stores' acknowledgements (~2-3 us per chunk, measured).
    // So per chunk: [barrier] expand -> [barrier] issue DMA(ch+1) -> depthwise COMPUTE -> wait DMA -> output stores.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // W(ch0) / params(ch0) / the input fragments have landed
    __syncthreads();
    for (int ch = ch0; ch < ch1; ++ch) {
        const float* P = (const float*)(Pl + (size_t)(ch & 1) * PBYTES);
        const float* wl = P + 2 * CC;
        // ---- expansion on the matrix cores -> Et (real pixels only)
        {
            const int n0 = kg * 4 * NI;
            f32x4 bi[NI];
#pragma unroll
            for (int q = 0; q < NI; ++q) bi[q] = *(const f32x4*)(P + n0 + q * 4);
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                const int mb = wave + mi * nwaves;
                if (mb < a.MBr && !COSY_DBG(a.dbg & 2)) {
                    f32x4 acc[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[ni] = bi[ni];          // log2(e) * BN0 bias: C operand of the first MFMA
#pragma unroll
                    for (int kb = 0; kb < KBN; ++kb)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            const raw_t wf = *(const raw_t*)(Wl + (size_t)(ni * KBN + kb) * 1024 + lane * 16);
                            mma(acc[ni], wf, xf[mi][kb]);
                        }
                    const int p = mb * 16 + prow;
                    if (p < a.NP) {
                        const int y = (int)(((unsigned)p * a.rcp_w) >> 16), x = p - y * a.W;
                        float v12[NI * 4];
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float t = acc[ni][r];               // = log2(e) * BN0(expand)
                                v12[ni * 4 + r] = t * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-t));   // = log2(e) * silu
                            }
                        const int ey = a.xpose ? x : y, ex = a.xpose ? y : x;      // position in the (walked) expanded image
                        float* dst = (float*)(Et + (size_t)((ey + a.lo) * TWin + ex + a.lo) * PITCH) + kg * 4 * NI;
                        store4(dst, v12); store4(dst + 4, v12 + 4); store4(dst + 8, v12 + 8);
                    }
                }
            }
        }
        __syncthreads();                       // Et complete; the weight buffer is free
        if (ch + 1 < ch1 && !COSY_DBG(a.dbg & 4)) issue_w(ch + 1);     // lands while the depthwise phase computes
        // ---- depthwise from Et: compute first, store after the DMA wait
        float sum[CPT], yv[R][CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) sum[c] = 0.f;
        if (has_unit) {
            float bi[CPT];
            load4(P + CC + cq * CPT, bi);
            float acc[R][CPT];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < CPT; ++c) acc[r][c] = bi[c];             // BN1 bias (its scale is in the taps)
#pragma unroll 1
            for (int kx = 0; kx < KS; ++kx) {
                float wc[KS][CPT];
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) load4(wl + (ky * KS + kx) * CC + cq * CPT, wc[ky]);
                const char* col = Et + ((size_t)(uyq * R * S) * TWin + ux * S + kx) * PITCH + cq * 16;
#pragma unroll
                for (int rr = 0; rr < NROW; ++rr) {
                    float v[CPT];
                    load4((const float*)(col + (size_t)rr * TWin * PITCH), v);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int ky = rr - r * S;
                        if (ky >= 0 && ky < KS) {
#pragma unroll
                            for (int c = 0; c < CPT; ++c) acc[r][c] += wc[ky][c] * v[c];
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    float v = acc[r][c];
                    v = v * sigmoid_t<T>(v);
                    yv[r][c] = v;
                    if (uyq * R + r < a.Ho) sum[c] += v;
                }
        }
        if constexpr (ROWMAP) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) {     // fixed-order tree over the 16 lanes of the row; lane 15 holds the total
                float v = sum[c];
                v += dpp_row_shr0<1>(v); v += dpp_row_shr0<2>(v); v += dpp_row_shr0<4>(v); v += dpp_row_shr0<8>(v);
                sum[c] = v;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CPT; ++c) red[tid * CPT + c] = sum[c];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // DMA(ch+1) landed (issued before any store of this chunk)
        if (has_unit) {
            if constexpr (ROWMAP) {          // chunked D: [sample][Cmid/16][HW][16]
                T* __restrict__ out = (T*)a.D + ((size_t)(b * (a.Cmid >> 4) + ch * (CC / 16) + (cq >> 2)) * a.Ho * a.Wo) * 16 + (cq & 3) * CPT;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int oy = uyq * R + r;
                    if (oy < a.Ho) store4(out + (a.xpose ? (size_t)ux * a.Ho + oy : (size_t)oy * a.Wo + ux) * 16, yv[r]);
                }
            } else {
                T* __restrict__ out = (T*)a.D + (size_t)b * a.Ho * a.Wo * a.Cmid + ch * CC + cq * CPT;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int oy = uyq * R + r;
                    if (oy < a.Ho) store4(out + ((size_t)oy * a.Wo + ux) * a.Cmid, yv[r]);
                }
            }
        }
        if constexpr (ROWMAP) {            // lane 15 of every quad row holds the quad's sums (it may have no unit of its own: 14-unit maps)
            if (tid < 16 * NG && (tid & 15) == 15 && !COSY_DBG(a.dbg & 1))
                *(f32x4*)(a.partial + (size_t)b * a.Cmid + ch * CC + cq * CPT) = f32x4{sum[0], sum[1], sum[2], sum[3]};
        }
        __syncthreads();   // red complete; everybody's DMA(ch+1) landed; all Et / parameter reads of this chunk are done
        if constexpr (!ROWMAP) reduce_squeeze_sums(red, stride, NG, CPT, tid, nthr, a.partial + (size_t)b * a.Cmid + ch * CC);
    }
}

struct FuseSmallPlan { int kbn, MBr, mpw, threads, THin, TWin, ncg, cpw; size_t lds; bool ok; int xpose, R, Ho, Wo; };
// 7x10 maps (240x320 crops) are walked TRANSPOSED with 5 output rows per unit: 7 columns x 2 row-units = 14 of the 16 lanes of a DPP row,
// i.e. the row-mapped form of the 8x8 maps (no LDS reduction, conflict-free tile reads, chunked D) instead of the plain one (113 vs ~75 us)
static bool fuse_small_xpose(int H, int W, int k, int s) {
    static const int on = tune_int("COSY_SMALL_XPOSE", 1);
    return on && s == 1 && H == 7 && W == 10 && (k == 3 || k == 5);
}
static FuseSmallPlan fuse_small_plan(int Cin, int Cmid, int H, int W, int Ho, int Wo, int k, int s, int esz) {
    FuseSmallPlan p{};
    p.xpose = fuse_small_xpose(H, W, k, s);
    if (p.xpose) { const int t = Ho; Ho = Wo; Wo = t; const int u = H; H = W; W = u; }      // walked dimensions from here on
    p.Ho = Ho; p.Wo = Wo;
    const int R = p.xpose ? 5 : s == 1 ? 4 : 2;
    p.R = R;
    p.kbn = cdiv(Cin, 32);
    p.MBr = cdiv(H * W, 16);
    p.threads = p.MBr <= 4 ? 256 : 512;
    p.mpw = cdiv(p.MBr, p.threads / 64);
    const int nyq = cdiv(Ho, R);
    p.THin = (nyq * R - 1) * s + k;
    if (p.THin < H + k - 1) p.THin = H + k - 1;
    p.TWin = (Wo - 1) * s + k;
    if (p.TWin < W + k - 1) p.TWin = W + k - 1;
    const int pj = (2 * 12 + k * k * 12 + 63) / 64;     // parameter block, see the kernel
    p.lds = (size_t)p.THin * p.TWin * et_pitch(4, s) + (size_t)3 * p.kbn * 1024 + (size_t)2 * pj * 1024 + (size_t)p.threads * 4 * 4;
    const int nchunks = Cmid / 48;
    static const int cpw_target = tune_int("COSY_SMALL_CPW", 15);
    p.cpw = nchunks <= cpw_target ? nchunks : cdiv(nchunks, cdiv(nchunks, cpw_target));   // ~15 chunks per workgroup (measured 3..29)
    p.ncg = cdiv(nchunks, p.cpw);
    // built for the 8x8 (7x10) maps of blocks 19-25: stride 1, one 16-pixel block per wave, 232 or 384 input channels.
    // (On the 16x16 maps the input registers + a 400-pixel fp32 tile leave one workgroup per CU: not built.)
    p.ok = esz == 2 && Cmid % 48 == 0 && (p.kbn == 8 || p.kbn == 12) && p.mpw == 1 && p.lds <= 80 * 1024 && (k == 3 || k == 5) && s == 1 &&
           12 * Wo * nyq <= p.threads && ((H == 8 && W == 8) || (H == 7 && W == 10) || (p.xpose && H == 10 && W == 7));    // the two maps it is tested on
    return p;
}
static int fuse_small_enabled() { static const int v = tune_int("COSY_FUSE_SMALL", 1); return v; }

// 8x8 maps: 16 spatial units per channel quad -> the row-mapped variant, which writes D in the chunked layout
static bool fuse_small_rowmap(int Ho, int Wo, int threads) {
    static const int on = tune_int("COSY_SMALL_ROWMAP", 1);
    return on && Wo * cdiv(Ho, 4) == 16 && Ho % 4 == 0 && threads >= 192;
}
static bool fuse_small_rowmap(const FuseSmallPlan& p) { return p.xpose || fuse_small_rowmap(p.Ho, p.Wo, p.threads); }
template <typename T, int KS, int KBN>
static int launch_fuse_small_m(const FuseSmallPlan& p, const FuseSKArgs& k, int B, hipStream_t s) {
    const dim3 grid((unsigned)(B * k.ncg)), block(p.threads);
    static const hipError_t attr_rc0 = hipFuncSetAttribute((const void*)mbconv_small_kernel<T, KS, 1, 4, KBN, 1, false>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static const hipError_t attr_rc1 = hipFuncSetAttribute((const void*)mbconv_small_kernel<T, KS, 1, 4, KBN, 1, true>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static const hipError_t attr_rc2 = hipFuncSetAttribute((const void*)mbconv_small_kernel<T, KS, 1, 5, KBN, 1, true>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    COSY_CHECK_HIP(attr_rc0);
    COSY_CHECK_HIP(attr_rc1);
    COSY_CHECK_HIP(attr_rc2);
    if (p.xpose) hipLaunchKernelGGL((mbconv_small_kernel<T, KS, 1, 5, KBN, 1, true>), grid, block, p.lds, s, k);
    else if (fuse_small_rowmap(k.Ho, k.Wo, p.threads)) hipLaunchKernelGGL((mbconv_small_kernel<T, KS, 1, 4, KBN, 1, true>), grid, block, p.lds, s, k);
    else hipLaunchKernelGGL((mbconv_small_kernel<T, KS, 1, 4, KBN, 1, false>), grid, block, p.lds, s, k);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}
template <typename T>
static int launch_fuse_small_t(const FuseArgs& a, hipStream_t s) {
    const FuseSmallPlan p = fuse_small_plan(a.Cin, a.Cmid, a.H, a.W, a.Ho, a.Wo, a.k, a.s, sizeof(T));
    FuseSKArgs k;
    k.X = a.X; k.Wp = a.Wp; k.b0 = a.b0; k.dww = a.dww; k.b1 = a.b1; k.D = a.D; k.partial = a.partial;
    k.zeros = a.zeros; k.H = a.H; k.W = a.W; k.NP = a.H * a.W; k.Cin = a.Cin; k.Cmid = a.Cmid; k.Ho = a.Ho; k.Wo = a.Wo; k.lo = a.pad_lo;
    k.THin = p.THin; k.TWin = p.TWin; k.nkb_total = pw_nkb_total(a.Cin, COSY_BF16); k.MBr = p.MBr; k.ncg = p.ncg; k.cpw = p.cpw;
    k.xpose = p.xpose; k.Ho = p.Ho; k.Wo = p.Wo;       // walked dimensions (== a.Ho / a.Wo unless transposed)
    k.rcp_w = (65536u + a.W - 1) / a.W;
    static const int dbg = tune_int("COSY_SMALL_DBG", 0);   // timing experiments only
    k.dbg = dbg;
    for (int q = 0; q < p.MBr * 16; ++q)
        if ((int)(((unsigned)q * k.rcp_w) >> 16) != q / a.W) { set_error("mbconv_small: reciprocal division inexact"); return COSY_EINVAL; }
    if constexpr (sizeof(T) == 2) {
        if (a.k == 3 && p.kbn == 8) return launch_fuse_small_m<T, 3, 8>(p, k, a.B, s);
        if (a.k == 3 && p.kbn == 12) return launch_fuse_small_m<T, 3, 12>(p, k, a.B, s);
        if (a.k == 5 && p.kbn == 8) return launch_fuse_small_m<T, 5, 8>(p, k, a.B, s);
        if (a.k == 5 && p.kbn == 12) return launch_fuse_small_m<T, 5, 12>(p, k, a.B, s);
    }
    set_error("mbconv_small: unsupported k=%d s=%d k-blocks=%d", a.k, a.s, p.kbn);
    return COSY_EINVAL;
}
static bool fuse_use_small(int Cin, int Cmid, int H, int W, int Ho, int Wo, int k, int s, int dtype) {
    if (!fuse_small_enabled() || dtype == COSY_F32 || cdiv(Cin, 32) <= 2) return false;
    return fuse_small_plan(Cin, Cmid, H, W, Ho, Wo, k, s, 2).ok;
}

// does launch_mbconv_small write D in the chunked layout [sample][Cmid/16][HW][16] for this shape? (the project GEMM must know)
bool small_writes_chunked(int Cin, int Cmid, int H, int W, int Ho, int Wo, int k, int s, int dtype) {
    if (!fuse_use_small(Cin, Cmid, H, W, Ho, Wo, k, s, dtype)) return false;
    const FuseSmallPlan p = fuse_small_plan(Cin, Cmid, H, W, Ho, Wo, k, s, 2);
    return fuse_small_rowmap(p);
}
// are the depthwise taps of this block to be handed over as w[kx][ky] (launch_mbconv_small walks the map transposed)?
bool small_transposed(int Cin, int Cmid, int H, int W, int k, int s, int dtype) {
    return small_supported(Cin, Cmid, k, s, dtype, H, W) && fuse_small_xpose(H, W, k, s);
}
int launch_mbconv_small(const FuseArgs& a, int dtype, hipStream_t s) {
    if (a.B == 0) return COSY_OK;
    COSY_REQUIRE(fuse_use_small(a.Cin, a.Cmid, a.H, a.W, a.Ho, a.Wo, a.k, a.s, dtype), "mbconv_small: unsupported shape Cin=%d Cmid=%d %dx%d",
                 a.Cin, a.Cmid, a.H, a.W);
    return COSY_DISPATCH_T(dtype, launch_fuse_small_t<T>(a, s));
}

// ==========================================================================================
// stem: 3x3 stride 2, 6 -> 40, static pad (lo 0, hi 1), BN + SiLU -- as an implicit GEMM on the matrix cores.
// K = 9 taps x 8 (NHWC8-padded) input channels = 72 -> 3 k-blocks of 32 (taps 9..11 are zero); the B-operand
// fragment of lane (pixel j, k-group kg) in k-block kb is exactly ONE 16-byte NHWC8 input pixel (tap 4*kb+kg), loaded
// straight from global memory: no LDS, no im2col.  Weights (40 -> 48 rows) are the A operand, so a lane ends up
// with 12 consecutive output channels of its pixel (80-byte NHWC rows are written completely by the 4 k-groups).
// For fp32 the same scheme runs with 16-deep k-blocks (2 taps each) on v_mfma_f32_16x16x4_f32.
// ==========================================================================================
static constexpr int STEM_G = 8;   // 16-pixel groups per wave (4 / 8 / 16 / 32: 180 / 170 / 176 / 179 us at 256 crops)
size_t stem_packed_elems(int dtype) { return (size_t)3 * (dtype == COSY_F32 ? 6 : 3) * 64 * (dtype == COSY_F32 ? 4 : 8); }
// w: reference layout (40, 6, 3, 3) -> fragment blocks [ni 0..2][kb][lane][EPL], rows permuted so lane (i>>2 = kg') holds
// channels kg'*12 + ni*4 + (i&3)
void stem_pack_weights(const float* w, int dtype, void* dst) {
    const int epl = dtype == COSY_F32 ? 4 : 8, kb_n = dtype == COSY_F32 ? 6 : 3, kdepth = dtype == COSY_F32 ? 16 : 32;
    size_t idx = 0;
    for (int ni = 0; ni < 3; ++ni)
        for (int kb = 0; kb < kb_n; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < epl; ++e, ++idx) {
                    const int i = lane & 15, kg = lane >> 4;
                    const int n = (i >> 2) * 12 + ni * 4 + (i & 3);
                    const int k = kb * kdepth + kg * epl + e;      // k = tap*8 + ci
                    const int tap = k / 8, ci = k % 8;
                    const float v = (n < 40 && tap < 9 && ci < 6) ? w[((size_t)n * 6 + ci) * 9 + tap] : 0.f;
                    if (dtype == COSY_F32) ((float*)dst)[idx] = v; else ((uint16_t*)dst)[idx] = dtype == COSY_BF16 ? f32_to_bf16_host(v) : f32_to_f16_host(v);
                }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const T* __restrict__ x, const T* __restrict__ wp, const float* __restrict__ scale,
                                                   const float* __restrict__ bias, T* __restrict__ out, int H, int W, int Ho, int Wo,
                                                   long n_groups, int gpw) {
    using raw_t = typename DT<T>::raw_t;
    constexpr int EPL = DT<T>::EPL;
    constexpr int KBN = sizeof(T) == 2 ? 3 : 6;          // k-blocks
    constexpr int TPB = sizeof(T) == 2 ? 4 : 2;          // taps per k-block
    constexpr int LPT = 16 / (int)sizeof(T) == 8 ? 1 : 2;  // 16-byte loads per tap (NHWC8 pixel = 16 B bf16 / 32 B fp32)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kg = lane >> 4;
    raw_t wf[3][KBN];
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb) wf[ni][kb] = *(const raw_t*)(wp + ((size_t)(ni * KBN + kb) * 64 + lane) * EPL);
    const int n0 = kg * 12;
    float sc[12], bi[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) { sc[q] = n0 + q < 40 ? scale[n0 + q] : 0.f; bi[q] = n0 + q < 40 ? bias[n0 + q] : 0.f; }
    const long g0 = ((long)blockIdx.x * 4 + wave) * gpw;
    // input fragments of one 16-pixel group: 3 (6) independent 16-byte loads per lane.  The next group's are issued before the
    // current group's MFMAs / epilogue / stores (the loop is not unrolled: without this every group exposed a full memory latency)
    auto load_group = [&](long g, raw_t* xf) {
        const long pix = min(g, n_groups - 1) * 16 + j;
        const int ox = (int)(pix % Wo);
        const long t2 = pix / Wo;
        const int oy = (int)(t2 % Ho), b = (int)(t2 / Ho);
        const T* xb = x + (size_t)b * H * W * 8;
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb) {
            // this lane's k-range inside the block: kg*EPL .. +EPL  ->  tap and channel offset
            const int kk = kg * EPL;                      // 0..31 (bf16) / 0..15 (fp32)
            const int tap = kb * TPB + kk / 8, ci0 = kk % 8;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int iy = oy * 2 + ky, ix = ox * 2 + kx;
#pragma unroll
            for (int e = 0; e < EPL; ++e) xf[kb][e] = 0;
            if (tap < 9 && iy < H && ix < W) xf[kb] = *(const raw_t*)(xb + ((size_t)iy * W + ix) * 8 + ci0);
        }
    };
    raw_t xcur[KBN], xnext[KBN];
    if (g0 < n_groups) load_group(g0, xcur);
#pragma unroll 1
    for (int gi = 0; gi < gpw; ++gi) {
        const long g = g0 + gi;
        if (g >= n_groups) break;
        if (gi + 1 < gpw) load_group(g + 1, xnext);
        const long pix = g * 16 + j;                 // linear output pixel over (b, oy, ox); Ho*Wo is a multiple of 16
        f32x4 acc[3];
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb) {
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) mma(acc[ni], wf[ni][kb], xcur[kb]);
        }
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb) xcur[kb] = xnext[kb];
        (void)LPT;
        float y[12];
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[ni][r] * sc[ni * 4 + r] + bi[ni * 4 + r];
                y[ni * 4 + r] = v * sigmoid_t<T>(v);
            }
        T* o = out + (size_t)pix * 40 + n0;
        if (kg < 3) {
            if constexpr (sizeof(T) == 2) { store8(o, y); store4(o + 8, y + 8); }
            else { store4(o, y); store4(o + 4, y + 4); store4(o + 8, y + 8); }
        } else {
            store4(o, y);                                 // channels 36..39
        }
    }
}
int launch_stem(const void* x, const void* wp, const float* scale, const float* bias, void* out, int B, int H, int W, int Ho,
                int Wo, int dtype, hipStream_t s) {
    if (B == 0) return COSY_OK;
    COSY_REQUIRE((Ho * Wo) % 16 == 0, "stem: Ho*Wo=%d must be a multiple of 16", Ho * Wo);
    const long n_groups = (long)B * Ho * Wo / 16;
    static const int gpw = tune_int("COSY_STEM_G", STEM_G);
    dim3 grid((unsigned)cdiv(n_groups, 4 * gpw));
    COSY_DISPATCH_STMT(dtype, hipLaunchKernelGGL(stem_kernel<T>, grid, dim3(256), 0, s, (const T*)x, (const T*)wp, scale, bias, (T*)out, H, W, Ho, Wo, n_groups, gpw));
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

}  // namespace cosy
