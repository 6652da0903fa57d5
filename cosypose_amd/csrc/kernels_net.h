// Launchers of the backbone kernels (kernels_net.hip).  T = float or bf16_t selected by `dtype`.
#pragma once
#include "cosy_common.h"

namespace cosy {

struct SeArgs;

// tile configuration of the pointwise-conv GEMM: WV waves as (WV/WN) x WN, each 64 rows x 16*NI columns:
// BN = 16*NI*WN, BM = 64*(WV/WN)
// KG > 1: in-workgroup split-K, the WV waves form KG groups of WV/KG waves that share one output tile (kernels_net.hip)
struct PwCfg { int NI, WN, WV = 4, KG = 1; };
static inline int pw_bn(PwCfg c) { return 16 * c.NI * c.WN; }
static inline int pw_bm(PwCfg c) { return 64 * (c.WV / c.KG / c.WN); }
PwCfg pw_choose_cfg(int N);
PwCfg pw_choose_cfg_late(int K, int N, int HW, bool gated, int dtype);
int pw_ring_stages(int K, PwCfg c, int dtype);   // LDS ring depth of the 4-wave GEMM tile for this layer (the squeeze-excite prologue exists for 3 stages only)
int pw_hl(int dtype);                     // weight fragment blocks per k-block: 2 for bf16 (hi + lo pairs), else 1
bool pw_gate_on_weights(int HW, int dtype);   // project GEMM: squeeze-excite gate folded into the weight fragments (else: multiplied into the activation rows)
int pw_kb(int dtype);                     // k elements per fragment block: 32 (bf16) / 16 (f32)
size_t pw_packed_elems(int K, int N, PwCfg c, int dtype, int hl = -1);      // hl: weight fragment blocks per k-block (< 0: pw_hl(dtype); 1: single values)
// host-side packing of a (N,K) fp32 weight into the kernel's fragment-block order (dst has elem size of dtype)
void pw_pack_weights(const float* w, int K, int N, PwCfg c, int dtype, void* dst, int hl = -1);

struct PwArgs {
    const void* A;       // (M,K) activations, NHWC rows
    const void* Wp;      // packed weights
    void* out;           // (M,N)
    const float* scale;  // (N_pad) folded BN scale
    const float* bias;   // (N_pad) folded BN bias
    const void* res;     // (M,N) residual or null
    const float* gate;   // (B,K) SE gate applied to A rows, or null
    int M, K, N, HW;     // HW = rows per sample (for the gate)
    int silu;
    const void* zeros;   // >= 16 zero bytes (global): source of padded rows/k for the LDS-DMA pipeline
    int a_chunked;       // 1: A is laid out [sample][K/16][HW][16] (what the wave front writes: a wave's row is one contiguous run)
    int out_chunked;     // 1: out is written as [sample][ceil(N/16)][HW][16] -- the block-input layout of the matrix-pipe wave fronts (round 6: their fragment
                         //    loads put 4 neighbouring pixels' 16-byte pieces into one 128-byte line, which the vector-memory address path takes at twice the
                         //    rate of NHWC rows: profiles/r06_ta_patterns.txt); channels N .. 16 ceil(N/16) - 1 of the last chunk are not written
    int res_chunked;     // 1: res is in that layout
    int out_perm_lw, out_perm_lp;   // out_chunked only, 0 = none: the pixels of a row of 2^lw pixels are stored in the order the fp32-FMA wave fronts' fragments want them --
                         //    pixel x at position (x mod P) * 16 + x / P, P = 2^lp = pixels per lane: fragment q = the 16 lanes' pixels p * P + q is then 16 NEIGHBOURING
                         //    32-byte pixels like a matrix-pipe front's segment (FuseArgs::x_perm); rows per sample in row-major order (HW a multiple of 2^lw)
    const struct SeArgs* se_fused;   // non-null: NO squeeze-excite launch ran -- every workgroup computes the gates of the samples under its
                                     // m-tile in its prologue from the squeeze partial sums (se_fused->gate == gate, written for probes)
};
int launch_pw_gemm(const PwArgs& a, PwCfg cfg, int dtype, hipStream_t s);
void pw_kernel_name(const PwArgs& a, PwCfg c, int dtype, char* buf, size_t n);
void small_kernel_name(int Cin, int k, int s, int dtype, int H, int W, char* buf, size_t n);

struct DwArgs {
    const void* in;      // (B,H,W,C)
    const float* w;      // (k*k, C) fp32 taps
    const float* scale;  // (C)
    const float* bias;   // (C)
    void* out;           // (B,Ho,Wo,C)
    float* partial;      // (B, n_tiles, C) per-tile sums of the activated output (SE squeeze)
    int B, H, W, C, Ho, Wo, k, s, pad_lo;
    const void* zeros;   // >= 16 zero bytes (global), source of the padding for the LDS-DMA staging
};
int dw_num_tiles(int C, int Ho, int Wo, int k);
int launch_dwconv(const DwArgs& a, int dtype, hipStream_t s);

// fused expand(1x1, MFMA) + depthwise front half of an MBConv block (the expanded tensor never leaves the CU)
struct FuseArgs {
    const void* X;       // (B,H,W,Cin) block input
    const void* Wp;      // expand weights packed with PwCfg{3,1} (48-channel tiles; small kernel) / PwCfg{1,1} (wave kernel)
    // wave kernel: s0/b0 = folded BN0 scale/bias, dww = fp32 depthwise taps (k*k, Cmid), s1/b1 = folded BN1 scale/bias.
    // small kernel: both BatchNorm scales are folded away at create time -- Wp holds W * s0 * log2(e) (rounded to the storage type
    // after the scaling), b0 = log2(e) * BN0 bias (the C operand of the first MFMA), dww = taps * s1 * ln 2, b1 = BN1 bias
    // (initialises the depthwise accumulators); s0 / s1 are unused (null).
    const float* s0; const float* b0;
    const float* dww;
    const float* s1; const float* b1;
    const float* wparams;   // wave kernel only: the five arrays above packed per 16-channel chunk (wave_pack_params); it reads nothing else
    void* D;             // (B,Ho,Wo,Cmid)
    float* partial;      // (B, n_tiles, Cmid)
    const void* zeros;
    int B, H, W, Cin, Cmid, Ho, Wo, k, s, pad_lo;
    // wave kernel only: pixel order of the block input X / of D inside a sample.  0 = row-major (y * W + x); 1 = column-major (x * H + y): the whole
    // resolution stage is stored transposed so that a wave that walks the map's columns (wave_walks_columns) reads and writes contiguous runs
    int x_colmajor, d_colmajor;
    int x_chunked;       // wave kernel only: X is laid out [sample][ceil(Cin/16)][H*W][16] (PwArgs::out_chunked of the block before)
    int x_perm;          // ... with the pixels of a row permuted for a lane's run of P pixels (PwArgs::out_perm_lw / out_perm_lp)
};
// tiled variant (LDS tile per workgroup) for high-resolution blocks whose row width the wave kernel is not built for
bool tile_supported(int Cin, int Cmid, int k, int s, int dtype);
int tile_num_tiles(int Cin, int Ho, int Wo, int k, int s, int dtype);
int launch_mbconv_tile(const FuseArgs& a, int dtype, hipStream_t s);
void tile_kernel_name(int Cin, int k, int s, int dtype, char* buf, size_t n);
// whole-image variant for the small maps of the late blocks (kernels_net.hip: mbconv_small_kernel); partial has ONE tile per sample
bool small_supported(int Cin, int Cmid, int k, int s, int dtype, int H, int W);
int launch_mbconv_small(const FuseArgs& a, int dtype, hipStream_t s);
bool small_transposed(int Cin, int Cmid, int H, int W, int k, int s, int dtype);   // taps wanted as w[kx][ky]
bool small_writes_chunked(int Cin, int Cmid, int H, int W, int Ho, int Wo, int k, int s, int dtype);   // D layout of launch_mbconv_small
// 8x8 maps with the depthwise taps on the matrix pipe (kernels_smx.hip): Wp packed with PwCfg{1,1} (16-channel tiles) from W * s0 * log2(e), wparams =
// small_mx_pack_params (log2(e) * BN0 bias, folded BatchNorm 1, Toeplitz tap fragments in the storage type); D chunked; partial has ONE tile per sample
bool small_mx_supported(int Cin, int Cmid, int k, int s, int dtype, int H, int W);
size_t small_mx_param_bytes(int Cmid, int k);
void small_mx_pack_params(const float* b0l2e, const float* dww, const float* s1, const float* b1, int Cmid, int k, int dtype, void* dst, int transposed = 0);
bool small_mx_transposed(int H, int W);    // 7x10 maps: the kernel walks the map's columns (taps packed as w[kx][ky])
void small_mx_kernel_name(int Cin, int k, int dtype, char* buf, size_t n);
int launch_mbconv_small_mx(const FuseArgs& a, int dtype, hipStream_t s);
// wave-autonomous variant (kernels_wave.hip): expanded rows in registers, no LDS ring / barriers; expand weights packed with
// PwCfg{1,1} (16-channel tiles, natural row order); partial has ONE tile per sample
bool wave_supported(int Cin, int Cmid, int k, int s, int dtype, int H, int W);
bool wave_walks_columns(int Cin, int Cmid, int k, int s, int dtype, int H, int W);   // the job's rows are the map's columns (transposed walk)
size_t wave_params_floats(int Cin, int Cmid, int k, int s, int dtype, int H, int W);
int wave_input_perm_lp(int Cin, int Cmid, int k, int s, int dtype, int H, int W);    // > 0: this block's wave front (fp32-FMA taps, full power-of-two rows) reads its input with the pixels of a row permuted for runs of 2^lp
bool wave_input_chunk_ok(int Cin, int Cmid, int k, int s, int dtype, int H, int W);   // the wave front of this block reads a chunked input [sample][C/16][HW][16] at the full address rate
bool wave_taps_on_mfma(int Cin, int Cmid, int k, int s, int dtype, int H, int W);     // the wave kernel applies this block's depthwise taps with small MFMAs (E and the taps in the storage type)
void wave_pack_params(const float* s0, const float* b0, const float* dww, const float* s1, const float* b1, int Cin, int Cmid, int k, int s,
                      int dtype, int H, int W, float* dst);
void wave_kernel_name(int Cin, int Cmid, int k, int s, int dtype, int H, int W, char* buf, size_t n);
int wave_max_tiles();   // upper bound of the row bands (partial-sum tiles per sample) a launch may use
int launch_mbconv_wave(const FuseArgs& a, int dtype, int* n_tiles_out, hipStream_t s);

// stem conv + block 0's depthwise front in one wave-autonomous kernel (kernels_stem.hip): the stem tensor never exists in memory.
// 16-bit storage types, 256-pixel-wide inputs; D is written chunked as [sample][3][Hs * Ws][16] (channels 40..47 zero)
struct StemFrontArgs {
    const void* X;          // (B, H, W, 8) network input
    const void* Wp;         // stem_front_pack_weights
    const float* params;    // stem_front_pack_params
    void* D; float* partial;
    void* dump;             // stem_front_dump_bytes() of scratch (finished rows outside a job's band land there)
    const void* zeros;      // the zero page: must lie behind X, within 4 GB of it (the padding taps are fetched from it by offset)
    int B, H, W;
};
bool stem_front_supported(int dtype, int H, int W);
int stem_front_tiles(int H);                 // upper bound of the squeeze partial-sum tiles per sample a launch writes
size_t stem_front_weight_elems();
size_t stem_front_param_floats();
size_t stem_front_dump_bytes();
void stem_front_pack_weights(const float* w_oihw /*(40,6,3,3)*/, int dtype, void* dst);
void stem_front_pack_params(const float* s0, const float* b0, const float* dww /*[tap][40]*/, const float* s1, const float* b1, float* dst);
int launch_stem_front(const StemFrontArgs& a, int dtype, int* n_tiles_out, hipStream_t s);


struct SeArgs {
    const float* partial;  // (B, n_tiles, C)
    int n_tiles;
    const float* w_red;    // (Cse, C)
    const float* b_red;    // (Cse)
    const float* w_exp;    // (C, Cse)
    const float* b_exp;    // (C)
    float* gate;           // (B, C)
    int B, C, Cse, HW;
};
int launch_se(const SeArgs& a, hipStream_t s);
// batched form for the late blocks: the two FCs as GEMMs over the batch (16-sample tiles share one read of the weights);
// wr_p (CseP, C), br_p (CseP), we_p (C, CseP): zero-padded copies, CseP = Cse rounded up to 16; redv: (B, CseP) scratch
bool se_batched_supported(int C, int Cse);
int launch_se_batched(const SeArgs& a, const float* wr_p, const float* br_p, const float* we_p, float* redv, hipStream_t s);

size_t stem_packed_elems(int dtype);
void stem_pack_weights(const float* w_oihw /*(40,6,3,3)*/, int dtype, void* dst);
int launch_stem(const void* x_nhwc8, const void* w_packed, const float* scale, const float* bias, void* out,
                int B, int H, int W, int Ho, int Wo, int dtype, hipStream_t s);
int launch_pool_fc(const void* head /*(B,HW,1536)*/, const float* fc_w /*(9,1536)*/, const float* fc_b, float* feat_or_null,
                   float* feat_scratch, float* pose, int B, int HW, int dtype, hipStream_t s);
// colH > 0: the activation's pixels are stored column-major (x * colH + y, colH = the map's height); the probes index them row-major
int launch_nhwc_to_nchw(const void* act /*(B,HW,C)*/, int B, int HW, int C, int dtype, float* out, hipStream_t s, int chunked = 0, int colH = 0, int perm_lw = 0, int perm_lp = 0);
int launch_taps(const void* act /*(B,HW,C)*/, int B, int HW, int C, int dtype, float* taps /*(B,9,16)*/, int tap_index,
                hipStream_t s, int colH = 0, int chunked = 0, int perm_lw = 0, int perm_lp = 0);   // chunked: [sample][ceil(C/16)][HW][16] (+ PwArgs::out_perm_*)
// out (B, H*W, C) row-major pixels <- in (B, W*H, C) column-major pixels: the exit of a resolution stage that is stored transposed
int launch_pixels_to_rowmajor(const void* in, void* out, int B, int H, int W, int C, int dtype, hipStream_t s);

}  // namespace cosy
