// cosy_net_t: packed EfficientNet-B3(6ch) weights + activation workspace, and the layer schedule.
// Also hosts the extern "C" boundary declared in include/cosyhip.h.
#include "kernels_net.h"
#include "raster_device.h"
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

namespace cosy {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// (k, s, expand, cin, cout): reference block strings cosypose/models/efficientnet_utils.py:259-264 scaled by
// width 1.2 / depth 1.4 (round_filters :60-72, round_repeats :75-80).  Mirrors cosypose_amd/arch.py.
struct BlkDef { int k, s, e, cin, cout; };
static const BlkDef B3[26] = {
    {3, 1, 1, 40, 24},   {3, 1, 1, 24, 24},
    {3, 2, 6, 24, 32},   {3, 1, 6, 32, 32},   {3, 1, 6, 32, 32},
    {5, 2, 6, 32, 48},   {5, 1, 6, 48, 48},   {5, 1, 6, 48, 48},
    {3, 2, 6, 48, 96},   {3, 1, 6, 96, 96},   {3, 1, 6, 96, 96},   {3, 1, 6, 96, 96},   {3, 1, 6, 96, 96},
    {5, 1, 6, 96, 136},  {5, 1, 6, 136, 136}, {5, 1, 6, 136, 136}, {5, 1, 6, 136, 136}, {5, 1, 6, 136, 136},
    {5, 2, 6, 136, 232}, {5, 1, 6, 232, 232}, {5, 1, 6, 232, 232}, {5, 1, 6, 232, 232}, {5, 1, 6, 232, 232}, {5, 1, 6, 232, 232},
    {3, 1, 6, 232, 384}, {3, 1, 6, 384, 384},
};
static const int STAGE_END[7] = {1, 4, 7, 12, 17, 23, 25};
enum { STEM_C = 40, HEAD_IN = 384, HEAD_C = 1536, IN_C = 6, N_POSE = 9 };
static const double BN_EPS = 1e-3;

// Conv2dStaticSamePadding(image_size=300): padding is fixed from the constructor's image size, not the input
// (efficientnet_utils.py:130-141): s=1 -> (k-1)/2 both sides; s=2 -> total k-2, lo = tot/2.
static void static_pad(int k, int s, int* lo, int* hi) {
    if (s == 1) { *lo = *hi = (k - 1) / 2; }
    else { const int tot = k - 2; *lo = tot / 2; *hi = tot - tot / 2; }
}
static int out_dim(int n, int k, int s) { int lo, hi; static_pad(k, s, &lo, &hi); return (n + lo + hi - k) / s + 1; }
static int se_ch(int cin) { return cin / 4 > 1 ? cin / 4 : 1; }

static long param_count() {
    long n = (long)STEM_C * IN_C * 9 + 4 * STEM_C;
    for (int i = 0; i < 26; ++i) {
        const BlkDef& b = B3[i];
        const int cmid = b.cin * b.e, cse = se_ch(b.cin);
        if (b.e != 1) n += (long)cmid * b.cin + 4 * cmid;
        n += (long)cmid * b.k * b.k + 4 * cmid + (long)cse * cmid + cse + (long)cmid * cse + cmid + (long)b.cout * cmid + 4 * b.cout;
    }
    return n + (long)HEAD_C * HEAD_IN + 4 * HEAD_C + N_POSE * HEAD_C + N_POSE;
}

struct PwLayer { int K = 0, N = 0; PwCfg cfg{4, 2}; void* Wp = nullptr; float* scale = nullptr; float* bias = nullptr; };
struct Block {
    BlkDef d; int cmid, cse, H, W, Ho, Wo, pad_lo, n_tiles, dw_tiles;   // n_tiles: most partial-sum tiles per sample any front of this block writes (sizing); dw_tiles: dwconv_kernel's
    bool skip;
    bool wave;            // front = mbconv_wave_kernel (kernels_wave.hip)
    bool small;           // front = mbconv_small_kernel (whole-image kernel of the late blocks)
    bool smx;             // ... in its matrix-pipe form (kernels_smx.hip: 8x8 maps, E and the taps in the storage type); params in wave_params
    bool tiled;           // front = mbconv_tile_kernel (LDS-tiled kernel: high-resolution blocks the wave kernel's row mapping does not fit)
    bool fused;           // wave || small || tiled: the expanded tensor never reaches HBM; otherwise pw_gemm_dma -> E -> dwconv
    PwLayer exp, proj;
    void* exp_wp_fused;   // expand weights packed in 16- (wave) or 48-channel (small) tiles for the fused front
    float *dw_w, *dw_scale, *dw_bias, *se_wr, *se_br, *se_we, *se_be;
    float* wave_params;           // wave kernel: BN0 / BN1 / taps packed per 16-channel chunk (wave_pack_params)
    float *b0_fold, *dw_w_fold;   // small kernel: log2(e) * BN0 bias; taps * BN1 scale * ln 2 (the BN0 scale is inside exp_wp_fused)
    bool se_batched;      // squeeze-excite as two batched GEMM kernels (late blocks) instead of one workgroup per sample
    bool se_fused;        // squeeze-excite inside the project GEMM's prologue: no launch of its own (blocks with small FC matrices)
    // Pixel order of the block's tensors inside a sample: row-major (y * W + x) or column-major (x * H + y).  A resolution stage whose wave
    // kernels all walk the map's COLUMNS (240x320 crops: 30x40 and 15x20 maps -- 15 / 30 pixels fill 16 / 32 lanes, 20 / 40 do not) is stored
    // column-major from the D of its stride-2 entry block on, so that those walks read and write contiguous runs; 1x1 convolutions, squeeze-excite
    // and residuals do not care about the order.  in_col: the block input X; out_col: D and the block output; to_rowmajor: the last block of
    // such a stage when what follows cannot read column-major -- its output goes through one re-ordering copy (launch_pixels_to_rowmajor).
    bool in_col, out_col, to_rowmajor;
    // Channel layout of the block INPUT (= the block before's output): NHWC rows, or chunked [sample][ceil(Cin/16)][H*W][16] for the wave fronts with the
    // taps on the matrix pipe (their fragment q = 16 neighbouring pixels x 32 channels: in the chunked layout 4 neighbouring lanes read one 128-byte line,
    // which the vector-memory address path takes at twice the rate of 4 NHWC rows -- profiles/r06_ta_patterns.txt).  The project GEMM of the block before
    // writes that layout (PwArgs::out_chunked), this block's project GEMM reads its residual from it (res_chunked).
    bool x_chunk;
    int x_perm_lp;        // x_chunk and the front is the fp32-FMA form: log2 of its pixels per lane -- the rows of the input are stored permuted (PwArgs::out_perm_*); else 0
    float *se_wr_p, *se_br_p, *se_we_p;   // zero-padded copies for the batched form: (CseP, Cmid), (CseP), (Cmid, CseP)
};

}  // namespace cosy

struct cosy_net {
    int dtype, H, W, maxB, esz, Hs, Ws, Hf, Wf;
    void* stem_w;
    float *stem_scale, *stem_bias, *fc_w, *fc_b;
    // stem conv + block 0's depthwise front as one kernel (kernels_stem.hip): 16-bit types, 256-pixel-wide inputs.  A forward that is asked for
    // the stem tensor itself (test probe -1, the per-stage taps) runs stem_kernel + dwconv instead -- both sets of weights exist.
    bool stem_fused;
    void* stemf_w; float* stemf_params; void* dump;
    cosy::Block blk[26];
    cosy::PwLayer head;
    void* X;
    int chunk, fuse;
    unsigned small_mask;  // bit i: MBConv block i may run the fused whole-image front kernel (mbconv_small_kernel)
    unsigned tile_mask;   // bit i: ... the LDS-tiled front kernel (mbconv_tile_kernel) when neither of the others is built for its shape
    unsigned wave_mask;   // bit i: ... the wave-autonomous front kernel (mbconv_wave_kernel); both only where the shape is built
    int se_batch_from;    // blocks >= this run the batched squeeze-excite kernels
    unsigned se_fuse_mask; // bit i: block i may compute its squeeze-excite gate in the project GEMM's prologue (Block::se_fused)
    int probe_layer;      // test probe (cosy_effnet_b3_set_probe): -2 = off
    float* probe_out;
    // activation workspaces: ws[0] holds max_batch samples; ws[1] (half size) serves the second half-batch when the
    // forward is split over two internal streams so that VALU-bound and MFMA/bandwidth-bound kernels co-reside
    struct WS { void *act[2], *E, *D, *Hd, *actc[2], *Ec, *Dc; float *partial, *gate, *featbuf, *redv; } ws[2];
    int nstreams, last_split;
    hipStream_t side[2];
    hipEvent_t ev_fork, ev_join[2];
    void* zeros;
    void* crop_taps;   // roi_align tap tables of cosy_crop_pack (maxB x (H + W) entries)
    void* wbase; void* abase;
    size_t wbytes, abytes;
    // profiling ring: PROF_SEGS forwards x (PROF_SLOTS+1) events
    int prof_on, prof_nslots, prof_seg;
    hipEvent_t* prof_ev;
    cosy_prof_rec_t* prof_rec;
};
enum { PROF_SEGS = 24, PROF_SLOTS = 1024 };

namespace cosy {

// simple bump allocator over one hipMalloc'd slab
struct Bump {
    char* base = nullptr; size_t off = 0;
    // host mirror of the weight slab (fill pass): every packed tensor is staged at its device offset and the slab goes up in ONE hipMemcpy at the
    // end of cosy_effnet_b3_create (round 5's profile: ~300 small blocking copies per engine, 19 % of the traced kernel time of a cold start)
    std::vector<char>* mirror = nullptr;
    void* take(size_t bytes) { off = (off + 255) & ~(size_t)255; void* p = base ? base + off : nullptr; off += bytes; return p; }
    void stage(void* dev, const void* src, size_t bytes) { memcpy(mirror->data() + ((char*)dev - base), src, bytes); }
};

static void fold_bn(const float* bn, int C, int Cpad, std::vector<float>& scale, std::vector<float>& bias) {
    scale.assign(Cpad, 0.f); bias.assign(Cpad, 0.f);
    for (int c = 0; c < C; ++c) {
        const double s = (double)bn[c] / sqrt((double)bn[3 * C + c] + BN_EPS);
        scale[c] = (float)s;
        bias[c] = (float)((double)bn[C + c] - (double)bn[2 * C + c] * s);
    }
}

// One pass = sizing (bump.base == nullptr) or filling.  Returns the number of blob floats consumed.
static void plan_pixel_order(cosy_net* n);
static void plan_channel_layout(cosy_net* n);
static long build_weights(cosy_net* n, const float* p, Bump& bump, bool fill, hipError_t* herr) {
    const float* p0 = p;
    auto up_f32 = [&](const std::vector<float>& v) -> float* {
        float* d = (float*)bump.take(v.size() * sizeof(float));
        if (fill) bump.stage(d, v.data(), v.size() * sizeof(float));
        return d;
    };
    auto mk_pw = [&](PwLayer& L, const float* w, int K, int N, const float* bn, int HW, bool gated) {
        L.K = K; L.N = N; L.cfg = n->esz == 2 ? pw_choose_cfg_late(K, N, HW, gated, n->dtype) : pw_choose_cfg(N);
        const size_t ne = pw_packed_elems(K, N, L.cfg, n->dtype);
        L.Wp = bump.take(ne * n->esz);
        const int npad = cdiv(N, pw_bn(L.cfg)) * pw_bn(L.cfg);
        std::vector<float> sc, bi;
        if (fill) {
            std::vector<char> tmp(ne * n->esz);
            pw_pack_weights(w, K, N, L.cfg, n->dtype, tmp.data());
            bump.stage(L.Wp, tmp.data(), tmp.size());
            fold_bn(bn, N, npad, sc, bi);
        } else { sc.assign(npad, 0.f); bi.assign(npad, 0.f); }
        L.scale = up_f32(sc); L.bias = up_f32(bi);
    };
    // stem: (40,6,3,3) -> MFMA fragment blocks (implicit GEMM, K = 9 taps x 8 channels)
    const float* stem_w_host = p;
    std::vector<float> stem_sc_host, stem_bi_host;
    {
        const size_t ne = stem_packed_elems(n->dtype);
        n->stem_w = bump.take(ne * n->esz);
        std::vector<float>& sc = stem_sc_host; std::vector<float>& bi = stem_bi_host;
        if (fill) {
            std::vector<char> tmp(ne * n->esz);
            stem_pack_weights(p, n->dtype, tmp.data());
            bump.stage(n->stem_w, tmp.data(), tmp.size());
        }
        p += STEM_C * IN_C * 9;
        if (fill) fold_bn(p, STEM_C, STEM_C, sc, bi); else { sc.assign(STEM_C, 0.f); bi.assign(STEM_C, 0.f); }
        p += 4 * STEM_C;
        n->stem_scale = up_f32(sc); n->stem_bias = up_f32(bi);
    }
    int h = n->Hs, w_ = n->Ws;
    for (int i = 0; i < 26; ++i) {
        Block& b = n->blk[i];
        b.d = B3[i]; b.cmid = b.d.cin * b.d.e; b.cse = se_ch(b.d.cin);
        b.H = h; b.W = w_; b.Ho = out_dim(h, b.d.k, b.d.s); b.Wo = out_dim(w_, b.d.k, b.d.s);
        int hi; static_pad(b.d.k, b.d.s, &b.pad_lo, &hi);
        b.skip = (b.d.s == 1 && b.d.cin == b.d.cout);  // id_skip, efficientnet.py:94
        b.n_tiles = b.dw_tiles = dw_num_tiles(b.cmid, b.Ho, b.Wo, b.d.k);
        // fused fronts by shape: the wave kernel where a variant holds the block's rows (or columns: transposed walk), the small-map kernel
        // on 8x8 / 7x10 maps (2-byte types), the tiled kernel on blocks 2-5 / 8 otherwise; whatever is left runs the shape-agnostic
        // unfused kernels (pw_gemm_dma -> E -> dwconv)
        b.wave = n->fuse && b.d.e != 1 && ((n->wave_mask >> i) & 1) && wave_supported(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W);
        b.small = !b.wave && n->fuse && b.d.e != 1 && ((n->small_mask >> i) & 1) && small_supported(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W) &&
                  (n->dtype != COSY_BF16 || (tune_int("COSY_SMALL_MX_BF16", 0) && small_mx_supported(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W)));
        // bf16's hi + lo weight pairs: only the matrix-pipe form carries them (kernels_smx.hip, parity-green) -- and it is OFF: the doubled weight ring (91-104 KB of LDS)
        // leaves one workgroup per CU, 98 / 63 / 191 us per block (19-23 / 24 / 25) against 90 / 72 / 113 us of the unfused pair (profiles/r06_dead_ends.txt)
        b.smx = b.small && small_mx_supported(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W);
        b.tiled = !b.wave && !b.small && n->fuse && b.d.e != 1 && ((n->tile_mask >> i) & 1) && tile_supported(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype);
        b.fused = b.wave || b.small || b.tiled;
        b.exp_wp_fused = nullptr;
        b.wave_params = nullptr;
        std::vector<float> exp_sc, exp_bi;       // folded BatchNorm 0 of the expansion (host copy for wave_pack_params)
        if (b.d.e != 1) {
            mk_pw(b.exp, p, b.d.cin, b.cmid, p + (size_t)b.cmid * b.d.cin, b.H * b.W, false);
            if (b.fused) {
                // Small kernel (blocks 19-25): BatchNorm 0 costs no instruction.  Its scale -- times log2(e), so that the SiLU that
                // follows is t / (1 + 2^-t) -- is folded into the expand weights BEFORE they are rounded to the storage type, its bias
                // (times log2 e) is the C operand of the first MFMA (b0_fold); the inverse factor ln 2 and BatchNorm 1's scale
                // ride in the depthwise taps (dw_w_fold), BatchNorm 1's bias initialises the depthwise accumulators.
                // (Measured on the wave kernel too: no gain there -- the freed VALU slots do not shorten its rows, and the MFMA
                // results then feed inline asm directly, which needs explicit wait states -- so it keeps its BatchNorms.)
                const PwCfg c48 = b.wave || b.smx ? PwCfg{1, 1} : PwCfg{3, 1};   // 16-channel tiles for the wave kernel, 48 for the small / tiled kernels
                const size_t ne = pw_packed_elems(b.d.cin, b.cmid, c48, n->dtype);
                b.exp_wp_fused = bump.take(ne * n->esz);
                std::vector<float> b0f(b.cmid, 0.f);
                if (fill) {
                    std::vector<float> sc0, bi0, ws(p, p + (size_t)b.cmid * b.d.cin);
                    if (b.small) {
                        fold_bn(p + (size_t)b.cmid * b.d.cin, b.cmid, b.cmid, sc0, bi0);
                        const double L2E = 1.4426950408889634;
                        for (int c = 0; c < b.cmid; ++c) {
                            b0f[c] = (float)((double)bi0[c] * L2E);
                            for (int k = 0; k < b.d.cin; ++k) ws[(size_t)c * b.d.cin + k] = (float)((double)p[(size_t)c * b.d.cin + k] * (double)sc0[c] * L2E);
                        }
                    }
                    std::vector<char> tmp(ne * n->esz);
                    pw_pack_weights(ws.data(), b.d.cin, b.cmid, c48, n->dtype, tmp.data());
                    bump.stage(b.exp_wp_fused, tmp.data(), tmp.size());
                }
                b.b0_fold = up_f32(b0f);
                if (b.wave) { if (fill) fold_bn(p + (size_t)b.cmid * b.d.cin, b.cmid, b.cmid, exp_sc, exp_bi); else { exp_sc.assign(b.cmid, 0.f); exp_bi.assign(b.cmid, 0.f); } }
                b.n_tiles = b.wave ? wave_max_tiles() : b.tiled ? tile_num_tiles(b.d.cin, b.Ho, b.Wo, b.d.k, b.d.s, n->dtype) : 1;
            }
            p += (size_t)b.cmid * b.d.cin + 4 * b.cmid;
        }
        {   // depthwise (Cmid,1,k,k) -> [tap][Cmid]
            const int kk = b.d.k * b.d.k;
            std::vector<float> w(kk * b.cmid), sc, bi;
            if (fill)
                for (int c = 0; c < b.cmid; ++c)
                    for (int t = 0; t < kk; ++t) w[t * b.cmid + c] = p[c * kk + t];
            p += (size_t)b.cmid * kk;
            if (fill) fold_bn(p, b.cmid, b.cmid, sc, bi); else { sc.assign(b.cmid, 0.f); bi.assign(b.cmid, 0.f); }
            p += 4 * b.cmid;
            b.dw_w = up_f32(w); b.dw_scale = up_f32(sc); b.dw_bias = up_f32(bi);
            if (i == 0 && n->stem_fused) {      // the fused stem + depthwise front: stem weights per 16-channel chunk, both BatchNorms + taps per chunk
                n->stemf_w = bump.take(stem_front_weight_elems() * n->esz);
                std::vector<float> sp(stem_front_param_floats(), 0.f);
                if (fill) {
                    std::vector<char> tmp(stem_front_weight_elems() * n->esz);
                    stem_front_pack_weights(stem_w_host, n->dtype, tmp.data());
                    bump.stage(n->stemf_w, tmp.data(), tmp.size());
                    stem_front_pack_params(stem_sc_host.data(), stem_bi_host.data(), w.data(), sc.data(), bi.data(), sp.data());
                }
                n->stemf_params = up_f32(sp);
                b.n_tiles = std::max(b.n_tiles, stem_front_tiles(n->H));
            }
            if (b.wave) {
                std::vector<float> wp(wave_params_floats(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W), 0.f);
                if (fill) wave_pack_params(exp_sc.data(), exp_bi.data(), w.data(), sc.data(), bi.data(), b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W, wp.data());
                b.wave_params = up_f32(wp);
            }
            if (b.smx) {      // b0f was uploaded as b.b0_fold above; its host copy is rebuilt here (log2 e * BN0 bias)
                std::vector<char> sp(small_mx_param_bytes(b.cmid, b.d.k), 0);
                if (fill) {
                    std::vector<float> sc0, bi0;
                    const float* pe = p - 4 * b.cmid - (size_t)b.cmid * kk - 4 * b.cmid;      // BatchNorm 0 of the expansion (4 * Cmid floats in front of the depthwise weights)
                    fold_bn(pe, b.cmid, b.cmid, sc0, bi0);
                    std::vector<float> b0l(b.cmid);
                    for (int c = 0; c < b.cmid; ++c) b0l[c] = (float)((double)bi0[c] * 1.4426950408889634);
                    small_mx_pack_params(b0l.data(), w.data(), sc.data(), bi.data(), b.cmid, b.d.k, n->dtype, sp.data(), small_mx_transposed(b.H, b.W));
                }
                b.wave_params = (float*)bump.take(sp.size());
                if (fill) bump.stage(b.wave_params, sp.data(), sp.size());
            }
            b.dw_w_fold = nullptr;
            if (b.small && !b.smx) {
                std::vector<float> wf(w.size(), 0.f);
                const bool xp = small_transposed(b.d.cin, b.cmid, b.H, b.W, b.d.k, b.d.s, n->dtype);
                if (fill)
                    for (int t = 0; t < kk; ++t)
                        for (int c = 0; c < b.cmid; ++c) {
                            const int ts = xp ? (t % b.d.k) * b.d.k + t / b.d.k : t;      // walked (ky, kx) -> stored w[ky][kx]
                            wf[t * b.cmid + c] = (float)((double)w[ts * b.cmid + c] * (double)sc[c] * 0.6931471805599453);
                        }
                b.dw_w_fold = up_f32(wf);
            }
        }
        {   // SE: reduce (Cse,Cmid), bias, expand (Cmid,Cse) -> stored transposed (Cse,Cmid), bias
            std::vector<float> wr(p, p + (fill ? (size_t)b.cse * b.cmid : 0)); if (!fill) wr.assign((size_t)b.cse * b.cmid, 0.f);
            p += (size_t)b.cse * b.cmid;
            std::vector<float> br(b.cse, 0.f); if (fill) br.assign(p, p + b.cse);
            p += b.cse;
            std::vector<float> we((size_t)b.cse * b.cmid, 0.f);
            if (fill)
                for (int c = 0; c < b.cmid; ++c)
                    for (int j = 0; j < b.cse; ++j) we[(size_t)j * b.cmid + c] = p[(size_t)c * b.cse + j];
            p += (size_t)b.cmid * b.cse;
            std::vector<float> be(b.cmid, 0.f); if (fill) be.assign(p, p + b.cmid);
            p += b.cmid;
            b.se_wr = up_f32(wr); b.se_br = up_f32(br); b.se_we = up_f32(we); b.se_be = up_f32(be);
            b.se_batched = i >= n->se_batch_from && se_batched_supported(b.cmid, b.cse);
            b.se_fused = false;      // decided below, once the project GEMM's tile is known
            b.se_wr_p = b.se_br_p = b.se_we_p = nullptr;
            if (b.se_batched) {
                const int csep = (b.cse + 15) & ~15;
                std::vector<float> wrp((size_t)csep * b.cmid, 0.f), brp(csep, 0.f), wep((size_t)b.cmid * csep, 0.f);
                if (fill) {
                    std::copy(wr.begin(), wr.end(), wrp.begin());
                    std::copy(br.begin(), br.end(), brp.begin());
                    for (int c = 0; c < b.cmid; ++c)
                        for (int j = 0; j < b.cse; ++j) wep[(size_t)c * csep + j] = we[(size_t)j * b.cmid + c];
                }
                b.se_wr_p = up_f32(wrp); b.se_br_p = up_f32(brp); b.se_we_p = up_f32(wep);
            }
        }
        mk_pw(b.proj, p, b.cmid, b.d.cout, p + (size_t)b.d.cout * b.cmid, b.Ho * b.Wo, true);
        // squeeze-excite inside the project GEMM's prologue: small FC matrices, and m-tiles that never straddle two samples (a straddling
        // tile computes two gates: at 240x320 crops -- maps of 1200 / 300 pixels under 128-row tiles -- the project GEMMs of blocks 13-17
        // went 64 -> 148 us for 13 us of squeeze-excite kernels saved; those sizes keep the kernels)
        b.se_fused = ((n->se_fuse_mask >> i) & 1) && (size_t)b.cse * b.cmid * 8 <= ((size_t)256 << 10) && b.cse <= 128 &&
                     (b.Ho * b.Wo) % pw_bm(b.proj.cfg) == 0 && b.proj.cfg.WV == 4 && b.proj.cfg.NI >= 3 && b.cmid > 2 * pw_kb(n->dtype) && pw_ring_stages(b.cmid, b.proj.cfg, n->dtype) == 3;   // (the 3-stage 4-wave tiles carry the prologue)
        p += (size_t)b.d.cout * b.cmid + 4 * b.d.cout;
        h = b.Ho; w_ = b.Wo;
    }
    n->Hf = h; n->Wf = w_;
    plan_pixel_order(n);
    plan_channel_layout(n);
    mk_pw(n->head, p, HEAD_IN, HEAD_C, p + (size_t)HEAD_C * HEAD_IN, n->Hf * n->Wf, false);
    p += (size_t)HEAD_C * HEAD_IN + 4 * HEAD_C;
    {
        std::vector<float> fw(N_POSE * HEAD_C, 0.f), fb(N_POSE, 0.f);
        if (fill) { fw.assign(p, p + N_POSE * HEAD_C); fb.assign(p + N_POSE * HEAD_C, p + N_POSE * HEAD_C + N_POSE); }
        p += N_POSE * HEAD_C + N_POSE;
        n->fc_w = up_f32(fw); n->fc_b = up_f32(fb);
    }
    return (long)(p - p0);
}

// Which resolution stages are stored column-major (Block::in_col / out_col / to_rowmajor).  Measured at 240x320 crops, 256 per forward, fp16
// (knock-out timing, profiles/r04_colmajor_stages.txt): with row-major storage the column walks' strided input loads and output stores cost
// 24-41 % of blocks 8-12 and 20-25 % of blocks 14-17 (the arithmetic alone scales with the pixel count: x1.25 against 256x256 crops, the kernels x1.5-1.66).
static void plan_pixel_order(cosy_net* n) {
    for (int i = 0; i < 26; ++i) n->blk[i].in_col = n->blk[i].out_col = n->blk[i].to_rowmajor = false;
    static const int allow = tune_int("COSY_COLMAJOR", 1);
    if (!allow) return;
    for (int e = 0; e < 26; ++e) {
        if (n->blk[e].d.s != 2 || !n->blk[e].wave) continue;          // the entry of a stage: a wave block writes its D in any order
        int l = e;
        while (l + 1 < 26 && n->blk[l + 1].d.s == 1) ++l;
        bool all = l > e;
        for (int i = e + 1; i <= l && all; ++i) {
            const Block& b = n->blk[i];
            all = b.wave && wave_walks_columns(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W);
        }
        if (!all) continue;
        for (int i = e; i <= l; ++i) n->blk[i].out_col = true;
        for (int i = e + 1; i <= l; ++i) n->blk[i].in_col = true;
        if (l + 1 < 26 && n->blk[l + 1].wave) n->blk[l + 1].in_col = true;      // the next stage's entry reads column-major as well as anything
        else n->blk[l].to_rowmajor = true;
    }
}

static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static void plan_channel_layout(cosy_net* n) {
    static const int allow = tune_int("COSY_X_CHUNKED", 1);
    for (int i = 0; i < 26; ++i) {
        Block& b = n->blk[i];
        b.x_chunk = allow && i >= 1 && n->esz == 2 && b.wave && !n->blk[i - 1].to_rowmajor &&
                    wave_input_chunk_ok(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W);
        // the fp32-FMA fronts (stride-2 blocks 2 / 5 / 8 at 256x256): a lane owns a run of P pixels, fragment q = the 16 lanes' pixels p * P + q -- 16 neighbours only
        // if the row is stored in that order
        b.x_perm_lp = 0;
        static const int allow_perm = tune_int("COSY_X_PERM", 1);
        if (allow && allow_perm && !b.x_chunk && i >= 1 && n->esz == 2 && b.wave && !b.in_col && !n->blk[i - 1].to_rowmajor && !n->blk[i - 1].out_col && (b.W & (b.W - 1)) == 0) {
            b.x_perm_lp = wave_input_perm_lp(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W);
            b.x_chunk = b.x_perm_lp > 0;
        }
    }
}

enum { EARLY_BLOCKS = 9 };  // stem + blocks 0..8 (feature maps >= 32x32 at 256^2) form the "early" segment

// Early segment runs in sample chunks through small buffers that are REUSED for every chunk, so the large
// high-resolution intermediates stay resident in the 256 MiB Infinity Cache / L2 between producer and consumer
// kernels instead of round-tripping through HBM; the late segment (small maps, big GEMMs) runs on the full batch.
static void layout_ws(cosy_net* n, Bump& b, cosy_net::WS& w, size_t B) {
    const size_t e = n->esz;
    const size_t Bc = std::min((size_t)n->chunk, B);
    size_t act_e = (size_t)n->Hs * n->Ws * STEM_C, ex_e = 0, dw_e = 0, act_l = 0, ex_l = 0, dw_l = 0, part = 0, gate = 0;
    for (int i = 0; i < 26; ++i) {
        const Block& k = n->blk[i];
        const bool early = i < EARLY_BLOCKS;
        size_t& act = early ? act_e : act_l; size_t& ex = early ? ex_e : ex_l; size_t& dw = early ? dw_e : dw_l;
        const size_t cpad = (size_t)((k.d.cout + 15) & ~15);      // (the chunked layout of a wave front's input pads the channels to whole 16-channel chunks)
        act = std::max(act, (size_t)k.Ho * k.Wo * cpad);
        if (i == EARLY_BLOCKS - 1) act_l = std::max(act_l, (size_t)k.Ho * k.Wo * cpad);  // hand-over tensor
        if (k.d.e != 1 && !k.fused) ex = std::max(ex, (size_t)k.H * k.W * k.cmid);
        if (k.to_rowmajor) ex = std::max(ex, (size_t)k.Ho * k.Wo * k.d.cout);      // the project GEMM writes there, the re-ordering copy into the output
        dw = std::max(dw, (size_t)k.Ho * k.Wo * (i == 0 && n->stem_fused ? (size_t)((k.cmid + 15) & ~15) : (size_t)k.cmid));     // (stem front: chunked D, 40 -> 48 channels)
        part = std::max(part, (size_t)k.n_tiles * k.cmid * (early ? Bc : B));
        gate = std::max(gate, (size_t)k.cmid);
    }
    w.actc[0] = b.take(Bc * act_e * e);
    w.actc[1] = b.take(Bc * act_e * e);
    w.Ec = b.take(Bc * ex_e * e + 256);
    w.Dc = b.take(Bc * dw_e * e);
    w.act[0] = b.take(B * act_l * e);
    w.act[1] = b.take(B * act_l * e);
    w.E = b.take(B * ex_l * e);
    w.D = b.take(B * dw_l * e);
    w.Hd = b.take(B * (size_t)n->Hf * n->Wf * HEAD_C * e);
    w.partial = (float*)b.take(part * sizeof(float));
    w.gate = (float*)b.take(B * gate * sizeof(float));
    w.featbuf = (float*)b.take(B * (size_t)HEAD_C * sizeof(float));
    w.redv = (float*)b.take(B * (size_t)128 * sizeof(float));
}
static void layout_workspace(cosy_net* n, Bump& b) {
    n->X = b.take((size_t)n->maxB * n->H * n->W * 8 * n->esz);
    n->zeros = b.take(256);   // stays zero: the workspace is memset at creation and nothing writes here (directly behind X: the stem front reaches it by a 32-bit offset)
    n->dump = b.take(stem_front_dump_bytes());
    n->crop_taps = b.take(crop_taps_bytes(n->maxB, n->H, n->W));
    layout_ws(n, b, n->ws[0], n->maxB);
    if (n->nstreams == 2) layout_ws(n, b, n->ws[1], (n->maxB + 1) / 2);
}

static const char* dt_name(int dtype) { return dtype == COSY_F32 ? "float" : dtype == COSY_BF16 ? "__bf16" : "_Float16"; }

static int net_forward(cosy_net* n, const cosy_net::WS& w, int x_off, int B, float* feat, float* pose, float* taps, hipStream_t s, bool allow_prof) {
    int rc;
    const double esz_d = n->esz;
    const size_t e = n->esz;
    const bool prof = allow_prof && n->prof_on && n->prof_seg < PROF_SEGS && !taps;
    hipEvent_t* ev = prof ? n->prof_ev + (size_t)n->prof_seg * (PROF_SLOTS + 1) : nullptr;
    int slot = 0;
    if (prof) COSY_CHECK_HIP(hipEventRecord(ev[0], s));
    // cbytes: the COMPULSORY part of `bytes` under SURVEY 8(d)'s block-fused model -- block inputs / outputs / residuals and weights
    // only; the block-internal tensors (expanded E, depthwise output D, squeeze sums, gates, the head activation) count as 0
    auto mark = [&](const char* kname, int layer, double bytes, double flops, double cbytes) -> int {
        if (!prof || slot >= PROF_SLOTS) return COSY_OK;
        if (n->prof_seg == 0) {
            cosy_prof_rec_t& r = n->prof_rec[slot];
            snprintf(r.name, sizeof(r.name), "%s", kname);
            r.layer = layer; r.bytes = bytes; r.flops = flops; r.cbytes = cbytes;
        }
        ++slot;
        COSY_CHECK_HIP(hipEventRecord(ev[slot], s));
        return COSY_OK;
    };
    char kn[64];
    auto pw_name = [&](const PwLayer& L, const PwArgs& a) { pw_kernel_name(a, L.cfg, n->dtype, kn, sizeof(kn)); };
    auto pw_bytes = [&](const PwArgs& a, int Bc) { return ((double)a.M * a.K + (double)a.K * a.N + (double)a.M * a.N * (a.res ? 2 : 1)) * esz_d + (a.gate ? (double)Bc * a.K * 4 : 0); };
    auto tap = [&](const void* act, int Bc, int b0, int HW, int C, int idx, int colH = 0, int chunked = 0, int lw = 0, int lp = 0) -> int {
        if (!taps) return COSY_OK;
        return launch_taps(act, Bc, HW, C, n->dtype, taps + (size_t)b0 * 9 * 16, idx, s, colH, chunked, lw, lp);
    };
    // test probe: the whole activation `layer` as fp32 NCHW (layer -1 stem, 0..25 block outputs, 26 head, 100+i depthwise output
    // D of block i, 200+i SE gate of block i as (B, Cmid))
    auto probe = [&](int layer, const void* act, int Bc, int b0, int HW, int C, int chunked, int colH = 0, int lw = 0, int lp = 0) -> int {
        if (n->probe_layer != layer || !n->probe_out) return COSY_OK;
        return launch_nhwc_to_nchw(act, Bc, HW, C, n->dtype, n->probe_out + (size_t)b0 * HW * C, s, chunked, colH, lw, lp);
    };
    // one MBConv block on Bc samples: [expand 1x1] -> depthwise (+squeeze partials) -> SE gate -> project 1x1 (+residual)
    // stem_x != nullptr (block 0 only): the front is the fused stem + depthwise kernel reading the network input; `in` is unused then
    auto run_block = [&](int i, const void* in, void* out, int Bc, void* Ebuf, void* Dbuf, int b0, const void* stem_x = nullptr) -> int {
        const Block& b = n->blk[i];
        const void* src = in;
        int se_tiles = b.fused ? b.n_tiles : b.dw_tiles;     // partial-sum tiles per sample the front kernel writes (the wave kernel decides per launch)
        if (stem_x) {
            StemFrontArgs f{};
            f.X = stem_x; f.Wp = n->stemf_w; f.params = n->stemf_params; f.D = Dbuf; f.partial = w.partial; f.dump = n->dump; f.zeros = n->zeros;
            f.B = Bc; f.H = n->H; f.W = n->W;
            if ((rc = launch_stem_front(f, n->dtype, &se_tiles, s))) return rc;
            snprintf(kn, sizeof(kn), "stem_front_kernel<%s, %d>", dt_name(n->dtype), n->W / 64);
            if ((rc = mark(kn, 0, ((double)Bc * n->H * n->W * 8 + (double)Bc * b.Ho * b.Wo * 48) * esz_d,
                           2.0 * Bc * n->Hs * n->Ws * STEM_C * IN_C * 9 + 2.0 * Bc * b.Ho * b.Wo * b.cmid * 9, (double)Bc * n->H * n->W * 8 * esz_d))) return rc;
        } else if (b.fused) {
            FuseArgs f{};
            f.X = in; f.Wp = b.exp_wp_fused;
            if (b.smx) f.wparams = b.wave_params;
            else if (b.small) { f.b0 = b.b0_fold; f.dww = b.dw_w_fold; f.b1 = b.dw_bias; }
            else { f.s0 = b.exp.scale; f.b0 = b.exp.bias; f.dww = b.dw_w; f.s1 = b.dw_scale; f.b1 = b.dw_bias; f.wparams = b.wave_params; }
            f.D = Dbuf; f.partial = w.partial; f.zeros = n->zeros;
            f.B = Bc; f.H = b.H; f.W = b.W; f.Cin = b.d.cin; f.Cmid = b.cmid; f.Ho = b.Ho; f.Wo = b.Wo; f.k = b.d.k; f.s = b.d.s; f.pad_lo = b.pad_lo;
            f.x_colmajor = b.in_col; f.d_colmajor = b.out_col; f.x_chunked = b.x_chunk; f.x_perm = b.x_perm_lp > 0;
            if ((rc = b.wave ? launch_mbconv_wave(f, n->dtype, &se_tiles, s) : b.tiled ? launch_mbconv_tile(f, n->dtype, s) : b.smx ? launch_mbconv_small_mx(f, n->dtype, s) : launch_mbconv_small(f, n->dtype, s))) return rc;
            if (b.wave) wave_kernel_name(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W, kn, sizeof(kn));
            else if (b.tiled) tile_kernel_name(b.d.cin, b.d.k, b.d.s, n->dtype, kn, sizeof(kn));
            else if (b.smx) small_mx_kernel_name(b.d.cin, b.d.k, n->dtype, kn, sizeof(kn));
            else small_kernel_name(b.d.cin, b.d.k, b.d.s, n->dtype, b.H, b.W, kn, sizeof(kn));
            if ((rc = mark(kn, i, ((double)Bc * b.H * b.W * b.d.cin + (double)Bc * b.Ho * b.Wo * b.cmid + (double)b.d.cin * b.cmid) * esz_d,
                           2.0 * Bc * b.H * b.W * b.d.cin * b.cmid + 2.0 * Bc * b.Ho * b.Wo * b.cmid * b.d.k * b.d.k,
                           ((double)Bc * b.H * b.W * b.d.cin + (double)b.d.cin * b.cmid) * esz_d))) return rc;
        } else {
        if (b.d.e != 1) {
            PwArgs a{};
            a.A = in; a.Wp = b.exp.Wp; a.out = Ebuf; a.scale = b.exp.scale; a.bias = b.exp.bias;
            a.M = Bc * b.H * b.W; a.K = b.d.cin; a.N = b.cmid; a.HW = b.H * b.W; a.silu = 1; a.zeros = n->zeros;
            if ((rc = launch_pw_gemm(a, b.exp.cfg, n->dtype, s))) return rc;
            pw_name(b.exp, a);
            if ((rc = mark(kn, i, pw_bytes(a, Bc), 2.0 * a.M * a.K * a.N, ((double)a.M * a.K + (double)a.K * a.N) * esz_d))) return rc;
            src = Ebuf;
        }
        DwArgs d{};
        d.in = src; d.w = b.dw_w; d.scale = b.dw_scale; d.bias = b.dw_bias; d.out = Dbuf; d.partial = w.partial;
        d.B = Bc; d.H = b.H; d.W = b.W; d.C = b.cmid; d.Ho = b.Ho; d.Wo = b.Wo; d.k = b.d.k; d.s = b.d.s; d.pad_lo = b.pad_lo; d.zeros = n->zeros;
        if ((rc = launch_dwconv(d, n->dtype, s))) return rc;
        snprintf(kn, sizeof(kn), "dwconv_kernel<%s, %d, %d>", dt_name(n->dtype), b.d.k, b.d.s);
        if ((rc = mark(kn, i, ((double)Bc * b.H * b.W * b.cmid + (double)Bc * b.Ho * b.Wo * b.cmid) * esz_d + (double)Bc * b.dw_tiles * b.cmid * 4,
                       2.0 * Bc * b.Ho * b.Wo * b.cmid * b.d.k * b.d.k, b.d.e == 1 ? (double)Bc * b.H * b.W * b.cmid * esz_d : 0.0))) return rc;
        }
#ifdef COSY_TUNE
        if (taps && tune_int("COSY_TAP_D", -1) == i) {       // experiment: probe the depthwise output of block i into tap slot 0
            if ((rc = launch_taps(Dbuf, Bc, b.Ho * b.Wo, b.cmid, n->dtype, taps, 0, s))) return rc;
        }
#endif
        SeArgs se{};
        se.partial = w.partial; se.n_tiles = se_tiles; se.w_red = b.se_wr; se.b_red = b.se_br; se.w_exp = b.se_we; se.b_exp = b.se_be;
        se.gate = w.gate; se.B = Bc; se.C = b.cmid; se.Cse = b.cse; se.HW = b.Ho * b.Wo;
        // Squeeze-excite: blocks 5-13 (b.se_fused; FC matrices <= 222 KB, project GEMMs of <= 2048 workgroups) have NO launch of their own -- every
        // workgroup of the project GEMM computes the gates of its samples in its prologue (kernels_net.hip); the late blocks (0.65 / 1.77 MB
        // of FC weights per gate) keep the batched kernels, where a 16-sample tile shares one read of them.
        if (!b.se_fused) {
            if ((rc = b.se_batched ? launch_se_batched(se, b.se_wr_p, b.se_br_p, b.se_we_p, w.redv, s) : launch_se(se, s))) return rc;
            if ((rc = mark(b.se_batched ? "se_fc1_kernel+se_fc2_kernel" : "se_kernel", i, (double)Bc * se_tiles * b.cmid * 4 + (double)Bc * b.cmid * 4 + 2.0 * b.cse * b.cmid * 4,
                           4.0 * Bc * b.cse * b.cmid, 2.0 * b.cse * b.cmid * 4))) return rc;
        }
        PwArgs a{};
        a.A = Dbuf; a.Wp = b.proj.Wp; a.out = b.to_rowmajor ? Ebuf : out; a.scale = b.proj.scale; a.bias = b.proj.bias;
        a.res = b.skip ? in : nullptr; a.gate = w.gate; a.se_fused = b.se_fused ? &se : nullptr;
        const int out_chunked = i + 1 < 26 && n->blk[i + 1].x_chunk;      // the next block's front wants its input chunked
        a.res_chunked = b.x_chunk; a.out_chunked = out_chunked;
        const int out_lp = out_chunked ? n->blk[i + 1].x_perm_lp : 0, out_lw = out_lp ? ilog2(b.Wo) : 0;      // (a block with a permuted input has stride 2: no residual reads it)
        a.out_perm_lw = out_lw; a.out_perm_lp = out_lp;
        a.M = Bc * b.Ho * b.Wo; a.K = b.cmid; a.N = b.d.cout; a.HW = b.Ho * b.Wo; a.silu = 0; a.zeros = n->zeros;
        // the wave front (and the row-mapped 8x8 kernel) write D as [sample][Cmid/16][HW][16]
        a.a_chunked = stem_x != nullptr || b.wave || b.smx || (b.small && small_writes_chunked(b.d.cin, b.cmid, b.H, b.W, b.Ho, b.Wo, b.d.k, b.d.s, n->dtype));
        if ((rc = probe(100 + i, Dbuf, Bc, b0, b.Ho * b.Wo, b.cmid, a.a_chunked, b.out_col ? b.Ho : 0))) return rc;
        if ((rc = launch_pw_gemm(a, b.proj.cfg, n->dtype, s))) return rc;
        if (n->probe_layer == 200 + i && n->probe_out)      // behind the GEMM: with the squeeze-excite in its prologue that is where the gate is written
            COSY_CHECK_HIP(hipMemcpyAsync(n->probe_out + (size_t)b0 * b.cmid, w.gate, (size_t)Bc * b.cmid * sizeof(float), hipMemcpyDeviceToDevice, s));
        pw_name(b.proj, a);
        if ((rc = mark(kn, i, pw_bytes(a, Bc), 2.0 * a.M * a.K * a.N, ((double)a.K * a.N + (double)a.M * a.N * (a.res ? 2 : 1)) * esz_d))) return rc;
        if (b.to_rowmajor) {
            if ((rc = launch_pixels_to_rowmajor(Ebuf, out, Bc, b.Ho, b.Wo, b.d.cout, n->dtype, s))) return rc;
            if ((rc = mark("pixels_to_rowmajor_kernel", i, 2.0 * Bc * b.Ho * b.Wo * b.d.cout * esz_d, 0.0, 0.0))) return rc;
        }
        return probe(i, out, Bc, b0, b.Ho * b.Wo, b.d.cout, out_chunked, b.out_col && !b.to_rowmajor ? b.Ho : 0, out_lw, out_lp);
    };
    auto out_is_chunked = [&](int i) -> int { return i + 1 < 26 && n->blk[i + 1].x_chunk; };
    auto out_perm_lp = [&](int i) -> int { return out_is_chunked(i) ? n->blk[i + 1].x_perm_lp : 0; };
    auto out_perm_lw = [&](int i) -> int { return out_perm_lp(i) ? ilog2(n->blk[i].Wo) : 0; };
    auto stage_tap_index = [&](int i) -> int { for (int q = 0; q < 7; ++q) if (STAGE_END[q] == i) return q + 1; return -1; };

    // ---- early segment, chunked
    const Block& last_e = n->blk[EARLY_BLOCKS - 1];
    const size_t handover = (size_t)last_e.Ho * last_e.Wo * (n->blk[EARLY_BLOCKS].x_chunk ? (size_t)((last_e.d.cout + 15) & ~15) : (size_t)last_e.d.cout) * e;
    const int chunk = std::min(n->chunk, n->maxB);
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = std::min(chunk, B - b0);
        const char* x = (const char*)n->X + (size_t)(x_off + b0) * n->H * n->W * 8 * e;
        // the stem tensor only exists when somebody wants to look at it (per-stage taps, test probe -1); otherwise the stem conv runs inside
        // block 0's front kernel
        const bool stemf = n->stem_fused && !taps && n->probe_layer != -1;
        if (!stemf) {
            if ((rc = launch_stem(x, n->stem_w, n->stem_scale, n->stem_bias, w.actc[0], Bc, n->H, n->W, n->Hs, n->Ws, n->dtype, s))) return rc;
            snprintf(kn, sizeof(kn), "stem_kernel<%s>", dt_name(n->dtype));
            if ((rc = mark(kn, -1, ((double)Bc * n->H * n->W * 8 + (double)Bc * n->Hs * n->Ws * STEM_C) * esz_d, 2.0 * Bc * n->Hs * n->Ws * STEM_C * IN_C * 9,
                           ((double)Bc * n->H * n->W * 8 + (double)Bc * n->Hs * n->Ws * STEM_C) * esz_d))) return rc;
            if ((rc = tap(w.actc[0], Bc, b0, n->Hs * n->Ws, STEM_C, 0))) return rc;
            if ((rc = probe(-1, w.actc[0], Bc, b0, n->Hs * n->Ws, STEM_C, 0))) return rc;
        }
        int cur = 0;
        for (int i = 0; i < EARLY_BLOCKS; ++i) {
            const Block& b = n->blk[i];
            void* out = (i == EARLY_BLOCKS - 1) ? (void*)((char*)w.act[0] + (size_t)b0 * handover) : w.actc[cur ^ 1];
            if ((rc = run_block(i, w.actc[cur], out, Bc, w.Ec, w.Dc, b0, i == 0 && stemf ? x : nullptr))) return rc;
            cur ^= 1;
            const int ti = stage_tap_index(i);
            if (ti >= 0 && (rc = tap(out, Bc, b0, b.Ho * b.Wo, b.d.cout, ti, b.out_col && !b.to_rowmajor ? b.Ho : 0, out_is_chunked(i), out_perm_lw(i), out_perm_lp(i)))) return rc;
        }
    }
    // ---- late segment, full batch
    int cur = 0;
    for (int i = EARLY_BLOCKS; i < 26; ++i) {
        const Block& b = n->blk[i];
        if ((rc = run_block(i, w.act[cur], w.act[cur ^ 1], B, w.E, w.D, 0))) return rc;
        cur ^= 1;
        const int ti = stage_tap_index(i);
        if (ti >= 0 && (rc = tap(w.act[cur], B, 0, b.Ho * b.Wo, b.d.cout, ti, b.out_col && !b.to_rowmajor ? b.Ho : 0, out_is_chunked(i), out_perm_lw(i), out_perm_lp(i)))) return rc;
    }
    PwArgs a{};
    a.A = w.act[cur]; a.Wp = n->head.Wp; a.out = w.Hd; a.scale = n->head.scale; a.bias = n->head.bias;
    a.M = B * n->Hf * n->Wf; a.K = HEAD_IN; a.N = HEAD_C; a.HW = n->Hf * n->Wf; a.silu = 1; a.zeros = n->zeros;
    if ((rc = launch_pw_gemm(a, n->head.cfg, n->dtype, s))) return rc;
    pw_name(n->head, a);
    if ((rc = mark(kn, 26, pw_bytes(a, B), 2.0 * a.M * a.K * a.N, ((double)a.M * a.K + (double)a.K * a.N) * esz_d))) return rc;
    if ((rc = tap(w.Hd, B, 0, n->Hf * n->Wf, HEAD_C, 8))) return rc;
    if ((rc = probe(26, w.Hd, B, 0, n->Hf * n->Wf, HEAD_C, 0))) return rc;
    if ((rc = launch_pool_fc(w.Hd, n->fc_w, n->fc_b, feat, w.featbuf, pose, B, n->Hf * n->Wf, n->dtype, s))) return rc;
    snprintf(kn, sizeof(kn), "pool_kernel<%s>+fc9_kernel", dt_name(n->dtype));
    if ((rc = mark(kn, 26, (double)B * n->Hf * n->Wf * HEAD_C * esz_d, 2.0 * B * HEAD_C * (n->Hf * n->Wf + N_POSE), (double)B * (HEAD_C + N_POSE) * 4))) return rc;
    if (prof) { n->prof_nslots = slot; ++n->prof_seg; }
    return COSY_OK;
}

// Whole-batch entry: one stream, or two half-batches on two internal streams (fork/join with events; capturable).
static int net_forward_top(cosy_net* n, int B, float* feat, float* pose, float* taps, hipStream_t s) {
    const bool dual = n->nstreams == 2 && B >= 32 && !taps && !n->prof_on && n->probe_layer == -2;
    n->last_split = dual ? (B + 1) / 2 : B;
    if (!dual) return net_forward(n, n->ws[0], 0, B, feat, pose, taps, s, true);
    const int B0 = n->last_split, B1 = B - B0;
    COSY_CHECK_HIP(hipEventRecord(n->ev_fork, s));
    int rc;
    for (int i = 0; i < 2; ++i) {
        COSY_CHECK_HIP(hipStreamWaitEvent(n->side[i], n->ev_fork, 0));
        const int off = i ? B0 : 0, cnt = i ? B1 : B0;
        if ((rc = net_forward(n, n->ws[i], off, cnt, feat ? feat + (size_t)off * HEAD_C : nullptr, pose + (size_t)off * N_POSE, nullptr,
                              n->side[i], false))) return rc;
        COSY_CHECK_HIP(hipEventRecord(n->ev_join[i], n->side[i]));
        COSY_CHECK_HIP(hipStreamWaitEvent(s, n->ev_join[i], 0));
    }
    return COSY_OK;
}

}  // namespace cosy

using namespace cosy;

extern "C" {

int cosy_version(void) { return COSY_VERSION; }
const char* cosy_last_error(void) { return g_err; }
long cosy_effnet_b3_param_count(void) { return param_count(); }

int cosy_effnet_b3_out_hw(int H, int W, int* oh, int* ow) {
    COSY_REQUIRE(H >= 32 && W >= 32 && oh && ow, "out_hw: bad arguments");
    int h = out_dim(H, 3, 2), w = out_dim(W, 3, 2);
    for (int i = 0; i < 26; ++i) { h = out_dim(h, B3[i].k, B3[i].s); w = out_dim(w, B3[i].k, B3[i].s); }
    *oh = h; *ow = w;
    return COSY_OK;
}

int cosy_effnet_b3_create(const float* host_params, size_t n_floats, int dtype, int H, int W, int max_batch, cosy_net_t** out) {
    COSY_REQUIRE(host_params && out, "create: null argument");
    COSY_REQUIRE(dtype == COSY_F32 || dtype == COSY_BF16 || dtype == COSY_F16, "create: dtype %d not supported (0=f32, 1=bf16, 2=f16)", dtype);
    // Supported crop sizes.  The fused fronts are built for the maps of 256x256 and 240x320 (the metric's and the reference's
    // crop size); any other size runs the shape-agnostic kernels (pw_gemm_dma / dwconv), which need: even sides (the stem's
    // stride-2 output is H/2 x W/2), a stem map that is a whole number of 16-pixel groups, and final maps of >= 16 pixels
    // (the gate rows of the samples under one GEMM tile must fit the LDS).  Anything else fails here, not later.
    COSY_REQUIRE(max_batch >= 1, "create: bad max_batch=%d", max_batch);
    COSY_REQUIRE(H >= 128 && W >= 128 && H <= 1024 && W <= 1024 && H % 16 == 0 && W % 16 == 0,
                 "create: crop size %dx%d not supported (sides must be multiples of 16 in [128, 1024])", H, W);
    if ((long)n_floats != param_count()) {
        set_error("create: parameter blob has %zu floats, expected %ld", n_floats, param_count());
        return COSY_ESIZE;
    }
    cosy_net* n = (cosy_net*)calloc(1, sizeof(cosy_net));
    if (!n) { set_error("create: host allocation failed"); return COSY_ENOMEM; }
    n->dtype = dtype; n->H = H; n->W = W; n->maxB = max_batch; n->esz = dtype == COSY_F32 ? 4 : 2;
    n->probe_layer = -2; n->probe_out = nullptr;
    n->Hs = out_dim(H, 3, 2); n->Ws = out_dim(W, 3, 2);
    {   // schedule knobs: fixed in the shipping build, env-overridable only under -DCOSY_TUNE (cosy_common.h)
        const int c = tune_int("COSY_EARLY_CHUNK", 0);   // measured: chunking the early segment is slower (kernels are issue-bound)
        n->chunk = c <= 0 ? max_batch : c;
        n->fuse = tune_int("COSY_FUSE", 1);
        // measured (256 crops): batched from block 19: +1.5 %, from 9: another +1.1 % over one-workgroup-per-sample everywhere; the early
        // blocks (Cmid <= 288, Cse <= 12) stay on se_kernel: two dependent launches cost what its one does
        n->se_batch_from = tune_int("COSY_SE_BATCH_FROM", 9);
        // round 4: blocks 5-13 compute the gate inside the project GEMM's prologue (no squeeze-excite launch at all).  Measured per block
        // at 256 crops (profiles/r04_se_fused_ab.txt): the prologue adds 6-11 us to the GEMM (5-8 dependent L2 round trips under the
        // DMA streams of the co-resident workgroups) against 8-14 us of squeeze-excite kernel(s): -2..-6 us per block, 9 launches
        // fewer per forward.  NOT for blocks 0-4 (their project GEMMs stream 8,000-16,000 workgroups and every one would redo the
        // gate: block 0 160 -> 458 us, blocks 2-4 +13..22 us), block 18 (+3 us) or blocks 19-25 (0.65 / 1.77 MB of FC weights per gate:
        // the size rule in build_weights keeps them on the batched kernels, where a 16-sample tile shares one read of them).
        // Same-box A/B against the round-3 tree (profiles/r04_vs_r03_layers.txt): blocks 5-13 gain 2-6 us each.  Blocks 14-17 (Cse = 34, Cmid = 816:
        // the largest prologue): fused or not makes no measurable difference (round 5, alternating same-call A/B: backbone 4.551 vs 4.553 ms,
        // profiles/r05_se_fused_ab.txt; round 4's two records disagreed) -> they keep the batched kernels.
        n->se_fuse_mask = (unsigned)tune_int("COSY_SE_FUSE_MASK", 0x3fe0);
        // the fused stem front reaches the zero page (directly behind X, layout_workspace) by a 32-bit offset from the chunk's X pointer
        // (launch_stem_front requires it below 2^32 - 2^24): an engine whose input buffer is larger than that (>= 4080 crops of 256x256 in a
        // 16-bit type, i.e. a capacity of 4096) keeps the unfused stem + block 0, which has no such limit, instead of failing every forward
        n->stem_fused = n->fuse && stem_front_supported(dtype, H, W) &&
                        (size_t)max_batch * H * W * 8 * n->esz + 256 < ((size_t)1 << 32) - ((size_t)1 << 24);
        n->stemf_w = nullptr; n->stemf_params = nullptr;
        // bf16 (round 6: hi + lo weight pairs in the GEMM, the wave fronts and the matrix-pipe form of the 8x8-map front): mbconv_small_kernel (7x10 maps) and the
        // LDS-tiled front do not carry the pairs -- their blocks run the unfused kernels (pw_gemm_dma -> E -> dwconv), which do
        const bool pairs = dtype == COSY_BF16;
        n->small_mask = (unsigned)tune_int("COSY_SMALL_MASK", 0x3f80000);      // (bf16: only where the matrix-pipe form exists, build_weights)
        n->tile_mask = pairs ? 0u : (unsigned)tune_int("COSY_TILE_MASK", 0x13c);         // blocks 2-5 and 8 (measured in round 1: it loses on the k=5 stride-1 blocks 6/7)   // blocks 19-25 (8x8 / 7x10 maps): whole-image kernel
        n->wave_mask = (unsigned)tune_int("COSY_WAVE_MASK", 0x3fffc);   // blocks 2-17: maps 16..128 pixels wide, stride per shape table
        n->nstreams = tune_int("COSY_STREAMS", 1) == 2 && max_batch >= 32 ? 2 : 1;   // measured: 2 streams x half batches is ~10 % slower
    }
    hipError_t herr = hipSuccess;
    Bump wb;
    const long used = build_weights(n, host_params, wb, false, &herr);
    if (used != param_count()) { set_error("create: internal blob walk mismatch %ld", used); free(n); return COSY_EINVAL; }
    n->wbytes = wb.off + 256;
    Bump ab;
    layout_workspace(n, ab);
    n->abytes = ab.off + 256;
    if (hipMalloc(&n->wbase, n->wbytes) != hipSuccess || hipMalloc(&n->abase, n->abytes) != hipSuccess) {
        set_error("create: hipMalloc of %zu + %zu bytes failed", n->wbytes, n->abytes);
        if (n->wbase) (void)hipFree(n->wbase);
        free(n);
        return COSY_ENOMEM;
    }
    wb.base = (char*)n->wbase; wb.off = 0;
    {
        std::vector<char> mirror(n->wbytes, 0);
        wb.mirror = &mirror;
        build_weights(n, host_params, wb, true, &herr);
        if (herr == hipSuccess) herr = hipMemcpy(n->wbase, mirror.data(), wb.off, hipMemcpyHostToDevice);      // the whole weight slab, one copy
        wb.mirror = nullptr;
    }
    ab.base = (char*)n->abase; ab.off = 0;
    layout_workspace(n, ab);
    if (herr == hipSuccess) herr = hipMemset(n->abase, 0, n->abytes);
    // hipMemset on device memory returns before the fill has run, and it runs on the NULL stream: the caller's streams (torch creates them
    // non-blocking) are not ordered behind it -- a forward launched right after create() could have its input / workspaces zeroed under it
    // (seen as a 1-in-15 bit mismatch of test_refinement_loop_with_on_device_renderer, which rebuilds four engines and launches at once).
    if (herr == hipSuccess) herr = hipDeviceSynchronize();
    if (herr != hipSuccess) {
        set_error("create: weight upload failed: %s", hipGetErrorString(herr));
        (void)hipFree(n->wbase); (void)hipFree(n->abase); free(n);
        return COSY_EHIP;
    }
    if (n->nstreams == 2) {
        for (int i = 0; i < 2; ++i) {
            if (hipStreamCreateWithFlags(&n->side[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&n->ev_join[i], hipEventDisableTiming) != hipSuccess) n->nstreams = 1;
        }
        if (hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming) != hipSuccess) n->nstreams = 1;
    }
    *out = n;
    return COSY_OK;
}

int cosy_effnet_b3_set_profiling(cosy_net_t* n, int enable) {
    COSY_REQUIRE(n, "set_profiling: null net");
    if (enable && !n->prof_ev) {
        const size_t ne = (size_t)PROF_SEGS * (PROF_SLOTS + 1);
        n->prof_ev = (hipEvent_t*)calloc(ne, sizeof(hipEvent_t));
        n->prof_rec = (cosy_prof_rec_t*)calloc(PROF_SLOTS, sizeof(cosy_prof_rec_t));
        if (!n->prof_ev || !n->prof_rec) { set_error("set_profiling: host allocation failed"); return COSY_ENOMEM; }
        for (size_t i = 0; i < ne; ++i) COSY_CHECK_HIP(hipEventCreate(&n->prof_ev[i]));
    }
    n->prof_on = enable ? 1 : 0;
    n->prof_seg = 0;
    return COSY_OK;
}

int cosy_effnet_b3_profile_read(cosy_net_t* n, cosy_prof_rec_t* recs, int cap, int* n_out) {
    COSY_REQUIRE(n && recs && n_out, "profile_read: null argument");
    COSY_REQUIRE(n->prof_ev, "profile_read: profiling was never enabled");
    const int ns = n->prof_nslots, segs = n->prof_seg;
    *n_out = 0;
    if (segs == 0) return COSY_OK;
    COSY_REQUIRE(cap >= ns, "profile_read: need room for %d records", ns);
    for (int i = 0; i < ns; ++i) { recs[i] = n->prof_rec[i]; recs[i].n = 0; recs[i].ms_avg = 0.f; recs[i].ms_min = 1e30f; }
    for (int g = 0; g < segs; ++g) {
        hipEvent_t* ev = n->prof_ev + (size_t)g * (PROF_SLOTS + 1);
        COSY_CHECK_HIP(hipEventSynchronize(ev[ns]));
        for (int i = 0; i < ns; ++i) {
            float ms = 0.f;
            COSY_CHECK_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            recs[i].ms_avg += ms; recs[i].n += 1;
            if (ms < recs[i].ms_min) recs[i].ms_min = ms;
        }
    }
    for (int i = 0; i < ns; ++i) recs[i].ms_avg /= (float)recs[i].n;
    *n_out = ns;
    n->prof_seg = 0;
    return COSY_OK;
}

int cosy_effnet_b3_set_probe(cosy_net_t* n, int layer, float* out) {
    COSY_REQUIRE(n, "set_probe: null net");
    COSY_REQUIRE(layer == -2 || out, "set_probe: null output");
    COSY_REQUIRE(layer >= -2 && layer < 226, "set_probe: bad layer %d", layer);
    n->probe_layer = layer; n->probe_out = layer == -2 ? nullptr : out;
    return COSY_OK;
}

int cosy_effnet_b3_block_info(const cosy_net_t* n, int i, int* dims) {
    COSY_REQUIRE(n && dims && i >= 0 && i < 26, "block_info: bad arguments");
    const Block& b = n->blk[i];
    // where the project GEMM applies the squeeze-excite gate: to the weight fragments (maps of a multiple of 64 pixels: a wave's 64
    // rows belong to one sample) or to the activation rows
    const int gate_w = pw_gate_on_weights(b.Ho * b.Wo, n->dtype);
    const int v[11] = {b.H, b.W, b.Ho, b.Wo, b.d.cin, b.cmid, b.d.cout, i == 0 && n->stem_fused ? 4 : b.wave ? (wave_taps_on_mfma(b.d.cin, b.cmid, b.d.k, b.d.s, n->dtype, b.H, b.W) ? 5 : 1) : b.smx ? 6 : b.small ? 2 : b.tiled ? 3 : 0, b.d.k, b.d.s, gate_w};
    for (int q = 0; q < 11; ++q) dims[q] = v[q];
    return COSY_OK;
}

int cosy_effnet_b3_destroy(cosy_net_t* n) {
    if (!n) return COSY_OK;
    if (n->prof_ev) {
        for (size_t i = 0; i < (size_t)PROF_SEGS * (PROF_SLOTS + 1); ++i) (void)hipEventDestroy(n->prof_ev[i]);
        free(n->prof_ev); free(n->prof_rec);
    }
    if (n->nstreams == 2) {
        for (int i = 0; i < 2; ++i) { (void)hipStreamDestroy(n->side[i]); (void)hipEventDestroy(n->ev_join[i]); }
        (void)hipEventDestroy(n->ev_fork);
    }
    (void)hipFree(n->wbase); (void)hipFree(n->abase);
    free(n);
    return COSY_OK;
}

size_t cosy_effnet_b3_workspace_bytes(const cosy_net_t* n) { return n ? n->abytes + n->wbytes : 0; }

int cosy_effnet_b3_set_input_nchw(cosy_net_t* n, const float* x, int B, cosy_stream_t stream) {
    COSY_REQUIRE(n && x, "set_input: null argument");
    COSY_REQUIRE(B >= 0 && B <= n->maxB, "set_input: batch %d exceeds max_batch %d", B, n->maxB);
    return launch_pack_nchw(n->X, n->dtype, x, B, n->H, n->W, (hipStream_t)stream);
}

int cosy_frames_u8_to_nhwc4(const unsigned char* images, float* out, int N, int h, int w, cosy_stream_t stream) {
    COSY_REQUIRE(images && out, "frames_u8_to_nhwc4: null argument");
    return launch_frames_u8_to_nhwc4(images, out, N, h, w, (hipStream_t)stream);
}
int cosy_frames_to_nhwc4(const float* images, float* out, int N, int h, int w, cosy_stream_t stream) {
    COSY_REQUIRE(images && out, "frames_to_nhwc4: null argument");
    return launch_frames_to_nhwc4(images, out, N, h, w, (hipStream_t)stream);
}

int cosy_crop_pack(cosy_net_t* n, const float* images, const int* im_id, const float* boxes_crop, const float* renders, int B,
                   int N, int h, int w, cosy_stream_t stream) {
    COSY_REQUIRE(n, "crop_pack: null net");
    if (B == 0) return COSY_OK;   // an empty batch carries null data pointers (an empty device tensor has none)
    COSY_REQUIRE(images && boxes_crop && renders, "crop_pack: null argument");
    COSY_REQUIRE(B >= 0 && B <= n->maxB, "crop_pack: batch %d exceeds max_batch %d", B, n->maxB);
    return launch_crop_pack(n->X, n->dtype, images, im_id, boxes_crop, renders, B, N, h, w, n->H, n->W, n->crop_taps, (hipStream_t)stream);
}

int cosy_render_crop_pack(cosy_net_t* n, const cosy_mesh_t* mesh, const cosy_shade_t* shade, const int* obj_id, const float* TCO,
                          const float* K_crop, const float* frames_nhwc4, const int* im_id, const float* boxes_crop, int B, int N, int h,
                          int w, void* scratch, cosy_stream_t stream) {
    COSY_REQUIRE(n, "render_crop_pack: null net");
    COSY_REQUIRE(B >= 0 && B <= n->maxB, "render_crop_pack: batch %d exceeds max_batch %d", B, n->maxB);
    return render_crop_pack(n->X, n->dtype, mesh, shade, obj_id, TCO, K_crop, frames_nhwc4, im_id, boxes_crop, B, N, h, w, n->H, n->W, scratch,
                            (hipStream_t)stream);
}

int cosy_effnet_b3_forward(cosy_net_t* n, int B, float* feat, float* pose9, float* taps, cosy_stream_t stream) {
    COSY_REQUIRE(n, "forward: null net");
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(pose9, "forward: null argument");
    COSY_REQUIRE(B >= 0 && B <= n->maxB, "forward: batch %d exceeds max_batch %d", B, n->maxB);
    return net_forward_top(n, B, feat, pose9, taps, (hipStream_t)stream);
}

int cosy_effnet_b3_features_nchw(cosy_net_t* n, int B, float* out, cosy_stream_t stream) {
    COSY_REQUIRE(n && out, "features_nchw: null argument");
    COSY_REQUIRE(B >= 0 && B <= n->maxB, "features_nchw: batch %d exceeds max_batch %d", B, n->maxB);
    const int B0 = n->last_split < B ? n->last_split : B;   // the head activation lives in two workspaces after a split forward
    int rc = launch_nhwc_to_nchw(n->ws[0].Hd, B0, n->Hf * n->Wf, HEAD_C, n->dtype, out, (hipStream_t)stream);
    if (rc || B0 == B) return rc;
    return launch_nhwc_to_nchw(n->ws[1].Hd, B - B0, n->Hf * n->Wf, HEAD_C, n->dtype, out + (size_t)B0 * n->Hf * n->Wf * HEAD_C, (hipStream_t)stream);
}

int cosy_crop_geometry(const float* pts_table, const int* obj_id, const float* K, const int* im_id, const float* TCO, int B, int P,
                       float z_min, int im_h, int im_w, int out_h, int out_w, float lamb, float* boxes_rend, float* boxes_crop,
                       float* K_crop, cosy_stream_t stream) {
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(pts_table && obj_id && K && TCO && boxes_rend && boxes_crop && K_crop, "crop_geometry: null argument");
    COSY_REQUIRE(B >= 0 && P >= 1, "crop_geometry: bad sizes B=%d P=%d", B, P);
    return launch_crop_geometry(pts_table, obj_id, K, im_id, TCO, B, P, z_min, im_h, im_w, out_h, out_w, lamb, boxes_rend,
                                boxes_crop, K_crop, (hipStream_t)stream);
}

int cosy_roi_align(const float* images, const int* im_id, const float* boxes, int B, int N, int C, int h, int w, int out_h,
                   int out_w, int sampling_ratio, float* out, cosy_stream_t stream) {
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(images && boxes && out, "roi_align: null argument");
    return launch_roi_align(images, im_id, boxes, B, N, C, h, w, out_h, out_w, sampling_ratio, out, (hipStream_t)stream);
}

int cosy_pose_update(const float* TCO_in, const float* K_crop, const float* pose9, int B, float* TCO_out, cosy_stream_t stream) {
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(TCO_in && K_crop && pose9 && TCO_out, "pose_update: null argument");
    return launch_pose_update(TCO_in, K_crop, pose9, B, TCO_out, (hipStream_t)stream);
}

int cosy_tco_init_from_boxes(const float* boxes, const float* K, const int* im_id, int B, float z, float* TCO, cosy_stream_t stream) {
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(boxes && K && TCO, "tco_init_from_boxes: null argument");
    return launch_tco_init_from_boxes(boxes, K, im_id, B, z, TCO, (hipStream_t)stream);
}

int cosy_tco_init_zup_autodepth(const float* boxes, const float* pts_table, const int* obj_id, const float* K, const int* im_id,
                                int B, int P, float* TCO, cosy_stream_t stream) {
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(boxes && pts_table && obj_id && K && TCO, "tco_init_zup_autodepth: null argument");
    return launch_tco_init_zup(boxes, pts_table, obj_id, K, im_id, B, P, TCO, (hipStream_t)stream);
}

int cosy_scatter_argmin(const float* dists, const int* ids, int M, int n_seg, int* out, cosy_stream_t stream) {
    COSY_REQUIRE(dists && ids && out, "scatter_argmin: null argument");
    return launch_scatter_argmin(dists, ids, M, n_seg, out, (hipStream_t)stream);
}

}  // extern "C"
