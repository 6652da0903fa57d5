// NOT BUILT -- record of a measured dead end of round 5 (profiles/r05_dead_ends.txt): block 0's project conv + block 1's depthwise as one
// wave-autonomous kernel.  It lived in cosypose_amd/csrc/kernels_stem.hip (it uses that file's helpers: sf_dpp_*, sf_silu4, SF_PF) behind the stem
// front; parity was green (storage-emulating oracle, block kind 5), the time was not: 216 us against 130 + 111 us for the two kernels it replaced, while
// block 1's project GEMM lost 12.6 us on the chunked D1 -- and the two-stream step got 1.2 % SLOWER.
// ==========================================================================================
// Block 0's project conv + block 1's depthwise front as ONE kernel (the same wave design; round 5):
//     D0 (chunked, 40 channels) x squeeze-excite gate -> project 1x1 40 -> 24 (MFMA) -> BN -> X1 (stored: block 1's residual), and in the same
//     registers -> block 1's depthwise 3x3 (expand_ratio 1: no expansion) -> BN -> SiLU -> D1 (chunked), squeeze sums
// replacing pw_gemm_dma (block 0's project) + dwconv_kernel (block 1): X1 is written once and never read back by a depthwise kernel, one launch
// less.  Reference: MBConvBlock.forward, efficientnet.py:89-91 (block 0: _bn2(_project_conv(x * gate))) and :79-84 (block 1: depthwise, BN, swish).
// Numerics are those of the two kernels it replaces: the gate multiplies the WEIGHT fragments in the storage type exactly as pw_gemm_dma does
// (fp16: packed half products; bf16: fp32 product rounded once), the k-blocks accumulate in the same order, and the depthwise reads X1 ROUNDED to
// the storage type (what dwconv_kernel would have loaded), so block 1's D is compared by the storage-emulating oracle like an unfused block's.
// job = one wavefront = (sample, 16-channel chunk of the 24 -> 32 channels, half row, row band); a workgroup = the two chunk jobs of one (sample,
// half, band).  Fragment (q, kb) of lane (p, kg) = 8 channels of pixel 16 q + p: 16 bytes of one chunk row of D0 -- 16 lanes = 512 contiguous bytes.
// ==========================================================================================
struct ProjFrontKArgs {
    const void* D0;           // [sample][3][HW][16] block 0's depthwise output (stem front)
    const float* gate;        // (B, 40) block 0's squeeze-excite gate
    const void* Wp;           // block 0's project weights as MFMA A fragments: [chunk 2][k-block 2][lane 64][8]
    const float* params;      // [chunk 2][4 + 9][16]: s2, b2 (block 0's BatchNorm 2), s1, b1 (block 1's BatchNorm 1), block 1's taps
    void* X1;                 // (B, HW, 24) block 0's output = block 1's input (NHWC)
    void* D1;                 // [sample][2][HW][16]
    float* partial;           // (B, n_tiles, 24)
    void* dump;
    int B, Hs, Ws, rsplit, rows_per, n_tiles;
};

template <typename T, int PPL>
__global__ __launch_bounds__(128, 2) void proj_front_kernel(ProjFrontKArgs a) {
    using raw_t = typename DT<T>::raw_t;
    constexpr int KBN = 2, NQ = PPL + 1, PF = SF_PF, SEGW = 16 * PPL;
    typedef T out_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float sf_smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, kg = lane >> 4;
    const int id = blockIdx.x, xcd = id & 7, g = id >> 3;
    const int wps = 2 * a.rsplit;
    const int b = (g / wps) * 8 + xcd, jrem = g % wps;
    const int ch = wave, seg = jrem & 1, band = jrem >> 1;
    if (b >= a.B) return;
    const int HW = a.Hs * a.Ws;

    float* P = sf_smem + wave * (PF + KBN * 256);
    const float* Pl = P + kg * 4;
    {
        const f32x4* PP = (const f32x4*)(a.params + (size_t)ch * PF);
        if (lane < PF / 4) *(f32x4*)(P + lane * 4) = PP[lane];
    }
    // weight fragments x this sample's gate, once per job: lane (row i, kg) holds k = 32 kb + 8 kg .. + 7
    char* Wl = (char*)(P + PF);
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
        raw_t w = *(const raw_t*)((const T*)a.Wp + ((size_t)(ch * 2 + kb) * 64 + lane) * 8);
        const int k0 = kb * 32 + kg * 8;
        float gq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gq[e] = k0 + e < 40 ? a.gate[(size_t)b * 40 + min(k0 + e, 39)] : 0.f;
        if constexpr (__is_same(T, f16_t)) {
            f16x8 g8;
#pragma unroll
            for (int e = 0; e < 8; ++e) g8[e] = (f16_t)gq[e];
            w = w * g8;                                     // packed half products, as pw_gemm_dma's weight-side gate
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (T)((float)w[e] * gq[e]);
        }
        *(raw_t*)(Wl + kb * 1024 + lane * 16) = w;
    }

    // fragment (q, kb): chunk c = 2 kb + (kg >> 1) of D0 (chunk 3 does not exist: k >= 48 meets zero weights -- read chunk 2 instead), 8 channels
    unsigned xo[KBN], so[KBN];
    const int xs = seg == 0 ? SEGW : SEGW - 1;                  // pixel beyond the seam
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
        const int c = min(2 * kb + (kg >> 1), 2);
        xo[kb] = (unsigned)((c * HW + seg * SEGW + p) * 32 + (kg & 1) * 16);
        so[kb] = (unsigned)((c * HW + xs) * 32 + (kg & 1) * 16);
    }
    const char* Db = (const char*)a.D0 + (size_t)b * 3 * HW * 32;
    raw_t x[NQ][KBN];
    auto load_row = [&](int sy) {
        const int syc = min(sy, a.Hs - 1);
        const char* rowp = Db + (size_t)syc * a.Ws * 32;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int kb = 0; kb < KBN; ++kb) {
                const unsigned off = q < PPL ? xo[kb] : so[kb];
                const int imm = q < PPL ? 512 * q : 0;
                x[q][kb] = *(const raw_t*)(rowp + off + imm);
            }
    };
    const int oy_a = band * a.rows_per, oy_b = min(a.Hs, oy_a + a.rows_per);
    const int sy0 = (max(oy_a - 1, 0) / 3) * 3;
    const int n3 = (oy_b - sy0) / 3 + 1;
    load_row(sy0);

    float acc[3][PPL][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < PPL; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[s][t][c] = 0.f;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool real = ch * 16 + kg * 4 < 24;                      // this lane's channel quad exists (chunk 1 holds channels 16..23 only)
    // the loop is entered with the memory queue in the state its back edge leaves: [fragment loads][X1 row stores][D1 row stores]
#pragma unroll
    for (int t = 0; t < 2 * PPL; ++t) *(out_t*)((T*)a.dump + (seg * SEGW + 16 * (t % PPL) + p) * 16 + kg * 4) = out_t{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    const float sl = seg == 1 ? 1.f : 0.f, sr = seg == 0 ? 1.f : 0.f;
    T* __restrict__ Dch = (T*)a.D1 + (size_t)(b * 2 + ch) * HW * 16;
    const int dlane = (seg * SEGW + p) * 16 + kg * 4;
    const int drow = a.Ws * 16;
    T* __restrict__ X1b = (T*)a.X1 + (size_t)b * HW * 24;
    // chunk 1 holds channels 16..23 only: its lanes kg = 2, 3 carry DUPLICATES of kg = 0, 1 (duplicated weight rows and parameters, proj_front_pack),
    // so every lane stores valid data -- the same value to the same address twice -- and no store needs a mask
    const int xlane = (seg * SEGW + p) * 24 + ch * 16 + (ch == 1 ? (kg & 1) : kg) * 4;
    const int xrow = a.Ws * 24;
    const long x_dump_rel = (T*)a.dump - X1b, d_dump_rel = (T*)a.dump - Dch;      // the dump row seen from the two output bases

    auto row = [&](auto uc, const int base, const long xbase_off, const long dbase_off) {
        constexpr int u = decltype(uc)::value;
        const int sy = base + u;
        asm volatile("" ::: "memory");
        // ---- A. block 0's output row sy: MFMAs, the next row's loads, BN, the row's store, rounding to the storage type
        f32x4 m[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            m[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KBN; ++kb) mma(m[q], *(const raw_t*)(Wl + kb * 1024 + lane * 16), x[q][kb]);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_row(sy + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float rv = sy < a.Hs ? 1.f : 0.f;
        const bool xvalid = sy >= oy_a && sy < oy_b;              // this band owns row sy of X1 (halo rows belong to the neighbour bands)
        const long xrow_off = xbase_off + u * xrow;      // (the 64-bit products are formed once per three rows, outside: a multiply in a select arm made hipcc branch)
        float sc0[4], bi0[4];
        load4(Pl + 0 * 16, sc0); load4(Pl + 1 * 16, bi0);
        float E[PPL][4];
        float seam[4];
        T* xrowp = X1b + (xvalid ? xrow_off : x_dump_rel) + xlane;      // a select between two ready offsets (hipcc turned a select of pointers into a branch here)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            out_t yo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = m[q][e] * sc0[e] + bi0[e];         // BatchNorm 2 of block 0 (no activation: efficientnet.py:90)
                if constexpr (__is_same(T, f16_t)) yo[e] = to_f16_sat(y); else yo[e] = (T)y;
                const float yr = (float)yo[e] * rv;                 // what a depthwise kernel would load from X1
                if (q < PPL) E[q][e] = yr; else seam[e] = yr;
            }
            if (q < PPL) *(out_t*)(xrowp + q * (16 * 24)) = yo;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- B. block 1's depthwise: row sy of X1 is tap row ky of output row sy + 1 - ky
#pragma unroll
        for (int t = 0; t < PPL; ++t) {
            asm volatile("" ::: "memory");
            float X3[3][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float lo = t > 0 ? sf_dpp_ror1(E[t - 1][c]) : seam[c] * sl;
                const float hi = t < PPL - 1 ? sf_dpp_rol1(E[t + 1][c]) : seam[c] * sr;
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
        // ---- C. output row sy - 1 of block 1's depthwise is complete
        {
            const long drow_off = dbase_off + u * drow;
            const int os = (u + 2) % 3;
            const int oy = sy - 1;
            const bool valid = oy >= oy_a && oy < oy_b;
            const float fv = valid ? 1.f : 0.f;
            float sc1[4], bi1[4];
            load4(Pl + 2 * 16, sc1); load4(Pl + 3 * 16, bi1);
            T* o = Dch + (valid ? drow_off : d_dump_rel) + dlane;
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
        const long xb = (long)base * xrow, db = (long)(base - 1) * drow;
        row(std::integral_constant<int, 0>{}, base, xb, db);
        row(std::integral_constant<int, 1>{}, base, xb, db);
        row(std::integral_constant<int, 2>{}, base, xb, db);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = sum[c];
        v += sf_dpp_mov0<0x111>(v); v += sf_dpp_mov0<0x112>(v); v += sf_dpp_mov0<0x114>(v); v += sf_dpp_mov0<0x118>(v);
        sum[c] = v;
    }
    if (p == 15 && real)
        *(f32x4*)(a.partial + ((size_t)b * a.n_tiles + band * 2 + seg) * 24 + ch * 16 + kg * 4) = f32x4{sum[0], sum[1], sum[2], sum[3]};
}

// ---- block 0's project + block 1's depthwise
size_t proj_front_weight_elems() { return (size_t)2 * 2 * 64 * 8; }
size_t proj_front_param_floats() { return (size_t)2 * SF_PF; }
// w: block 0's project weights (24, 40); s2 / b2: its folded BatchNorm 2 (24); dww: block 1's depthwise taps [tap][24]; s1 / b1: its folded BatchNorm 1 (24)
void proj_front_pack(const float* w, const float* s2, const float* b2, const float* dww, const float* s1, const float* b1, int dtype, void* wdst, float* pdst) {
    // 32 rows: channels 24..31 are COPIES of 16..23 (the kernel's lanes without a channel quad of their own compute duplicates; k >= 40 is zero)
    std::vector<float> w32((size_t)32 * 40);
    for (int n = 0; n < 32; ++n)
        for (int k = 0; k < 40; ++k) w32[(size_t)n * 40 + k] = w[(size_t)(n < 24 ? n : n - 8) * 40 + k];
    pw_pack_weights(w32.data(), 40, 32, PwCfg{1, 1}, dtype, wdst);            // [n-tile 2][k-block 2][lane][8]
    for (int ch = 0; ch < 2; ++ch)
        for (int c = 0; c < 16; ++c) {
            float* d = pdst + (size_t)ch * SF_PF + c;
            const int cc = ch * 16 + c < 24 ? ch * 16 + c : ch * 16 + c - 8;
            d[0 * 16] = s2[cc]; d[1 * 16] = b2[cc]; d[2 * 16] = s1[cc]; d[3 * 16] = b1[cc];
            for (int t = 0; t < 9; ++t) d[(4 + t) * 16] = dww[(size_t)t * 24 + cc];
        }
}
int launch_proj_front(const ProjFrontArgs& f, int dtype, int* n_tiles_out, hipStream_t s) {
    static const int rsplit = std::min(std::max(tune_int("COSY_PROJ_RSPLIT", SF_RSPLIT), 1), (int)SF_RSPLIT_MAX);
    *n_tiles_out = 2 * rsplit;
    if (f.B == 0) return COSY_OK;
    COSY_REQUIRE(dtype != COSY_F32 && (f.Ws == 128 || f.Ws == 160), "proj_front: unsupported map %dx%d / dtype %d", f.Hs, f.Ws, dtype);
    ProjFrontKArgs k;
    k.D0 = f.D0; k.gate = f.gate; k.Wp = f.Wp; k.params = f.params; k.X1 = f.X1; k.D1 = f.D1; k.partial = f.partial; k.dump = f.dump;
    k.B = f.B; k.Hs = f.Hs; k.Ws = f.Ws; k.rsplit = rsplit; k.rows_per = cdiv(k.Hs, k.rsplit); k.n_tiles = 2 * rsplit;
    const long wgs_per_xcd = (long)cdiv(f.B, 8) * 2 * k.rsplit;
    const dim3 grid((unsigned)(wgs_per_xcd * 8)), block(128);
    const size_t lds = (size_t)2 * (SF_PF + 2 * 256) * sizeof(float);
    if (f.Ws == 128) {
        if (dtype == COSY_BF16) hipLaunchKernelGGL((proj_front_kernel<bf16_t, 4>), grid, block, lds, s, k);
        else hipLaunchKernelGGL((proj_front_kernel<f16_t, 4>), grid, block, lds, s, k);
    } else {
        if (dtype == COSY_BF16) hipLaunchKernelGGL((proj_front_kernel<bf16_t, 5>), grid, block, lds, s, k);
        else hipLaunchKernelGGL((proj_front_kernel<f16_t, 5>), grid, block, lds, s, k);
    }
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

