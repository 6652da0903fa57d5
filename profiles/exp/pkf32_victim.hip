// Stand-alone victim / aggressor pair for the round-4 finding "a wave's packed-fp32 results are wrong while another wave of its SIMD issues 16-bit MFMAs"
// (profiles/r04_raster_streams.txt).  The round-4 stand-alone attempt (mfma_victim.hip) had no packed-fp32 instruction in its victim and did not reproduce.
//
//   victim      : every lane runs a chain of packed-fp32 instructions on its own data and, beside it, the same arithmetic with scalar instructions
//                 (IEEE fma / mul / add are exact per element, so packed == scalar bit for bit on a healthy machine).  Forms:
//                   pkfma  : v_pk_fma_f32 (inline asm, plain operands)            pkmuladd : v_pk_mul_f32 + v_pk_add_f32 (inline asm)
//                   pkvec  : float2 ext-vector C++ (what hipcc's SLP vectoriser emits: pk ops with op_sel swizzles chosen by the compiler)
//                   scalar : v_fma_f32 only (control)
//                 at 1 / 2 / 4 victim waves per SIMD (the workgroup's dynamic LDS sets how many fit beside the aggressor).
//   aggressor   : persistent waves (256 CUs x AW workgroups of 256 threads = AW waves per SIMD) of back-to-back MFMAs on ANOTHER stream:
//                   none | f16 VGPR acc | f16 AGPR acc | bf16 VGPR acc | f32 (16x16x4) VGPR acc | f16 VGPR acc + dependent VALU/trans work between the MFMAs
//   check       : every victim launch is compared bit for bit (a) packed vs scalar inside the lane, (b) with the quiet run of the same launch.
// build (cross-compiles here): hipcc --offload-arch=gfx950 -O3 -o profiles/exp/pkf32_victim profiles/exp/pkf32_victim.hip ; run on the GPU box: profiles/exp/pkf32_victim [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { AG_NONE = 0, AG_F16_V, AG_F16_A, AG_BF16_V, AG_F32_V, AG_F16_V_DEP, AG_N };
static const char* ag_name[AG_N] = {"none", "f16 16x16x32 VGPR acc", "f16 16x16x32 AGPR acc", "bf16 16x16x32 VGPR acc", "f32 16x16x4 VGPR acc", "f16 VGPR acc + dependent VALU/exp"};

template <int MODE>
__global__ __launch_bounds__(256, 4) void aggressor(const _Float16* __restrict__ src, float* __restrict__ dst, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a = *(const f16x8*)(src + ((blockIdx.x * 256 + threadIdx.x) % 4096) * 8), b = *(const f16x8*)(src + lane * 8);
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int k = 0; k < iters; ++k) {
        if constexpr (MODE == AG_F16_V || MODE == AG_F16_V_DEP) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_f16 %1, %5, %4, %1\n v_mfma_f32_16x16x32_f16 %2, %4, %4, %2\n v_mfma_f32_16x16x32_f16 %3, %5, %5, %3"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        } else if constexpr (MODE == AG_F16_A) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_f16 %1, %5, %4, %1\n v_mfma_f32_16x16x32_f16 %2, %4, %4, %2\n v_mfma_f32_16x16x32_f16 %3, %5, %5, %3"
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
        } else if constexpr (MODE == AG_BF16_V) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %5, %4, %1\n v_mfma_f32_16x16x32_bf16 %2, %4, %4, %2\n v_mfma_f32_16x16x32_bf16 %3, %5, %5, %3"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        } else if constexpr (MODE == AG_F32_V) {
            const float fa = (float)a[0], fb = (float)b[1];
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %5, %4, %1\n v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n v_mfma_f32_16x16x4_f32 %3, %5, %5, %3"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(fa), "v"(fb));
        }
        if constexpr (MODE == AG_F16_V_DEP) {     // what the wave-autonomous fronts do with their MFMA results: BN, SiLU (exp + rcp), DPP neighbours
            c0[0] = c0[0] * 0.999f + c1[1];
            c2[2] = c2[2] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-c3[3] * 1e-3f));
            c1[0] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c0[1]), 0x111, 0xf, 0xf, true));
        } else {
            asm volatile("" : "+v"(c0), "+v"(c1));      // keep the loop body as written
        }
    }
    dst[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

enum { V_PKFMA = 0, V_PKMULADD, V_PKVEC, V_PKFMA_SGPR, V_PKFMA_OPSEL, V_PKFMA_NEG, V_PAIR_PLAIN, V_PAIR_SWZ, V_PAIR_NOP, V_SRC1_SWZ, V_SRC1_SWZ_NEG, V_EXACT, V_SCALAR, V_N };
static const char* v_name[V_N] = {"v_pk_fma_f32 v,v,v,v (asm, no modifiers)", "v_pk_mul_f32 + v_pk_add_f32 (asm, no modifiers)", "float2 C++ (compiler's pk forms: SGPR operands, op_sel, neg)",
                                  "v_pk_fma_f32 with an SGPR-pair source (asm)", "v_pk_fma_f32 with op_sel / op_sel_hi swizzle (asm)", "v_pk_fma_f32 with neg_lo / neg_hi (asm)",
                                  "two DEPENDENT v_pk_fma_f32 back to back, no modifiers (asm)", "v_pk_fma_f32 -> dependent v_pk_fma_f32 reading the halves SWAPPED (op_sel), back to back (asm)",
                                  "the same swapped pair with s_nop 4 between the two (asm)",
                                  "v_pk_fma_f32 -> dependent v_pk_fma_f32 taking the result as SRC1 with swapped halves (asm)", "the same with neg_lo / neg_hi on src2 (asm)",
                                  "hipcc's exact sequence: pk_fma, two scalar v_fma, dependent pk_fma (src1 swapped, src2 negated) (asm)", "scalar v_fma_f32 (control)"};

// out[i] = packed-chain result (2 floats), chk[i] = number of chain steps at which the lane's packed result differed from its scalar twin
template <int FORM>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, float* __restrict__ out, int* __restrict__ chk, int n, int steps) {
    extern __shared__ char pad[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (threadIdx.x == 0 && steps < 0) pad[0] = 1;      // the LDS allocation is what limits the workgroups per CU
    f32x2 x = {in[2 * i], in[2 * i + 1]}, y = {in[(2 * i + 7) % (2 * n)], in[(2 * i + 13) % (2 * n)]};
    f32x2 p = x;            // packed chain
    float s0 = x[0], s1 = x[1];   // scalar twin
    int bad = 0;
    for (int k = 0; k < steps; ++k) {
        float m0 = 0.75f + 0.001f * (float)(k & 7);
        asm volatile("" : "+v"(m0));           // the two multipliers are formed apart: the SLP vectoriser must not put packed forms into the control
        float m1 = 1.25f - 0.002f * (float)(k & 3);
        asm volatile("" : "+v"(m1));
        const f32x2 m = {m0, m1};
        if constexpr (FORM == V_PKFMA) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(m), "v"(y));
        } else if constexpr (FORM == V_PKMULADD) {
            asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p) : "v"(m), "v"(y));
        } else if constexpr (FORM == V_PKVEC) {
            p = p * m + y;          // contracts to v_pk_fma_f32 (fp-contract=fast is hipcc's default)
            p = f32x2{p[1], p[0]} * m - y;      // a swizzled form (op_sel)
        } else if constexpr (FORM == V_PKFMA_SGPR) {
            // the multiplier pair from SGPRs (wave-uniform: a function of the loop counter), as hipcc emits for uniform operands
            const float u0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m0)));
            const float u1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m1)));
            const f32x2 ms = {u0, u1};
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "s"(ms), "v"(y));
        } else if constexpr (FORM == V_PKFMA_OPSEL) {
            // lo = p.hi * m.lo + y.lo, hi = p.lo * m.hi + y.hi
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(p) : "v"(m), "v"(y));
        } else if constexpr (FORM == V_PKFMA_NEG) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(p) : "v"(m), "v"(y));
        } else if constexpr (FORM == V_PAIR_PLAIN) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(m), "v"(y));
        } else if constexpr (FORM == V_PAIR_SWZ) {
            // second: lo = p.hi * m.lo + y.lo, hi = p.lo * m.hi + y.hi -- it needs BOTH halves of the first one's result for each of its halves
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(p) : "v"(m), "v"(y));
        } else if constexpr (FORM == V_SRC1_SWZ) {
            // second: lo = m.lo * p.hi + y.lo, hi = m.hi * p.lo + y.hi  (the first one's result enters as src1)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(p) : "v"(m), "v"(y));
        } else if constexpr (FORM == V_SRC1_SWZ_NEG) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(p) : "v"(m), "v"(y));
        } else if constexpr (FORM == V_EXACT) {
            float d0 = s0, d1 = s1;
            asm volatile("v_pk_fma_f32 %0, %0, %3, %4\n v_fma_f32 %1, %1, %5, %6\n v_fma_f32 %2, %2, %7, %8\n"
                         "v_pk_fma_f32 %0, %3, %0, %4 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
                         : "+v"(p), "+v"(d0), "+v"(d1) : "v"(m), "v"(y), "v"(m0), "v"(y[0]), "v"(m1), "v"(y[1]));
        } else if constexpr (FORM == V_PAIR_NOP) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 4\n v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(p) : "v"(m), "v"(y));
        } else {
            float q0 = __builtin_fmaf(p[0], m0, y[0]);
            asm volatile("" : "+v"(q0));       // (kept apart: the SLP vectoriser would pair the two into a v_pk_fma_f32)
            float q1 = __builtin_fmaf(p[1], m1, y[1]);
            asm volatile("" : "+v"(q1));
            p = f32x2{q0, q1};
        }
        if constexpr (FORM == V_PKMULADD) {
            float t0 = s0 * m0, t1 = s1 * m1;
            asm volatile("" : "+v"(t0), "+v"(t1));        // two roundings, as the packed pair: no contraction into an fma
            s0 = t0 + y[0]; s1 = t1 + y[1];
        } else if constexpr (FORM == V_PKFMA_OPSEL) {
            const float t0 = __builtin_fmaf(s1, m0, y[0]), t1 = __builtin_fmaf(s0, m1, y[1]);
            s0 = t0; s1 = t1;
        } else if constexpr (FORM == V_PKFMA_NEG) {
            s0 = __builtin_fmaf(s0, m0, -y[0]); s1 = __builtin_fmaf(s1, m1, -y[1]);
        } else if constexpr (FORM == V_PAIR_PLAIN) {
            s0 = __builtin_fmaf(s0, m0, y[0]); s1 = __builtin_fmaf(s1, m1, y[1]);
            asm volatile("" : "+v"(s0), "+v"(s1));
            s0 = __builtin_fmaf(s0, m0, y[0]); s1 = __builtin_fmaf(s1, m1, y[1]);
        } else if constexpr (FORM == V_PAIR_SWZ || FORM == V_PAIR_NOP || FORM == V_SRC1_SWZ) {
            s0 = __builtin_fmaf(s0, m0, y[0]); s1 = __builtin_fmaf(s1, m1, y[1]);
            asm volatile("" : "+v"(s0), "+v"(s1));
            const float t0 = __builtin_fmaf(s1, m0, y[0]), t1 = __builtin_fmaf(s0, m1, y[1]);
            s0 = t0; s1 = t1;
        } else if constexpr (FORM == V_SRC1_SWZ_NEG || FORM == V_EXACT) {
            s0 = __builtin_fmaf(s0, m0, y[0]); s1 = __builtin_fmaf(s1, m1, y[1]);
            asm volatile("" : "+v"(s0), "+v"(s1));
            const float t0 = __builtin_fmaf(s1, m0, -y[0]), t1 = __builtin_fmaf(s0, m1, -y[1]);
            s0 = t0; s1 = t1;
        } else if constexpr (FORM == V_PKVEC) {
            s0 = __builtin_fmaf(s0, m[0], y[0]); s1 = __builtin_fmaf(s1, m[1], y[1]);
            const float t0 = __builtin_fmaf(s1, m[0], -y[0]), t1 = __builtin_fmaf(s0, m[1], -y[1]);
            s0 = t0; s1 = t1;
        } else {
            s0 = __builtin_fmaf(s0, m[0], y[0]); s1 = __builtin_fmaf(s1, m[1], y[1]);
        }
        asm volatile("" : "+v"(s0), "+v"(s1));     // keep the twin scalar and un-merged with the packed chain
        bad += (__float_as_uint(p[0]) != __float_as_uint(s0)) | (__float_as_uint(p[1]) != __float_as_uint(s1));
        // keep values bounded: fold back into [0.5, 2)
        if ((k & 15) == 15) {
            p[0] = __uint_as_float((__float_as_uint(p[0]) & 0x007fffffu) | 0x3f800000u); p[1] = __uint_as_float((__float_as_uint(p[1]) & 0x007fffffu) | 0x3f800000u);
            s0 = p[0]; s1 = p[1];
        }
    }
    out[2 * i] = p[0]; out[2 * i + 1] = p[1];
    chk[i] = bad;
}

template <int FORM> static void launch_victim(int grid, size_t lds, hipStream_t s, const float* in, float* out, int* chk, int n, int steps) {
    static bool once = (hipFuncSetAttribute((const void*)victim<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    hipLaunchKernelGGL(victim<FORM>, dim3(grid), dim3(256), lds, s, in, out, chk, n, steps);
}
static void run_victim(int form, int grid, size_t lds, hipStream_t s, const float* in, float* out, int* chk, int n, int steps) {
    switch (form) {
        case V_PKFMA: launch_victim<V_PKFMA>(grid, lds, s, in, out, chk, n, steps); break;
        case V_PKMULADD: launch_victim<V_PKMULADD>(grid, lds, s, in, out, chk, n, steps); break;
        case V_PKVEC: launch_victim<V_PKVEC>(grid, lds, s, in, out, chk, n, steps); break;
        case V_PKFMA_SGPR: launch_victim<V_PKFMA_SGPR>(grid, lds, s, in, out, chk, n, steps); break;
        case V_PKFMA_OPSEL: launch_victim<V_PKFMA_OPSEL>(grid, lds, s, in, out, chk, n, steps); break;
        case V_PKFMA_NEG: launch_victim<V_PKFMA_NEG>(grid, lds, s, in, out, chk, n, steps); break;
        case V_PAIR_PLAIN: launch_victim<V_PAIR_PLAIN>(grid, lds, s, in, out, chk, n, steps); break;
        case V_PAIR_SWZ: launch_victim<V_PAIR_SWZ>(grid, lds, s, in, out, chk, n, steps); break;
        case V_PAIR_NOP: launch_victim<V_PAIR_NOP>(grid, lds, s, in, out, chk, n, steps); break;
        case V_SRC1_SWZ: launch_victim<V_SRC1_SWZ>(grid, lds, s, in, out, chk, n, steps); break;
        case V_SRC1_SWZ_NEG: launch_victim<V_SRC1_SWZ_NEG>(grid, lds, s, in, out, chk, n, steps); break;
        case V_EXACT: launch_victim<V_EXACT>(grid, lds, s, in, out, chk, n, steps); break;
        default: launch_victim<V_SCALAR>(grid, lds, s, in, out, chk, n, steps); break;
    }
}
static void run_aggressor(int mode, int grid, hipStream_t s, const _Float16* src, float* dst, int iters) {
    switch (mode) {
        case AG_F16_V: hipLaunchKernelGGL(aggressor<AG_F16_V>, dim3(grid), dim3(256), 0, s, src, dst, iters); break;
        case AG_F16_A: hipLaunchKernelGGL(aggressor<AG_F16_A>, dim3(grid), dim3(256), 0, s, src, dst, iters); break;
        case AG_BF16_V: hipLaunchKernelGGL(aggressor<AG_BF16_V>, dim3(grid), dim3(256), 0, s, src, dst, iters); break;
        case AG_F32_V: hipLaunchKernelGGL(aggressor<AG_F32_V>, dim3(grid), dim3(256), 0, s, src, dst, iters); break;
        case AG_F16_V_DEP: hipLaunchKernelGGL(aggressor<AG_F16_V_DEP>, dim3(grid), dim3(256), 0, s, src, dst, iters); break;
        default: break;
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 12;
    const int n = 1 << 20, steps = 256;
    std::vector<float> h(2 * n); std::vector<_Float16> hs(4096 * 8);
    for (int i = 0; i < 2 * n; ++i) h[i] = 0.5f + (float)((i * 2654435761u) % 4093) / 4093.f;
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = (_Float16)(0.01f * (float)((i * 7) % 13));
    float *in, *out, *adst; int* chk; _Float16* asrc;
    CK(hipMalloc(&in, 2 * n * 4)); CK(hipMalloc(&out, 2 * n * 4)); CK(hipMalloc(&chk, n * 4)); CK(hipMalloc(&asrc, hs.size() * 2)); CK(hipMalloc(&adst, (size_t)4096 * 256 * 4));
    CK(hipMemcpy(in, h.data(), 2 * n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(asrc, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs; victim: %d lanes x %d chain steps per launch, %d launches per cell (%d rounds x 8); cell = launches whose output differs from the quiet run / lanes whose packed chain left its scalar twin\n",
           prop.name, n_cu, n, steps, rounds * 8, rounds);
    std::vector<float> ref(2 * n), o(2 * n); std::vector<int> c(n);
    const int occs[3] = {1, 2, 4};
    const int ag_waves[2] = {2, 4};     // aggressor waves per SIMD
    for (int form = 0; form < V_N; ++form) {
        printf("\nvictim = %s\n", v_name[form]);
        printf("  %-36s %-10s", "aggressor", "agg w/SIMD");
        for (int oi = 0; oi < 3; ++oi) printf("  victim %d w/SIMD", occs[oi]);
        printf("\n");
        for (int ag = 0; ag < AG_N; ++ag)
            for (int awi = 0; awi < (ag == AG_NONE ? 1 : 2); ++awi) {
                printf("  %-36s %-10d", ag_name[ag], ag == AG_NONE ? 0 : ag_waves[awi]);
                for (int oi = 0; oi < 3; ++oi) {
                    // LDS per victim workgroup so that exactly occs[oi] of them fit a CU beside the (LDS-free) aggressor: 160 KB / occ, minus a margin
                    const size_t lds = (size_t)(160 * 1024 / occs[oi]) - 2048;
                    run_victim(form, n / 256, lds, s0, in, out, chk, n, steps);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(ref.data(), out, 2 * n * 4, hipMemcpyDeviceToHost));
                    long bad_launch = 0, bad_lanes = 0, launches = 0;
                    for (int r = 0; r < rounds; ++r) {
                        if (ag != AG_NONE) run_aggressor(ag, n_cu * ag_waves[awi], s1, asrc, adst, 60000);     // a few ms: outlasts the 8 victim launches
                        for (int k = 0; k < 8; ++k) {
                            run_victim(form, n / 256, lds, s0, in, out, chk, n, steps);
                            CK(hipStreamSynchronize(s0));
                            CK(hipMemcpy(o.data(), out, 2 * n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), chk, n * 4, hipMemcpyDeviceToHost));
                            bad_launch += memcmp(o.data(), ref.data(), 2 * n * 4) != 0;
                            for (int i = 0; i < n; ++i) bad_lanes += c[i] != 0;
                            ++launches;
                        }
                        CK(hipDeviceSynchronize());
                    }
                    printf("  %6ld / %-8ld", bad_launch, bad_lanes);
                    fflush(stdout);
                }
                printf("\n");
            }
    }
    return 0;
}
