// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction of v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32,
// v_cvt_pk_bf16_f32 and a SiLU chain, at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_bench.hip -o valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_IT 256
#define UNR 16
template <int OP>
__global__ void k(float* out, float seed) {
    float a[UNR];
    f32x2 p[UNR];
    for (int i = 0; i < UNR; ++i) { a[i] = seed + i * 0.001f + threadIdx.x * 1e-6f; p[i] = f32x2{a[i], a[i] + 0.5f}; }
    const float c0 = 0.999f, c1 = 0.0001f;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f32x2{c0, c0}), "v"(f32x2{c1, c1}));
            if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(c0)); a[i] = __uint_as_float(r); }
            if (OP == 5) {   // SiLU chain as the kernels compute it: t = exp2(-x*log2e); y = x * rcp(1 + t)
                float t;
                asm volatile("v_mul_f32 %0, %1, %2\n v_exp_f32 %0, %0\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n v_mul_f32 %1, %1, %0"
                             : "=&v"(t), "+v"(a[i]) : "v"(-1.4426950f));
            }
            if (OP == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f32x2{c0, c0}));
            if (OP == 8) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(0x3f803f80u), "v"(0x3c003c00u));   // acc += a.lo*b.lo + a.hi*b.hi (bf16 pairs)
            if (OP == 9) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(0x3c003c00u), "v"(0x20002000u));
            if (OP == 10) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(0x3f803f80u), "v"(0x3c003c00u));
        }
    }
    float s = 0;
    for (int i = 0; i < UNR; ++i) s += a[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) out[0] = s;
}
template <int OP>
static void run(const char* name, int per_it) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int dev; hipGetDevice(&dev); hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
    const double ghz = pr.clockRate / 1e6;
    for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD; wps blocks per CU
        const int grid = pr.multiProcessorCount * wps;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms / 5 * 1e-3 * ghz * 1e9;                       // cycles of one launch
        const double inst_per_simd = (double)N_IT * UNR * per_it * wps;   // wave instructions issued on one SIMD
        printf("%-22s waves/SIMD %d: %.2f cycles per wave-instruction (clock %.2f GHz nominal)\n", name, wps, cyc / inst_per_simd, ghz);
    }
}
int main() {
    run<0>("v_fma_f32", 1); run<6>("v_mul_f32", 1); run<1>("v_pk_fma_f32", 1); run<7>("v_pk_mul_f32", 1); run<2>("v_exp_f32", 1); run<3>("v_rcp_f32", 1);
    run<4>("v_cvt_pk_bf16_f32", 1); run<5>("silu chain (5 instr)", 5);
    run<8>("v_dot2_f32_bf16", 1); run<9>("v_dot2_f32_f16", 1); run<10>("v_dot2c_f32_bf16", 1);
    return 0;
}
