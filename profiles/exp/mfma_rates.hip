// Issue / latency cost of the MFMA shapes the fused fronts use, per wave instruction in shader-clock cycles (s_memtime deltas, one workgroup per CU slot):
//   v_mfma_f32_16x16x32_f16 (expansion), v_mfma_f32_4x4x4_16B_f16 (depthwise taps), v_mfma_f32_16x16x16_f16 (transposition),
// as a DEPENDENT chain on one accumulator and as 4 independent accumulators, with 1, 2 and 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/exp/mfma_rates profiles/exp/mfma_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND, int ILP> __global__ void rate_kernel(unsigned long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a8, b8; f16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.001f * (lane + i)); b8[i] = (_Float16)(0.002f * (lane - i)); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    f32x4 acc[ILP];
    for (int j = 0; j < ILP; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / ILP; ++u)
#pragma unroll
            for (int j = 0; j < ILP; ++j) {
                if constexpr (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[j], 0, 0, 0);
                else if constexpr (KIND == 1) acc[j] = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[j], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < ILP; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND, int ILP> static double run(int waves_per_simd) {
    const int iters = 2000, threads = 64 * 4 * waves_per_simd, blocks = 256;      // one workgroup per CU, waves spread over the 4 SIMDs
    unsigned long long* d; float* sink;
    CK(hipMalloc(&d, blocks * 64 * 8)); CK(hipMalloc(&sink, (size_t)blocks * threads * 4));
    hipLaunchKernelGGL((rate_kernel<KIND, ILP>), dim3(blocks), dim3(threads), 0, 0, d, sink, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * (threads / 64));
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto v : h) m += (double)v; m /= h.size();
    CK(hipFree(d)); CK(hipFree(sink));
    // s_memtime counts at a fixed 100 MHz on this part: scale to shader cycles at 2.4 GHz
    return m / (iters * 16.0);
}
__global__ void spin_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t - t0;
}
int main() {
    {   // calibration: s_memtime ticks against the HIP event clock
        unsigned long long* d; CK(hipMalloc(&d, 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, d, 1000ull);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, d, 100000000ull); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h = 0; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        printf("s_memtime: %llu ticks in %.3f ms = %.1f MHz\n", h, ms, h / (ms * 1e3));
    }
    const char* names[3] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_4x4x4_16B_f16", "v_mfma_f32_16x16x16_f16"};
    printf("s_memtime ticks per wave instruction (multiply by the shader-clock / s_memtime ratio; the RATIOS between rows are what matters)\n");
    printf("%-28s %-12s %10s %10s %10s\n", "instruction", "chain", "1 wave/SIMD", "2", "4");
    double r[3][2][3];
    for (int w = 0; w < 3; ++w) {
        const int wps = 1 << w;
        r[0][0][w] = run<0, 1>(wps); r[0][1][w] = run<0, 4>(wps);
        r[1][0][w] = run<1, 1>(wps); r[1][1][w] = run<1, 4>(wps);
        r[2][0][w] = run<2, 1>(wps); r[2][1][w] = run<2, 4>(wps);
    }
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 2; ++c) printf("%-28s %-12s %10.3f %10.3f %10.3f\n", names[k], c ? "4 independent" : "dependent", r[k][c][0], r[k][c][1], r[k][c][2]);
    return 0;
}
