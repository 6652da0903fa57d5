// Round 6: what does a global_load_dwordx4 cost the vector-memory address path when only some of its lanes are active?
// The wave fronts' last k-block of a block input whose channel count is not a multiple of 32 (Cin = 136: 8 of its 32 channels exist) is a load whose
// lanes kg >= 1 re-read valid bytes for nothing.  Every wave loads the same L1-resident 4 KB (a block-14 input row's worth) NL times per iteration,
// 16 waves per CU; variants: all 64 lanes, exec = lanes 0-15, exec = lanes 0-31, all lanes with the surplus ones aimed at lane 0's address.
// hipcc --offload-arch=gfx950 -O3 ta_mask.hip -o ta_mask
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define N_IT 512
template <int MODE>
__global__ __launch_bounds__(256) void k(const char* src, float* out) {
    const int lane = threadIdx.x & 63;
    // fragment addressing of the wave fronts: lane = (p = lane & 15, kg = lane >> 4): pixel p at 272-byte stride, 16 bytes at 16 kg
    int voff = (lane & 15) * 272 + (lane >> 4) * 16;
    if (MODE == 3 && lane >= 16) voff = 0;
    if (MODE == 4) voff = lane * 16;                                   // fully contiguous 1 KB
    if (MODE == 5) voff = (lane & 15) * 32 + (lane >> 4) * 1024;       // pixel-fastest, 32-byte pixel stride (the chunked D layout), k-groups 1 KB apart
    if (MODE == 6) voff = (lane & 15) * 64 + (lane >> 4) * 16;         // pixel-fastest, 64-byte pixels (Cin = 32)
    if (MODE == 7) voff = (lane >> 2) * 272 + (lane & 3) * 16;         // k-group-fastest: a quad of lanes = 64 contiguous bytes of one pixel
    if (MODE == 8) voff = (lane & 15) * 16 + (lane >> 4) * 256;        // channel-group-major layout: 16 pixels x 16 bytes contiguous per k-group (= contiguous 1 KB)
    if (MODE == 9) voff = (lane & 15) * 16 + (lane >> 4) * 4352;       // the same with the k-groups a row apart
    if (MODE == 10) voff = (lane & 15) * 192 + (lane >> 4) * 16;       // Cin = 96
    if (MODE == 11) voff = (lane & 15) * 128 + (lane >> 4) * 16;       // 128-byte pixels
    const unsigned long long mask = MODE == 1 ? 0xffffull : MODE == 2 ? 0xffffffffull : ~0ull;
    f32x4 a = {0, 0, 0, 0}, b = a, c = a, d = a;
    for (int it = 0; it < N_IT; ++it) {
        asm volatile("s_mov_b64 exec, %5\n"
                     "global_load_dwordx4 %0, %4, %6\n global_load_dwordx4 %1, %4, %6 offset:64\n global_load_dwordx4 %2, %4, %6 offset:128\n global_load_dwordx4 %3, %4, %6 offset:192\n"
                     "s_mov_b64 exec, -1\n s_waitcnt vmcnt(0)"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(voff), "s"(mask), "s"(src) : "memory");
    }
    const float s = a[0] + b[1] + c[2] + d[3];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE> static void run(const char* name) {
    char* src; float* out;
    hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20); hipMalloc(&out, 64);
    for (int wpc : {4, 16}) {          // waves per CU
        const int blocks = 256 * wpc / 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 5, loads_per_cu = (double)wpc * N_IT * 4;
        printf("%-44s %2d waves/CU  %8.1f us   %6.1f ns per load instruction and CU  (= %5.1f cycles at 2.4 GHz)\n", name, wpc, us, us * 1e3 / loads_per_cu, us * 1e3 / loads_per_cu * 2.4);
    }
}
int main() {
    run<0>("all 64 lanes");
    run<1>("exec = lanes 0-15");
    run<2>("exec = lanes 0-31");
    run<3>("64 lanes, lanes 16-63 at lane 0's address");
    run<4>("lane L at 16 L (contiguous 1 KB)");
    run<5>("pixel-fastest, 32-byte pixels (chunked D)");
    run<6>("pixel-fastest, 64-byte pixels (Cin 32)");
    run<7>("k-group-fastest: quad = 64 B of one pixel");
    run<8>("[k-group][pixel][8 ch]: contiguous 1 KB");
    run<9>("[k-group][pixel][8 ch], k-groups 4352 B apart");
    run<10>("pixel-fastest, 192-byte pixels (Cin 96)");
    run<11>("pixel-fastest, 128-byte pixels");
    return 0;
}
