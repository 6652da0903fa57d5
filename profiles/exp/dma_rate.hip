// Round 6: L2 -> CU fill rate of one CU by ordinary vector loads (global_load_dwordx4 -> VGPR) and by LDS-DMA (global_load_lds_dwordx4), contiguous 1-KB
// fragments (the full-rate address pattern of profiles/exp/ta_mask.hip), from a working set that lives in the L2 / MALL (not in the 32-KB L1): every workgroup streams
// its own 32-KB slice again and again.  Question: would a register-staged operand path (load -> ds_write) beat the LDS-DMA ring of the late 1x1-conv GEMMs?
// hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SLICE (32 * 1024)      // per workgroup: 4 workgroups x 32 CUs x 32 KB = 4 MB per XCD at most (its L2), beyond the CU's L1 with the neighbours' slices
#define N_PASS 256
#define BIG (1024 * 1024)        // MODE 5 / 6 / 7: a 1-MB region (beyond the L1, inside the L2) read by every workgroup / one per XCD / 64 KB of its own per workgroup
template <int MODE>
__global__ __launch_bounds__(256) void k(const char* src, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // MODE 3 / 4: LDS-DMA where every workgroup of the chip reads the SAME 32 KB (what the weight operand of a GEMM is) / the same 32 KB per XCD
    const size_t slice = MODE == 3 ? 0 : MODE == 4 ? (blockIdx.x & 7) : (blockIdx.x % 1024);
    const char* base = src + slice * SLICE + wave * (SLICE / 4);
    f32x4 acc = {0, 0, 0, 0};
    if (MODE >= 5) {
        const int span = MODE == 7 ? 64 * 1024 : BIG;                       // bytes this workgroup walks per pass
        const char* b2 = src + (MODE == 5 ? 0 : MODE == 6 ? (size_t)(blockIdx.x & 7) * BIG : (size_t)(blockIdx.x % 512) * span);
        for (int pass = 0; pass < (MODE == 7 ? 64 : 4); ++pass)
            for (int off = wave * 8192; off < span; off += 4 * 8192) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b2 + off + u * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(lds + wave * 8192 + u * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        acc += *(f32x4*)(lds + threadIdx.x * 16);
        if (acc[0] == 12345.678f) out[0] = acc[1];
        return;
    }
    for (int pass = 0; pass < N_PASS; ++pass) {
        for (int off = 0; off < SLICE / 4; off += 8 * 1024) {          // 8 fragments of 1 KB in flight per wave (a wave's quarter of the slice = one batch)
            if (MODE == 0) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(base + off + u * 1024 + lane * 16) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            } else if (MODE == 1) {                                   // register-staged: load -> ds_write (what a register operand path would do)
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(base + off + u * 1024 + lane * 16) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u) *(volatile f32x4*)(lds + wave * 8192 + u * 1024 + lane * 16) = v[u];
            } else {                                                   // LDS-DMA (MODE 2: own slice; 3: one slice for the chip; 4: one per XCD)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + u * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(lds + wave * 8192 + u * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    if (MODE != 0) acc += *(f32x4*)(lds + threadIdx.x * 16);
    if (acc[0] == 12345.678f) out[0] = acc[1];
}
template <int MODE> static void run(const char* name) {
    char* src; float* out;
    (void)hipMalloc(&src, (size_t)1024 * SLICE); (void)hipMemset(src, 0, (size_t)1024 * SLICE); (void)hipMalloc(&out, 64);
    for (int wgs_per_cu : {1, 2, 4}) {
        const int blocks = 256 * wgs_per_cu;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, src, out);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, src, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = MODE >= 5 ? 3.0 * blocks * (MODE == 7 ? 64.0 * 1024 * 64 : (double)BIG * 4) : 3.0 * blocks * (double)SLICE * N_PASS;
        printf("%-40s %d workgroups / CU   %7.1f GB/s per CU   %6.2f TB/s chip\n", name, wgs_per_cu, bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12);
    }
}
int main() {
    run<0>("global_load_dwordx4 -> VGPR");
    run<1>("global_load_dwordx4 -> VGPR -> ds_write");
    run<2>("global_load_lds_dwordx4 (LDS-DMA)");
    run<3>("LDS-DMA, all workgroups the same 32 KB");
    run<4>("LDS-DMA, the same 32 KB per XCD");
    run<5>("LDS-DMA, the same 1 MB for the chip");
    run<6>("LDS-DMA, the same 1 MB per XCD");
    run<7>("LDS-DMA, 64 KB of its own per workgroup");
    return 0;
}
