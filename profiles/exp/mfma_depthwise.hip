// Feasibility micro-benchmark for DESIGN section 9's "depthwise taps on the matrix pipe": a 5x5 depthwise convolution over a 16x16 map x 16 channels per
// wavefront (the shape of blocks 14-17's jobs), from expanded rows held in registers in the MFMA C layout of the wave kernel (lane (p = pixel, kg): 4 channels),
//   V : as mbconv_wave_kernel does it -- DPP halo moves, 25 x 4 fp32 FMAs per row, taps read from LDS, input-stationary over 5 open rows;
//   M : v_mfma_f32_4x4x4_16B_f16 -- 16 blocks = 16 channels, A = per-channel Toeplitz matrices of one tap row (4 output pixels x 4 input pixels), B = 4 input pixels x
//       4 rows of the expanded map (f16, staged through a wave-private LDS tile [channel][row][x], written 2 bytes at a time from the C layout), fp32 accumulation;
//       the outputs leave in the instruction's layout (lane = (channel, row), 4 pixels) and are transposed back through LDS to the (pixel, channel quad) layout a
//       16-byte D store needs.
// Both include their conversions and LDS traffic; both are verified against each other (f16-rounded inputs and taps, fp32 accumulation: equal up to summation order).
// build: hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops -o profiles/exp/mfma_depthwise profiles/exp/mfma_depthwise.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CTRL> __device__ __forceinline__ float dpp0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// synthetic expanded value of (sample-ish seed s, row y, pixel x, channel c), exactly representable in f16
__device__ __host__ inline float e_val(int s, int y, int x, int c) { return (float)(((s * 7 + y * 13 + x * 5 + c * 3) & 31) - 16) * 0.0625f; }
__device__ __host__ inline float w_val(int ky, int kx, int c) { return (float)(((ky * 5 + kx) * 3 + c) % 11 - 5) * 0.125f; }

// ---- layout probe of v_mfma_f32_4x4x4_16B_f16: D = A x B per block with unknown lane mapping; the host tries the candidate mappings
__global__ void probe_kernel(const _Float16* a, const _Float16* b, float* d) {
    const int lane = threadIdx.x;
    f16x4 av = *(const f16x4*)(a + lane * 4), bv = *(const f16x4*)(b + lane * 4);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, c, 0, 0, 0);
    *(f32x4*)(d + lane * 4) = c;
}

// ---- V: the wave kernel's tap phase
__global__ __launch_bounds__(256, 4) void dw_valu_kernel(float* out, int reps) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, kg = lane >> 4;
    float* taps = smem + wave * 25 * 16;
    for (int i = lane; i < 25 * 16; i += 64) taps[i] = w_val(i / 16 / 5, (i / 16) % 5, i % 16);
    const float* tl = taps + kg * 4;
    const int seed = blockIdx.x * 4 + wave;
    float acc[5][4], total[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[s][c] = 0.f;
        for (int base = 0; base < 20; base += 5) {
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int iy = base + u;
                asm volatile("" ::: "memory");
                float E[5][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = e_val(seed + r, iy & 15, p, kg * 4 + c) * (iy < 16 ? 1.f : 0.f);
                    asm volatile("" : "+v"(v) :: "memory");     // the row's values appear here (in the real kernel: out of the expand MFMAs), not hoisted over earlier rows
                    E[2][c] = v; E[1][c] = dpp0<0x111>(v); E[0][c] = dpp0<0x112>(v); E[3][c] = dpp0<0x101>(v); E[4][c] = dpp0<0x102>(v);
                }
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    const int os = (u + 2 - ky + 5) % 5;
                    asm volatile("" ::: "memory");                 // keep the tap reads of one tap row together (no hoisting of all 25 quads)
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx) {
                        const f32x4 w = *(const f32x4*)(tl + (ky * 5 + kx) * 16);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[os][c] += w[c] * E[kx][c];
                    }
                }
                const int os = (u + 3) % 5, oy = iy - 2;
#pragma unroll
                for (int c = 0; c < 4; ++c) { total[c] += acc[os][c] * ((oy >= 0 && oy < 16) ? (float)(1 + ((oy * 16 + p) & 3)) : 0.f); acc[os][c] = 0.f; }
            }
        }
    }
    *(f32x4*)(out + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane) * 4) = f32x4{total[0], total[1], total[2], total[3]};
}

// ---- M: taps on the matrix pipe
enum { TP = 20, RING = 8, PLANE = RING * TP + 4 };      // f16 ring [16 channels][8 rows][20 x (2 + 16 + 2)] + a skew per channel plane; row y lives in slot (y + 2) % 8
__global__ __launch_bounds__(256, 4) void dw_mfma_kernel(float* out, int reps) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, kg = lane >> 4;
    _Float16* T = (_Float16*)smem + (size_t)wave * (16 * PLANE + 4 * 16 * 16);
    _Float16* T2 = T + 16 * PLANE;                                  // outputs of one 4-row block [row 4][x 16][ch 16]
    for (int i = lane; i < 16 * PLANE; i += 64) T[i] = (_Float16)0.f;   // the x halo columns stay zero
    const int seed = blockIdx.x * 4 + wave;
    const int cb = lane >> 2, ij = lane & 3;                        // MFMA lane roles: block (channel) cb, row / column index ij
    f16x4 A[5][2];                                                  // Toeplitz blocks of tap row ky: A[i][k] = w[ky][4 m + k - i]
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int kx = 4 * m + k - ij; A[ky][m][k] = (_Float16)((kx >= 0 && kx < 5) ? w_val(ky, kx, cb) : 0.f); }
    float total[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        asm volatile("" ::: "memory");
        // expanded row y: C layout (lane = pixel p, 4 channels) -> ring slot (y + 2) % 8, 2 bytes at a time; rows outside the map are zero rows
        auto put_row = [&](int y) {
            const int slot = (y + 2) & 7;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                T[(kg * 4 + c) * PLANE + slot * TP + p + 2] = (_Float16)((y >= 0 && y < 16) ? e_val(seed + r, y, p, kg * 4 + c) : 0.f);
        };
        for (int y = -2; y < 2; ++y) put_row(y);
        for (int yb = 0; yb < 4; ++yb) {
            for (int y = 4 * yb + 2; y < 4 * yb + 6; ++y) put_row(y);          // the ring now holds rows 4 yb - 2 .. 4 yb + 5
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const _Float16* rowp = T + cb * PLANE + ((4 * yb + ij + ky) & 7) * TP;     // input row 4 yb + j + ky - 2
#pragma unroll
                for (int gm = 0; gm < 5; ++gm) {                                           // pixels 4 gm - 2 .. 4 gm + 1
                    const f16x4 bv = *(const f16x4*)(rowp + 4 * gm);
                    if (gm < 4) acc[gm] = __builtin_amdgcn_mfma_f32_4x4x4f16(A[ky][0], bv, acc[gm], 0, 0, 0);
                    if (gm > 0) acc[gm - 1] = __builtin_amdgcn_mfma_f32_4x4x4f16(A[ky][1], bv, acc[gm - 1], 0, 0, 0);
                }
            }
            // outputs: lane (channel cb, row j = ij) holds pixel i of group g in acc[g][i] -> T2[j][x][ch], then back in the (pixel, channel quad) layout
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) T2[(ij * 16 + 4 * g + i) * 16 + cb] = (_Float16)acc[g][i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 4 * yb + j;
                const f16x4 o = *(const f16x4*)(T2 + (j * 16 + p) * 16 + kg * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) total[c] += (float)o[c] * (float)(1 + ((y * 16 + p) & 3));
            }
        }
    }
    *(f32x4*)(out + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane) * 4) = f32x4{total[0], total[1], total[2], total[3]};
}

int main() {
    // ---- probe
    std::vector<_Float16> ha(256), hb(256); std::vector<float> hd(256);
    for (int i = 0; i < 256; ++i) { ha[i] = (_Float16)((i * 7 % 13) - 6); hb[i] = (_Float16)((i * 5 % 11) - 5); }
    _Float16 *da, *db; float* dd;
    CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost));
    // hypothesis: A[blk][i][k] = a[4 blk + i][k], B[blk][k][j] = b[4 blk + j][k]; D[blk][i][j] at (lane 4 blk + j, reg i) [H1] or (lane 4 blk + i, reg j) [H2]
    double e1 = 0, e2 = 0;
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0;
                for (int k = 0; k < 4; ++k) s += (float)ha[(4 * blk + i) * 4 + k] * (float)hb[(4 * blk + j) * 4 + k];
                e1 = fmax(e1, fabs(s - hd[(4 * blk + j) * 4 + i])); e2 = fmax(e2, fabs(s - hd[(4 * blk + i) * 4 + j]));
            }
    printf("layout probe v_mfma_f32_4x4x4_16B_f16: max |err| under H1 (D[i][j] in lane 4b+j, reg i) %.3g, under H2 (lane 4b+i, reg j) %.3g\n", e1, e2);

    // ---- the two depthwise versions: grids that fill the chip at 4 waves per SIMD, several rounds
    const int blocks = 256 * 4 * 4, reps = 8;
    float *o1, *o2;
    const size_t n = (size_t)blocks * 4 * 64 * 4;
    CK(hipMalloc(&o1, n * 4)); CK(hipMalloc(&o2, n * 4));
    const size_t ldsV = 4 * 25 * 16 * 4, ldsM = 4 * (16 * PLANE + 4 * 16 * 16) * 2;
    CK(hipFuncSetAttribute((const void*)dw_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1_;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1_));
    float msV = 0, msM = 0;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(dw_valu_kernel, dim3(blocks), dim3(256), ldsV, 0, o1, reps); CK(hipEventRecord(e1_)); CK(hipEventSynchronize(e1_));
        CK(hipEventElapsedTime(&msV, e0, e1_));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(dw_mfma_kernel, dim3(blocks), dim3(256), ldsM, 0, o2, reps); CK(hipEventRecord(e1_)); CK(hipEventSynchronize(e1_));
        CK(hipEventElapsedTime(&msM, e0, e1_));
    }
    CK(hipGetLastError());
    std::vector<float> h1(n), h2(n);
    CK(hipMemcpy(h1.data(), o1, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, n * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (size_t i = 0; i < n; ++i) { worst = fmax(worst, fabs((double)h1[i] - h2[i])); scale = fmax(scale, fabs((double)h1[i])); }
    const double tiles = (double)blocks * 4 * reps;
    printf("5x5 depthwise over 16x16x16 tiles, %d workgroups x 4 waves x %d tiles (LDS per workgroup: V %zu B, M %zu B):\n", blocks, reps, ldsV, ldsM);
    printf("  V (DPP + 100 FMAs per row, taps from LDS)      %8.3f ms = %6.1f ns per tile\n", msV, msV * 1e6 / tiles);
    printf("  M (v_mfma_f32_4x4x4_16B_f16, LDS transposes)   %8.3f ms = %6.1f ns per tile   ratio V / M = %.2f\n", msM, msM * 1e6 / tiles, msV / msM);
    printf("  checksums: max |V - M| = %.4g at scale %.4g (M rounds its outputs to f16 before the checksum)\n", worst, scale);
    return 0;
}
