// Do waves that issue 16-bit MFMAs (v_mfma_f32_16x16x32_f16, accumulator in VGPRs, 4 waves per SIMD like the wave-autonomous fronts) disturb the results of
// ordinary VALU kernels that share their SIMDs?  Victims: (1) IEEE divisions / sqrt on loaded values, (2) gather loads through an index table, (3) readlane
// broadcasts, (4) 64-bit integer keys.  build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mv mfma_victim.hip && /tmp/mv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 4) void aggressor(const _Float16* __restrict__ src, float* __restrict__ dst, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a = *(const f16x8*)(src + ((blockIdx.x * 256 + threadIdx.x) % 4096) * 8), b = *(const f16x8*)(src + lane * 8);
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int k = 0; k < iters; ++k) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
        c0[0] = c0[0] * 0.999f + c1[1]; c2[2] = c2[2] * 0.999f - c3[3];     // dependent VALU on the results, as BN + SiLU are
    }
    dst[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void victim(const float* __restrict__ in, const int* __restrict__ idx, float* __restrict__ out, unsigned long long* __restrict__ keys, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j0 = idx[i], j1 = idx[(i + 1) % n], j2 = idx[(i + 2) % n];
    const float ax = in[j0 * 3], ay = in[j0 * 3 + 1], az = in[j0 * 3 + 2];          // gathers like the rasteriser's projected vertices
    const float bx = in[j1 * 3], by = in[j1 * 3 + 1], bz = in[j1 * 3 + 2];
    const float cx = in[j2 * 3], cy = in[j2 * 3 + 1], cz = in[j2 * 3 + 2];
    const float area = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
    const float inv = 1.f / area;
    const float w0 = ((cx - bx) * (ay - by) - (cy - by) * (ax - bx)) * inv, w1 = ((ax - cx) * (by - cy) - (ay - cy) * (bx - cx)) * inv, w2 = 1.f - w0 - w1;
    const float iz = (w0 / az + w1 / bz) + w2 / cz;
    const float z = 1.f / iz;
    const float nn = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    const float bc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), (blockIdx.x * 7) & 63));
    out[i] = z + (nn > 0.f ? w0 / nn : 0.f) + bc;
    keys[i] = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned int)(j0 * 2654435761u);
}
int main() {
    const int n = 1 << 20, nv = 1 << 16;
    std::vector<float> h(nv * 3); std::vector<int> hi(n); std::vector<_Float16> hs(4096 * 8);
    for (int i = 0; i < nv * 3; ++i) h[i] = 0.5f + (float)((i * 2654435761u) % 4093) / 4093.f + (float)(i % 5);
    for (int i = 0; i < n; ++i) hi[i] = (int)((i * 40503u + 17u) % nv);
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = (_Float16)(0.01f * (float)((i * 7) % 13));
    float *in, *out, *ref, *adst; int* idx; unsigned long long *keys, *kref; _Float16* asrc;
    hipMalloc(&in, nv * 12); hipMalloc(&idx, n * 4); hipMalloc(&out, n * 4); hipMalloc(&ref, n * 4); hipMalloc(&keys, n * 8); hipMalloc(&kref, n * 8);
    hipMalloc(&asrc, hs.size() * 2); hipMalloc(&adst, (size_t)8192 * 256 * 4);
    hipMemcpy(in, h.data(), nv * 12, hipMemcpyHostToDevice); hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(asrc, hs.data(), hs.size() * 2, hipMemcpyHostToDevice);
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    victim<<<n / 256, 256, 0, s0>>>(in, idx, ref, kref, n);
    hipDeviceSynchronize();
    std::vector<float> r(n), o(n); std::vector<unsigned long long> kr(n), ko(n);
    hipMemcpy(r.data(), ref, n * 4, hipMemcpyDeviceToHost); hipMemcpy(kr.data(), kref, n * 8, hipMemcpyDeviceToHost);
    long bad[2] = {0, 0}, badk[2] = {0, 0};
    for (int round = 0; round < 200; ++round) {
        const int load = round & 1;
        if (load) aggressor<<<8192, 256, 0, s1>>>(asrc, adst, 400);
        for (int k = 0; k < 8; ++k) victim<<<n / 256, 256, 0, s0>>>(in, idx, out, keys, n);
        hipDeviceSynchronize();
        hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ko.data(), keys, n * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) { bad[load] += memcmp(&o[i], &r[i], 4) != 0; badk[load] += ko[i] != kr[i]; }
    }
    printf("victim values differing from the quiet reference (float outputs / 64-bit keys), 100 x %d each: quiet rounds %ld / %ld, beside the MFMA aggressor %ld / %ld\n", n, bad[0], badk[0], bad[1], badk[1]);
    return 0;
}
