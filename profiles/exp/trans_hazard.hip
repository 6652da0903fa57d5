// Is a hipcc-scheduled v_rcp / v_sqrt / division sequence reproducible while another stream hammers the transcendental unit?
// build: hipcc --offload-arch=gfx950 -O3 -o trans_hazard trans_hazard.hip ; run: ./trans_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
__global__ void victim(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = in[i], b = in[(i + 7) % n], c = in[(i + 13) % n];
    // the rasteriser's arithmetic shape: divisions, reciprocal, sqrt, short dependency chains
    const float area = (b - a) * (c - a) - (c - b) * (b + a);
    const float inv = 1.f / area;
    const float w0 = (a * b - c) * inv, w1 = (b * c - a) * inv, w2 = (c * a - b) * inv;
    const float iz = (w0 / a + w1 / b) + w2 / c;
    const float z = 1.f / iz;
    const float nn = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    out[i] = z + (nn > 0.f ? w0 / nn : 0.f);
}
__global__ void aggressor(float* __restrict__ buf, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = buf[i];
    for (int k = 0; k < iters; ++k) { v = __expf(-v) + 1.f; v = __frcp_rn(v); v = v * 1.37f + 0.1f; }
    buf[i] = v;
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0.5f + 0.37f * (float)((i * 2654435761u) % 1000) / 1000.f + (float)(i % 7);
    float *in, *out, *ref, *agg;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMalloc(&ref, n * 4); hipMalloc(&agg, (size_t)(1 << 24) * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(agg, 0, (size_t)(1 << 24) * 4);
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    victim<<<n / 256, 256, 0, s0>>>(in, ref, n);
    hipDeviceSynchronize();
    std::vector<float> r(n), o(n);
    hipMemcpy(r.data(), ref, n * 4, hipMemcpyDeviceToHost);
    long bad_quiet = 0, bad_loaded = 0;
    for (int round = 0; round < 200; ++round) {
        const bool load = round & 1;
        if (load) aggressor<<<(1 << 24) / 256, 256, 0, s1>>>(agg, 1 << 24, 64);
        for (int k = 0; k < 8; ++k) victim<<<n / 256, 256, 0, s0>>>(in, out, n);
        hipDeviceSynchronize();
        hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; ++i) bad += memcmp(&o[i], &r[i], 4) != 0;
        (load ? bad_loaded : bad_quiet) += bad;
    }
    printf("values differing from the quiet reference: quiet rounds %ld, rounds with the transcendental aggressor on another stream %ld (of %d x 100 each)\n", bad_quiet, bad_loaded, n);
    return 0;
}
