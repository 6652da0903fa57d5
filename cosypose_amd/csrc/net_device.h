// Device-side helpers shared by the backbone kernel files (gfx950, wave64).
#pragma once
#include "kernels_net.h"

namespace cosy {

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int EPL = 4, KB = 16; typedef f32x4 raw_t; };
template <> struct DT<bf16_t> { static constexpr int EPL = 8, KB = 32; typedef bf16x8 raw_t; };
template <> struct DT<f16_t> { static constexpr int EPL = 8, KB = 32; typedef f16x8 raw_t; };

__device__ __forceinline__ void mma(f32x4& c, bf16x8 a, bf16x8 b) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void mma(f32x4& c, f16x8 a, f16x8 b) { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void mma(f32x4& c, f32x4 a, f32x4 b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ float sigmoid_t(float x);
template <> __device__ __forceinline__ float sigmoid_t<float>(float x) { return 1.f / (1.f + expf(-x)); }
template <> __device__ __forceinline__ float sigmoid_t<bf16_t>(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
template <> __device__ __forceinline__ float sigmoid_t<f16_t>(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ void to_f32(const f32x4& r, float* v) { v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3]; }
__device__ __forceinline__ void to_f32(const bf16x8& r, float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
__device__ __forceinline__ void from_f32(f32x4& r, const float* v) { r = f32x4{v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void from_f32(bf16x8& r, const float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
}

__device__ __forceinline__ void to_f32(const f16x8& r, float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
// fp16 saturates instead of overflowing to inf (activations beyond +-65504 would otherwise poison the squeeze sums)
__device__ __forceinline__ f16_t to_f16_sat(float x) { return (f16_t)__builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f); }
__device__ __forceinline__ void from_f32(f16x8& r, const float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = to_f16_sat(v[i]);
}
__device__ __forceinline__ void load8(const f16_t* p, float* v) { to_f32(*(const f16x8*)p, v); }
__device__ __forceinline__ void store8(f16_t* p, const float* v) { f16x8 r; from_f32(r, v); *(f16x8*)p = r; }
__device__ __forceinline__ void store4(f16_t* p, const float* v) {
    *(f16x4*)p = f16x4{to_f16_sat(v[0]), to_f16_sat(v[1]), to_f16_sat(v[2]), to_f16_sat(v[3])};
}
__device__ __forceinline__ void load4(const f16_t* p, float* v) {
    f16x4 a = *(const f16x4*)p; v[0] = (float)a[0]; v[1] = (float)a[1]; v[2] = (float)a[2]; v[3] = (float)a[3];
}
// load / store 8 consecutive channels as fp32
__device__ __forceinline__ void load8(const float* p, float* v) {
    f32x4 a = ((const f32x4*)p)[0], b = ((const f32x4*)p)[1];
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16_t* p, float* v) { to_f32(*(const bf16x8*)p, v); }
__device__ __forceinline__ void store8(float* p, const float* v) {
    ((f32x4*)p)[0] = f32x4{v[0], v[1], v[2], v[3]};
    ((f32x4*)p)[1] = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) { bf16x8 r; from_f32(r, v); *(bf16x8*)p = r; }
__device__ __forceinline__ void store4(float* p, const float* v) { *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void store4(bf16_t* p, const float* v) {
    *(bf16x4*)p = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}
__device__ __forceinline__ void load4(const float* p, float* v) { f32x4 a = *(const f32x4*)p; v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; }
__device__ __forceinline__ void load4(const bf16_t* p, float* v) {
    bf16x4 a = *(const bf16x4*)p; v[0] = (float)a[0]; v[1] = (float)a[1]; v[2] = (float)a[2]; v[3] = (float)a[3];
}

// lane i <- lane i-N of its 16-lane DPP row; lanes without a source read 0 (bound_ctrl:0)
template <int N> __device__ __forceinline__ float dpp_row_shr0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
}

// Fixed-order (deterministic) reduction of per-thread squeeze sums parked in LDS as red[thread][CPT]: output (g, e) =
// sum over the threads t = g, g + NG, ... < stride of red[t][e].  Spread over up to 4 threads per output (each sums
// every 4th contribution, two xor-shuffles combine them) instead of one thread walking all of them: the serial walk
// (32 dependent LDS reads) sat on the critical path of every chunk while the other waves waited at the next barrier.
__device__ __forceinline__ void reduce_squeeze_sums(const float* red, int stride, int NG, int CPT, int tid, int nthr, float* out) {
    const int outputs = NG * CPT;
    if (outputs > nthr) {       // tiny maps (fewer threads than outputs): every thread walks several outputs serially
        for (int o = tid; o < outputs; o += nthr) {
            const int g = o / CPT, e = o - g * CPT;
            float sacc = 0.f;
            for (int t = g; t < stride; t += NG) sacc += red[t * CPT + e];
            out[o] = sacc;
        }
        return;
    }
    int parts = 4;
    while (parts > 1 && outputs * parts > nthr) parts >>= 1;
    const int o = tid / parts, part = tid - o * parts;
    float sacc = 0.f;
    if (o < outputs) {
        const int g = o / CPT, e = o - g * CPT;
        for (int t = g + part * NG; t < stride; t += NG * parts) sacc += red[t * CPT + e];
    }
    if (parts == 4) { sacc += __shfl_xor(sacc, 1); sacc += __shfl_xor(sacc, 2); }
    else if (parts == 2) sacc += __shfl_xor(sacc, 1);
    if (o < outputs && part == 0) out[o] = sacc;
}

__device__ __forceinline__ void lds_ld8(const bf16_t* p, float* v) {
    const uint4 u = *(const uint4*)p;
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void lds_ld8(const float* p, float* v) { load8(p, v); }
__device__ __forceinline__ void lds_ld8(const f16_t* p, float* v) { load8(p, v); }

}  // namespace cosy
