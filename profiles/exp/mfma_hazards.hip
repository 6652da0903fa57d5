// What hipcc pads between dependent MFMA / VALU / LDS instructions on gfx950 (compile with -save-temps and read the s_nop it inserts): the wait states the
// hand-ordered fused row of kernels_wave.hip (row_px) writes out itself.  hipcc --offload-arch=gfx950 -O3 -c mfma_hazards.hip -save-temps
// Result (ROCm 7.2): 4x4x4 result -> VALU / VMEM read: s_nop 4; -> the next MFMA taking it as SrcC: 2 states; 16x16x16 / 16x16x32 result -> VALU read: s_nop 7,
// -> the next MFMA SrcC: 0; VALU write -> MFMA A / B read: 2 states; v_exp / v_rcp -> VALU read: s_nop 0.
#include <hip/hip_runtime.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// A: 4x4x4 -> dependent 4x4x4 (SrcC) back to back
__global__ void hazA(f16x4* p, f32x4* o) {
  f16x4 a = p[threadIdx.x], b = p[threadIdx.x + 64], a2 = p[threadIdx.x + 128], b2 = p[threadIdx.x + 192];
  f32x4 c = o[threadIdx.x];
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a2, b2, c, 0, 0, 0);
  o[threadIdx.x] = c;
}
// B: 4x4x4 -> VALU read
__global__ void hazB(f16x4* p, f32x4* o, float s) {
  f16x4 a = p[threadIdx.x], b = p[threadIdx.x + 64];
  f32x4 c = o[threadIdx.x];
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  c = c * s + 1.0f;
  o[threadIdx.x] = c;
}
// C: 16x16x16 -> VALU read ; 16x16x32 -> VALU read
__global__ void hazC(f16x4* p, f16x8* q, f32x4* o, float s) {
  f16x4 a = p[threadIdx.x], b = p[threadIdx.x + 64];
  f32x4 c = o[threadIdx.x];
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  c = c * s + 1.0f;
  f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(q[threadIdx.x], q[threadIdx.x + 64], c, 0, 0, 0);
  d = d * s + 2.0f;
  o[threadIdx.x] = d;
}
// D: VALU write -> MFMA A/B ; DPP write -> MFMA; bpermute -> MFMA
__global__ void hazD(f16x4* p, f32x4* o, float s, int sel) {
  f16x4 a = p[threadIdx.x], b = p[threadIdx.x + 64];
  f32x4 c = o[threadIdx.x];
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2 bi = __builtin_bit_cast(i32x2, b);
  bi[0] = __builtin_amdgcn_update_dpp(0, bi[0], 0x90, 0xf, 0xf, false);
  bi[1] = __builtin_amdgcn_ds_bpermute(sel, bi[1]);
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, __builtin_bit_cast(f16x4, bi), c, 0, 0, 0);
  o[threadIdx.x] = c;
}
// E: 16x16x16 result -> cvt -> store ; 4x4x4 result -> 16x16x32 SrcC ; 16x16x32 -> 4x4x4 srcC
__global__ void hazE(f16x4* p, f16x8* q, f32x4* o) {
  f16x4 a = p[threadIdx.x], b = p[threadIdx.x + 64];
  f32x4 c = o[threadIdx.x];
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(q[threadIdx.x], q[threadIdx.x + 64], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  o[threadIdx.x] = c;
}
// F: MFMA result used as A operand of next MFMA (4x4x4 D -> cvt... no: direct f32 can't). MFMA D -> ds_bpermute data? (VALU between). 4x4x4 -> store directly
__global__ void hazF(f16x4* p, f32x4* o) {
  f16x4 a = p[threadIdx.x], b = p[threadIdx.x + 64];
  f32x4 c = o[threadIdx.x];
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  o[threadIdx.x] = c;
}
// G: trans -> VALU consumer (v_exp -> v_add)
__global__ void hazG(float* o) {
  float x = o[threadIdx.x];
  x = __builtin_amdgcn_exp2f(x) + 1.0f;
  x = __builtin_amdgcn_rcpf(x) * 3.0f;
  o[threadIdx.x] = x;
}
