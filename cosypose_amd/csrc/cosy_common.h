// Shared declarations for libcosyhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/cosyhip.h"

namespace cosy {

void set_error(const char* fmt, ...);

#define COSY_CHECK_HIP(expr)                                                          \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            cosy::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return COSY_EHIP;                                                         \
        }                                                                             \
    } while (0)

#define COSY_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            cosy::set_error(__VA_ARGS__);       \
            return COSY_EINVAL;                 \
        }                                       \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Tuning / phase knock-out knobs exist only in the -DCOSY_TUNE build (lib/libcosyhip_tune.so: experiments, timing only --
// results may be meaningless).  The shipping library reads no environment variable and compiles no knock-out branch.
#ifdef COSY_TUNE
#include <stdlib.h>
static inline int tune_int(const char* name, int dflt) { const char* v = getenv(name); return v ? (int)strtol(v, nullptr, 0) : dflt; }
#define COSY_DBG(expr) (expr)
#else
static inline int tune_int(const char*, int dflt) { return dflt; }
#define COSY_DBG(expr) 0
#endif

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

static inline int dtype_size(int dtype) { return dtype == COSY_F32 ? 4 : 2; }
// statement form: COSY_DISPATCH_STMT(dtype, hipLaunchKernelGGL(kernel<T>, ...));
#define COSY_DISPATCH_STMT(dtype, ...)                                        \
    do {                                                                      \
        if ((dtype) == COSY_F32) { using T = float; __VA_ARGS__; }            \
        else if ((dtype) == COSY_BF16) { using T = cosy::bf16_t; __VA_ARGS__; } \
        else { using T = cosy::f16_t; __VA_ARGS__; }                          \
    } while (0)
// run `expr` with T bound to the storage type selected by `dtype`
#define COSY_DISPATCH_T(dtype, expr)                                    \
    ((dtype) == COSY_F32 ? [&] { using T = float; return (expr); }()    \
     : (dtype) == COSY_BF16 ? [&] { using T = cosy::bf16_t; return (expr); }() \
                            : [&] { using T = cosy::f16_t; return (expr); }())

// ---- kernels_geom.hip ----
int launch_crop_geometry(const float* pts_table, const int* obj_id, const float* K, const int* im_id, const float* TCO,
                         int B, int P, float z_min, int im_h, int im_w, int out_h, int out_w, float lamb,
                         float* boxes_rend, float* boxes_crop, float* K_crop, hipStream_t s);
int launch_roi_align(const float* images, const int* im_id, const float* boxes, int B, int N, int C, int h, int w,
                     int out_h, int out_w, int sampling, float* out, hipStream_t s);
int launch_pose_update(const float* TCO_in, const float* K_crop, const float* pose9, int B, float* TCO_out, hipStream_t s);
int launch_tco_init_from_boxes(const float* boxes, const float* K, const int* im_id, int B, float z, float* TCO, hipStream_t s);
int launch_tco_init_zup(const float* boxes, const float* pts_table, const int* obj_id, const float* K, const int* im_id,
                        int B, int P, float* TCO, hipStream_t s);
int launch_scatter_argmin(const float* dists, const int* ids, int M, int n_seg, int* out, hipStream_t s);
// crop + render pack into the NHWC8 network input (T = float or bf16_t), see kernels_geom.hip
int launch_frames_to_nhwc4(const float* images, float* out, int N, int h, int w, hipStream_t s);
int launch_frames_u8_to_nhwc4(const unsigned char* images, float* out, int N, int h, int w, hipStream_t s);
int launch_crop_pack(void* x_nhwc8, int dtype, const float* frames_nhwc4, const int* im_id, const float* boxes,
                     const float* renders, int B, int N, int h, int w, int H, int W, void* taps_ws, hipStream_t s);
size_t crop_taps_bytes(int B, int H, int W);    // scratch of the per-crop roi_align tap tables (taps_ws above; may be null)
int launch_pack_nchw(void* x_nhwc8, int dtype, const float* x_nchw6, int B, int H, int W, hipStream_t s);

}  // namespace cosy
