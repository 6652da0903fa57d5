// Symmetric pose distances, the losses' argmin over symmetric ground truths, and ADD / ADD-S point distances
// (SURVEY 8a-12, 8f-2).  Reference: cosypose/lib3d/symmetric_distances.py:19-57, cosypose_ops.py:34-82,
// distances.py:5-21, csrc/cosypose_cext.cpp:247-259.
//
// All of this is small, latency/L2-bound work (a few thousand points per object): one workgroup per sample (or per
// 256 ground-truth points for ADD-S), points read straight from the per-object table (no B x P x 3 gather, no
// (B,S,P,3) intermediates as in the reference), reductions in a FIXED order (deterministic, batch-invariant).
// fp32 with contraction off, so that every per-point value equals the CPU restatement's bit for bit; only the order
// of the P-term sums differs (~1e-7 relative).  Index results (best symmetry, assigned ground truth, nearest point)
// follow the reference's tie rule: strict <, first index wins.
#include "cosy_common.h"

#pragma clang fp contract(off)

namespace cosy {

namespace {

__device__ __forceinline__ void mat4_mul(const float* A, const float* Bm, float* C) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += A[i * 4 + k] * Bm[k * 4 + j];
            C[i * 4 + j] = acc;
        }
}
// transform_pts (lib3d/transform_ops.py:7-21): R p + t
__device__ __forceinline__ void xform_pt(const float* T, float x, float y, float z, float* q) {
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = ((T[i * 4 + 0] * x + T[i * 4 + 1] * y) + T[i * 4 + 2] * z) + T[i * 4 + 3];
}

// sum over the 256 threads of a workgroup in a fixed order: lanes by xor-shuffle tree, then waves 0..3 in sequence.
// `scratch` = 4 floats of LDS per reduced value.  Returns the total in every thread.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();   // scratch may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    return ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
}

__global__ __launch_bounds__(256) void symmetric_distance_kernel(const float* __restrict__ T1, const float* __restrict__ T2,
                                                                 const int* __restrict__ obj, const float* __restrict__ pts,
                                                                 const float* __restrict__ sym, const int* __restrict__ n_sym,
                                                                 int P, int S, int mode, float* __restrict__ min_dists,
                                                                 int* __restrict__ best_sym, float* __restrict__ S12) {
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int o = obj ? obj[b] : b;
    const float* p = pts + (size_t)o * P * 3;
    float t1[16], t2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { t1[i] = T1[(size_t)b * 16 + i]; t2[i] = T2[(size_t)b * 16 + i]; }
    const int ns = mode == 0 ? (n_sym ? n_sym[o] : S) : S;
    int best = -1;
    float best_c = 0.f, best_d = 0.f;
    for (int s = 0; s < ns; ++s) {
        float sm[16], M[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) sm[i] = sym[((size_t)o * S + s) * 16 + i];
        mat4_mul(t1, sm, M);
        float sum_n = 0.f, sum_sq = 0.f;
        for (int i = tid; i < P; i += 256) {
            const float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
            float q1[3], q2[3];
            xform_pt(M, x, y, z, q1);
            xform_pt(t2, x, y, z, q2);
            const float dx = q1[0] - q2[0], dy = q1[1] - q2[1], dz = q1[2] - q2[2];
            const float sq = (dx * dx + dy * dy) + dz * dz;
            sum_sq += sq;
            sum_n += sqrtf(sq);
        }
        sum_n = block_sum(sum_n, red);
        sum_sq = block_sum(sum_sq, red + 4);
        const float c = mode == 0 ? sum_n / (float)P : sum_sq / (float)P;
        if (best < 0 || c < best_c) { best = s; best_c = c; best_d = sum_n / (float)P; }
    }
    if (tid == 0) { min_dists[b] = best_d; best_sym[b] = best; }
    if (tid < 16 && best >= 0) S12[(size_t)b * 16 + tid] = sym[((size_t)o * S + best) * 16 + tid];
}

// mean over the 3P coordinates of |pred p - gt_s p| for every possible ground truth s; min with first-wins
__device__ __forceinline__ void co_symmetric(const float* pred /*regs*/, const float* __restrict__ gt /*S,4,4*/,
                                             const float* __restrict__ p, int S, int P, float* red, float& loss, int& arg) {
    arg = -1; loss = 0.f;
    for (int s = 0; s < S; ++s) {
        float g[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = gt[(size_t)s * 16 + i];
        float acc = 0.f;
        for (int i = threadIdx.x; i < P; i += 256) {
            const float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
            float q1[3], q2[3];
            xform_pt(pred, x, y, z, q1);
            xform_pt(g, x, y, z, q2);
            acc += fabsf(q1[0] - q2[0]); acc += fabsf(q1[1] - q2[1]); acc += fabsf(q1[2] - q2[2]);
        }
        const float l = block_sum(acc, red) / (float)(3 * P);
        if (arg < 0 || l < loss) { arg = s; loss = l; }
    }
}

__global__ __launch_bounds__(256) void loss_co_symmetric_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                                const float* __restrict__ pts, const int* __restrict__ obj,
                                                                int S, int P, float* __restrict__ loss, int* __restrict__ min_id,
                                                                float* __restrict__ assign) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* p = pts + (size_t)(obj ? obj[b] : b) * P * 3;
    float pr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pr[i] = pred[(size_t)b * 16 + i];
    float l; int a;
    co_symmetric(pr, gt + (size_t)b * S * 16, p, S, P, red, l, a);
    if (threadIdx.x == 0) { loss[b] = l; if (min_id) min_id[b] = a; }
    if (assign && threadIdx.x < 16 && a >= 0) assign[(size_t)b * 16 + threadIdx.x] = gt[((size_t)b * S + a) * 16 + threadIdx.x];
}

__global__ __launch_bounds__(256) void loss_refiner_disentangled_kernel(const float* __restrict__ gt, const float* __restrict__ TCO_in,
                                                                        const float* __restrict__ out9, const float* __restrict__ K_crop,
                                                                        const float* __restrict__ pts, const int* __restrict__ obj,
                                                                        int S, int P, float* __restrict__ loss) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* p = pts + (size_t)(obj ? obj[b] : b) * P * 3;
    const float* g0 = gt + (size_t)b * S * 16;
    float Ti[16], o[9], G[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { Ti[i] = TCO_in[(size_t)b * 16 + i]; G[i] = g0[i]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = out9[(size_t)b * 9 + i];
    const float fx = K_crop[(size_t)b * 9], fy = K_crop[(size_t)b * 9 + 4];
    // ortho6d -> dR (rotations.py:6-21), columns x y z
    const float na = sqrtf((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]);
    const float x0 = o[0] / na, x1 = o[1] / na, x2 = o[2] / na;
    float z0 = x1 * o[5] - x2 * o[4], z1 = x2 * o[3] - x0 * o[5], z2 = x0 * o[4] - x1 * o[3];
    const float nz = sqrtf((z0 * z0 + z1 * z1) + z2 * z2);
    z0 /= nz; z1 /= nz; z2 /= nz;
    const float y0 = z1 * x2 - z2 * x1, y1 = z2 * x0 - z0 * x2, y2 = z0 * x1 - z1 * x0;
    const float dR[9] = {x0, y0, z0, x1, y1, z1, x2, y2, z2};
    float total = 0.f;
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {
        float pr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pr[i] = G[i];
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc += dR[i * 3 + k] * Ti[k * 4 + j];
                    pr[i * 4 + j] = acc;
                }
        } else if (t == 1) {
            pr[3] = (o[6] / fx + Ti[3] / Ti[11]) * G[11];
            pr[7] = (o[7] / fy + Ti[7] / Ti[11]) * G[11];
        } else {
            pr[11] = o[8] * Ti[11];
        }
        float l; int a;
        co_symmetric(pr, g0, p, S, P, red, l, a);
        total = t == 0 ? l : total + l;
    }
    if (threadIdx.x == 0) loss[b] = total;
}

constexpr int ADD_CHUNK = 2048;   // predicted points staged in LDS per pass (24 KB)
__global__ __launch_bounds__(256) void dists_add_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        const float* __restrict__ pts, const int* __restrict__ obj, int P,
                                                        int symmetric, float* __restrict__ out) {
    __shared__ float pp[ADD_CHUNK * 3];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const float* p = pts + (size_t)(obj ? obj[b] : b) * P * 3;
    float tp[16], tg[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { tp[k] = pred[(size_t)b * 16 + k]; tg[k] = gt[(size_t)b * 16 + k]; }
    float g[3] = {0.f, 0.f, 0.f};
    if (i < P) xform_pt(tg, p[i * 3], p[i * 3 + 1], p[i * 3 + 2], g);
    float r[3];
    if (!symmetric) {
        float q[3] = {0.f, 0.f, 0.f};
        if (i < P) xform_pt(tp, p[i * 3], p[i * 3 + 1], p[i * 3 + 2], q);
        r[0] = g[0] - q[0]; r[1] = g[1] - q[1]; r[2] = g[2] - q[2];
    } else {
        float best = 0.f;
        bool have = false;
        r[0] = r[1] = r[2] = 0.f;
        for (int j0 = 0; j0 < P; j0 += ADD_CHUNK) {
            const int nj = min(ADD_CHUNK, P - j0);
            __syncthreads();
            for (int j = threadIdx.x; j < nj; j += 256) {
                float q[3];
                xform_pt(tp, p[(j0 + j) * 3], p[(j0 + j) * 3 + 1], p[(j0 + j) * 3 + 2], q);
                pp[j * 3] = q[0]; pp[j * 3 + 1] = q[1]; pp[j * 3 + 2] = q[2];
            }
            __syncthreads();
            for (int j = 0; j < nj; ++j) {   // all lanes read the same LDS address: broadcast
                const float dx = g[0] - pp[j * 3], dy = g[1] - pp[j * 3 + 1], dz = g[2] - pp[j * 3 + 2];
                const float sq = (dx * dx + dy * dy) + dz * dz;
                if (!have || sq < best) { best = sq; have = true; r[0] = dx; r[1] = dy; r[2] = dz; }
            }
        }
    }
    if (i < P) {
        float* o = out + ((size_t)b * P + i) * 3;
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
    }
}

// expand_ids_for_symmetry on the device: one workgroup scans the per-item counts, then every item writes its run
__global__ __launch_bounds__(256) void expand_ids_kernel(const int* __restrict__ n_sym_item, int B, int* __restrict__ ids_expand,
                                                         int* __restrict__ sym_ids, int* __restrict__ total) {
    __shared__ int part[256];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 256) {
        const int n = base + tid < B ? n_sym_item[base + tid] : 0;
        part[tid] = n;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {   // Hillis-Steele inclusive scan
            const int v = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const int start = carry + part[tid] - n;
        for (int k = 0; k < n; ++k) { ids_expand[start + k] = base + tid; sym_ids[start + k] = k; }
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0 && total) *total = carry;
}

}  // namespace

}  // namespace cosy

using namespace cosy;

extern "C" {

int cosy_symmetric_distance(const float* T1, const float* T2, const int* obj_id, const float* pts_table, const float* sym_table,
                            const int* n_sym, int B, int P, int S, int mode, float* min_dists, int* best_sym, float* S12,
                            cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(B >= 0 && P > 0 && S > 0 && (mode == 0 || mode == 1), "cosy_symmetric_distance: B=%d P=%d S=%d mode=%d", B, P, S, mode);
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(T1 && T2 && pts_table && sym_table && min_dists && best_sym && S12, "cosy_symmetric_distance: null pointer");
    hipLaunchKernelGGL(symmetric_distance_kernel, dim3(B), dim3(256), 0, s, T1, T2, obj_id, pts_table, sym_table, n_sym, P, S, mode,
                       min_dists, best_sym, S12);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

int cosy_loss_co_symmetric(const float* TCO_possible_gt, const float* TCO_pred, const float* pts_table, const int* obj_id, int B,
                           int S, int P, float* loss, int* min_id, float* TCO_assign, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(B >= 0 && P > 0 && S > 0, "cosy_loss_co_symmetric: B=%d S=%d P=%d", B, S, P);
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(TCO_possible_gt && TCO_pred && pts_table && loss, "cosy_loss_co_symmetric: null pointer");
    hipLaunchKernelGGL(loss_co_symmetric_kernel, dim3(B), dim3(256), 0, s, TCO_possible_gt, TCO_pred, pts_table, obj_id, S, P, loss,
                       min_id, TCO_assign);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

int cosy_loss_refiner_disentangled(const float* TCO_possible_gt, const float* TCO_input, const float* refiner_outputs,
                                   const float* K_crop, const float* pts_table, const int* obj_id, int B, int S, int P, float* loss,
                                   cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(B >= 0 && P > 0 && S > 0, "cosy_loss_refiner_disentangled: B=%d S=%d P=%d", B, S, P);
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(TCO_possible_gt && TCO_input && refiner_outputs && K_crop && pts_table && loss, "cosy_loss_refiner_disentangled: null pointer");
    hipLaunchKernelGGL(loss_refiner_disentangled_kernel, dim3(B), dim3(256), 0, s, TCO_possible_gt, TCO_input, refiner_outputs, K_crop,
                       pts_table, obj_id, S, P, loss);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

int cosy_dists_add(const float* TXO_pred, const float* TXO_gt, const float* pts_table, const int* obj_id, int B, int P, int symmetric,
                   float* dists, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(B >= 0 && P > 0, "cosy_dists_add: B=%d P=%d", B, P);
    if (B == 0) return COSY_OK;
    COSY_REQUIRE(TXO_pred && TXO_gt && pts_table && dists, "cosy_dists_add: null pointer");
    hipLaunchKernelGGL(dists_add_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, s, TXO_pred, TXO_gt, pts_table, obj_id, P, symmetric ? 1 : 0,
                       dists);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

int cosy_expand_ids_for_symmetry(const int* n_sym_item, int B, int* ids_expand, int* sym_ids, int* total, cosy_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    COSY_REQUIRE(B >= 0, "cosy_expand_ids_for_symmetry: B=%d", B);
    if (B == 0) {
        if (total) COSY_CHECK_HIP(hipMemsetAsync(total, 0, sizeof(int), s));
        return COSY_OK;
    }
    COSY_REQUIRE(n_sym_item && ids_expand && sym_ids, "cosy_expand_ids_for_symmetry: null pointer");
    hipLaunchKernelGGL(expand_ids_kernel, dim3(1), dim3(256), 0, s, n_sym_item, B, ids_expand, sym_ids, total);
    COSY_CHECK_HIP(hipGetLastError());
    return COSY_OK;
}

}  // extern "C"
