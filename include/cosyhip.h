/*
 * cosyhip.h -- C ABI of libcosyhip.so: the MI355X (gfx950) implementation of CosyPose's
 * render-and-compare pose-refinement hot path.
 *
 * Each entry point replaces one reference interface (file:line relative to the reference
 * checkout of ylabbe/cosypose); the Python shim in cosypose_amd/ binds them with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary;
 *   - every pointer is a DEVICE pointer unless the parameter name starts with `host_`;
 *   - all geometry tensors are fp32 row-major: poses (B,4,4), intrinsics (B,3,3),
 *     boxes (B,4) = x1,y1,x2,y2 in pixels; frames are fp32 NCHW (N,3,h,w) in [0,1];
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no
 *     allocation and no synchronisation; the caller owns and sizes every output;
 *   - return value: COSY_OK (0) or a negative COSY_E* code; cosy_last_error() gives the
 *     message of the last failure on the calling thread.  Nothing throws across the ABI;
 *   - the library owns only cosy_net_t (packed weights + activation workspace);
 *     distinct nets may be used concurrently from distinct threads/streams.
 */
#ifndef COSYHIP_H
#define COSYHIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COSY_VERSION 100 /* 0.1.0 */

enum { COSY_OK = 0, COSY_EINVAL = -1, COSY_ENOMEM = -2, COSY_EHIP = -3, COSY_ESIZE = -4 };

/* storage/compute type of the backbone activations and weights (accumulation is always fp32) */
enum { COSY_F32 = 0, COSY_BF16 = 1, COSY_F16 = 2 };

typedef struct cosy_net cosy_net_t;
typedef void* cosy_stream_t; /* hipStream_t */

int cosy_version(void);
const char* cosy_last_error(void);

/* ---- backbone: EfficientNet-B3(in_channels=6) + global-avg-pool + Linear(1536,9) ----------
 * Replaces EfficientNet.extract_features (cosypose/models/efficientnet.py:174-190), MBConvBlock.forward
 * (:71-98), Conv2dStaticSamePadding (cosypose/models/efficientnet_utils.py:123-146) and
 * PosePredictor.net_forward (cosypose/models/pose.py:81-87), eval mode.
 *
 * host_params: flat fp32 blob of the reference state_dict in this order (BN tensors as
 * weight,bias,running_mean,running_var; num_batches_tracked omitted):
 *   backbone._conv_stem.weight[40,6,3,3], backbone._bn0;
 *   for i in 0..25: [_expand_conv.weight[Cmid,Cin,1,1], _bn0]  (absent when expand==1),
 *                   _depthwise_conv.weight[Cmid,1,k,k], _bn1,
 *                   _se_reduce.weight[Cse,Cmid,1,1], _se_reduce.bias, _se_expand.weight[Cmid,Cse,1,1], _se_expand.bias,
 *                   _project_conv.weight[Cout,Cmid,1,1], _bn2;
 *   backbone._conv_head.weight[1536,384,1,1], backbone._bn1; pose_fc.weight[9,1536], pose_fc.bias[9].
 * cosy_effnet_b3_param_count() floats in total (10,798,441). */
long cosy_effnet_b3_param_count(void);
int cosy_effnet_b3_out_hw(int H, int W, int* out_h, int* out_w);
int cosy_effnet_b3_create(const float* host_params, size_t n_floats, int dtype, int H, int W, int max_batch,
                          cosy_net_t** out);
int cosy_effnet_b3_destroy(cosy_net_t* net);
size_t cosy_effnet_b3_workspace_bytes(const cosy_net_t* net);

/* Fill the net's 6-channel input from a reference-layout tensor x (B,6,H,W) fp32 NCHW
 * (what torch.cat((images_crop, renders), 1) produces, pose.py:104). */
int cosy_effnet_b3_set_input_nchw(cosy_net_t* net, const float* x, int B, cosy_stream_t stream);

/* Frames (N,3,h,w) fp32 planar -> (N,h,w,4) fp32 interleaved (RGB + pad), the layout cosy_crop_pack samples from.
 * Done once per PosePredictor.forward call (the frames do not change across iterations); out holds N*h*w*4 floats. */
int cosy_frames_to_nhwc4(const float* images, float* out, int N, int h, int w, cosy_stream_t stream);
/* the same from uint8 frames (N,3,h,w) as the datasets deliver them: out = value / 255.f (training/pose_forward_loss.py:24: images.float() / 255.) */
int cosy_frames_u8_to_nhwc4(const unsigned char* images, float* out, int N, int h, int w, cosy_stream_t stream);

/* Fused replacement of deepim_crops_robust's roi_align (cosypose/lib3d/cropping.py:73-74,
 * torchvision 0.4.2 semantics, sampling_ratio=4) + torch.cat with the renders (pose.py:104):
 * writes observed crop -> channels 0..2 and renders (B,3,H,W fp32 NCHW) -> channels 3..5 of
 * the net input.  frames_nhwc4 (N,h,w,4) from cosy_frames_to_nhwc4; im_id (B) int32 index into the frames
 * or NULL for identity (the reference gathers images[im_ids] first, cosypose/integrated/pose_predictor.py:41).
 * The 16 bilinear samples of a pixel are evaluated in separable form (same sum, fp32 rounding differs ~1e-7). */
int cosy_crop_pack(cosy_net_t* net, const float* frames_nhwc4, const int* im_id, const float* boxes_crop,
                   const float* renders, int B, int N, int h, int w, cosy_stream_t stream);

/* Run the backbone on the current input: feat (B,1536) fp32 or NULL, pose9 (B,9) fp32,
 * taps (B,9,16) fp32 or NULL = per-stage probes [mean, mean|x|, 14 strided samples] of the
 * stem, the 7 stage outputs and the head activation (test hook; same probe as the oracle). */
int cosy_effnet_b3_forward(cosy_net_t* net, int B, float* feat, float* pose9, float* taps, cosy_stream_t stream);

/* After cosy_effnet_b3_forward: copy the head activation out as the reference's backbone(x) would return it,
 * (B,1536,h,w) fp32 NCHW (what EfficientNet.forward yields, efficientnet.py:192-204). */
int cosy_effnet_b3_features_nchw(cosy_net_t* net, int B, float* out, cosy_stream_t stream);

/* ---- test probes (parity tests of the fused kernels; never on the hot path) -----------------
 * cosy_effnet_b3_set_probe: the NEXT forwards also copy one whole activation out as fp32 NCHW into `out`
 * (caller-allocated, B x C x HW floats): layer -1 = stem output, 0..25 = output of MBConv block i
 * (MBConvBlock.forward's return value, cosypose/models/efficientnet.py:71-98), 26 = head activation,
 * 100+i = depthwise output of block i (after _bn1 + swish, :83), 200+i = squeeze-excite gate of block i as (B, Cmid)
 * (sigmoid(_se_expand(...)), :85-88); layer -2 switches the probe off.
 * cosy_effnet_b3_block_info: dims[11] = {H, W, Ho, Wo, Cin, Cmid, Cout, front kernel (0 unfused, 1 wave, 2 small, 3 tiled, 4 block 0 behind the
 * fused stem: the stem tensor stays fp32 in registers and is never stored -- a forward with probe -1 or per-stage taps runs the unfused kernels,
 * 5 wave / 6 small with the depthwise taps applied by small MFMAs: the expanded values and the taps are rounded to the storage type), k, s,
 * SE gate applied to the project weights (1) or to the activation rows (0)}. */
int cosy_effnet_b3_set_probe(cosy_net_t* net, int layer, float* out);
int cosy_effnet_b3_block_info(const cosy_net_t* net, int block, int* dims);

/* ---- measurement hook ---------------------------------------------------------------------
 * With profiling enabled every launch of cosy_effnet_b3_forward is bracketed by HIP events recorded on the
 * launch stream (no synchronisation, one hipEventRecord per kernel; up to 24 forwards are retained).
 * cosy_effnet_b3_profile_read blocks until the recorded events have completed and returns one record per
 * launch slot of the schedule: kernel name (as rocprofv3 prints it, abbreviated), layer index, number of
 * timed launches, mean/min duration, and the ALGORITHMIC bytes and flops of one launch (tensor sizes
 * in+out+weights).  Reading resets the accumulation. */
typedef struct {
    char name[64];
    int layer;      /* -1 stem, 0..25 MBConv block, 26 head */
    int n;          /* launches timed */
    float ms_avg, ms_min;
    double bytes;   /* algorithmic HBM bytes of one launch */
    double flops;   /* algorithmic flops of one launch */
    double cbytes;  /* the compulsory part of `bytes`: block inputs / outputs / residuals and weights only -- the tensors that stay
                     * inside an MBConv block (expanded E, depthwise output D, squeeze sums, gates) count as 0 */
} cosy_prof_rec_t;
int cosy_effnet_b3_set_profiling(cosy_net_t* net, int enable);
int cosy_effnet_b3_profile_read(cosy_net_t* net, cosy_prof_rec_t* recs, int cap, int* n_out);

/* ---- geometry ---------------------------------------------------------------------------
 * PosePredictor.crop_inputs without the pixel work (cosypose/models/pose.py:45-67):
 * project_points_robust + boxes_from_uv (cosypose/lib3d/camera_geometry.py:18-42), deepim_boxes
 * (cosypose/lib3d/cropping.py:7-47, lamb=1.4, not clamped), get_K_crop_resize (camera_geometry.py:45-87).
 * pts_table (n_obj,P,3) = mesh_db points after sample_points(P, deterministic=True)
 * (cosypose/lib3d/mesh_ops.py:31-41); obj_id (B) int32 rows of pts_table; K is (N,3,3) indexed
 * by im_id (B) or (B,3,3) when im_id is NULL. */
int cosy_crop_geometry(const float* pts_table, const int* obj_id, const float* K, const int* im_id, const float* TCO,
                       int B, int P, float z_min, int im_h, int im_w, int out_h, int out_w, float lamb,
                       float* boxes_rend, float* boxes_crop, float* K_crop, cosy_stream_t stream);

/* torchvision.ops.roi_align(images, [im_id|boxes], (out_h,out_w), sampling_ratio) as the reference
 * calls it (cropping.py:74): out (B,C,out_h,out_w) fp32 NCHW. */
int cosy_roi_align(const float* images, const int* im_id, const float* boxes, int B, int N, int C, int h, int w,
                   int out_h, int out_w, int sampling_ratio, float* out, cosy_stream_t stream);

/* PosePredictor.update_pose for pose_dim=9 (pose.py:69-79): compute_rotation_matrix_from_ortho6d
 * (cosypose/lib3d/rotations.py:6-21) + apply_imagespace_predictions (cosypose/lib3d/cosypose_ops.py:10-31). */
int cosy_pose_update(const float* TCO_in, const float* K_crop, const float* pose9, int B, float* TCO_out,
                     cosy_stream_t stream);

/* TCO_init_from_boxes(z_range=(z,z)) (cosypose_ops.py:121-135) and
 * TCO_init_from_boxes_zup_autodepth (cosypose_ops.py:138-173), as used by
 * CoarseRefinePosePredictor.make_TCO_init (pose_predictor.py:65-74). */
int cosy_tco_init_from_boxes(const float* boxes, const float* K, const int* im_id, int B, float z, float* TCO,
                             cosy_stream_t stream);
int cosy_tco_init_zup_autodepth(const float* boxes, const float* pts_table, const int* obj_id, const float* K,
                                const int* im_id, int B, int P, float* TCO, cosy_stream_t stream);

/* ---- index / assignment ops adjacent to the loop (bit-exact) --------------------------------
 * scatter_argmin (cosypose/csrc/cosypose_cext.cpp:218-245): per segment id in [0,n_seg) the index of the
 * smallest distance, first index wins on ties; out[s] = -1 for an empty segment. dists (M) fp32,
 * ids (M) int32 on the device. */
int cosy_scatter_argmin(const float* dists, const int* ids, int M, int n_seg, int* out, cosy_stream_t stream);

/* expand_ids_for_symmetry (cosypose/csrc/cosypose_cext.cpp:247-259): for item n (in order), for k < n_sym_item[n]:
 * ids_expand[m] = n, sym_ids[m] = k.  n_sym_item (B) int32 = n_symmetries[labels[n]] (the label lookup is the caller's
 * dict).  The caller sizes the outputs with M = sum(n_sym_item); *total (optional, device) receives M. */
int cosy_expand_ids_for_symmetry(const int* n_sym_item, int B, int* ids_expand, int* sym_ids, int* total, cosy_stream_t stream);

/* ---- symmetric pose distances / losses' argmin / ADD(-S) (fp32; index outputs follow the reference's tie rule) ----
 * Points come from a per-object table pts_table (n_obj,P,3) indexed by obj_id (B) int32; obj_id == NULL means the
 * table is already per sample, (B,P,3) -- the layout the reference passes.
 *
 * cosy_symmetric_distance: symmetric_distance_batched (mode 0) / symmetric_distance_batched_fast (mode 1),
 * cosypose/lib3d/symmetric_distances.py:19-57.  sym_table (n_obj,S,4,4) is the identity-padded symmetry table,
 * n_sym (n_obj) the real counts (mode 0 scans only those, like expand_ids_for_symmetry + scatter_argmin: strict <,
 * first wins; mode 1 scans all S rows and takes the first minimum of the mean SQUARED distance, like argmin).
 * -> min_dists (B), best_sym (B) int32, S12 (B,4,4) = sym_table[obj, best]. */
int cosy_symmetric_distance(const float* T1, const float* T2, const int* obj_id, const float* pts_table, const float* sym_table,
                            const int* n_sym, int B, int P, int S, int mode, float* min_dists, int* best_sym, float* S12,
                            cosy_stream_t stream);
/* loss_CO_symmetric with l1 (cosypose/lib3d/cosypose_ops.py:34-46), forward value: per sample the minimum over the S
 * possible ground truths of mean |pred points - gt points| (first minimum wins, torch.min) -> loss (B), min_id (B)
 * int32 (optional), TCO_assign (B,4,4) (optional). */
int cosy_loss_co_symmetric(const float* TCO_possible_gt, const float* TCO_pred, const float* pts_table, const int* obj_id, int B,
                           int S, int P, float* loss, int* min_id, float* TCO_assign, cosy_stream_t stream);
/* loss_refiner_CO_disentangled (cosypose_ops.py:49-82), forward value -> loss (B). */
int cosy_loss_refiner_disentangled(const float* TCO_possible_gt, const float* TCO_input, const float* refiner_outputs,
                                   const float* K_crop, const float* pts_table, const int* obj_id, int B, int S, int P, float* loss,
                                   cosy_stream_t stream);
/* dists_add (symmetric = 0) / dists_add_symmetric (symmetric = 1), cosypose/lib3d/distances.py:5-21 -> dists (B,P,3):
 * gt point minus predicted point (ADD), or minus the NEAREST predicted point (ADD-S; first minimum of the squared
 * distance wins). */
int cosy_dists_add(const float* TXO_pred, const float* TXO_gt, const float* pts_table, const int* obj_id, int B, int P,
                   int symmetric, float* dists, cosy_stream_t stream);

/* ---- training step of the refiner network, fp32, activations NHWC = rows x channels --------------
 * Every piece of cosypose/training/train_pose.py:317-331's step: the 1x1 convolutions and their gradients on the library's own
 * fp32 MFMA GEMMs (cosy_train_gemm / cosy_wgrad below), BatchNorm, depthwise, squeeze-excite scaling, loss gradient, clip +
 * Adam, the squeeze-excite and pose FCs (cosy_se_train_*, cosy_fc_small_*): no rocBLAS on the step.  `workspace` is a device buffer
 * of cosy_train_workspace_bytes() bytes shared by the reductions (deterministic two-stage sums, no atomics). */
size_t cosy_train_workspace_bytes(void);
/* crop + render pack (as cosy_crop_pack) into a caller-owned NHWC8 buffer of element type `dtype` */
int cosy_crop_pack_to(void* x_nhwc8, int dtype, const float* frames_nhwc4, const int* im_id, const float* boxes_crop,
                      const float* renders, int B, int N, int h, int w, int H, int W, cosy_stream_t stream);
/* the same through the per-crop tap tables and the LDS-tiled kernel that cosy_crop_pack runs on the net's own buffer (bit-identical
 * to cosy_crop_pack_to): `workspace` = cosy_crop_pack_workspace_bytes(B, H, W) bytes of device scratch */
size_t cosy_crop_pack_workspace_bytes(int B, int H, int W);
int cosy_crop_pack_to_ws(void* x_nhwc8, int dtype, const float* frames_nhwc4, const int* im_id, const float* boxes_crop,
                         const float* renders, int B, int N, int h, int w, int H, int W, void* workspace, cosy_stream_t stream);
/* nn.BatchNorm2d in train mode (efficientnet.py:49-68; eps 1e-3, momentum 0.01): batch mean / 1/sqrt(biased var + eps)
 * per channel over the M rows; running_mean/var (optional) updated in place with the unbiased variance. */
int cosy_bn_train_stats(const float* x, long M, int C, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                        float* running_var, void* workspace, cosy_stream_t stream);
/* out = act((x-mean)*rstd*gamma+beta) [* rowscale[row/HW]] [+ res];  act 0 none / 1 Swish; rowscale = drop_connect's
 * per-sample mask/keep_prob (efficientnet_utils.py:83-92), res = the block's skip input. */
int cosy_bn_train_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, long M, int C,
                        int act, const float* rowscale, int HW, const float* res, float* out, cosy_stream_t stream);
/* backward of the above (+ SwishImplementation.backward, efficientnet_utils.py:44-48): dgamma, dbeta (accumulated when
 * `accumulate`), dx.  `sums` = 2*C floats of scratch. */
int cosy_bn_train_backward(const float* dout, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, long M, int C, int act, const float* rowscale, int HW, float* dgamma, float* dbeta,
                           int accumulate, float* dx, float* sums, void* workspace, cosy_stream_t stream);
/* the same with the incoming gradient formed on the fly as dout * cgate[sample][c] + cadd[sample][c] * cadd_scale (sample = row / HW; cgate,
 * cadd (B,C)): the squeeze-excite backward of an MBConv block (gradient through the gate multiply + the pooled mean) feeds BatchNorm 1's
 * backward without the sum being stored */
int cosy_bn_train_backward_gated(const float* dout, const float* cgate, const float* cadd, float cadd_scale, const float* x, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, long M, int C, int act, const float* rowscale, int HW,
                                 float* dgamma, float* dbeta, int accumulate, float* dx, float* sums, void* workspace, cosy_stream_t stream);
/* depthwise convolution with the reference's static "same" padding; weights transposed to (k*k, C) */
int cosy_dw_train_forward(const float* x, const float* wt, int B, int H, int W, int C, int k, int stride, float* out,
                          cosy_stream_t stream);
int cosy_dw_train_backward_data(const float* dy, const float* wt, int B, int H, int W, int C, int k, int stride, float* dx,
                                cosy_stream_t stream);
/* ... + add (B,H,W,C): dx = conv-transpose(dy) + add in one pass (stride 1): the skip connection's gradient of an MBConv block without expansion */
int cosy_dw_train_backward_data_add(const float* dy, const float* wt, const float* add, int B, int H, int W, int C, int k, int stride, float* dx,
                                    cosy_stream_t stream);
int cosy_dw_train_backward_weight(const float* x, const float* dy, int B, int H, int W, int C, int k, int stride, float* dwt,
                                  void* workspace, cosy_stream_t stream);
/* module_layout = 1: the gradient is written as (C, k*k) = the module's _depthwise_conv.weight (C,1,k,k) instead of [tap][channel] */
int cosy_dw_train_backward_weight_ex(const float* x, const float* dy, int B, int H, int W, int C, int k, int stride, float* dw, int module_layout,
                                     void* workspace, cosy_stream_t stream);
/* 1x1 convolutions of the training step on the library's own fp32 MFMA GEMMs (no rocBLAS on the path):
 *   cosy_train_gemm: out (M,N) = A (M,K) . op(W) (+ add (M,N)); op(W) = W^T for W stored (N,K) [w_is_kn = 0: the forward of
 *                    F.conv2d with a 1x1 kernel, efficientnet.py:81,90,188], W for W stored (K,N) [w_is_kn = 1: the data
 *                    gradient dX = dY . W].  K and N multiples of 4.  The weights are packed on the device per call.
 *   cosy_wgrad:      dW (N,K) = dY^T (N,M) . X (M,K), dY (M,N), X (M,K) row-major, any shape; deterministic (fixed-order
 *                    combine of per-slab partial tiles).
 * cosy_wgrad_tall / _supported are round 1's names for the same kernel (kept: every shape is supported now). */
int cosy_train_gemm(const float* A, const float* W, int w_is_kn, long M, int K, int N, const float* add, float* out, void* workspace,
                    cosy_stream_t stream);
int cosy_wgrad(const float* dY, const float* X, long M, int N, int K, float* dW, void* workspace, cosy_stream_t stream);
/* The weights of EVERY 1x1 convolution packed in one launch per step (they change once per step, in Adam): cosy_train_pack_plan fills `plan`
 * (host, 8 x int64 per entry: see kernels_train.hip) for n weights given as in cosy_train_gemm and returns the pool size in floats and the grid;
 * the caller uploads the plan, cosy_train_pack_all packs into `pool` (pool_floats floats), cosy_train_gemm_packed is cosy_train_gemm on entry e's
 * packed form at pool + plan[e][1]. */
int cosy_train_pack_plan(int n, const float* const* W, const int* K, const int* N, const int* w_is_kn, long long* plan, long long* pool_floats,
                         long long* n_blocks);
int cosy_train_pack_all(const long long* plan_dev, int n, long long n_blocks, float* pool, cosy_stream_t stream);
int cosy_train_gemm_packed(const float* A, const float* pool, long long packed_offset, long M, int K, int N, const float* add, float* out,
                           cosy_stream_t stream);

int cosy_wgrad_tall_supported(long M, int N, int K);
int cosy_wgrad_tall(const float* dY, const float* X, long M, int N, int K, float* dW, void* workspace, cosy_stream_t stream);
/* per-sample reductions / broadcasts over the HW pixels of a (B,HW,C) activation: mean (adaptive_avg_pool2d),
 * sum of a*a2 (gradient of the squeeze-excite gate), a*g[b,c] (+ add[b,c]*add_scale), v[b,c]*scale broadcast */
int cosy_rows_mean(const float* a, int B, int HW, int C, float* out, void* workspace, cosy_stream_t stream);
/* Squeeze-excite without the unscaled activation in memory (efficientnet.py:84-90 in train mode): the per-sample means / dot products take the
 * BatchNorm INPUT `raw` and recompute swish(bn(raw)) per element (the arithmetic of cosy_bn_train_apply), and cosy_bn_train_apply_gated writes
 * swish(bn(raw)) * cgate[sample][c] -- the project conv's input -- directly. */
int cosy_rows_mean_bn(const float* raw, const float* mean, const float* rstd, const float* gamma, const float* beta, int B, int HW, int C, float* out,
                      void* workspace, cosy_stream_t stream);
int cosy_rows_dot_bn(const float* a, const float* raw, const float* mean, const float* rstd, const float* gamma, const float* beta, int B, int HW, int C,
                     float* out, void* workspace, cosy_stream_t stream);
int cosy_bn_train_apply_gated(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, long M, int C, int act,
                              const float* cgate, int HW, float* out, cosy_stream_t stream);
int cosy_rows_dot(const float* a, const float* a2, int B, int HW, int C, float* out, void* workspace, cosy_stream_t stream);
int cosy_rows_scale(const float* a, const float* g, const float* add, float add_scale, int B, int HW, int C, float* out,
                    cosy_stream_t stream);
int cosy_rows_broadcast(const float* v, float scale, int B, int HW, int C, float* out, cosy_stream_t stream);
/* elementwise Swish (kind 0) / sigmoid (kind 1) and their gradients */
/* Squeeze-excite of an MBConv block in the training step (efficientnet.py:85-88 and its autograd), one launch forward, two backward:
 *   forward : h_pre (B,Cse) = pooled (B,C) w_reduce^T + b_reduce;  gate (B,C) = sigmoid(swish(h_pre) w_expand^T + b_expand)
 *   backward: from dgate (B,C): dpooled (B,C), dw_reduce (Cse,C), db_reduce (Cse), dw_expand (C,Cse), db_expand (C); sums over the batch in
 *             sample order (deterministic).  Weight layouts are the module's own (_se_reduce.weight (Cse,C,1,1), _se_expand.weight (C,Cse,1,1)).
 * cosy_fc_small_*: a Linear layer with J <= 16 outputs (models/pose.py:84: pose_fc, J = 9): y = x w^T + bias and its three gradients.
 * They replace torch.addmm / matmul + elementwise launches (rocBLAS at 64 rows). */
int cosy_se_train_forward(const float* pooled, const float* w_reduce, const float* b_reduce, const float* w_expand, const float* b_expand, int B, int C,
                          int Cse, float* h_pre, float* gate, cosy_stream_t stream);
int cosy_se_train_backward(const float* dgate, const float* gate, const float* h_pre, const float* pooled, const float* w_reduce, const float* w_expand,
                           int B, int C, int Cse, float* dpooled, float* dw_reduce, float* db_reduce, float* dw_expand, float* db_expand, void* workspace,
                           cosy_stream_t stream);
int cosy_fc_small_forward(const float* x, const float* w, const float* bias, int B, int C, int J, float* y, cosy_stream_t stream);
int cosy_fc_small_backward(const float* dy, const float* x, const float* w, int B, int C, int J, float* dx, float* dw, float* db, cosy_stream_t stream);
int cosy_act_forward(const float* x, long n, int kind, float* out, cosy_stream_t stream);
int cosy_act_backward(const float* x, const float* dy, long n, int kind, float* dx, cosy_stream_t stream);
/* stem 3x3 stride-2 patches of the NHWC8 input as GEMM rows: cols (B*Ho*Wo, 54), column = (ky*3+kx)*6 + c */
int cosy_stem_im2col(const float* x_nhwc8, int B, int H, int W, float* cols, cosy_stream_t stream);
/* the same with a row stride ld >= 54 (columns 54..ld-1 are written as zeros): ld = 56 gives the 16-byte aligned rows
 * cosy_train_gemm wants */
int cosy_stem_im2col_ld(const float* x_nhwc8, int B, int H, int W, int ld, float* cols, cosy_stream_t stream);
/* gradient of loss_refiner_CO_disentangled (cosypose_ops.py:49-82) wrt refiner_outputs, times the upstream dloss (B) */
int cosy_loss_refiner_disentangled_backward(const float* TCO_possible_gt, const float* TCO_input, const float* refiner_outputs,
                                            const float* K_crop, const float* pts_table, const int* obj_id, int B, int S, int P,
                                            const float* dloss, float* d_refiner_outputs, cosy_stream_t stream);
/* torch.nn.utils.clip_grad_norm_(max_norm, 2) on a flat gradient buffer -> norm_and_coef[0] = total norm,
 * [1] = min(1, max_norm/(norm+1e-6)) (1 when max_norm <= 0); torch.optim.Adam step on flat buffers, gradients scaled by
 * norm_and_coef[1] when given (train_pose.py:325-329). */
int cosy_grad_norm_clip(const float* grads, long n, float max_norm, float* norm_and_coef, void* workspace, cosy_stream_t stream);
int cosy_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, const float* norm_and_coef, cosy_stream_t stream);

/* ---- on-device mesh rasteriser behind renderer.render (interface of BulletBatchRenderer.render,
 * cosypose/rendering/bullet_batch_renderer.py:46-90; camera model of simulator/camera.py:9-33).  verts / colors
 * (n_obj,V,3), faces (n_obj,F,3) int32 (padded), n_faces (n_obj); per crop obj_id, TCO (B,4,4), K (B,3,3) ->
 * rgb (B,3,H,W) in [0,1] (black background, non-finite poses -> black) and optional depth (B,H,W) in metres.
 * Shading = vertex colour x (ambient + diffuse |n.l|), l = light direction in the camera frame (PyBullet's OpenGL shading
 * is third-party: pixel values parity-unpinned).  `scratch`: cosy_render_scratch_bytes(B,V,H,W) bytes. */
size_t cosy_render_scratch_bytes(int B, int V, int H, int W);
int cosy_render_meshes(const float* verts, const float* colors, const int* faces, const int* n_faces, const int* obj_id,
                       const float* TCO, const float* K, int B, int V, int F, int H, int W, float ambient, float diffuse,
                       float light_x, float light_y, float light_z, float* rgb, float* depth, void* scratch, cosy_stream_t stream);

/* The object set on the device (one row per label, padded to V vertices / F faces) and the shading model.
 * cosy_shade_t.smooth = 1 selects the OpenGL-like model that PyBullet's hardware renderer has in structure
 * (bullet_scene_renderer.py:38-60 -> getCameraImage(ER_BULLET_HARDWARE_OPENGL)): texture x vertex colour, interpolated
 * vertex normals, one-sided Lambert + Blinn-Phong highlight, light fixed in the world frame (= the object's frame: every
 * object is rendered at TWO = identity, bullet_batch_renderer.py:56-59), 8-bit output (quantize = 1:
 * `images.float() / 255`, bullet_batch_renderer.py:83-84).  The shader's constants are third-party: pixel values stay
 * parity-unpinned. */
typedef struct {
    const float* verts;   /* (n_obj,V,3) metres, object frame */
    const float* colors;  /* (n_obj,V,3) in [0,1] */
    const float* normals; /* (n_obj,V,3) unit vertex normals, or NULL (flat shading only) */
    const float* uvs;     /* (n_obj,V,2) texture coordinates, or NULL */
    const float* tex;     /* (n_obj,TH,TW,4) RGB(+pad) fp32 in [0,1], or NULL */
    const int* faces;     /* (n_obj,F,3) */
    const int* n_faces;   /* (n_obj) */
    int V, F, TH, TW;
} cosy_mesh_t;
typedef struct {
    float ambient, diffuse, specular, shininess;
    float light[3];       /* unit vector from the surface towards the light */
    int light_frame;      /* 0: camera frame, 1: object (PyBullet world) frame */
    int smooth;           /* 0: flat two-sided face normals, 1: interpolated vertex normals */
    int quantize;         /* 1: round colours to multiples of 1/255 */
} cosy_shade_t;
int cosy_render_meshes_ex(const cosy_mesh_t* mesh, const cosy_shade_t* shade, const int* obj_id, const float* TCO, const float* K, int B,
                          int H, int W, float* rgb, float* depth, void* scratch, cosy_stream_t stream);

/* Render + crop + pack in one pass (replaces renderer.render -> cosy_crop_pack when the renderer is this library's): the
 * resolve pass of the rasteriser also samples the observed frame (roi_align, as cosy_crop_pack) and writes the network's
 * 8-channel NHWC input pixel directly -- no fp32 (B,3,H,W) render tensor is written or read.  K = K_crop (B,3,3); the
 * render resolution is the network's input resolution. */
int cosy_render_crop_pack(cosy_net_t* net, const cosy_mesh_t* mesh, const cosy_shade_t* shade, const int* obj_id, const float* TCO,
                          const float* K_crop, const float* frames_nhwc4, const int* im_id, const float* boxes_crop, int B, int N, int h,
                          int w, void* scratch, cosy_stream_t stream);
int cosy_render_crop_pack_to(void* x_nhwc8, int dtype, const cosy_mesh_t* mesh, const cosy_shade_t* shade, const int* obj_id,
                             const float* TCO, const float* K_crop, const float* frames_nhwc4, const int* im_id, const float* boxes_crop,
                             int B, int N, int h, int w, int H, int W, void* scratch, cosy_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* COSYHIP_H */
