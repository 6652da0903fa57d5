"""Tensor-level mirrors of the reference's lib3d functions on the hot path, executed by
the HIP kernels in libcosyhip.so.  Inputs/outputs are torch tensors on a ROCm device
(fp32, contiguous copies are made when needed).  Same names and argument meaning as:

  crop_geometry              = PosePredictor.crop_inputs' boxes/K part      cosypose/models/pose.py:45-67
  roi_align                  = torchvision.ops.roi_align as called at       cosypose/lib3d/cropping.py:74
  get_K_crop_resize, boxes_from_uv, project_points_robust, deepim_boxes     -> folded into crop_geometry
  compute_rotation_matrix_from_ortho6d + apply_imagespace_predictions
                             = update_pose                                  cosypose/models/pose.py:69-79
  TCO_init_from_boxes, TCO_init_from_boxes_zup_autodepth                    cosypose/lib3d/cosypose_ops.py:121-173
  scatter_argmin                                                             cosypose/lib3d/symmetric_distances.py:13-16
  loss_CO_symmetric, loss_refiner_CO_disentangled (forward values)           cosypose/lib3d/cosypose_ops.py:34-82
"""
import torch

from . import _lib
from ._lib import lib, check, ptr, stream, require_device, ints_to_device


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _i32(t, device):
    if t is None:
        return None
    return ints_to_device(t, device)


def _dst(out, shape, dev):
    """a caller-provided destination (contiguous fp32 tensor of `shape` on `dev`, e.g. the rows of a result tensor) or a fresh one"""
    if out is None:
        return torch.empty(*shape, device=dev)
    assert tuple(out.shape) == tuple(shape) and out.dtype == torch.float32 and out.is_contiguous() and out.device == dev, (out.shape, shape)
    return out


def crop_geometry(point_table, obj_ids, K, TCO, im_size, render_size, im_ids=None, lamb=1.4, z_min=0.1, out=None):
    """-> boxes_rend (B,4), boxes_crop (B,4), K_crop (B,3,3).  K is (B,3,3), or (N,3,3) with im_ids (B,).
    out = (boxes_rend, boxes_crop, K_crop) destinations to write into (each may be None)."""
    require_device(point_table, K, TCO)
    TCO, K, point_table = _f32(TCO), _f32(K), _f32(point_table)
    dev = TCO.device
    obj_ids, im_ids = _i32(obj_ids, dev), _i32(im_ids, dev)
    B = TCO.shape[0]
    assert TCO.shape == (B, 4, 4) and obj_ids.shape == (B,)
    assert K.shape == ((im_ids is None and B or K.shape[0]), 3, 3)
    o = out if out is not None else (None, None, None)
    boxes_rend, boxes_crop, K_crop = _dst(o[0], (B, 4), dev), _dst(o[1], (B, 4), dev), _dst(o[2], (B, 3, 3), dev)
    check(lib().cosy_crop_geometry(ptr(point_table), ptr(obj_ids), ptr(K), ptr(im_ids), ptr(TCO), B, point_table.shape[1],
                                   z_min, int(im_size[0]), int(im_size[1]), int(render_size[0]), int(render_size[1]),
                                   lamb, ptr(boxes_rend), ptr(boxes_crop), ptr(K_crop), stream()))
    return boxes_rend, boxes_crop, K_crop


def roi_align(images, boxes, output_size, sampling_ratio=4, im_ids=None):
    """images (N,C,h,w); boxes (B,4) xyxy (+ im_ids) or (B,5) with the batch index in column 0."""
    require_device(images, boxes)
    images, boxes = _f32(images), _f32(boxes)
    if boxes.shape[1] == 5:
        im_ids = boxes[:, 0].to(torch.int32)
        boxes = boxes[:, 1:].contiguous()
    im_ids = _i32(im_ids, images.device)
    B = boxes.shape[0]
    N, C, h, w = images.shape
    out = torch.empty(B, C, int(output_size[0]), int(output_size[1]), device=images.device)
    check(lib().cosy_roi_align(ptr(images), ptr(im_ids), ptr(boxes), B, N, C, h, w, int(output_size[0]), int(output_size[1]),
                               int(sampling_ratio), ptr(out), stream()))
    return out


def update_pose(TCO, K_crop, pose_outputs, out=None):
    require_device(TCO, K_crop, pose_outputs)
    TCO, K_crop, pose_outputs = _f32(TCO), _f32(K_crop), _f32(pose_outputs)
    assert pose_outputs.shape[-1] == 9
    out = _dst(out, TCO.shape, TCO.device)
    check(lib().cosy_pose_update(ptr(TCO), ptr(K_crop), ptr(pose_outputs), TCO.shape[0], ptr(out), stream()))
    return out


def TCO_init_from_boxes(z_range, boxes, K, im_ids=None):
    assert len(z_range) == 2 and boxes.dim() == 2 and boxes.shape[-1] == 4
    require_device(boxes, K)
    boxes, K = _f32(boxes), _f32(K)
    im_ids = _i32(im_ids, boxes.device)
    z = float(torch.as_tensor(z_range, dtype=torch.float32).mean())
    TCO = torch.empty(boxes.shape[0], 4, 4, device=boxes.device)
    check(lib().cosy_tco_init_from_boxes(ptr(boxes), ptr(K), ptr(im_ids), boxes.shape[0], z, ptr(TCO), stream()))
    return TCO


def TCO_init_from_boxes_zup_autodepth(boxes_2d, point_table, obj_ids, K, im_ids=None):
    require_device(boxes_2d, point_table, K)
    boxes_2d, point_table, K = _f32(boxes_2d), _f32(point_table), _f32(K)
    dev = boxes_2d.device
    obj_ids, im_ids = _i32(obj_ids, dev), _i32(im_ids, dev)
    TCO = torch.empty(boxes_2d.shape[0], 4, 4, device=dev)
    check(lib().cosy_tco_init_zup_autodepth(ptr(boxes_2d), ptr(point_table), ptr(obj_ids), ptr(K), ptr(im_ids),
                                            boxes_2d.shape[0], point_table.shape[1], ptr(TCO), stream()))
    return TCO


def scatter_argmin(dists, ids_expand, n_segments=None):
    """Segmented argmin, first index wins on ties (bit-exact vs cosypose_cext.scatter_argmin)."""
    require_device(dists)
    dists = _f32(dists)
    ids = _i32(ids_expand, dists.device)
    if n_segments is None:
        n_segments = int(ids.max().item()) + 1 if ids.numel() else 0
    out = torch.empty(n_segments, dtype=torch.int32, device=dists.device)
    check(lib().cosy_scatter_argmin(ptr(dists), ptr(ids), dists.shape[0], n_segments, ptr(out), stream()))
    return out


def loss_CO_symmetric(TCO_possible_gt, TCO_pred, points):
    """Forward value of the reference's loss_CO_symmetric with l1 (cosypose_ops.py:34-46): (loss (B,), TCO_assign (B,4,4)).
    No autograd graph is built (the training step, SURVEY 8a-13, is not part of this library yet)."""
    bsz = TCO_possible_gt.shape[0]
    assert TCO_possible_gt.dim() == 4 and TCO_possible_gt.shape[-2:] == (4, 4)
    assert TCO_pred.shape == (bsz, 4, 4)
    assert points.dim() == 3 and points.shape[0] == bsz and points.shape[-1] == 3
    require_device(TCO_possible_gt, TCO_pred, points)
    gt, pred, pts = _f32(TCO_possible_gt), _f32(TCO_pred), _f32(points)
    loss = torch.empty(bsz, device=gt.device)
    assign = torch.empty(bsz, 4, 4, device=gt.device)
    check(lib().cosy_loss_co_symmetric(ptr(gt), ptr(pred), ptr(pts), None, bsz, gt.shape[1], pts.shape[1], ptr(loss), None,
                                       ptr(assign), stream()))
    return loss, assign


def loss_refiner_CO_disentangled(TCO_possible_gt, TCO_input, refiner_outputs, K_crop, points):
    """Forward value of the refiner's disentangled loss (cosypose_ops.py:49-82): (B,)."""
    bsz = TCO_possible_gt.shape[0]
    assert TCO_input.shape == (bsz, 4, 4) and refiner_outputs.shape == (bsz, 9) and K_crop.shape == (bsz, 3, 3)
    assert points.dim() == 3 and points.shape[0] == bsz and points.shape[-1] == 3
    assert TCO_possible_gt.dim() == 4 and TCO_possible_gt.shape[-2:] == (4, 4)
    require_device(TCO_possible_gt, TCO_input, refiner_outputs, K_crop, points)
    gt, Ti, out9, K, pts = (_f32(t) for t in (TCO_possible_gt, TCO_input, refiner_outputs, K_crop, points))
    loss = torch.empty(bsz, device=gt.device)
    check(lib().cosy_loss_refiner_disentangled(ptr(gt), ptr(Ti), ptr(out9), ptr(K), ptr(pts), None, bsz, gt.shape[1], pts.shape[1],
                                               ptr(loss), stream()))
    return loss
