"""Python face of the parity oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package `cosypose_amd` never does.

Three things live here:
  * ctypes bindings to oracle/cosy_oracle.c (plain-C fp32 restatement of the
    reference's hot path, each function citing the reference file:line);
  * `roi_align_numpy`: a second, independently written restatement of
    torchvision 0.4.2 roi_align used to cross-check the C one (the reference's
    only third-party arithmetic on the path; PARITY UNPINNED, see the C header);
  * `TorchRef`: the same path written with stock torch CPU ops (F.conv2d,
    F.batch_norm ...) -- the arithmetic the reference's PyTorch-CPU path
    executes -- used to time the CPU baseline and to validate the C restatement
    at full size.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libcosy_oracle.so')
_SRC = os.path.join(_HERE, 'cosy_oracle.c')


_SO_SAN = os.path.join(os.path.dirname(_SO), 'libcosy_oracle_san.so')


def build(force=False, sanitize=False):
    """gcc-compile the C restatement into oracle/_build/ (idempotent).  sanitize=True: the AddressSanitizer +
    UndefinedBehaviorSanitizer build (tests/test_oracle_sanitizers.py runs the golden tests on it in a subprocess with libasan
    preloaded; COSY_ORACLE_SANITIZE=1 makes lib() load it)."""
    so = _SO_SAN if sanitize else _SO
    if not force and os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(_SRC):
        return so
    os.makedirs(os.path.dirname(so), exist_ok=True)
    opt = ['-O1', '-g', '-fno-omit-frame-pointer', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined'] if sanitize else ['-O3']
    cmd = ['gcc'] + opt + ['-march=x86-64-v2', '-fopenmp', '-fPIC', '-shared', '-fno-fast-math', '-ffp-contract=off', '-o', so, _SRC, '-lm']
    subprocess.check_call(cmd)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        san = os.environ.get('COSY_ORACLE_SANITIZE') == '1'
        so = build(sanitize=san)
        _lib = ctypes.CDLL(so)
        _lib.cosy_oracle_b3_param_count.restype = ctypes.c_long
        _lib.cosy_oracle_b3_forward.restype = ctypes.c_int
        _lib.cosy_oracle_scatter_argmin.restype = ctypes.c_int
        _lib.cosy_oracle_expand_ids_for_symmetry.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def set_threads(n):
    lib().cosy_oracle_set_threads(ctypes.c_int(int(n)))


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------

def project_points_robust(pts, K, TCO, z_min=0.1):
    pts, pp = _f(pts); K, kp = _f(K); TCO, tp = _f(TCO)
    B, P = pts.shape[:2]
    uv = np.empty((B, P, 2), np.float32)
    lib().cosy_oracle_project_points_robust(pp, kp, tp, B, P, ctypes.c_float(z_min), uv.ctypes.data_as(ctypes.c_void_p))
    return uv


def boxes_from_uv(uv):
    uv, up = _f(uv)
    B, P = uv.shape[:2]
    out = np.empty((B, 4), np.float32)
    lib().cosy_oracle_boxes_from_uv(up, B, P, out.ctypes.data_as(ctypes.c_void_p))
    return out


def get_K_crop_resize(K, boxes, crop_resize):
    K, kp = _f(K); boxes, bp = _f(boxes)
    out = np.empty_like(K)
    lib().cosy_oracle_get_K_crop_resize(kp, bp, K.shape[0], int(crop_resize[0]), int(crop_resize[1]),
                                        out.ctypes.data_as(ctypes.c_void_p))
    return out


def crop_geometry(pts, K, TCO, im_hw, out_hw, lamb=1.4):
    """boxes_rend, boxes_crop, K_crop of PosePredictor.crop_inputs (pose.py:45-67)."""
    pts, pp = _f(pts); K, kp = _f(K); TCO, tp = _f(TCO)
    B, P = pts.shape[:2]
    br = np.empty((B, 4), np.float32); bc = np.empty((B, 4), np.float32); kc = np.empty((B, 3, 3), np.float32)
    lib().cosy_oracle_crop_geometry(pp, kp, tp, B, P, int(im_hw[0]), int(im_hw[1]), int(out_hw[0]), int(out_hw[1]),
                                    ctypes.c_float(lamb), br.ctypes.data_as(ctypes.c_void_p),
                                    bc.ctypes.data_as(ctypes.c_void_p), kc.ctypes.data_as(ctypes.c_void_p))
    return br, bc, kc


def roi_align(images, rois, output_size, sampling_ratio=4):
    """torchvision-0.4.2 semantics.  images (N,C,h,w), rois (R,5) -> (R,C,PH,PW)."""
    images, ip = _f(images); rois, rp = _f(rois)
    N, C, h, w = images.shape
    R = rois.shape[0]
    PH, PW = int(output_size[0]), int(output_size[1])
    out = np.empty((R, C, PH, PW), np.float32)
    lib().cosy_oracle_roi_align(ip, N, C, h, w, rp, R, PH, PW, int(sampling_ratio), out.ctypes.data_as(ctypes.c_void_p))
    return out


def roi_align_numpy(images, rois, output_size, sampling_ratio=4):
    """Independent vectorised restatement of the same spec (fp32 step by step)."""
    images = np.asarray(images, np.float32); rois = np.asarray(rois, np.float32)
    N, C, h, w = images.shape
    PH, PW = output_size
    g = sampling_ratio
    out = np.zeros((rois.shape[0], C, PH, PW), np.float32)
    f32 = np.float32
    for n, roi in enumerate(rois):
        bi = int(roi[0]); x1, y1, x2, y2 = roi[1:]
        rw = max(f32(x2 - x1), f32(1)); rh = max(f32(y2 - y1), f32(1))
        bh = f32(rh / f32(PH)); bw = f32(rw / f32(PW))
        ph = np.arange(PH, dtype=np.float32)[:, None]; iy = np.arange(g, dtype=np.float32)[None, :]
        pw = np.arange(PW, dtype=np.float32)[:, None]; ix = np.arange(g, dtype=np.float32)[None, :]
        ys = ((y1 + ph * bh).astype(np.float32) + ((iy + f32(.5)) * bh).astype(np.float32) / f32(g)).astype(np.float32).reshape(-1)
        xs = ((x1 + pw * bw).astype(np.float32) + ((ix + f32(.5)) * bw).astype(np.float32) / f32(g)).astype(np.float32).reshape(-1)

        def axis(v, size):
            valid = ~((v < -1.0) | (v > size))
            v = np.where(v <= 0, f32(0), v).astype(np.float32)
            lo = v.astype(np.int32)
            edge = lo >= size - 1
            lo = np.where(edge, size - 1, lo); hi = np.where(edge, size - 1, lo + 1)
            v = np.where(edge, lo.astype(np.float32), v)
            l = (v - lo.astype(np.float32)).astype(np.float32); hh = (f32(1) - l).astype(np.float32)
            return valid, lo, hi, l, hh
        vy, yl, yh, ly, hy = axis(ys, h)
        vx, xl, xh, lx, hx = axis(xs, w)
        img = images[bi]
        w1 = (hy[:, None] * hx[None, :]).astype(np.float32); w2 = (hy[:, None] * lx[None, :]).astype(np.float32)
        w3 = (ly[:, None] * hx[None, :]).astype(np.float32); w4 = (ly[:, None] * lx[None, :]).astype(np.float32)
        valid = (vy[:, None] & vx[None, :]).astype(np.float32)
        val = (w1 * img[:, yl][:, :, xl] + w2 * img[:, yl][:, :, xh] + w3 * img[:, yh][:, :, xl] + w4 * img[:, yh][:, :, xh])
        val = (val * valid).astype(np.float32).reshape(C, PH, g, PW, g)
        # sequential fp32 accumulation in (iy, ix) order, as the C++ loop does
        acc = np.zeros((C, PH, PW), np.float32)
        for a in range(g):
            for b in range(g):
                acc = (acc + val[:, :, a, :, b]).astype(np.float32)
        out[n] = acc / f32(g * g)
    return out


def update_pose(TCO, K_crop, pose9):
    TCO, tp = _f(TCO); K_crop, kp = _f(K_crop); pose9, pp = _f(pose9)
    out = np.empty_like(TCO)
    lib().cosy_oracle_update_pose(tp, kp, pp, TCO.shape[0], out.ctypes.data_as(ctypes.c_void_p))
    return out


def ortho6d_to_R(p6):
    p6, pp = _f(p6)
    out = np.empty((p6.shape[0], 3, 3), np.float32)
    lib().cosy_oracle_ortho6d_to_R(pp, p6.shape[0], out.ctypes.data_as(ctypes.c_void_p))
    return out


def tco_init_from_boxes(boxes, K, z=1.0):
    boxes, bp = _f(boxes); K, kp = _f(K)
    out = np.empty((boxes.shape[0], 4, 4), np.float32)
    lib().cosy_oracle_tco_init_from_boxes(bp, kp, boxes.shape[0], ctypes.c_float(z), out.ctypes.data_as(ctypes.c_void_p))
    return out


def tco_init_zup_autodepth(boxes, pts, K):
    boxes, bp = _f(boxes); K, kp = _f(K); pts, pp = _f(pts)
    out = np.empty((boxes.shape[0], 4, 4), np.float32)
    lib().cosy_oracle_tco_init_zup_autodepth(bp, pp, kp, boxes.shape[0], pts.shape[1], out.ctypes.data_as(ctypes.c_void_p))
    return out


def scatter_argmin(dists, ids, n_seg):
    dists, dp = _f(dists); ids, ip = _i(ids)
    out = np.empty(n_seg, np.int32)
    rc = lib().cosy_oracle_scatter_argmin(dp, ip, dists.shape[0], out.ctypes.data_as(ctypes.c_void_p), int(n_seg))
    assert rc == 0
    return out


def expand_ids_for_symmetry(n_sym_item):
    n, npp = _i(n_sym_item)
    M = int(n.sum())
    a = np.empty(M, np.int32); b = np.empty(M, np.int32)
    m = lib().cosy_oracle_expand_ids_for_symmetry(npp, n.shape[0], a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p))
    assert m == M
    return a, b


def symmetric_distance(T1, T2, obj, pts, sym, n_sym, fast=False):
    """-> (min_dists (B,), best_sym (B,) int32, S12 (B,4,4)); fast=False: symmetric_distance_batched, True: ..._fast"""
    T1, t1 = _f(T1); T2, t2 = _f(T2); obj, op = _i(obj); pts, pp = _f(pts); sym, sp = _f(sym); n_sym, npp = _i(n_sym)
    B, P, S = T1.shape[0], pts.shape[1], sym.shape[1]
    d = np.empty(B, np.float32); best = np.empty(B, np.int32); S12 = np.empty((B, 4, 4), np.float32)
    lib().cosy_oracle_symmetric_distance(t1, t2, op, pp, sp, npp, B, P, S, int(bool(fast)), d.ctypes.data_as(ctypes.c_void_p),
                                         best.ctypes.data_as(ctypes.c_void_p), S12.ctypes.data_as(ctypes.c_void_p))
    return d, best, S12


def loss_co_symmetric(gt, pred, points):
    gt, gp = _f(gt); pred, pp = _f(pred); points, ptp = _f(points)
    B, S, P = gt.shape[0], gt.shape[1], points.shape[1]
    loss = np.empty(B, np.float32); mid = np.empty(B, np.int32); assign = np.empty((B, 4, 4), np.float32)
    lib().cosy_oracle_loss_co_symmetric(gp, pp, ptp, B, S, P, loss.ctypes.data_as(ctypes.c_void_p), mid.ctypes.data_as(ctypes.c_void_p),
                                        assign.ctypes.data_as(ctypes.c_void_p))
    return loss, mid, assign


def loss_refiner_disentangled(gt, TCO_in, out9, K_crop, points):
    gt, gp = _f(gt); TCO_in, tp = _f(TCO_in); out9, op = _f(out9); K_crop, kp = _f(K_crop); points, ptp = _f(points)
    B, S, P = gt.shape[0], gt.shape[1], points.shape[1]
    loss = np.empty(B, np.float32)
    lib().cosy_oracle_loss_refiner_disentangled(gp, tp, op, kp, ptp, B, S, P, loss.ctypes.data_as(ctypes.c_void_p))
    return loss


def dists_add(pred, gt, points, symmetric=False):
    pred, pp = _f(pred); gt, gp = _f(gt); points, ptp = _f(points)
    B, P = points.shape[:2]
    out = np.empty((B, P, 3), np.float32)
    lib().cosy_oracle_dists_add(pp, gp, ptp, B, P, int(bool(symmetric)), out.ctypes.data_as(ctypes.c_void_p))
    return out


def rasterize(verts, colors, faces, n_faces, obj, TCO, K, H, W, ambient=0.6, diffuse=0.4, light_dir=(0., 0., -1.), normals=None, uvs=None,
              tex=None, specular=0.0, shininess=1.0, light_frame=0, smooth=0, quantize=0):
    """CPU twin of the HIP mesh rasteriser -> rgb (B,3,H,W), depth (B,H,W), zbuf (B,H,W) uint64.  tex: (n_obj,TH,TW,4)."""
    verts, vp = _f(verts); colors, cp = _f(colors); faces, fp = _i(faces); n_faces, nfp = _i(n_faces); obj, op = _i(obj)
    TCO, tp = _f(TCO); K, kp = _f(K)
    shade, sp = _f(np.array([ambient, diffuse, specular, shininess, *light_dir, light_frame, smooth, quantize], np.float32))
    nrm, uv, tx, TH, TW = None, None, None, 0, 0
    if normals is not None:
        normals, nrm = _f(normals)
    if uvs is not None and tex is not None:
        uvs, uv = _f(uvs); tex, tx = _f(tex); TH, TW = tex.shape[1], tex.shape[2]
    B, V, F = TCO.shape[0], verts.shape[1], faces.shape[1]
    rgb = np.empty((B, 3, H, W), np.float32); depth = np.empty((B, H, W), np.float32); zb = np.empty((B, H, W), np.uint64)
    lib().cosy_oracle_rasterize_ex(vp, cp, nrm, uv, tx, TH, TW, fp, nfp, op, tp, kp, B, V, F, H, W, sp,
                                   rgb.ctypes.data_as(ctypes.c_void_p), depth.ctypes.data_as(ctypes.c_void_p), zb.ctypes.data_as(ctypes.c_void_p))
    return rgb, depth, zb


# ----------------------------------------------------------------------------
# EfficientNet-B3: parameter blob + forward
# ----------------------------------------------------------------------------

# (k, s, expand, cin, cout) -- SURVEY Appendix A / efficientnet_utils.py:259-264 scaled for B3
B3_BLOCKS = [
    (3, 1, 1, 40, 24), (3, 1, 1, 24, 24),
    (3, 2, 6, 24, 32), (3, 1, 6, 32, 32), (3, 1, 6, 32, 32),
    (5, 2, 6, 32, 48), (5, 1, 6, 48, 48), (5, 1, 6, 48, 48),
    (3, 2, 6, 48, 96), (3, 1, 6, 96, 96), (3, 1, 6, 96, 96), (3, 1, 6, 96, 96), (3, 1, 6, 96, 96),
    (5, 1, 6, 96, 136), (5, 1, 6, 136, 136), (5, 1, 6, 136, 136), (5, 1, 6, 136, 136), (5, 1, 6, 136, 136),
    (5, 2, 6, 136, 232), (5, 1, 6, 232, 232), (5, 1, 6, 232, 232), (5, 1, 6, 232, 232), (5, 1, 6, 232, 232), (5, 1, 6, 232, 232),
    (3, 1, 6, 232, 384), (3, 1, 6, 384, 384),
]


def state_dict_keys(prefix='backbone.'):
    """Reference state_dict keys in flat-blob order (BN: weight,bias,running_mean,running_var)."""
    bn = ('weight', 'bias', 'running_mean', 'running_var')
    keys = [prefix + '_conv_stem.weight'] + [prefix + '_bn0.' + s for s in bn]
    for i, (k, s, e, cin, cout) in enumerate(B3_BLOCKS):
        p = f'{prefix}_blocks.{i}.'
        if e != 1:
            keys += [p + '_expand_conv.weight'] + [p + '_bn0.' + s for s in bn]
        keys += [p + '_depthwise_conv.weight'] + [p + '_bn1.' + s for s in bn]
        keys += [p + '_se_reduce.weight', p + '_se_reduce.bias', p + '_se_expand.weight', p + '_se_expand.bias']
        keys += [p + '_project_conv.weight'] + [p + '_bn2.' + s for s in bn]
    keys += [prefix + '_conv_head.weight'] + [prefix + '_bn1.' + s for s in bn]
    keys += ['pose_fc.weight', 'pose_fc.bias']
    return keys


def flatten_state_dict(sd):
    """PosePredictor state_dict (numpy or torch values) -> flat fp32 blob."""
    parts = []
    for k in state_dict_keys():
        v = sd[k]
        v = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
        parts.append(v.astype(np.float32).reshape(-1))
    blob = np.concatenate(parts)
    assert blob.size == lib().cosy_oracle_b3_param_count(), (blob.size, lib().cosy_oracle_b3_param_count())
    return blob


def b3_out_hw(H, W):
    oh, ow = ctypes.c_int(), ctypes.c_int()
    lib().cosy_oracle_b3_out_hw(int(H), int(W), ctypes.byref(oh), ctypes.byref(ow))
    return oh.value, ow.value


def b3_forward(x, blob, want_taps=False):
    """x (B,6,H,W) NCHW fp32 -> feat (B,1536), pose (B,9) [, taps (B,9,16)]."""
    x, xp = _f(x); blob, bp = _f(blob)
    B, C, H, W = x.shape
    assert C == 6
    feat = np.empty((B, 1536), np.float32); pose = np.empty((B, 9), np.float32)
    taps = np.empty((B, 9, 16), np.float32) if want_taps else None
    rc = lib().cosy_oracle_b3_forward(xp, B, H, W, bp, feat.ctypes.data_as(ctypes.c_void_p),
                                      pose.ctypes.data_as(ctypes.c_void_p),
                                      taps.ctypes.data_as(ctypes.c_void_p) if want_taps else None)
    assert rc == 0, rc
    return (feat, pose, taps) if want_taps else (feat, pose)


def taps_from_stage_tensors(tensors):
    """Same probe as the C forward's `taps`, from 9 stage tensors (each (C,H,W) numpy)."""
    out = np.empty((9, 16), np.float32)
    for i, t in enumerate(tensors):
        f = np.asarray(t, np.float32).reshape(-1)
        n = f.size
        out[i, 0] = np.float32(f.astype(np.float64).sum() / n)
        out[i, 1] = np.float32(np.abs(f.astype(np.float64)).sum() / n)
        for q in range(14):
            out[i, 2 + q] = f[(q * 2 + 1) * n // 29]
    return out


# ----------------------------------------------------------------------------
# Whole loop: PosePredictor.forward (pose.py:89-132) on the C restatement
# ----------------------------------------------------------------------------

def pose_predictor_forward(images, K, obj_ids, TCO, points_table, blob, render_fn, n_iterations=1,
                           render_size=(240, 320), backbone=None):
    """images (B,3,h,w) already gathered per object (pose_predictor.py:41), points_table
    (n_obj,2000,3) = mesh_db points after sample_points(2000, deterministic=True).
    render_fn(n, TCO_input, K_crop) -> (B,3,H,W) in [0,1].  `backbone(x)->(feat,pose)`
    defaults to the C forward; pass TorchRef(...).net_forward for speed."""
    images = np.asarray(images, np.float32)
    B, _, h, w = images.shape
    pts = np.asarray(points_table, np.float32)[np.asarray(obj_ids)]
    outputs = {}
    TCO_in = np.asarray(TCO, np.float32)
    for n in range(n_iterations):
        br, bc, kc = crop_geometry(pts, K, TCO_in, (h, w), render_size)
        rois = np.concatenate([np.arange(B, dtype=np.float32)[:, None], bc], 1)
        crop = roi_align(images, rois, render_size, 4)
        rend = np.asarray(render_fn(n, TCO_in, kc), np.float32)
        x = np.concatenate([crop, rend], 1)
        _, pose = (backbone(x) if backbone is not None else b3_forward(x, blob))
        TCO_out = update_pose(TCO_in, kc, pose)
        outputs[f'iteration={n + 1}'] = dict(TCO_input=TCO_in, TCO_output=TCO_out, K_crop=kc, pose=pose,
                                             boxes_rend=br, boxes_crop=bc, images_crop=crop)
        TCO_in = TCO_out
    return outputs


class TorchRef:
    """The backbone + head on stock torch CPU ops, from a reference-keyed state_dict.

    Mirrors EfficientNet.extract_features (efficientnet.py:174-190), MBConvBlock.forward
    (:71-98), Conv2dStaticSamePadding(image_size=300) (efficientnet_utils.py:123-146) and
    PosePredictor.net_forward (pose.py:81-87), eval mode."""

    def __init__(self, sd, dtype=None):
        import torch
        self.torch = torch
        # private copies: train mode updates the running statistics in place and must not touch the caller's arrays
        self.sd = {k: (v if hasattr(v, 'detach') else torch.from_numpy(np.asarray(v))).detach().float().clone()
                   for k, v in sd.items() if not k.endswith('num_batches_tracked')}

    @staticmethod
    def _pad(k, s):
        if s == 1:
            return (k - 1) // 2, (k - 1) // 2
        tot = k - 2
        return tot // 2, tot - tot // 2

    def _conv(self, x, w, k, s, groups=1, bias=None):
        F = self.torch.nn.functional
        lo, hi = self._pad(k, s)
        if lo or hi:
            x = F.pad(x, (lo, hi, lo, hi))
        return F.conv2d(x, w, bias, stride=s, groups=groups)

    train = False        # True: batch-statistics BatchNorm, running stats updated in place (momentum 0.01)
    drop = None          # train mode: {block index: (keep_prob, (B,) 0/1 mask)} for drop_connect (efficientnet_utils.py:83-92)

    def _bn(self, x, p):
        F = self.torch.nn.functional
        sd = self.sd
        return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                            self.train, 0.01, 1e-3)

    def extract_features(self, x, stages=None):
        torch = self.torch; sd = self.sd
        sw = lambda t: t * torch.sigmoid(t)
        x = sw(self._bn(self._conv(x, sd['backbone._conv_stem.weight'], 3, 2), 'backbone._bn0'))
        if stages is not None:
            stages.append(x)
        ends = {1, 4, 7, 12, 17, 23, 25}
        for i, (k, s, e, cin, cout) in enumerate(B3_BLOCKS):
            p = f'backbone._blocks.{i}.'
            inp = x
            if e != 1:
                x = sw(self._bn(self._conv(x, sd[p + '_expand_conv.weight'], 1, 1), p + '_bn0'))
            x = sw(self._bn(self._conv(x, sd[p + '_depthwise_conv.weight'], k, s, groups=x.shape[1]), p + '_bn1'))
            q = x.mean((2, 3), keepdim=True)
            q = self._conv(sw(self._conv(q, sd[p + '_se_reduce.weight'], 1, 1, bias=sd[p + '_se_reduce.bias'])),
                           sd[p + '_se_expand.weight'], 1, 1, bias=sd[p + '_se_expand.bias'])
            x = torch.sigmoid(q) * x
            x = self._bn(self._conv(x, sd[p + '_project_conv.weight'], 1, 1), p + '_bn2')
            if s == 1 and cin == cout:
                if self.train and self.drop and i in self.drop:
                    keep, mask = self.drop[i]
                    x = x / keep * mask.reshape(-1, 1, 1, 1)
                x = x + inp
            if stages is not None and i in ends:
                stages.append(x)
        x = sw(self._bn(self._conv(x, sd['backbone._conv_head.weight'], 1, 1), 'backbone._bn1'))
        if stages is not None:
            stages.append(x)
        return x

    def net_forward(self, x):
        torch = self.torch
        with torch.no_grad():
            x = torch.as_tensor(np.asarray(x, np.float32)) if not torch.is_tensor(x) else x
            f = self.extract_features(x).flatten(2).mean(-1)
            pose = torch.nn.functional.linear(f, self.sd['pose_fc.weight'], self.sd['pose_fc.bias'])
        return f.numpy(), pose.numpy()


    # ---- storage emulation (test infrastructure for the 16-bit throughput modes) ----------------------------------------
    # The same network with every value rounded to the 16-bit storage type exactly where the HIP path stores one:
    # network input, stem output, block inputs / outputs, the depthwise output D, the expanded tensor E of the UNFUSED blocks
    # (the fused fronts keep E in fp32 registers / LDS; the small kernel also folds BN0's scale into its expand weights before
    # rounding them; front kind 5 rounds E and the depthwise taps: they are the 16-bit operands of its tap MFMAs),
    # the head activation; the 1x1-conv and stem weights; and the
    # squeeze-excite gate where the project GEMM applies it -- to the weight fragments (maps with Ho*Wo % 64 == 0:
    # bf16 round(w * g), fp16 w * half(g) in half arithmetic) or to the activation rows (other maps).  Accumulation, BatchNorm,
    # SiLU, the depthwise taps, the squeeze sums (taken BEFORE D is rounded) and the SE FCs are fp32, as on the device.
    # What is left between this and the device result is fp32 summation order and the transcendental approximations,
    # i.e. ~1e-6 before rounding -> a small fraction of values one storage ulp apart.
    def _rnd(self, t, storage):
        torch = self.torch
        if storage == 'bf16':
            return t.bfloat16().float()
        if storage == 'fp16':
            return t.clamp(-65504.0, 65504.0).half().float()
        return t

    # bf16 (round 6): the 1x1-conv WEIGHTS are error-compensated pairs -- hi = bf16(w), lo = bf16(w - hi), two MFMAs per fragment against the same
    # activation fragment, fp32 accumulation -- so what multiplies the activations is hi + lo (16 significant bits); the depthwise taps and the expanded
    # values of the matrix-pipe fronts (kinds 5 / 6) are fp16 operands in EVERY 16-bit mode (the small MFMA's operands are a register format, not storage).
    hilo = True          # False: bf16 weights as single 8-bit values (the library before round 6)

    def _rndw(self, t, storage):
        if storage == 'bf16' and self.hilo:
            hi = t.bfloat16().float()
            return hi + (t - hi).bfloat16().float()
        return self._rnd(t, storage)

    def _rnd_mx(self, t, storage):
        """operand format of the depthwise MFMAs (front kinds 5 / 6): fp16, saturating, for both 16-bit storage types"""
        return self._rnd(t, 'fp16' if (storage == 'bf16' and self.hilo) else storage)

    def stem_emulated(self, x, storage, round_output=True):
        """x (B,6,H,W) fp32 (rounded to the storage type here, as the crop kernel does) -> stem output.  round_output=False: the stem tensor as
        the fused stem + block-0 front holds it (kernels_stem.hip: fp32 rows in registers, never stored)"""
        R = lambda t: self._rnd(t, storage)
        sw = lambda t: t * self.torch.sigmoid(t)
        y = sw(self._bn(self._conv(R(x), R(self.sd['backbone._conv_stem.weight']), 3, 2), 'backbone._bn0'))
        return R(y) if round_output else y

    def block_emulated(self, i, x, storage, fused, gate_on_weights=None):
        """MBConv block i on a block input that is already in the storage type -> (D, gate (B,Cmid), block output).
        gate_on_weights: where the device's project GEMM applies the SE gate (cosy_effnet_b3_block_info); default = the tile
        kernel's rule (maps of a multiple of 64 pixels: weights, else activation rows)."""
        torch = self.torch; sd = self.sd
        R = lambda t: self._rnd(t, storage)
        sw = lambda t: t * torch.sigmoid(t)
        k, s, e, cin, cout = B3_BLOCKS[i]
        p = f'backbone._blocks.{i}.'
        inp = x
        if e != 1 and fused in (2, 6):
            # small kernel (front kind 2, mbconv_small_kernel): BN0's scale times log2(e) is folded into the expand weights
            # BEFORE they are rounded to the storage type, its bias enters as the MFMA's C operand, the expanded tensor stays fp32
            # (effnet.hip: build_weights).  t = log2(e) * BN0(expand); the device evaluates silu as (t / (1 + 2^-t)) * ln 2.
            g64 = sd[p + '_bn0.weight'].double() / torch.sqrt(sd[p + '_bn0.running_var'].double() + 1e-3)
            s0 = g64.float().double()
            b0 = (sd[p + '_bn0.bias'].double() - sd[p + '_bn0.running_mean'].double() * g64).float().double()
            L2E = 1.4426950408889634
            We = self._rndw((sd[p + '_expand_conv.weight'].double() * (s0 * L2E)[:, None, None, None]).float(), storage)
            t = self._conv(x, We, 1, 1) + (b0 * L2E).float()[None, :, None, None]
            x = sw(t * 0.6931471805599453)
            if fused == 6:                                       # the small kernel's matrix-pipe form (kernels_smx.hip): E rounded to the tap MFMAs' operand type
                x = self._rnd_mx(x, storage)
        elif e != 1:
            x = sw(self._bn(self._conv(x, self._rndw(sd[p + '_expand_conv.weight'], storage), 1, 1), p + '_bn0'))
            if not fused:                                        # unfused blocks store the expanded tensor
                x = R(x)
            elif fused == 5:                                     # front kind 5 (wave kernel, taps on the matrix pipe): the small MFMA's operand
                x = self._rnd_mx(x, storage)
        dw_w = sd[p + '_depthwise_conv.weight']
        if fused in (5, 6):                                      # ... and the taps too (products exact, fp32 accumulation)
            dw_w = self._rnd_mx(dw_w, storage)
        d32 = sw(self._bn(self._conv(x, dw_w, k, s, groups=x.shape[1]), p + '_bn1'))
        q = d32.mean((2, 3), keepdim=True)
        q = self._conv(sw(self._conv(q, sd[p + '_se_reduce.weight'], 1, 1, bias=sd[p + '_se_reduce.bias'])),
                       sd[p + '_se_expand.weight'], 1, 1, bias=sd[p + '_se_expand.bias'])
        g = torch.sigmoid(q)[:, :, 0, 0]                       # (B, Cmid)
        D = R(d32)
        W = self._rndw(sd[p + '_project_conv.weight'], storage)[:, :, 0, 0]        # (Cout, Cmid)
        hw = D.shape[2] * D.shape[3]
        if gate_on_weights is None:
            gate_on_weights = hw % 64 == 0
        if gate_on_weights:                                      # gate folded into the weight fragments, per sample
            if storage == 'fp16':
                Wg = (W.half()[None] * g.half()[:, None, :]).float()
            else:
                Wg = self._rndw(W[None] * g[:, None, :], storage)      # bf16: (hi + lo) * g in fp32, re-split into a pair
            y = torch.einsum('bnk,bkhw->bnhw', Wg, D)
        else:                                                    # gate applied to the activation rows
            if storage == 'fp16':
                Dg = (D.half() * g.half()[:, :, None, None]).float()
            else:
                Dg = R(D * g[:, :, None, None])
            y = torch.einsum('nk,bkhw->bnhw', W, Dg)
        y = self._bn(y, p + '_bn2')
        if s == 1 and cin == cout:
            y = y + inp
        return D, g, R(y)

    def head_emulated(self, x, storage):
        R = lambda t: self._rnd(t, storage)
        sw = lambda t: t * self.torch.sigmoid(t)
        return R(sw(self._bn(self._conv(x, self._rndw(self.sd['backbone._conv_head.weight'], storage), 1, 1), 'backbone._bn1')))

    def extract_features_emulated(self, x, storage, fused, probes=None, gate_w=None):
        """x (B,6,H,W) fp32; storage 'bf16' | 'fp16'; fused[i] = front kernel of block i as cosy_effnet_b3_block_info reports it
        (0 unfused: E is stored; 1 wave, 2 small: E never stored; 4: block 0 behind the fused stem, the stem tensor never stored; 5 wave with the
        depthwise taps on the matrix pipe: E and the taps rounded to the storage type, never stored).
        probes: dict filled with {-1: stem, i: block output, 100+i: D of block i, 200+i: gate (B,Cmid), 26: head}.
        NOTE: two evaluations of a 26-block network that round at every layer decorrelate with depth (a value one ulp apart
        perturbs the next layer's roundings), so the END-TO-END distance between this and the device grows to the size of
        the storage type's own rounding noise; kernels are therefore checked block by block on the DEVICE's block inputs
        (tests/test_gpu_parity.py: test_fused_kernels_vs_storage_emulation), where the distance stays at isolated ulps."""
        put = (lambda k, v: probes.__setitem__(k, v.clone())) if probes is not None else (lambda k, v: None)
        x_in = x
        x = self.stem_emulated(x, storage)
        put(-1, x)
        for i in range(len(B3_BLOCKS)):
            if i == 0 and fused[0] == 4:      # block 0 behind the fused stem: the stem tensor is never stored (fp32 in registers)
                x = self.stem_emulated(x_in, storage, round_output=False)
            D, g, x = self.block_emulated(i, x, storage, fused[i], None if gate_w is None else gate_w[i])
            put(100 + i, D); put(200 + i, g); put(i, x)
        x = self.head_emulated(x, storage)
        put(26, x)
        return x

    def net_forward_emulated(self, x, storage, fused, probes=None, gate_w=None):
        torch = self.torch
        with torch.no_grad():
            x = torch.as_tensor(np.asarray(x, np.float32)) if not torch.is_tensor(x) else x
            f = self.extract_features_emulated(x, storage, fused, probes, gate_w).flatten(2).mean(-1)
            pose = torch.nn.functional.linear(f, self.sd['pose_fc.weight'], self.sd['pose_fc.bias'])
        return f.numpy(), pose.numpy()


    # ---- training step: train-mode forward, disentangled loss, backward -- torch-CPU autograd over the
    # functional restatement above; pinned against the reference's own loss / gradients (tests/golden/*train*)
    @staticmethod
    def ortho6d(p6, torch):
        """compute_rotation_matrix_from_ortho6d, lib3d/rotations.py:6-21"""
        a, b = p6[:, 0:3], p6[:, 3:6]
        x = a / torch.norm(a, p=2, dim=1, keepdim=True)
        z = torch.cross(x, b, dim=1)
        z = z / torch.norm(z, p=2, dim=1, keepdim=True)
        y = torch.cross(z, x, dim=1)
        return torch.stack((x, y, z), dim=-1)

    def disentangled_loss(self, gt, TCO_in, out9, K_crop, points):
        """loss_refiner_CO_disentangled + loss_CO_symmetric(l1), lib3d/cosypose_ops.py:34-82 -> (B,)"""
        torch = self.torch

        def tp(T, pts):   # transform_pts
            if T.dim() == 4:
                return (T[..., :3, :3].unsqueeze(2) @ pts.unsqueeze(1).unsqueeze(-1)).squeeze(-1) + T[..., :3, 3].unsqueeze(2)
            return (T[:, :3, :3].unsqueeze(1) @ pts.unsqueeze(-1)).squeeze(-1) + T[:, :3, 3].unsqueeze(1)

        def co_sym(pred):
            d = (tp(pred, points).unsqueeze(1) - tp(gt, points)).flatten(-2, -1).abs().mean(-1)
            return d.min(dim=1)[0]
        dR = self.ortho6d(out9[:, 0:6], torch)
        g0 = gt[:, 0]
        orn = g0.clone(); orn[:, :3, :3] = dR @ TCO_in[:, :3, :3]
        xy = g0.clone()
        z_gt, z_in = g0[:, 2, [3]], TCO_in[:, 2, [3]]
        fxfy = K_crop[:, [0, 1], [0, 1]]
        xy[:, :2, 3] = ((out9[:, 6:8] / fxfy) + (TCO_in[:, :2, 3] / z_in.repeat(1, 2))) * z_gt.repeat(1, 2)
        zz = g0.clone(); zz[:, [2], [3]] = out9[:, [8]] * z_in
        return co_sym(orn) + co_sym(xy) + co_sym(zz)

    def train_forward_backward(self, x, gt, TCO_in, K_crop, points, drop=None):
        """x (B,6,H,W) network input.  -> loss (float), pose (B,9), {param name: grad}.  Running BN stats in self.sd
        are updated in place, exactly as one nn.Module forward in train mode would."""
        torch = self.torch
        t = lambda a: a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a, np.float32))
        names = [k for k in self.sd if not (k.endswith('running_mean') or k.endswith('running_var'))]
        for k in names:
            self.sd[k] = self.sd[k].detach().clone().requires_grad_(True)
        self.train, self.drop = True, drop
        try:
            f = self.extract_features(t(x)).flatten(2).mean(-1)
            pose = torch.nn.functional.linear(f, self.sd['pose_fc.weight'], self.sd['pose_fc.bias'])
            loss = self.disentangled_loss(t(gt), t(TCO_in), pose, t(K_crop), t(points)).mean()
            loss.backward()
        finally:
            self.train, self.drop = False, None
        grads = {k: self.sd[k].grad.detach().numpy() for k in names}
        for k in names:
            self.sd[k] = self.sd[k].detach()
        return float(loss.item()), pose.detach().numpy(), grads
