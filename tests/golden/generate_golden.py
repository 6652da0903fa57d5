#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (ylabbe/cosypose,
mounted read-only at /root/reference) on seeded synthetic inputs.

Run in the build container only:   python tests/golden/generate_golden.py
The GPU box never sees /root/reference; it only sees the .npz files written here.
Inputs are regenerated from seeds by cosypose_amd.synthetic on both sides, so the
fixtures hold (almost) only the reference's OUTPUTS.

Stubs needed to import the reference here:
  * pinocchio / eigenpy / transforms3d / trimesh: imported by cosypose.lib3d at module
    load, never executed on the hot path -> empty stub modules;
  * cosypose.config: asserts a local_data/ dir and $CONDA_PREFIX -> stub with the 3
    path constants pose.py imports;
  * torchvision.ops.roi_align: third-party, not installed.  Bound to the oracle's C
    restatement of torchvision 0.4.2 (oracle/cosy_oracle.c).  This is the one piece of
    hot-path arithmetic the reference cannot supply here: crops are PARITY UNPINNED.
No reference source is copied: the reference is imported and executed in place.
"""
import argparse
import os
import sys
import types
import pathlib

HERE = pathlib.Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = pathlib.Path('/root/reference')
sys.dont_write_bytecode = True
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / 'oracle'))

import numpy as np
import torch

import cosy_oracle as O
from cosypose_amd import synthetic as syn


def install_stubs():
    for name in ('pinocchio', 'eigenpy', 'transforms3d', 'transforms3d.euler', 'trimesh', 'simplejson',
                 'torchnet', 'colorama'):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules['eigenpy'].switchToNumpyArray = lambda: None
    sys.modules['transforms3d'].euler = sys.modules['transforms3d.euler']
    tv = types.ModuleType('torchvision'); ops = types.ModuleType('torchvision.ops')

    def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1):
        assert spatial_scale == 1.0
        out = O.roi_align(input.detach().cpu().numpy(), boxes.detach().cpu().numpy(), output_size, sampling_ratio)
        return torch.from_numpy(out)
    ops.roi_align = roi_align
    tv.ops = ops
    sys.modules['torchvision'] = tv; sys.modules['torchvision.ops'] = ops
    cfg = types.ModuleType('cosypose.config')
    cfg.PROJECT_DIR = REF; cfg.LOCAL_DATA_DIR = pathlib.Path('/tmp'); cfg.DEBUG_DATA_DIR = pathlib.Path('/tmp')
    sys.modules['cosypose.config'] = cfg
    sys.path.insert(0, str(REF))


class FakeRenderer:
    """renderer.render contract (bullet_batch_renderer.py:46-90): float (B,3,H,W) in [0,1]."""

    def __init__(self, seed):
        self.seed = seed
        self.calls = 0

    def render(self, obj_infos, TCO, K, resolution):
        r = syn.make_renders(self.seed + self.calls, len(obj_infos), *resolution)
        self.calls += 1
        return torch.from_numpy(r)


def build_ref_model(sd_np, mesh_points, labels):
    import argparse as ap
    from cosypose.training.pose_models_cfg import create_model_pose, check_update_config
    from cosypose.lib3d.rigid_mesh_database import BatchedMeshes
    n_obj = len(labels)
    infos = {l: dict(label=l, n_points=mesh_points.shape[1], n_sym=1) for l in labels}
    sym = torch.eye(4).reshape(1, 1, 4, 4).repeat(n_obj, 1, 1, 1)
    mesh_db = BatchedMeshes(infos, labels, torch.from_numpy(mesh_points), sym).float()
    cfg = check_update_config(ap.Namespace(backbone_str='efficientnet-b3', n_pose_dims=9))
    model = create_model_pose(cfg, renderer=FakeRenderer(0), mesh_db=mesh_db)
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    assert not missing.unexpected_keys
    assert all(k.endswith('num_batches_tracked') for k in missing.missing_keys), missing.missing_keys
    model.eval()
    model.cfg = cfg
    return model, mesh_db


def g_functions(out):
    """Per-function goldens: SURVEY 8a rows a-3, a-4 (boxes), a-5, a-9, a-10, a-11 inits."""
    from cosypose.lib3d.camera_geometry import project_points_robust, boxes_from_uv, get_K_crop_resize
    from cosypose.lib3d.cropping import deepim_crops_robust
    from cosypose.lib3d.rotations import compute_rotation_matrix_from_ortho6d
    from cosypose.lib3d.cosypose_ops import (apply_imagespace_predictions, TCO_init_from_boxes,
                                             TCO_init_from_boxes_zup_autodepth)
    B, P, h, w = 6, 2000, 480, 640
    pts = syn.make_mesh_points(11, B, P)
    K = syn.make_K(B, h, w); K[:, 0, 0] *= 1.01  # fx != fy
    TCO = syn.make_TCO(12, B)
    TCO[4, 2, 3] = 0.05   # some points behind z_min -> exercises the clamp
    TCO[5, :2, 3] = (0.9, -0.6)  # projects far outside the frame
    images = syn.make_frames(13, B, h, w)
    t = lambda a: torch.from_numpy(np.asarray(a))
    uv = project_points_robust(t(pts), t(K), t(TCO))
    boxes = boxes_from_uv(uv)
    boxes_crop, crops = deepim_crops_robust(images=t(images), obs_boxes=boxes, K=t(K), TCO_pred=t(TCO),
                                            O_vertices=t(pts), output_size=(240, 320), lamb=1.4)
    K_crop = get_K_crop_resize(K=t(K).clone(), boxes=boxes_crop, orig_size=(h, w), crop_resize=(240, 320))
    K_crop_sq = get_K_crop_resize(K=t(K).clone(), boxes=boxes_crop, orig_size=(h, w), crop_resize=(256, 256))
    rs = np.random.RandomState(14)
    pose9 = rs.normal(size=(B, 9)).astype(np.float32); pose9[:, 8] = rs.uniform(0.8, 1.2, B)
    dR = compute_rotation_matrix_from_ortho6d(t(pose9[:, :6]))
    TCO_out = apply_imagespace_predictions(t(TCO), K_crop, t(pose9[:, 6:9]), dR)
    _, _, det_boxes = syn.make_detections(15, B, 1, 1, h, w)
    T0 = TCO_init_from_boxes(z_range=(1.0, 1.0), boxes=t(det_boxes), K=t(K))
    T1 = TCO_init_from_boxes_zup_autodepth(t(det_boxes), t(pts), t(K))
    out.update(fn_uv_sample=uv.numpy()[:, ::97], fn_boxes_rend=boxes.numpy(), fn_boxes_crop=boxes_crop.numpy(),
               fn_K_crop=K_crop.numpy(), fn_K_crop_sq=K_crop_sq.numpy(), fn_pose9=pose9, fn_dR=dR.numpy(),
               fn_TCO_out=TCO_out.numpy(), fn_det_boxes=det_boxes, fn_TCO_init_v0=T0.numpy(),
               fn_TCO_init_zup=T1.numpy(), fn_TCO=TCO, fn_K=K,
               fn_crops_sample=crops.numpy()[:, :, ::7, ::11])  # via the stubbed roi_align (unpinned)


def g_backbone(out, model, sd_np):
    """a-7/a-8: per-stage probes, features and pose of the reference backbone."""
    for name, (H, W), seed in (('240x320', (240, 320), 21), ('256x256', (256, 256), 22)):
        x = np.random.RandomState(seed).random_sample((2, 6, H, W)).astype(np.float32)
        stages = []
        bb = model.backbone
        hooks = [bb._bn0.register_forward_hook(lambda m, i, o: None)]
        with torch.no_grad():
            xt = torch.from_numpy(x)
            y = bb._swish(bb._bn0(bb._conv_stem(xt)))
            stages.append(y)
            ends = {1, 4, 7, 12, 17, 23, 25}
            for idx, block in enumerate(bb._blocks):
                y = block(y, drop_connect_rate=0.2 * idx / len(bb._blocks))
                if idx in ends:
                    stages.append(y)
            y = bb._swish(bb._bn1(bb._conv_head(y)))
            stages.append(y)
            feat = y.flatten(2).mean(-1)
            whole = model.net_forward(xt)['pose']
            pose = model.pose_fc(feat)
            assert torch.equal(whole, pose)
        for h_ in hooks:
            h_.remove()
        taps = np.stack([O.taps_from_stage_tensors([s[b].numpy() for s in stages]) for b in range(2)])
        out[f'bb_{name}_taps'] = taps
        out[f'bb_{name}_feat'] = feat.numpy()
        out[f'bb_{name}_pose'] = pose.numpy()
        out[f'bb_{name}_shapes'] = np.array([list(s.shape[1:]) for s in stages])


def g_forward(out, model):
    """a-1: PosePredictor.forward, B in {1,3}, n_iter in {1,4}, two frame sizes."""
    cases = (('b1_n1_480', 1, 1, (480, 640), 31), ('b3_n4_480', 3, 4, (480, 640), 32), ('b3_n1_540', 3, 1, (540, 720), 33))
    n_obj = len(model.mesh_db.labels)
    for name, B, n_it, (h, w), seed in cases:
        obj, _, _ = syn.make_detections(seed, B, 1, n_obj, h, w)
        labels = model.mesh_db.labels[obj]
        images = syn.make_frames(seed + 100, B, h, w)
        K = syn.make_K(B, h, w)
        TCO = syn.make_TCO(seed + 200, B)
        model.renderer = FakeRenderer(seed * 1000)
        with torch.no_grad():
            o = model(images=torch.from_numpy(images), K=torch.from_numpy(K), labels=labels,
                      TCO=torch.from_numpy(TCO), n_iterations=n_it)
        out[f'fw_{name}_obj'] = obj
        for n in range(1, n_it + 1):
            it = o[f'iteration={n}']
            for k in ('TCO_input', 'TCO_output', 'K_crop', 'boxes_rend', 'boxes_crop'):
                out[f'fw_{name}_it{n}_{k}'] = it[k].numpy()
            out[f'fw_{name}_it{n}_pose'] = it['model_outputs']['pose'].numpy()


def g_predictor(out, model):
    """a-11: CoarseRefinePosePredictor.get_predictions, D=5 detections over 2 frames, bsz_objects=2."""
    import pandas as pd
    import cosypose.utils.tensor_collection as tc
    from cosypose.integrated.pose_predictor import CoarseRefinePosePredictor
    h, w, D, N = 480, 640, 5, 2
    n_obj = len(model.mesh_db.labels)
    obj, im, boxes = syn.make_detections(41, D, N, n_obj, h, w)
    images = syn.make_frames(42, N, h, w); K = syn.make_K(N, h, w)
    for init in ('v0', 'z-up+auto-depth'):
        model.cfg.init_method = init
        model.renderer = FakeRenderer(4300)
        infos = pd.DataFrame(dict(label=model.mesh_db.labels[obj], batch_im_id=im, score=np.linspace(1, 0.5, D)))
        det = tc.PandasTensorCollection(infos=infos, bboxes=torch.from_numpy(boxes))
        pred = CoarseRefinePosePredictor(coarse_model=model, refiner_model=model, bsz_objects=2)
        final, allp = pred.get_predictions(torch.from_numpy(images), torch.from_numpy(K), detections=det,
                                           n_coarse_iterations=1, n_refiner_iterations=2)
        tag = 'v0' if init == 'v0' else 'zup'
        out[f'cr_{tag}_keys'] = np.array(list(allp.keys()))
        out[f'cr_{tag}_final_poses'] = final.poses.numpy()
        out[f'cr_{tag}_final_labels'] = np.array(list(final.infos['label']))
        for k, v in allp.items():
            kk = k.replace('/', '_').replace('=', '')
            for tname in ('poses', 'poses_input', 'K_crop', 'boxes_rend', 'boxes_crop'):
                out[f'cr_{tag}_{kk}_{tname}'] = getattr(v, tname).numpy()
    model.cfg.init_method = 'v0'
    out['cr_obj'] = obj; out['cr_im'] = im; out['cr_boxes'] = boxes


def g_cext(out):
    """a-12: scatter_argmin from the reference's own C++ (compiled into oracle/_ref/ by oracle/build_ref.sh)."""
    ref_dir = REPO / 'oracle' / '_ref'
    sys.path.insert(0, str(ref_dir))
    try:
        import cosypose_cext
    except ImportError:
        print('oracle/_ref/cosypose_cext not built (run oracle/build_ref.sh); skipping cext goldens')
        return
    rs = np.random.RandomState(51)
    n_seg, M = 37, 400
    ids = np.sort(np.concatenate([np.arange(n_seg), rs.randint(0, n_seg, M - n_seg)])).astype(np.int32)
    d = rs.uniform(0, 1, M).astype(np.float32)
    d[rs.randint(0, M, 60)] = 0.25  # ties -> first index must win
    out['cext_dists'] = d; out['cext_ids'] = ids
    out['cext_argmin'] = np.asarray(cosypose_cext.scatter_argmin(d, ids)).astype(np.int32)
    labels = [f'obj_{i % 5}' for i in range(12)]
    n_sym = {f'obj_{i}': i + 1 for i in range(5)}
    a, b = cosypose_cext.expand_ids_for_symmetry(labels, n_sym)
    out['cext_expand_ids'] = np.asarray(a).astype(np.int32); out['cext_sym_ids'] = np.asarray(b).astype(np.int32)
    out['cext_expand_nsym'] = np.array([n_sym[l] for l in labels], np.int32)


def _rand_poses(rs, n, t_scale=0.05, z=0.8):
    """random rigid transforms (n,4,4) float32: QR rotations (det +1), translations in front of the camera"""
    T = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    for i in range(n):
        q, r = np.linalg.qr(rs.randn(3, 3))
        q = q * np.sign(np.diag(r))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        T[i, :3, :3] = q.astype(np.float32)
        T[i, :3, 3] = (rs.randn(3) * t_scale + np.array([0, 0, z])).astype(np.float32)
    return T


def g_distances(out):
    """a-12 / 8f-2: symmetric distances, loss_CO_symmetric argmin, the refiner's disentangled loss (forward), ADD / ADD-S
    point distances -- outputs of the reference's lib3d/symmetric_distances.py, cosypose_ops.py and distances.py."""
    sys.path.insert(0, str(REPO / 'oracle' / '_ref'))
    import cosypose_cext  # noqa: F401  (the reference imports it at module load)
    from cosypose.lib3d.rigid_mesh_database import BatchedMeshes
    from cosypose.lib3d import symmetric_distances as sdist
    from cosypose.lib3d import cosypose_ops as cops
    from cosypose.lib3d import distances as dst
    rs = np.random.RandomState(77)
    n_obj, P, S = 4, 257, 4
    n_sym = np.array([1, 2, 4, 3], np.int32)
    labels_all = [f'obj_{i:06d}' for i in range(1, n_obj + 1)]
    pts = (rs.uniform(-1, 1, (n_obj, P, 3)) * rs.uniform(0.03, 0.12, (n_obj, 1, 3))).astype(np.float32)
    sym = np.tile(np.eye(4, dtype=np.float32), (n_obj, S, 1, 1))
    for o in range(n_obj):
        for k in range(1, n_sym[o]):                     # discrete rotations about z + a small offset: distinct symmetries
            a = 2 * np.pi * k / n_sym[o]
            sym[o, k, :3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
            sym[o, k, :3, 3] = (rs.randn(3) * 0.002).astype(np.float32)
    infos = {l: dict(label=l, n_points=P, n_sym=int(n_sym[i])) for i, l in enumerate(labels_all)}
    mesh_db = BatchedMeshes(infos, np.array(labels_all), torch.from_numpy(pts), torch.from_numpy(sym)).float()
    B = 9
    obj = rs.randint(0, n_obj, B).astype(np.int32); obj[:4] = [0, 1, 2, 3]
    labels = np.array(labels_all)[obj]
    T2 = _rand_poses(rs, B)
    T1 = T2.copy()
    for b in range(B):      # T1 = T2 . (some symmetry of the object) . small perturbation -> a non-trivial best symmetry
        k = rs.randint(0, n_sym[obj[b]])
        d = _rand_poses(rs, 1, t_scale=0.003, z=0.0)[0]
        d[:3, :3] = np.eye(3, dtype=np.float32) + 0.02 * rs.randn(3, 3).astype(np.float32)
        T1[b] = (T2[b] @ np.linalg.inv(sym[obj[b], k]) @ d).astype(np.float32)
    out['sd_pts'] = pts; out['sd_sym'] = sym; out['sd_nsym'] = n_sym; out['sd_obj'] = obj; out['sd_T1'] = T1; out['sd_T2'] = T2
    with torch.no_grad():
        d0, S0 = sdist.symmetric_distance_batched(torch.from_numpy(T1), torch.from_numpy(T2), labels, mesh_db)
        d1, S1 = sdist.symmetric_distance_batched_fast(torch.from_numpy(T1), torch.from_numpy(T2), labels, mesh_db)
    out['sd_batched_dists'] = d0.numpy(); out['sd_batched_S12'] = S0.numpy()
    out['sd_fast_dists'] = d1.numpy(); out['sd_fast_S12'] = S1.numpy()

    # loss_CO_symmetric / loss_refiner_CO_disentangled (cosypose_ops.py:34-82)
    points = torch.from_numpy(pts[obj])                                     # (B,P,3)
    gt = torch.from_numpy(T2).unsqueeze(1) @ torch.from_numpy(sym[obj])      # (B,S,4,4) possible GTs = T . S_k
    pred = torch.from_numpy(T1)
    with torch.no_grad():
        loss, assign = cops.loss_CO_symmetric(gt, pred, points)
    out['lc_gt'] = gt.numpy(); out['lc_pred'] = pred.numpy(); out['lc_loss'] = loss.numpy(); out['lc_assign'] = assign.numpy()
    refiner_outputs = torch.from_numpy((rs.randn(B, 9) * 0.1 + np.array([1, 0, 0, 0, 1, 0, 0, 0, 1])).astype(np.float32))
    K_crop = torch.from_numpy(np.tile(np.array([[600., 0, 160], [0, 610., 120], [0, 0, 1]], np.float32), (B, 1, 1)))
    with torch.no_grad():
        ld = cops.loss_refiner_CO_disentangled(gt, pred, refiner_outputs, K_crop, points)
    out['lc_refiner_outputs'] = refiner_outputs.numpy(); out['lc_K_crop'] = K_crop.numpy(); out['lc_disentangled'] = ld.numpy()

    # ADD / ADD-S point distances (distances.py:5-21)
    with torch.no_grad():
        da = dst.dists_add(pred, torch.from_numpy(T2), points)
        ds = dst.dists_add_symmetric(pred, torch.from_numpy(T2), points)
    out['add_dists'] = da.numpy(); out['adds_dists'] = ds.numpy()



class _Meter:
    def __init__(self): self.vals = []
    def add(self, v): self.vals.append(v)


def g_training(out):
    """a-13: one training step of the refiner as the reference runs it (train_pose.py:317-331): h_pose forward in train
    mode (batch-statistics BatchNorm; drop_connect forced to 0 for determinism), disentangled loss, backward,
    clip_grad_norm_(0.5), Adam(lr=3e-4)."""
    import types as _t
    from collections import defaultdict
    from cosypose.training import pose_forward_loss as pfl
    pfl.cast = lambda x: x
    sd = syn.golden_state_dict(0)
    n_obj = 21
    labels_all = np.array([f'obj_{i:06d}' for i in range(1, n_obj + 1)])
    mesh_points = syn.make_mesh_points(7, n_obj, 2500)
    model, mesh_db = build_ref_model(sd, mesh_points, labels_all)
    model.renderer = FakeRenderer(900)
    model.train()
    model.backbone._global_params = model.backbone._global_params._replace(drop_connect_rate=0.0)
    B = 4
    frames, K, TCO, obj = syn.make_training_batch(61, B)
    labels = labels_all[obj]
    pts = torch.from_numpy(mesh_points[obj])
    from cosypose.lib3d.camera_geometry import project_points, boxes_from_uv
    uv = project_points(pts, torch.from_numpy(K), torch.from_numpy(TCO))
    bboxes = boxes_from_uv(uv)
    data = _t.SimpleNamespace(images=torch.from_numpy(frames), K=torch.from_numpy(K), TCO=torch.from_numpy(TCO),
                              objects=[dict(name=l) for l in labels], bboxes=bboxes)
    cfg = argparse.Namespace(n_points_loss=600, loss_disentangled=True, n_pose_dims=9, init_method='v0')
    captured = []
    hook = model.pose_fc.register_forward_hook(lambda m, i, o: captured.append(o.detach().clone()))
    np.random.seed(123)                       # sample_points(deterministic=False) draws from the global numpy RNG
    out['tr_point_ids'] = np.random.RandomState(123).choice(2500, size=600, replace=False)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=0.)
    opt.zero_grad()
    loss = pfl.h_pose(model=model, mesh_db=mesh_db, data=data, meters=defaultdict(_Meter), cfg=cfg, n_iterations=1,
                      input_generator='fixed')
    loss.backward()
    hook.remove()
    out['tr_bboxes'] = bboxes.numpy(); out['tr_loss'] = np.float32(loss.item()); out['tr_pose'] = captured[0].numpy()
    names = [n for n, p in model.named_parameters()]
    out['tr_param_names'] = np.array(names)
    out['tr_grad_norms'] = np.array([p.grad.norm().item() for n, p in model.named_parameters()], np.float32)
    for n, p in model.named_parameters():
        if p.numel() <= 14000:
            out['tr_grad/' + n] = p.grad.numpy().copy()
    bn = dict(model.named_buffers())
    for n in ('backbone._bn0', 'backbone._blocks.0._bn1', 'backbone._blocks.10._bn0', 'backbone._blocks.25._bn2', 'backbone._bn1'):
        out[f'tr_bn/{n}.running_mean'] = bn[n + '.running_mean'].numpy().copy()
        out[f'tr_bn/{n}.running_var'] = bn[n + '.running_var'].numpy().copy()
    total = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=0.5, norm_type=2)
    out['tr_total_grad_norm'] = np.float32(float(total))
    opt.step()
    for n, p in model.named_parameters():
        if p.numel() <= 2000:
            out['tr_after_adam/' + n] = p.detach().numpy().copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=str(HERE / 'reference_golden.npz'))
    ap.add_argument('--dist-out', default=str(HERE / 'reference_golden_dist.npz'))
    ap.add_argument('--only-dist', action='store_true', help='only (re)generate the distance-op fixtures')
    ap.add_argument('--train-out', default=str(HERE / 'reference_golden_train.npz'))
    ap.add_argument('--only-train', action='store_true', help='only (re)generate the training-step fixtures')
    args = ap.parse_args()
    assert REF.exists(), 'reference checkout not mounted; goldens can only be generated in the build container'
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if not args.only_dist:
        tr = {}
        g_training(tr)
        np.savez_compressed(args.train_out, **tr)
        print('wrote', args.train_out, os.path.getsize(args.train_out) / 1e6, 'MB,', len(tr), 'arrays')
        if args.only_train:
            return
    dist = {}
    g_distances(dist)
    np.savez_compressed(args.dist_out, **dist)
    print('wrote', args.dist_out, os.path.getsize(args.dist_out) / 1e6, 'MB,', len(dist), 'arrays')
    if args.only_dist:
        return
    out = {}
    sd = syn.golden_state_dict(0)
    n_obj = 21
    labels = np.array([f'obj_{i:06d}' for i in range(1, n_obj + 1)])
    mesh_points = syn.make_mesh_points(7, n_obj, 2500)
    model, mesh_db = build_ref_model(sd, mesh_points, labels)
    # the ids sample_points(2000, deterministic=True) draws (mesh_ops.py:31-41) for Nmax=2500
    ref_sd = model.state_dict()
    out['sd_keys'] = np.array(list(ref_sd.keys()))
    out['sd_shapes'] = np.array([list(v.shape) + [-1] * (4 - v.dim()) for v in ref_sd.values()])
    out['sample_ids_2500'] = np.random.RandomState(0).choice(2500, size=2000, replace=False)
    g_functions(out)
    g_backbone(out, model, sd)
    g_forward(out, model)
    g_predictor(out, model)
    g_cext(out)
    np.savez_compressed(args.out, **out)
    print('wrote', args.out, os.path.getsize(args.out) / 1e6, 'MB,', len(out), 'arrays')


if __name__ == '__main__':
    main()
