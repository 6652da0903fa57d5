"""Microbenchmark of the crop + pack kernel at the headline shapes (256 crops of 256x256 out of 16 frames 512x512, boxes as the
loop produces them): COSY_TUNE_LIB=1 COSY_CROP_DBG=<mask> python profiles/exp/crop_bench.py"""
import os, sys
import numpy as np, torch, pandas as pd
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from cosypose_amd import synthetic as syn, tensor_collection as tc, _lib
from cosypose_amd.mesh_db import BatchedMeshes
from cosypose_amd.pose_predictor import CoarseRefinePosePredictor
from cosypose_amd._lib import lib, check, ptr, stream

H = W = 256; D = 256; n_obj = 21
labels = np.array([f'obj_{i:06d}' for i in range(1, n_obj + 1)])
pts = syn.make_mesh_points(7, n_obj, 2500)
infos = {l: dict(label=l, n_points=2500, n_sym=1) for l in labels}
mesh_db = BatchedMeshes(infos, labels, torch.from_numpy(pts), torch.eye(4).reshape(1, 1, 4, 4).repeat(n_obj, 1, 1, 1)).float().cuda()
frames, K, det = bench.make_scene(syn, torch, tc, pd, labels, 1, D, 16, 512, 512, n_obj)
g = torch.Generator(device='cuda'); g.manual_seed(1)
renders = [torch.rand(D, 3, H, W, device='cuda', generator=g) for _ in range(5)]
rend = bench.SyntheticRenderer(renders)
coarse = bench.build_model(0, mesh_db, (H, W), 'fp16', rend); refiner = bench.build_model(1, mesh_db, (H, W), 'fp16', rend)
pred = CoarseRefinePosePredictor(coarse_model=coarse, refiner_model=refiner, bsz_objects=D)
final, allp = pred.get_predictions(frames, K, detections=det, n_coarse_iterations=1, n_refiner_iterations=4)
for key in ('coarse/iteration=1', 'refiner/iteration=4'):
    boxes = allp[key].boxes_crop.contiguous()
    wh = (boxes[:, 2:] - boxes[:, :2]).cpu().numpy()
    im_ids = _lib.ints_to_device(det.infos['batch_im_id'].values, 'cuda')
    n_im, _, h, w = frames.shape
    frames4 = torch.empty(n_im, h, w, 4, device='cuda')
    check(lib().cosy_frames_to_nhwc4(ptr(frames.contiguous()), ptr(frames4), n_im, h, w, stream()))
    net = refiner._net(D, torch.device('cuda'))
    r = renders[0]
    for _ in range(3):
        check(lib().cosy_crop_pack(net, ptr(frames4), ptr(im_ids), ptr(boxes), ptr(r), D, n_im, h, w, stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        check(lib().cosy_crop_pack(net, ptr(frames4), ptr(im_ids), ptr(boxes), ptr(r), D, n_im, h, w, stream()))
    e1.record(); e1.synchronize()
    print(f"{key}: box w/h median {np.median(wh[:, 0]):.0f}x{np.median(wh[:, 1]):.0f} max {wh.max():.0f}; crop_pack (taps + pack) "
          f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us  dbg={os.environ.get('COSY_CROP_DBG', '0')}", flush=True)
