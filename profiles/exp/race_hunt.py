"""stress of test_refinement_loop_with_on_device_renderer's concurrent part: which rows / fields / iterations differ (GPU box)"""
import argparse, os, sys
import numpy as np, torch, pandas as pd
sys.path.insert(0, os.getcwd())
from cosypose_amd import synthetic as syn, tensor_collection as tc
from cosypose_amd.mesh_db import BatchedMeshes
from cosypose_amd.pose_models_cfg import create_model_pose, check_update_config
from cosypose_amd.pose_predictor import CoarseRefinePosePredictor
from cosypose_amd.rasterizer import RenderMeshes, HipBatchRenderer
MODE = sys.argv[1] if len(sys.argv) > 1 else 'plain'
dev = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to('cuda', dt)
labels = np.array([f'obj_{i:06d}' for i in range(1, 6)])
v, f, c = syn.make_render_meshes(7, 5)
meshes = RenderMeshes(labels, v, f, c).cuda()
renderer = HipBatchRenderer(meshes)
if os.environ.get('FORCE_STREAMS') == '1':
    renderer.concurrent_streams_safe = True      # take the concurrent path although the class declines it
pts = np.stack([vv[np.random.RandomState(0).choice(len(vv), 2500, replace=len(vv) < 2500)] for vv in v])
mesh_db = BatchedMeshes({l: dict(label=l, n_sym=1) for l in labels}, labels, torch.from_numpy(pts), torch.eye(4).reshape(1, 1, 4, 4).repeat(5, 1, 1, 1)).float().cuda()
cfg = check_update_config(argparse.Namespace(backbone_str='efficientnet-b3', n_pose_dims=9))
m = create_model_pose(cfg, renderer, mesh_db)
m.load_state_dict({k: torch.from_numpy(vv) for k, vv in syn.golden_state_dict(0).items()}, strict=False)
m.cfg = cfg
m = m.cuda().eval()
if os.environ.get('RS'):
    m.render_size = tuple(int(v) for v in os.environ['RS'].split('x'))
if os.environ.get('SYN') == '2':       # the HIP rasteriser behind the reference's interface only (render -> images -> cosy_crop_pack)
    class RenderOnly:
        def __init__(self, r): self.r = r
        def render(self, **kw): return self.r.render(**kw)
    m.renderer = RenderOnly(renderer)
if os.environ.get('SYN') == '3':       # the rasteriser RUNS (its kernels share the chip with the other lanes) but its images are discarded: a constant image goes in
    class RunAndDiscard:
        def __init__(self, r, H, W): self.r, self.img = r, torch.rand(1, 3, H, W, device='cuda')
        def render(self, **kw):
            self.r.render(**kw)
            return self.img.expand(len(kw['obj_infos']), -1, -1, -1)
    m.renderer = RunAndDiscard(renderer, *m.render_size)
REC = None
if os.environ.get('SYN') == '5':       # record what the rasteriser is given and what it returns, per call
    class Recorder:
        def __init__(self, r): self.r = r
        def render(self, **kw):
            rgb = self.r.render(**kw)
            if REC is not None:
                REC.setdefault(tuple(o['name'] for o in kw['obj_infos']), []).append((kw['TCO'].clone(), kw['K'].clone(), rgb.clone()))
            return rgb
    m.renderer = Recorder(renderer)
if os.environ.get('SYN') == '1':
    class Fixed:
        def __init__(self, H, W): self.img = torch.rand(256, 3, H, W, device='cuda')
        def render(self, obj_infos, TCO, K, resolution): return self.img[:1].expand(len(obj_infos), -1, -1, -1)
    m.renderer = Fixed(*m.render_size)
images, K = dev(syn.make_frames(3, 2, 480, 640)), dev(syn.make_K(2, 480, 640))
NDET, CH = int(os.environ.get('NDET', 7)), int(os.environ.get('CH', 2))
obj, im, boxes = syn.make_detections(5, NDET, 2, 5, 480, 640)
det = tc.PandasTensorCollection(infos=pd.DataFrame(dict(label=labels[obj], batch_im_id=im, score=1.0)), bboxes=dev(boxes))
pred = CoarseRefinePosePredictor(coarse_model=m, refiner_model=m, bsz_objects=CH if os.environ.get('SYN') == '5' else max(4, 2 * CH))
pred3 = CoarseRefinePosePredictor(coarse_model=m, refiner_model=m, bsz_objects=CH, n_streams=3)
bad = 0
if MODE == 'load':
    # the SEQUENTIAL predictor (one stream, one engine) while an unrelated kernel stream keeps the chip busy: is a forward's result load-dependent?
    side = torch.cuda.Stream()
    A = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)
    E_ = torch.randn(64 << 20, device='cuda')
    for dtype in ('fp32', 'fp16'):
        m.compute_dtype = dtype
        want, want_all = pred.get_predictions(images, K, detections=det, n_coarse_iterations=1, n_refiner_iterations=2)
        torch.cuda.synchronize()
        for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
            with torch.cuda.stream(side):
                for _ in range(12):
                    (A @ A); E_.mul_(1.0001).sigmoid_()
            got, got_all = pred.get_predictions(images, K, detections=det, n_coarse_iterations=1, n_refiner_iterations=2)
            torch.cuda.synchronize()
            for k in want_all:
                for tname in ('poses',):
                    a, b = getattr(got_all[k], tname), getattr(want_all[k], tname)
                    if not torch.equal(a, b):
                        rows = [int(r) for r in torch.nonzero((a != b).flatten(1).any(1)).flatten()]
                        print(f'load: round {rnd} {dtype}: {k}.{tname} rows {rows} maxdiff {float((a - b).abs().max()):.3e}')
                        bad += 1
    print('mismatches under load:', bad)
    sys.exit(0)
for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    for dtype in ('fp16',):
        m.compute_dtype = dtype
        REC = {} if os.environ.get('SYN') == '5' else None
        want, want_all = pred.get_predictions(images, K, detections=det, n_coarse_iterations=1, n_refiner_iterations=2)
        REC_WANT = REC
        if MODE == 'sync':
            torch.cuda.synchronize()
        for it in range(2 if MODE != 'steady' else 6):
            REC = {} if os.environ.get('SYN') == '5' else None
            got, got_all = pred3.get_predictions(images, K, detections=det, n_coarse_iterations=1, n_refiner_iterations=2)
            torch.cuda.synchronize()
            if REC is not None:
                for key, calls in REC.items():
                    for j, ((t1, k1, r1), (t0, k0, r0)) in enumerate(zip(calls, REC_WANT[key])):
                        same_in = torch.equal(t1, t0) and torch.equal(k1, k0)
                        if same_in and not torch.equal(r1, r0):
                            d = (r1 != r0)
                            print(f'round {rnd} call {it}: RASTERISER differs on identical input, render #{j} of its chunk: {int(d.sum())} values in samples {[int(x) for x in torch.nonzero(d.flatten(1).any(1)).flatten()][:6]} maxdiff {float((r1 - r0).abs().max()):.3e}')
                            bad += 1
                        if not same_in:
                            break
            for k in want_all:
                for tname in ('poses_input', 'K_crop', 'boxes_rend', 'boxes_crop', 'poses'):
                    a, b = getattr(got_all[k], tname), getattr(want_all[k], tname)
                    if not torch.equal(a, b):
                        rows = [int(r) for r in torch.nonzero((a != b).flatten(1).any(1)).flatten()]
                        if tname == 'poses': print(f'round {rnd} {dtype} call {it}: {k}.{tname} rows {rows} maxdiff {float((a - b).abs().max()):.3e}')
                        bad += 1
print('mismatches:', bad)
