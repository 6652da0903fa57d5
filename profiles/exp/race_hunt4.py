"""rasteriser on one stream while backbone forwards run on two others: is render() (rgb + depth) reproducible?  Which part differs?"""
import os, sys, argparse
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from cosypose_amd import synthetic as syn
from cosypose_amd.rasterizer import RenderMeshes, HipBatchRenderer
from cosypose_amd.efficientnet import NetEngine
from cosypose_amd._lib import lib, check, ptr, stream
from cosypose_amd.pose_models_cfg import create_model_pose, check_update_config
dev = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to('cuda', dt)
labels = np.array([f'obj_{i:06d}' for i in range(1, 6)])
v, f, c = syn.make_render_meshes(7, 5)
print('faces per mesh', [len(x) for x in f], 'verts', [len(x) for x in v])
meshes = RenderMeshes(labels, v, f, c).cuda()
renderer = HipBatchRenderer(meshes)
B, H, W = 32, 240, 320
obj = np.random.RandomState(0).randint(0, 5, B)
infos = [dict(name=labels[o]) for o in obj]
TCO = dev(syn.make_TCO(11, B, z_range=(0.5, 1.0), xy=0.05))
K = dev(np.tile(np.array([[520., 0, 158.3], [0, 515., 121.7], [0, 0, 1]], np.float32), (B, 1, 1)))
cfg = check_update_config(argparse.Namespace(backbone_str='efficientnet-b3', n_pose_dims=9))
m = create_model_pose(cfg, None, None)
m.load_state_dict({k: torch.from_numpy(vv) for k, vv in syn.golden_state_dict(0).items()}, strict=False)
m = m.cuda().eval()
engines = [NetEngine(m.backbone, m.pose_fc) for _ in range(2)]
x = torch.rand(B, 6, H, W, device='cuda')
def fwd(e):
    h = e.ensure(B, H, W, os.environ.get('DT', 'fp16'), x.device)
    pose = torch.empty(B, 9, device='cuda')
    check(lib().cosy_effnet_b3_set_input_nchw(h, ptr(x), B, stream()))
    check(lib().cosy_effnet_b3_forward(h, B, None, ptr(pose), None, stream()))
    return pose
for e in engines: fwd(e)
import cosypose_amd.rasterizer as rz
if os.environ.get('FILL'):           # outputs pre-filled with a marker: a stale (never written / not visible) output value shows up as the marker
    _empty = torch.empty
    class _T:
        def __getattr__(self, k): return getattr(torch, k)
        def empty(self, *a, **k):
            if k.get('dtype') == torch.uint8: return torch.empty(*a, **k)
            return torch.full(a if not isinstance(a[0], (tuple, list)) else tuple(a[0]), -7.0, **k)
    rz.torch = _T()
rgb0, d0 = renderer.render(infos, TCO, K, resolution=(H, W), render_depth=True)
torch.cuda.synchronize()
lanes = [torch.cuda.Stream() for _ in range(3)]
bad_rgb = bad_d = 0
for rnd in range(int(os.environ.get('ROUNDS', 40))):
    outs = []
    for l in lanes: l.wait_stream(torch.cuda.current_stream())
    for rep in range(3):
        with torch.cuda.stream(lanes[1]): fwd(engines[0])
        with torch.cuda.stream(lanes[2]): fwd(engines[1])
        with torch.cuda.stream(lanes[0]):
            for _ in range(4):
                outs.append(renderer.render(infos, TCO, K, resolution=(H, W), render_depth=True))
    torch.cuda.synchronize()
    for j, (rgb, d) in enumerate(outs):
        if not torch.equal(d, d0):
            bad_d += 1
            if bad_d <= 4:
                dd = d != d0
                rel = ((d[dd] - d0[dd]).abs() / d0[dd].abs().clamp_min(1e-9))
                print(f'   relative depth differences: min {float(rel.min()):.2e} median {float(rel.median()):.2e} max {float(rel.max()):.2e}')
                print(f'round {rnd} render {j}: DEPTH differs in {int(dd.sum())} pixels of samples {[int(r) for r in torch.nonzero(dd.flatten(1).any(1)).flatten()][:6]}; marker there: {int((d[dd] == -7).sum())}, got==0 there: {int((d[dd] == 0).sum())}, want==0 there: {int((d0[dd] == 0).sum())}')
        elif not torch.equal(rgb, rgb0):
            bad_rgb += 1
            if bad_rgb <= 4:
                print(f'round {rnd} render {j}: depth equal, RGB differs in {int((rgb != rgb0).sum())} values; of them marker (-7): {int((rgb[rgb != rgb0] == -7).sum())}')
print('renders with differing depth:', bad_d, ' with equal depth but differing rgb:', bad_rgb, 'of', 12 * int(os.environ.get('ROUNDS', 40)))
