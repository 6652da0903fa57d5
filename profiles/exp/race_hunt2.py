"""is the rasteriser (z-buffer -> shaded image / -> fused render + crop + pack) bit-reproducible when three HIP streams render at once?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from cosypose_amd import synthetic as syn
from cosypose_amd.rasterizer import RenderMeshes, HipBatchRenderer
from cosypose_amd._lib import lib, check, ptr, stream, COSY_F16, COSY_F32
dev = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to('cuda', dt)
labels = np.array([f'obj_{i:06d}' for i in range(1, 6)])
v, f, c = syn.make_render_meshes(7, 5)
meshes = RenderMeshes(labels, v, f, c).cuda()
renderer = HipBatchRenderer(meshes)
B, H, W = 32, 240, 320
sets = []
for s in range(3):
    obj = np.random.RandomState(s).randint(0, 5, B)
    TCO = dev(syn.make_TCO(11 + s, B, z_range=(0.5, 1.0), xy=0.05))
    K = dev(np.tile(np.array([[520., 0, 158.3], [0, 515., 121.7], [0, 0, 1]], np.float32), (B, 1, 1)))
    sets.append(([dict(name=labels[o]) for o in obj], TCO, K))
frames4 = torch.rand(2, 480, 640, 4, device='cuda')
im_ids = dev(np.zeros(B), torch.int32)
boxes = dev(np.tile(np.array([100., 80., 420., 320.], np.float32), (B, 1)))
def run(i, mode, dtype):
    infos, TCO, K = sets[i]
    if mode == 'render':
        rgb, depth = renderer.render(infos, TCO, K, resolution=(H, W), render_depth=True)
        return torch.cat([rgb.flatten(1), depth.flatten(1)], 1)
    x8 = torch.zeros(B, H, W, 8, device='cuda', dtype=torch.float16 if dtype == COSY_F16 else torch.float32)
    renderer.render_crop_pack(infos, TCO, K, frames4, im_ids, boxes, (H, W), x8=x8, dtype=dtype)
    return x8.float().flatten(1)
lanes = [torch.cuda.Stream() for _ in range(3)]
for mode, dtype in (('render', None), ('pack', COSY_F32), ('pack', COSY_F16)):
    want = [run(i, mode, dtype) for i in range(3)]
    torch.cuda.synchronize()
    bad = 0
    for rnd in range(60):
        got = [None] * 3
        for i, l in enumerate(lanes):
            l.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(l):
                for _ in range(3):
                    got[i] = run(i, mode, dtype)
        torch.cuda.synchronize()
        for i in range(3):
            if not torch.equal(got[i], want[i]):
                d = (got[i] != want[i])
                bad += 1
                if bad <= 4:
                    print(f'{mode} {dtype} round {rnd} lane {i}: {int(d.sum())} values differ in samples {[int(r) for r in torch.nonzero(d.any(1)).flatten()][:8]}, maxdiff {float((got[i] - want[i]).abs().max()):.3e}')
    print(mode, dtype, 'mismatching (round, lane) pairs:', bad, 'of 180')
