"""List the non-cosy device ops (torch elementwise / reduce / copies) of ONE training step: what is left to remove."""
import sys, os, types, argparse
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import SyntheticRenderer, build_model
from cosypose_amd import synthetic as syn, train_engine, pose_forward_loss as pfl
from cosypose_amd.mesh_db import BatchedMeshes
B, n_obj, h, w, H, W = 64, 21, 480, 640, 240, 320
labels = np.array([f'obj_{i:06d}' for i in range(1, n_obj + 1)])
pts = syn.make_mesh_points(7, n_obj, 2600)
infos = {l: dict(label=l, n_points=2600, n_sym=1) for l in labels}
mesh_db = BatchedMeshes(infos, labels, torch.from_numpy(pts), torch.eye(4).reshape(1, 1, 4, 4).repeat(n_obj, 1, 1, 1)).float().cuda()
frames, K, TCO, obj = syn.make_training_batch(100, B, n_obj, h, w)
renderer = SyntheticRenderer([torch.rand(B, 3, H, W, device='cuda') for _ in range(3)])
model = build_model(1, mesh_db, (H, W), 'fp32', renderer).train()
pin = lambda t: t.contiguous().pin_memory()
data = types.SimpleNamespace(images=pin(torch.from_numpy(frames)), K=pin(torch.from_numpy(K)), TCO=pin(torch.from_numpy(TCO)),
                             objects=[dict(name=l) for l in labels[obj]], bboxes=pin(torch.rand(B, 4) * 100 + torch.tensor([100., 100, 300, 300])))
cfg = argparse.Namespace(n_points_loss=2600, loss_disentangled=True, n_pose_dims=9, init_method='v0')
opt = train_engine.FlatAdam(model, lr=3e-4, clip_grad_norm=0.5)
class M:
    def add(self, v): pass
meters = defaultdict(M)
def step():
    opt.zero_grad()
    loss = pfl.h_pose(model=model, mesh_db=mesh_db, data=data, meters=meters, cfg=cfg, n_iterations=1, input_generator='fixed')
    loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, 'self_device_time_total', 0) or 0
    if t > 0 and 'cosy::' not in e.key:
        rows.append((t, e.count, e.key[:110]))
for t, n, k in sorted(rows, reverse=True)[:40]:
    print(f'{t:9.1f} us  x{n:4d}  {k}')
