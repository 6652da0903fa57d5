"""do the backbone's kernels write outside their own buffers?  canaries (small and large torch allocations) around the forward's allocations"""
import os, sys, argparse
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from cosypose_amd import synthetic as syn
from cosypose_amd.efficientnet import NetEngine
from cosypose_amd._lib import lib, check, ptr, stream
from cosypose_amd.pose_models_cfg import create_model_pose, check_update_config
B, H, W = int(os.environ.get('B', 32)), 240, 320
dtype = os.environ.get('DT', 'fp16')
cfg = check_update_config(argparse.Namespace(backbone_str='efficientnet-b3', n_pose_dims=9))
m = create_model_pose(cfg, None, None)
m.load_state_dict({k: torch.from_numpy(vv) for k, vv in syn.golden_state_dict(0).items()}, strict=False)
m = m.cuda().eval()
PAT = 1234.5
engines = [NetEngine(m.backbone, m.pose_fc) for _ in range(2)]
x = torch.rand(B, 6, H, W, device='cuda')
lanes = [torch.cuda.Stream() for _ in range(2)]
def fwd(e):
    h = e.ensure(B, H, W, dtype, x.device)
    pose = torch.empty(B, 9, device='cuda')
    check(lib().cosy_effnet_b3_set_input_nchw(h, ptr(x), B, stream()))
    check(lib().cosy_effnet_b3_forward(h, B, None, ptr(pose), None, stream()))
    return pose
for e in engines: fwd(e)            # the engines' device memory exists now: what is allocated next may sit right behind it
torch.cuda.synchronize()
small = [torch.full((n,), PAT, device='cuda') for n in ([64, 128, 288, 512, 1024] * 400)]       # 256 B ... 4 KB blocks
big = [torch.full((1 << 20,), PAT, device='cuda') for _ in range(64)] + [torch.full((8 << 20,), PAT, device='cuda') for _ in range(12)]   # 4 MB and 32 MB blocks
keep = small[::2]; del small
for rnd in range(20):
    for l in lanes: l.wait_stream(torch.cuda.current_stream())
    poses = []
    for rep in range(4):
        for l, e in zip(lanes, engines):
            with torch.cuda.stream(l): poses.append(fwd(e))
    torch.cuda.synchronize()
hit = [(i, int((t != PAT).sum())) for i, t in enumerate(keep) if not bool((t == PAT).all())]
hitb = [(i, int((t != PAT).sum())) for i, t in enumerate(big) if not bool((t == PAT).all())]
print('B', B, dtype, 'small canaries modified:', hit[:10], 'big canaries modified:', hitb[:10])
