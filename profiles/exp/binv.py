#!/usr/bin/env python3
"""Batch-invariance probe: first 3 samples alone vs inside a batch of 256; prints per-stage tap differences."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from cosypose_amd import synthetic as syn, arch
from cosypose_amd.efficientnet import EfficientNet, NetEngine
from cosypose_amd._lib import lib, check, ptr, stream

def fwd(x, dtype='bf16'):
    net = EfficientNet.from_name('efficientnet-b3', in_channels=6)
    fc = torch.nn.Linear(arch.HEAD_C, 9)
    sd = {k: torch.from_numpy(v) for k, v in syn.golden_state_dict(1).items()}
    net.load_state_dict({k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}, strict=False)
    fc.load_state_dict({'weight': sd['pose_fc.weight'], 'bias': sd['pose_fc.bias']})
    net, fc = net.cuda().eval(), fc.cuda()
    eng = NetEngine(net, fc)
    B = x.shape[0]
    h = eng.ensure(B, x.shape[2], x.shape[3], dtype, x.device)
    pose = torch.empty(B, 9, device='cuda'); taps = torch.zeros(B, 9, 16, device='cuda')
    check(lib().cosy_effnet_b3_set_input_nchw(h, ptr(x), B, stream()))
    check(lib().cosy_effnet_b3_forward(h, B, None, ptr(pose), ptr(taps), stream()))
    torch.cuda.synchronize()
    out = taps.cpu().numpy().copy(), pose.cpu().numpy().copy()
    eng.release()
    return out

g = torch.Generator(device='cuda'); g.manual_seed(5)
x = torch.rand(256, 6, 256, 256, device='cuda', generator=g)
sel = [0, 131, 255]
for env in sys.argv[1:] or ['']:
    for kv in env.split():
        k, v = kv.split('='); os.environ[k] = v
    tb, pb = fwd(x)
    ts, ps = fwd(x[sel].contiguous())
    print(f'[{env}] pose equal: {np.array_equal(pb[sel], ps)}; per-stage tap max|diff| (stem, stage1..7, head):',
          ' '.join(f'{np.abs(tb[sel][:, i] - ts[:, i]).max():.2e}' for i in range(9)))
    for kv in env.split():
        os.environ.pop(kv.split('=')[0], None)
