#!/usr/bin/env python3
"""A/B of backbone schedules with the experiment build (COSY_TUNE_LIB=1, lib/libcosyhip_tune.so): the same weights and
input through two settings of the COSY_* knobs; prints the deviation of features / pose outputs between them and against
the fp32 engine.  Usage: COSY_TUNE_LIB=1 python profiles/ab_check.py "COSY_WAVE_MASK=0" "COSY_WAVE_MASK=0x3fffc" [--crop 256x256]"""
import os
import sys
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def run(envs, dtype, H, W, B, x):
    import torch
    from cosypose_amd import synthetic as syn
    from cosypose_amd.efficientnet import EfficientNet, NetEngine
    from cosypose_amd import arch
    from cosypose_amd._lib import lib, check, ptr, stream
    for kv in envs.split():
        k, v = kv.split('=')
        os.environ[k] = v
    net = EfficientNet.from_name('efficientnet-b3', in_channels=6)
    fc = torch.nn.Linear(arch.HEAD_C, 9)
    sd = {k: torch.from_numpy(v) for k, v in syn.golden_state_dict(1).items()}
    net.load_state_dict({k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}, strict=False)
    fc.load_state_dict({'weight': sd['pose_fc.weight'], 'bias': sd['pose_fc.bias']})
    net, fc = net.cuda().eval(), fc.cuda()
    eng = NetEngine(net, fc)
    h = eng.ensure(B, H, W, dtype, x.device)
    pose = torch.empty(B, 9, device=x.device); feat = torch.empty(B, arch.HEAD_C, device=x.device)
    check(lib().cosy_effnet_b3_set_input_nchw(h, ptr(x), B, stream()))
    check(lib().cosy_effnet_b3_forward(h, B, ptr(feat), ptr(pose), None, stream()))
    torch.cuda.synchronize()
    for kv in envs.split():
        os.environ.pop(kv.split('=')[0], None)
    out = feat.cpu().numpy().copy(), pose.cpu().numpy().copy()
    eng.release()
    return out


def main():
    import torch
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    crop = next((a.split('=')[1] for a in sys.argv[1:] if a.startswith('--crop=')), '256x256')
    B = int(next((a.split('=')[1] for a in sys.argv[1:] if a.startswith('--batch=')), '19'))
    H, W = (int(v) for v in crop.split('x'))
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    x = torch.rand(B, 6, H, W, device='cuda', generator=g)
    ref = run('', 'fp32', H, W, B, x)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    for dtype in ('bf16', 'fp16'):
        outs = [run(e, dtype, H, W, B, x) for e in args]
        for e, o in zip(args, outs):
            print(f'{crop} {dtype} [{e}] vs fp32: feat {rel(o[0], ref[0]):.3e} pose {rel(o[1], ref[1]):.3e}  finite={np.isfinite(o[0]).all()}')
        for e, o in zip(args[1:], outs[1:]):
            print(f'{crop} {dtype} [{e}] vs [{args[0]}]: feat {rel(o[0], outs[0][0]):.3e} pose {rel(o[1], outs[0][1]):.3e}')


if __name__ == '__main__':
    main()
