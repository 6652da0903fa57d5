"""Bit-level A/B of two builds of libcosyhip.so (COSY_TUNE_LIB=<path> selects the library of a process): dumps the depthwise output D of the given blocks, their
squeeze-excite gates and the pose head's output for seeded inputs; `--compare a.npz b.npz` lists every tensor that differs.
   COSY_TUNE_LIB=$PWD/cosypose_amd/lib/libcosyhip_dev.so python profiles/exp/ab_bits.py --out gpurun_out/x/dev.npz [--crop 256x256] [--blocks 9-17] [--B 24]"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out'); ap.add_argument('--crop', default='256x256'); ap.add_argument('--blocks', default='9-17'); ap.add_argument('--B', type=int, default=24)
    ap.add_argument('--dtypes', default='fp16,bf16')
    ap.add_argument('--compare', nargs=2)
    a = ap.parse_args()
    if a.compare:
        x, y = (dict(np.load(f)) for f in a.compare)
        bad = 0
        for k in sorted(x):
            if k not in y:
                print('missing', k); bad += 1; continue
            if x[k].shape != y[k].shape or not np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)):
                d = np.abs(x[k].astype(np.float64) - y[k].astype(np.float64))
                print(f'DIFF {k}: {int((x[k] != y[k]).sum())} of {x[k].size} values, max abs {d.max():.3e}, max |x| {np.abs(x[k]).max():.3e}, nan {int(np.isnan(y[k]).sum())}'); bad += 1
        print(f'{len(x)} tensors compared, {bad} differ')
        return 1 if bad else 0
    import ctypes
    import torch
    from cosypose_amd import synthetic as syn, arch
    from cosypose_amd._lib import lib, check, ptr, stream
    from cosypose_amd.mesh_db import BatchedMeshes
    from cosypose_amd.pose_models_cfg import create_model_pose, check_update_config
    H, W = (int(v) for v in a.crop.split('x'))
    lo, hi = (int(v) for v in a.blocks.split('-'))
    labels = np.array([f'obj_{i:06d}' for i in range(1, 22)])
    pts = syn.make_mesh_points(7, 21, 2500)
    mesh_db = BatchedMeshes({l: dict(label=l, n_points=2500, n_sym=1) for l in labels}, labels, torch.from_numpy(pts), torch.eye(4).reshape(1, 1, 4, 4).repeat(21, 1, 1, 1)).float().cuda()
    cfg = check_update_config(argparse.Namespace(backbone_str='efficientnet-b3', n_pose_dims=9))
    m = create_model_pose(cfg, None, mesh_db)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.golden_state_dict(0).items()}, strict=False)
    m = m.cuda().eval()
    m.render_size = (H, W)
    B = a.B
    x = torch.from_numpy(np.concatenate([syn.make_renders(71, B, H, W), syn.make_renders(72, B, H, W)], 1)).cuda()
    out = {}
    for dt in a.dtypes.split(','):
        m.compute_dtype = dt
        h = m._net(B, x.device)
        pose = torch.empty(B, 9, device='cuda')
        check(lib().cosy_effnet_b3_set_input_nchw(h, ptr(x), B, stream()))
        check(lib().cosy_effnet_b3_forward(h, B, None, ptr(pose), None, stream()))
        torch.cuda.synchronize()
        out[f'{dt}/pose'] = pose.cpu().numpy()
        for i in range(lo, hi + 1):
            dims = (ctypes.c_int * 11)()
            check(lib().cosy_effnet_b3_block_info(h, i, dims))
            Ho, Wo, cmid = dims[2], dims[3], dims[5]
            for layer, shape in ((100 + i, (B, cmid, Ho, Wo)), (200 + i, (B, cmid))):
                buf = torch.empty(shape, device='cuda')
                check(lib().cosy_effnet_b3_set_probe(h, layer, ptr(buf)))
                check(lib().cosy_effnet_b3_forward(h, B, None, ptr(pose), None, stream()))
                torch.cuda.synchronize()
                check(lib().cosy_effnet_b3_set_probe(h, -2, None))
                out[f'{dt}/L{layer}'] = buf.cpu().numpy()
        print(dt, 'pose[0]', out[f'{dt}/pose'][0][:4], 'finite', bool(np.isfinite(out[f'{dt}/pose']).all()))
    np.savez(a.out, **out)


if __name__ == '__main__':
    sys.exit(main())
