#!/usr/bin/env python3
"""Golden vectors for the training input generators' pose noise: outputs of the REFERENCE's own add_noise
(cosypose/lib3d/transform_ops.py:35-51), imported from /root/reference and executed in place.

add_noise draws its Euler angles and translations from the global numpy RNG and builds the rotation with
transforms3d.euler.euler2mat (default axes 'sxyz'), a third-party package that is not installed here.  The stub bound in
its place is NOT this repository's closed form: it is scipy.spatial.transform.Rotation.from_euler('xyz', ...) (lower case =
extrinsic = static axes, the same convention), and the generator also checks it against the composition of elementary
rotations Rz(ak) @ Ry(aj) @ Rx(ai) that 'sxyz' stands for.  Everything else -- the order of the RNG draws, the float casts,
R @ R_noise, t + noise -- is the reference's code.

Run in the build container only:   python tests/golden/generate_golden_noise.py  ->  reference_golden_noise.npz
"""
import pathlib
import sys

import numpy as np
import torch

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import generate_golden as G      # the import stubs of the other generators

from scipy.spatial.transform import Rotation


def euler2mat(ai, aj, ak, axes='sxyz'):
    assert axes == 'sxyz'
    return Rotation.from_euler('xyz', [ai, aj, ak]).as_matrix()


def elementary(ai, aj, ak):
    c, s = np.cos, np.sin
    Rx = np.array([[1, 0, 0], [0, c(ai), -s(ai)], [0, s(ai), c(ai)]])
    Ry = np.array([[c(aj), 0, s(aj)], [0, 1, 0], [-s(aj), 0, c(aj)]])
    Rz = np.array([[c(ak), -s(ak), 0], [s(ak), c(ak), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def main():
    G.install_stubs()
    sys.modules['transforms3d.euler'].euler2mat = euler2mat
    rs = np.random.RandomState(3)
    for _ in range(50):                       # the stub against the definition of static-frame x-y-z angles
        a = rs.uniform(-np.pi, np.pi, 3)
        assert np.abs(euler2mat(*a) - elementary(*a)).max() < 1e-12
    from cosypose.lib3d.transform_ops import add_noise
    from cosypose_amd import synthetic as syn
    out = {}
    cases = [('gt_noise', 11, 7, [15, 15, 15], [0.01, 0.01, 0.05]), ('trans_only', 12, 5, [0, 0, 0], [0.01, 0.01, 0.05]),
             ('one', 13, 1, [15, 15, 15], [0.01, 0.01, 0.05])]
    for name, seed, B, e_std, t_std in cases:
        TCO = torch.from_numpy(syn.make_TCO(seed, B))
        np.random.seed(1000 + seed)
        res = add_noise(TCO, euler_deg_std=e_std, trans_std=t_std)
        out[f'{name}_seed'] = np.array([seed, 1000 + seed, B])
        out[f'{name}_euler_std'] = np.array(e_std, np.float64); out[f'{name}_trans_std'] = np.array(t_std, np.float64)
        out[f'{name}_out'] = res.numpy()
        # the RNG state the reference leaves behind: the port must consume exactly as many draws
        out[f'{name}_next_draw'] = np.array([np.random.normal()])
    np.savez(HERE / 'reference_golden_noise.npz', **out)
    print('wrote', HERE / 'reference_golden_noise.npz', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
