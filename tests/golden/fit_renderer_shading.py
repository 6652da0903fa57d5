#!/usr/bin/env python3
"""Fit the shading constants of the on-device rasteriser's 'opengl' model (cosypose_amd/rasterizer.py: OPENGL_LIKE) to renders
of PyBullet's own pipeline -- the renderer the reference's checkpoints were trained against
(cosypose/rendering/bullet_scene_renderer.py:38-60, cosypose/simulator/camera.py:9-33,80-92).

PyBullet is a third-party renderer whose shader constants are not part of the reference, so the rasteriser's pixel values are
PARITY UNPINNED until this has been run on a box that has PyBullet:

    python tests/golden/fit_renderer_shading.py --out shading_fit.json        # needs `import pybullet`

It renders a handful of synthetic vertex-coloured meshes at seeded poses with PyBullet (TinyRenderer by default,
--hardware-opengl for the EGL path the reference uses), renders the same scenes with the rasteriser's scalar CPU twin
(oracle/cosy_oracle.c: cosy_oracle_rasterize_ex -- identical arithmetic to the HIP kernels, bit for bit on depth / faces), and
minimises the mean squared colour difference over the pixels both renderers cover, in (ambient, diffuse, specular, shininess,
light direction).  Paste the result into OPENGL_LIKE.

What is tested WITHOUT PyBullet (tests/test_host_logic.py: test_shading_fit_recovers_known_constants): the fitter itself -- targets
produced by the twin with hidden constants are recovered.  What is NOT tested in the build image (no PyBullet there): `render_pybullet`.
Camera conventions: pixel i spans [i, i+1) in K coordinates; OpenGL camera looks down -z, so view = diag(1,-1,-1,1) @ TCO; the
projection is the calibrated-camera matrix for (K, near, far) with the image origin at the top-left.
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'oracle'))

PARAMS = ('ambient', 'diffuse', 'specular', 'shininess', 'light_theta', 'light_phi')


def light_dir(theta, phi):
    return np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])


def vertex_normals(v, f):
    n = np.zeros_like(v, dtype=np.float64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    return (n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-12)).astype(np.float32)


def render_twin(scene, p):
    """the rasteriser's CPU twin with the 'opengl' structure (smooth normals, object-frame light, 8-bit output) -> (rgb, mask)"""
    import cosy_oracle as O
    rgb, depth, _ = O.rasterize(scene['verts'], scene['colors'], scene['faces'], scene['n_faces'], scene['obj'], scene['TCO'], scene['K'],
                                scene['H'], scene['W'], ambient=p['ambient'], diffuse=p['diffuse'], specular=p['specular'], shininess=p['shininess'],
                                light_dir=tuple(light_dir(p['light_theta'], p['light_phi'])), normals=scene['normals'], light_frame=1, smooth=1,
                                quantize=1)
    return rgb, depth > 0


def make_scene(seed=0, n_obj=3, n_poses=6, H=64, W=64):
    from cosypose_amd import synthetic as syn
    v, f, c = syn.make_render_meshes(seed, n_obj, n_lat=12, n_lon=16)
    V, F = max(len(x) for x in v), max(len(x) for x in f)
    verts = np.zeros((n_obj, V, 3), np.float32); colors = np.zeros((n_obj, V, 3), np.float32); faces = np.zeros((n_obj, F, 3), np.int32)
    normals = np.zeros((n_obj, V, 3), np.float32)
    for i in range(n_obj):
        verts[i, :len(v[i])] = v[i]; colors[i, :len(v[i])] = c[i]; faces[i, :len(f[i])] = f[i]; normals[i, :len(v[i])] = vertex_normals(v[i], f[i])
    TCO = syn.make_TCO(seed + 1, n_poses, z_range=(0.35, 0.6), xy=0.03)
    K = np.tile(np.array([[W * 1.6, 0, W / 2 - 0.7], [0, W * 1.6, H / 2 + 0.4], [0, 0, 1]], np.float32), (n_poses, 1, 1))
    return dict(verts=verts, colors=colors, faces=faces, normals=normals, n_faces=np.array([len(x) for x in f], np.int32),
                obj=(np.arange(n_poses) % n_obj).astype(np.int32), TCO=TCO, K=K, H=H, W=W, mesh_lists=(v, f, c))


def fit(scene, target_rgb, target_mask, x0=None, maxiter=400):
    """least squares over the commonly covered pixels; Nelder-Mead (the 8-bit output makes the loss piecewise constant)"""
    from scipy.optimize import minimize
    x0 = np.array(x0 if x0 is not None else [0.4, 0.6, 0.05, 16.0, 0.3, 0.0], np.float64)
    lo = np.array([0.0, 0.0, 0.0, 1.0, 0.0, -np.pi]); hi = np.array([1.5, 1.5, 1.0, 128.0, np.pi, np.pi])

    def loss(x):
        x = np.clip(x, lo, hi)
        rgb, mask = render_twin(scene, dict(zip(PARAMS, x)))
        both = mask & target_mask
        if both.sum() == 0:
            return 1e3
        d = (rgb - target_rgb)[np.broadcast_to(both[:, None], rgb.shape)]
        return float(np.mean(d * d))
    best = None
    for start in (x0, x0 * [1, 1, 1, 2, 1, 1] + [0, 0, 0.1, 0, 0.5, 1.0]):
        r = minimize(loss, start, method='Nelder-Mead', options=dict(maxiter=maxiter, xatol=1e-3, fatol=1e-9,
                                                                     initial_simplex=None))
        if best is None or r.fun < best.fun:
            best = r
    x = np.clip(best.x, lo, hi)
    out = dict(zip(PARAMS, (float(v) for v in x)))
    out['light_dir'] = [float(v) for v in light_dir(out['light_theta'], out['light_phi'])]
    out['mse'] = float(best.fun)
    return out


def proj_from_K(K, h, w, near, far):
    """OpenGL projection (column-major list) of a calibrated camera: x right, y down in the image, origin top-left"""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    P = np.array([[2 * fx / w, 0, 1 - 2 * cx / w, 0],
                  [0, 2 * fy / h, 2 * cy / h - 1, 0],
                  [0, 0, -(far + near) / (far - near), -2 * far * near / (far - near)],
                  [0, 0, -1, 0]], np.float64)
    return P.T.reshape(-1).tolist()


def render_pybullet(scene, hardware_opengl=False):
    """the same scenes through PyBullet.  NOT exercised in the build image (PyBullet is not installed there)."""
    import pybullet as pb
    cid = pb.connect(pb.DIRECT)
    v_l, f_l, c_l = scene['mesh_lists']
    bodies = []
    tmp = tempfile.mkdtemp()
    for i, (v, f, c) in enumerate(zip(v_l, f_l, c_l)):
        path = os.path.join(tmp, f'obj_{i}.obj')
        with open(path, 'w') as fh:                       # OBJ with per-vertex colours (v x y z r g b), as meshlab writes them
            for p, col in zip(v, c):
                fh.write('v %.7f %.7f %.7f %.4f %.4f %.4f\n' % (*p, *col))
            for t in f:
                fh.write('f %d %d %d\n' % (t[0] + 1, t[1] + 1, t[2] + 1))
        vis = pb.createVisualShape(pb.GEOM_MESH, fileName=path, physicsClientId=cid)
        bodies.append(pb.createMultiBody(baseVisualShapeIndex=vis, basePosition=[0, 0, -100 - i], physicsClientId=cid))
    H, W = scene['H'], scene['W']
    rgbs, masks = [], []
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    for n in range(len(scene['TCO'])):
        for i, b in enumerate(bodies):                    # the posed object at the world origin, the others out of sight
            pb.resetBasePositionAndOrientation(b, [0, 0, 0] if i == scene['obj'][n] else [0, 0, -100 - i], [0, 0, 0, 1], physicsClientId=cid)
        view = (flip @ scene['TCO'][n].astype(np.float64)).T.reshape(-1).tolist()
        _, _, rgba, _, seg = pb.getCameraImage(W, H, viewMatrix=view, projectionMatrix=proj_from_K(scene['K'][n], H, W, 0.01, 10.0),
                                               renderer=pb.ER_BULLET_HARDWARE_OPENGL if hardware_opengl else pb.ER_TINY_RENDERER,
                                               shadow=0, physicsClientId=cid)
        rgba = np.asarray(rgba, np.uint8).reshape(H, W, 4)
        seg = np.asarray(seg).reshape(H, W)
        rgbs.append(rgba[..., :3].transpose(2, 0, 1).astype(np.float32) / 255.0)
        masks.append(seg >= 0)
    pb.disconnect(cid)
    return np.stack(rgbs), np.stack(masks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='shading_fit.json')
    ap.add_argument('--hardware-opengl', action='store_true', help="PyBullet's EGL/OpenGL renderer (what the reference uses) instead of TinyRenderer")
    ap.add_argument('--resolution', type=int, default=96)
    a = ap.parse_args()
    try:
        import pybullet  # noqa: F401
    except ImportError:
        raise SystemExit('PyBullet is not installed: run this on a box that has it (pip install pybullet)')
    scene = make_scene(0, H=a.resolution, W=a.resolution)
    rgb, mask = render_pybullet(scene, a.hardware_opengl)
    res = fit(scene, rgb, mask)
    json.dump(res, open(a.out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
