"""On-device batch renderer with the interface of the reference's BulletBatchRenderer
(cosypose/rendering/bullet_batch_renderer.py:46-90): `render(obj_infos, TCO, K, resolution, render_depth=False)` ->
float (B,3,H,W) in [0,1] on the device [, depth (B,H,W) in metres, 0 = background].

The reference renders with PyBullet/OpenGL in worker processes and copies every image host -> device in each iteration of
the refinement loop; this one is a HIP z-buffer rasteriser (csrc/kernels_raster.hip) so the loop never leaves the GPU.
Camera model, near plane, background and the treatment of non-finite poses follow the reference; the SHADING of
PyBullet's OpenGL pipeline cannot be reproduced (third party) -- pixel values are parity-unpinned, see the kernel header.
Loading meshes from .ply/.obj (trimesh in the reference) is out of scope: meshes come as arrays.
"""
import numpy as np
import torch

from ._lib import lib, check, ptr, stream, require_device, ints_to_device


class RenderMeshes:
    """Padded triangle meshes of the object set: verts (n_obj,V,3) metres, colors (n_obj,V,3) in [0,1],
    faces (n_obj,F,3) int32, n_faces (n_obj,) -- one row per label."""

    def __init__(self, labels, verts_list, faces_list, colors_list=None):
        self.labels = np.asarray(labels)
        self.label_to_id = {l: i for i, l in enumerate(self.labels)}
        n = len(labels)
        V = max(len(v) for v in verts_list); F = max(len(f) for f in faces_list)
        verts = np.zeros((n, V, 3), np.float32); colors = np.full((n, V, 3), 0.7, np.float32); faces = np.zeros((n, F, 3), np.int32)
        for i, (v, f) in enumerate(zip(verts_list, faces_list)):
            verts[i, :len(v)] = v; faces[i, :len(f)] = f
            if colors_list is not None:
                colors[i, :len(v)] = colors_list[i]
        self.verts, self.colors, self.faces = torch.from_numpy(verts), torch.from_numpy(colors), torch.from_numpy(faces)
        self.n_faces = torch.tensor([len(f) for f in faces_list], dtype=torch.int32)

    def cuda(self):
        for k in ('verts', 'colors', 'faces', 'n_faces'):
            setattr(self, k, getattr(self, k).cuda().contiguous())
        return self


class HipBatchRenderer:
    def __init__(self, meshes, ambient=0.6, diffuse=0.4, light_dir=(0.0, 0.0, -1.0)):
        self.meshes = meshes
        self.ambient, self.diffuse = float(ambient), float(diffuse)
        l = np.asarray(light_dir, np.float64); l = l / np.linalg.norm(l)
        self.light = tuple(float(v) for v in l)
        self._scratch = None

    def render(self, obj_infos, TCO, K, resolution=(240, 320), render_depth=False):
        m = self.meshes
        require_device(m.verts, TCO, K)
        TCO = torch.as_tensor(TCO).detach().float().contiguous()
        K = torch.as_tensor(K).detach().float().contiguous()
        bsz = len(TCO)
        assert TCO.shape == (bsz, 4, 4) and K.shape == (bsz, 3, 3) and len(obj_infos) == bsz
        H, W = min(resolution), max(resolution)          # bullet_batch_renderer.py:34: images are (min(res), max(res))
        dev = TCO.device
        obj = ints_to_device(np.fromiter((m.label_to_id[o['name']] for o in obj_infos), dtype=np.int32, count=bsz), dev)
        rgb = torch.empty(bsz, 3, H, W, device=dev)
        depth = torch.empty(bsz, H, W, device=dev) if render_depth else None
        V, F = m.verts.shape[1], m.faces.shape[1]
        need = lib().cosy_render_scratch_bytes(bsz, V, H, W)
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != dev:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=dev)
        check(lib().cosy_render_meshes(ptr(m.verts), ptr(m.colors), ptr(m.faces), ptr(m.n_faces), ptr(obj), ptr(TCO), ptr(K), bsz, V, F,
                                       H, W, self.ambient, self.diffuse, *self.light, ptr(rgb), ptr(depth), ptr(self._scratch), stream()))
        return (rgb, depth) if render_depth else rgb
