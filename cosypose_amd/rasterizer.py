"""On-device batch renderer with the interface of the reference's BulletBatchRenderer
(cosypose/rendering/bullet_batch_renderer.py:46-90): `render(obj_infos, TCO, K, resolution, render_depth=False)` ->
float (B,3,H,W) in [0,1] on the device [, depth (B,H,W) in metres, 0 = background].

The reference renders with PyBullet/OpenGL in worker processes and copies every image host -> device in each iteration of
the refinement loop; this one is a HIP z-buffer rasteriser (csrc/kernels_raster.hip) so the loop never leaves the GPU.
Camera model, near plane, background and the treatment of non-finite poses follow the reference; the SHADING of
PyBullet's OpenGL pipeline cannot be reproduced (third party) -- pixel values are parity-unpinned, see the kernel header.
Loading meshes from .ply/.obj (trimesh in the reference) is out of scope: meshes come as arrays.
"""
import ctypes

import numpy as np
import torch

from ._lib import lib, check, ptr, stream, require_device, ints_to_device, MeshSet, Shade


def vertex_normals(verts, faces):
    """Area-weighted unit vertex normals (V,3) of a triangle mesh -- what a .obj/.ply loader (trimesh in the reference's
    URDF pipeline) hands to OpenGL when the file carries none."""
    v, f = np.asarray(verts, np.float64), np.asarray(faces, np.int64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    norm = np.linalg.norm(n, axis=1, keepdims=True)
    return (n / np.where(norm > 0, norm, 1.0)).astype(np.float32)


class RenderMeshes:
    """Padded triangle meshes of the object set: verts (n_obj,V,3) metres, colors (n_obj,V,3) in [0,1],
    faces (n_obj,F,3) int32, n_faces (n_obj,) -- one row per label; optional unit vertex normals (computed from the faces
    when not given), texture coordinates uvs (n_obj,V,2) and textures (n_obj,TH,TW,3) in [0,1] (resized by the caller to a
    common size)."""

    def __init__(self, labels, verts_list, faces_list, colors_list=None, normals_list=None, uvs_list=None, textures=None):
        self.labels = np.asarray(labels)
        self.label_to_id = {l: i for i, l in enumerate(self.labels)}
        n = len(labels)
        V = max(len(v) for v in verts_list); F = max(len(f) for f in faces_list)
        verts = np.zeros((n, V, 3), np.float32); colors = np.full((n, V, 3), 0.7, np.float32); faces = np.zeros((n, F, 3), np.int32)
        normals = np.zeros((n, V, 3), np.float32)
        for i, (v, f) in enumerate(zip(verts_list, faces_list)):
            verts[i, :len(v)] = v; faces[i, :len(f)] = f
            if colors_list is not None:
                colors[i, :len(v)] = colors_list[i]
            normals[i, :len(v)] = normals_list[i] if normals_list is not None else vertex_normals(v, f)
        self.verts, self.colors, self.faces = torch.from_numpy(verts), torch.from_numpy(colors), torch.from_numpy(faces)
        self.normals = torch.from_numpy(normals)
        self.n_faces = torch.tensor([len(f) for f in faces_list], dtype=torch.int32)
        self.uvs = self.tex = None
        if textures is not None:
            assert uvs_list is not None, 'textures need texture coordinates'
            uvs = np.zeros((n, V, 2), np.float32)
            for i, uv in enumerate(uvs_list):
                uvs[i, :len(uv)] = uv
            tex = np.asarray(textures, np.float32)
            assert tex.ndim == 4 and tex.shape[0] == n and tex.shape[3] == 3, tex.shape
            self.uvs = torch.from_numpy(uvs)
            self.tex = torch.from_numpy(np.concatenate([tex, np.zeros(tex.shape[:3] + (1,), np.float32)], axis=3))   # RGB + pad: one 16-byte texel

    def cuda(self):
        for k in ('verts', 'colors', 'faces', 'n_faces', 'normals', 'uvs', 'tex'):
            if getattr(self, k) is not None:
                setattr(self, k, getattr(self, k).cuda().contiguous())
        return self

    def c_struct(self):
        TH, TW = (self.tex.shape[1], self.tex.shape[2]) if self.tex is not None else (0, 0)
        return MeshSet(ptr(self.verts), ptr(self.colors), ptr(self.normals), ptr(self.uvs), ptr(self.tex), ptr(self.faces), ptr(self.n_faces),
                       self.verts.shape[1], self.faces.shape[1], TH, TW)


# The shading of PyBullet's hardware renderer is third-party and not part of the reference; its STRUCTURE is (see
# csrc/raster_device.h).  These constants are placeholders of the usual OpenGL fixed-function magnitudes, to be fitted against
# a few PyBullet renders of the object set when a box with PyBullet is at hand.
OPENGL_LIKE = dict(ambient=0.4, diffuse=0.6, specular=0.05, shininess=32.0, light_dir=(0.0, 0.0, 1.0), light_frame='object',
                   smooth=True, quantize=True)
FLAT = dict(ambient=0.6, diffuse=0.4, specular=0.0, shininess=1.0, light_dir=(0.0, 0.0, -1.0), light_frame='camera', smooth=False,
            quantize=False)


class HipBatchRenderer:
    """shading='flat' (round 1's model, default) or 'opengl' (PyBullet-like: smooth normals, texture, highlight, world-frame
    light, 8-bit output) or a dict with the keys of `OPENGL_LIKE`; ambient / diffuse / light_dir override single entries.

    `concurrent_streams_safe`: renderers may set this to False to make CoarseRefinePosePredictor run a call's chunks one after the other whatever
    n_streams says.  This class was such a case for a few hours of round 4: its renders lost triangles while the 16-bit backbone ran on another HIP
    stream -- a wave's PACKED-FP32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, which hipcc's SLP vectoriser had put all over the
    rasteriser) give wrong results on gfx950 while another wave of the SIMD issues 16-bit MFMAs with VGPR accumulators.  The rasteriser's and the
    geometry kernels' objects are now built without that target feature (cosypose_amd/build.py NO_PACKED_FP32; profiles/r04_raster_streams.txt)
    and the renders are bit-reproducible beside the backbone (0 of 480; the three-stream loop 0 of 480 calls)."""
    concurrent_streams_safe = True

    def __init__(self, meshes, ambient=None, diffuse=None, light_dir=None, shading='flat'):
        self.meshes = meshes
        cfg = dict(FLAT if shading == 'flat' else OPENGL_LIKE if shading == 'opengl' else shading)
        for k, v in (('ambient', ambient), ('diffuse', diffuse), ('light_dir', light_dir)):
            if v is not None:
                cfg[k] = v
        l = np.asarray(cfg['light_dir'], np.float64); l = l / np.linalg.norm(l)
        self.ambient, self.diffuse = float(cfg['ambient']), float(cfg['diffuse'])
        self.light = tuple(float(v) for v in l)
        self.shade = Shade(self.ambient, self.diffuse, float(cfg['specular']), float(cfg['shininess']), (ctypes.c_float * 3)(*self.light),
                           1 if cfg['light_frame'] == 'object' else 0, int(bool(cfg['smooth'])), int(bool(cfg['quantize'])))
        self._scratch = {}       # per HIP stream: renders issued on different streams may overlap

    def _prepare(self, obj_infos, TCO, K, H, W):
        m = self.meshes
        require_device(m.verts, TCO, K)
        TCO = torch.as_tensor(TCO).detach().float().contiguous()
        K = torch.as_tensor(K).detach().float().contiguous()
        bsz = len(TCO)
        assert TCO.shape == (bsz, 4, 4) and K.shape == (bsz, 3, 3) and len(obj_infos) == bsz
        dev = TCO.device
        obj = ints_to_device(np.fromiter((m.label_to_id[o['name']] for o in obj_infos), dtype=np.int32, count=bsz), dev)
        need = lib().cosy_render_scratch_bytes(bsz, m.verts.shape[1], H, W)
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        scratch = self._scratch.get(key)
        if scratch is None or scratch.numel() < need:
            scratch = self._scratch[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        return TCO, K, obj, bsz, dev, scratch

    def render(self, obj_infos, TCO, K, resolution=(240, 320), render_depth=False):
        H, W = min(resolution), max(resolution)          # bullet_batch_renderer.py:34: images are (min(res), max(res))
        TCO, K, obj, bsz, dev, scratch = self._prepare(obj_infos, TCO, K, H, W)
        rgb = torch.empty(bsz, 3, H, W, device=dev)
        depth = torch.empty(bsz, H, W, device=dev) if render_depth else None
        mesh = self.meshes.c_struct()
        check(lib().cosy_render_meshes_ex(ctypes.byref(mesh), ctypes.byref(self.shade), ptr(obj), ptr(TCO), ptr(K), bsz, H, W, ptr(rgb),
                                          ptr(depth), ptr(scratch), stream()))
        return (rgb, depth) if render_depth else rgb

    def render_crop_pack(self, obj_infos, TCO, K_crop, frames4, im_ids, boxes_crop, resolution, net=None, x8=None, dtype=None):
        """Render + crop + pack in one pass (cosy_render_crop_pack): writes the network's NHWC8 input -- of the inference
        engine `net`, or the caller's buffer `x8` of element type `dtype` -- without materialising the render."""
        H, W = resolution
        TCO, K_crop, obj, bsz, dev, scratch = self._prepare(obj_infos, TCO, K_crop, H, W)
        n_im, h, w = frames4.shape[0], frames4.shape[1], frames4.shape[2]
        mesh = self.meshes.c_struct()
        if net is not None:
            check(lib().cosy_render_crop_pack(net, ctypes.byref(mesh), ctypes.byref(self.shade), ptr(obj), ptr(TCO), ptr(K_crop), ptr(frames4),
                                              ptr(im_ids), ptr(boxes_crop), bsz, n_im, h, w, ptr(scratch), stream()))
        else:
            check(lib().cosy_render_crop_pack_to(ptr(x8), dtype, ctypes.byref(mesh), ctypes.byref(self.shade), ptr(obj), ptr(TCO), ptr(K_crop),
                                                 ptr(frames4), ptr(im_ids), ptr(boxes_crop), bsz, n_im, h, w, H, W, ptr(scratch), stream()))
