"""PosePredictor: the render-and-compare loop, same surface as the reference's
cosypose/models/pose.py:18-132, executed by HIP kernels (libcosyhip.so).

Differences that are deliberate and invisible to callers of the reference API:
  * the observed crop (roi_align) and the renderer's output are written straight into one
    channels-last 6-channel buffer (no torch.cat, no (B,3,H,W) crop tensor);
  * `forward(..., im_ids=...)` lets the caller pass the N frames once plus a per-object frame
    index instead of replicating frames per object (pose_predictor.py:41 `images[im_ids]`);
  * `compute_dtype` ('fp32' parity mode, 'bf16' throughput mode) selects the backbone's
    storage type; geometry and the pose update are always fp32.
There is no CPU / eager fallback: tensors must live on a ROCm device.
"""
import torch
from torch import nn

from . import lib3d, arch, train_engine
from ._lib import lib, check, ptr, stream, require_device, ints_to_device, CosyHipError, COSY_F32
from .efficientnet import EnginePool, packed_cuda


class PosePredictor(nn.Module):
    def __init__(self, backbone, renderer, mesh_db, render_size=(240, 320), pose_dim=9):
        super().__init__()
        self.backbone = backbone
        self.renderer = renderer
        self.mesh_db = mesh_db
        self.render_size = render_size
        self.pose_dim = pose_dim

        n_features = backbone.n_features
        self.heads = dict()
        self.pose_fc = nn.Linear(n_features, pose_dim, bias=True)
        self.heads['pose'] = self.pose_fc

        self.debug = False
        self.tmp_debug = dict()
        self.compute_dtype = 'fp32'
        self.drop_connect_rate = train_engine.DROP_CONNECT_RATE   # train mode only (efficientnet.py:182-185)
        self.__dict__['_engines'] = EnginePool(backbone, self.pose_fc)   # one engine per HIP stream

    # caches hung on the module by this package (name tables, the flat optimizer that re-homed the parameters, ...) are not state: a copy or a
    # pickle of the model must not drag a 43 MB optimizer (or another model's tensor tables) along; engines are rebuilt lazily by the copy
    def __getstate__(self):
        state = self.__dict__.copy()
        for k in [k for k in state if k.startswith('_cosy_')]:
            del state[k]
        state['_engines'] = None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)        # torch's back-compat defaults for older pickles (hook dicts), then the state itself
        self.__dict__['_engines'] = EnginePool(self.backbone, self.pose_fc)

    cuda = packed_cuda      # one packed host -> device copy per dtype (efficientnet.packed_cuda)

    def enable_debug(self):
        self.debug = True

    def disable_debug(self):
        self.debug = False

    # ---- pieces of the loop, reference signatures -------------------------------------------
    def _geometry(self, K, TCO, labels, im_size, im_ids=None, out=None):
        if self.pose_dim != 9:
            raise ValueError(f'pose_dim={self.pose_dim} not supported')
        table = self.mesh_db.point_table(2000)
        if table.device != TCO.device:
            raise CosyHipError(f'mesh_db lives on {table.device} but poses on {TCO.device}; call mesh_db.cuda()')
        obj_ids = labels if torch.is_tensor(labels) else self.mesh_db.object_ids(labels, TCO.device)
        return lib3d.crop_geometry(table, obj_ids, K, TCO, im_size, self.render_size, im_ids=im_ids, lamb=1.4, out=out)

    def crop_inputs(self, images, K, TCO, labels):
        bsz, nchannels, h, w = images.shape
        assert K.shape == (bsz, 3, 3)
        assert TCO.shape == (bsz, 4, 4)
        assert len(labels) == bsz
        boxes_rend, boxes_crop, K_crop = self._geometry(K, TCO, labels, (h, w))
        images_cropped = lib3d.roi_align(images, boxes_crop, self.render_size, sampling_ratio=4)
        if self.debug:
            self.tmp_debug.update(boxes_rend=boxes_rend, boxes_crop=boxes_crop)
        return images_cropped, K_crop.detach(), boxes_rend, boxes_crop

    def update_pose(self, TCO, K_crop, pose_outputs):
        if self.pose_dim != 9:
            raise ValueError(f'pose_dim={self.pose_dim} not supported')
        return lib3d.update_pose(TCO, K_crop, pose_outputs)

    def _net(self, B, device):
        H, W = self.render_size
        return self._engines.current(device).ensure(B, H, W, self.compute_dtype, device)

    def net_forward(self, x, return_features=False):
        """x = cat(images_crop, renders) (B,6,H,W) -> {'pose': (B,9)}."""
        require_device(x)
        x = x.detach().float().contiguous()
        B = x.shape[0]
        assert tuple(x.shape[1:]) == (6,) + tuple(self.render_size), x.shape
        h = self._net(B, x.device)
        pose = torch.empty(B, self.pose_dim, device=x.device)
        feat = torch.empty(B, arch.HEAD_C, device=x.device) if return_features else None
        check(lib().cosy_effnet_b3_set_input_nchw(h, ptr(x), B, stream()))
        check(lib().cosy_effnet_b3_forward(h, B, ptr(feat), ptr(pose), None, stream()))
        outputs = dict(pose=pose)
        if return_features:
            outputs['features'] = feat
        return outputs

    # ---- the loop ----------------------------------------------------------------------------
    @staticmethod
    def frames_to_nhwc4(images):
        """(N,3,h,w) float frames -> the interleaved (N,h,w,4) fp32 copy the crop kernels read (one 16-byte load per pixel).  forward() makes
        it per call; a driver that calls several models / chunks on the same frames makes it once and hands it over (`frames_nhwc4=`)."""
        require_device(images)
        n_im, _, h, w = images.shape
        frames4 = torch.empty(n_im, h, w, 4, device=images.device, dtype=torch.float32)
        if images.dtype == torch.uint8:          # the datasets' frames: value / 255.f inside the kernel (== images.float() / 255.)
            images = images.detach().contiguous()
            check(lib().cosy_frames_u8_to_nhwc4(ptr(images), ptr(frames4), n_im, h, w, stream()))
        else:
            images = images.detach().float().contiguous()
            check(lib().cosy_frames_to_nhwc4(ptr(images), ptr(frames4), n_im, h, w, stream()))
        return frames4

    def forward(self, images, K, labels, TCO, n_iterations=1, im_ids=None, out=None, frames_nhwc4=None):
        """(reference: models/pose.py:89-132.)  `out` (an extension, inference only): {iteration number: {'TCO_output' | 'K_crop' | 'boxes_rend' |
        'boxes_crop': destination tensor}} -- the geometry and pose-update kernels then write those outputs straight into the given
        (contiguous fp32) tensors, e.g. this chunk's rows of a batch-wide result, instead of fresh ones that have to be copied;
        `frames_nhwc4`: the result of frames_to_nhwc4(images) when the caller already has it."""
        require_device(images, K, TCO)
        if self.pose_dim != 9:
            raise ValueError(f'pose_dim={self.pose_dim} not supported')
        bsz = TCO.shape[0]
        n_im, nchannels, h, w = images.shape
        assert TCO.shape == (bsz, 4, 4)
        assert len(labels) == bsz
        if im_ids is None:
            assert n_im == bsz and K.shape == (bsz, 3, 3)
        else:
            assert K.shape == (n_im, 3, 3) and len(im_ids) == bsz
            im_ids = ints_to_device(im_ids, TCO.device)
        if images.dtype != torch.uint8:          # uint8 frames (an extension: the training batches) are scaled by 1/255 inside the frame conversion
            images = images.detach().float().contiguous()
        K = K.detach().float().contiguous()
        dev = TCO.device
        # frames -> interleaved (N,h,w,4) once per call: the crop kernel then fetches a pixel's RGB with one 16-byte load
        if frames_nhwc4 is None:
            frames4 = self.frames_to_nhwc4(images)
        else:
            frames4 = frames_nhwc4
            assert frames4.shape == (n_im, h, w, 4) and frames4.dtype == torch.float32 and frames4.is_contiguous() and frames4.device == dev
        obj_ids = self.mesh_db.object_ids(labels, dev)
        train = self.training and torch.is_grad_enabled()
        net = None if train else self._net(bsz, dev)
        H, W = self.render_size

        outputs = dict()
        TCO_input = TCO
        for n in range(n_iterations):
            TCO_input = TCO_input.detach().float().contiguous()
            dst = (out or {}).get(n + 1, {})
            boxes_rend, boxes_crop, K_crop = self._geometry(K, TCO_input, obj_ids, (h, w), im_ids=im_ids,
                                                            out=(dst.get('boxes_rend'), dst.get('boxes_crop'), dst.get('K_crop')))
            # a renderer of this library renders straight into the network input (cosy_render_crop_pack): no (B,3,H,W) fp32
            # render tensor exists; any other renderer keeps the reference's interface (renderer.render -> images)
            fused = hasattr(self.renderer, 'render_crop_pack') and not self.debug
            obj_infos = [dict(name=l) for l in labels]
            renders = None
            if not fused:
                renders = self.renderer.render(obj_infos=obj_infos, TCO=TCO_input, K=K_crop, resolution=self.render_size)
                require_device(renders)
                renders = renders.detach().float().contiguous()
                assert renders.shape == (bsz, 3, H, W), renders.shape
            if train:
                # train mode: fp32, batch-statistics BatchNorm, drop_connect; `pose` carries the autograd
                # graph of the parameters (train_engine.backbone_train), the pose update itself is not differentiated
                # (the reference's default loss, loss_refiner_CO_disentangled, only consumes model_outputs['pose'])
                x8 = torch.empty(bsz, H, W, 8, device=dev)
                if fused:
                    self.renderer.render_crop_pack(obj_infos, TCO_input, K_crop, frames4, im_ids, boxes_crop, (H, W), x8=x8, dtype=COSY_F32)
                else:
                    check(lib().cosy_crop_pack_to(ptr(x8), COSY_F32, ptr(frames4), ptr(im_ids), ptr(boxes_crop), ptr(renders), bsz, n_im,
                                                  h, w, H, W, stream()))
                drop = train_engine.make_drop_connect_scales(bsz, dev, self.drop_connect_rate)
                pose = train_engine.backbone_train(self, x8, drop)
            else:
                if fused:
                    self.renderer.render_crop_pack(obj_infos, TCO_input, K_crop, frames4, im_ids, boxes_crop, (H, W), net=net)
                else:
                    check(lib().cosy_crop_pack(net, ptr(frames4), ptr(im_ids), ptr(boxes_crop), ptr(renders), bsz, n_im, h, w, stream()))
                pose = torch.empty(bsz, self.pose_dim, device=dev)
                check(lib().cosy_effnet_b3_forward(net, bsz, None, ptr(pose), None, stream()))
            model_outputs = dict(pose=pose)
            TCO_output = lib3d.update_pose(TCO_input, K_crop, pose.detach(), out=dst.get('TCO_output'))

            outputs[f'iteration={n+1}'] = {
                'TCO_input': TCO_input,
                'TCO_output': TCO_output,
                'K_crop': K_crop,
                'model_outputs': model_outputs,
                'boxes_rend': boxes_rend,
                'boxes_crop': boxes_crop,
            }
            TCO_input = TCO_output

            if self.debug:
                self.tmp_debug.update(outputs[f'iteration={n+1}'])
                images_f = images.float() / 255. if images.dtype == torch.uint8 else images
                self.tmp_debug.update(images=images_f, renders=renders,
                                      images_crop=lib3d.roi_align(images_f, boxes_crop, self.render_size, 4, im_ids=im_ids))
        return outputs
