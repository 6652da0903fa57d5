"""EfficientNet-B3 backbone module: holds the parameters under the reference's state_dict
keys and runs them through libcosyhip.so.

Drop-in for `EfficientNet.from_name('efficientnet-b3', in_channels=6)`
(cosypose/models/efficientnet.py:206-210): `load_state_dict` accepts reference checkpoints
unchanged (`_conv_stem.weight`, `_bn0.*`, `_blocks.{i}._{expand_conv,bn0,depthwise_conv,bn1,
se_reduce,se_expand,project_conv,bn2}.*`, `_conv_head.weight`, `_bn1.*`), `forward(x)`
returns the (B,1536,h,w) feature map.  ONE restriction the reference does not have: the crop sides must be multiples of 16
in [128, 1024] (the fused kernels tile the 2x..32x downsampled maps exactly; 240x320 and 256x256 -- what every released CosyPose
model and the benchmark use -- qualify, EfficientNet-B3's nominal 300x300 does not): other sizes raise ValueError before any
device work (pad or resize the crop to a supported size).  This module's forward is the inference (eval-mode) engine; the
train-mode network (batch-statistics BatchNorm, drop_connect, backward: SURVEY 8a-13) runs through
cosypose_amd.train_engine on the same parameters.
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import arch
from ._lib import lib, check, ptr, stream, require_device, COSY_F32, COSY_BF16, COSY_F16, CosyHipError

_DTYPES = {'fp32': COSY_F32, 'f32': COSY_F32, 'float32': COSY_F32, 'bf16': COSY_BF16, 'bfloat16': COSY_BF16,
           'fp16': COSY_F16, 'f16': COSY_F16, 'float16': COSY_F16, 'half': COSY_F16}


class _Block(nn.Module):
    def __init__(self, k, s, expand, cin, cout):
        super().__init__()
        cmid, cse = cin * expand, arch.se_channels(cin)
        bn = lambda c: nn.BatchNorm2d(c, momentum=0.01, eps=arch.BN_EPS)
        if expand != 1:
            self._expand_conv = nn.Conv2d(cin, cmid, 1, bias=False)
            self._bn0 = bn(cmid)
        self._depthwise_conv = nn.Conv2d(cmid, cmid, k, stride=s, groups=cmid, bias=False)
        self._bn1 = bn(cmid)
        self._se_reduce = nn.Conv2d(cmid, cse, 1)
        self._se_expand = nn.Conv2d(cse, cmid, 1)
        self._project_conv = nn.Conv2d(cmid, cout, 1, bias=False)
        self._bn2 = bn(cout)


def packed_cuda(module, device=None):
    """module.cuda(device) with ONE host -> device copy per dtype instead of one per tensor: the CPU parameters and buffers of the tree are
    concatenated (each at a 256-byte boundary) in page-locked memory, uploaded, and every tensor's `.data` re-pointed at its slice of the
    device slab (round 5's profile: a 574-key state dict x 2 models = ~2,900 blocking copy launches, 15 ms of every cold start).  The
    tensor OBJECTS stay (optimizers, name tables and engines keyed on them keep working); whatever is not a CPU tensor, and everything
    torch's own Module.cuda does besides (submodules' non-tensor state), goes through nn.Module.cuda afterwards, which finds the tensors
    already on the device and leaves them alone."""
    idx = device if isinstance(device, int) else None if device is None else torch.device(device).index
    dev = torch.device('cuda', torch.cuda.current_device() if idx is None else idx)
    seen, groups = set(), {}
    for t in list(module.parameters()) + list(module.buffers()):
        if t is None or id(t) in seen or t.device.type != 'cpu' or t.numel() == 0 or not t.is_contiguous():
            continue
        seen.add(id(t))
        groups.setdefault(t.dtype, []).append(t)
    for dtype, ts in groups.items():
        if len(ts) < 2:
            continue
        esz = ts[0].element_size()
        offs, n = [], 0
        for t in ts:
            offs.append(n)
            n += -(-t.numel() * esz // 256) * 256 // esz
        host = torch.zeros(n, dtype=dtype).pin_memory()
        for t, o in zip(ts, offs):
            host[o:o + t.numel()].copy_(t.detach().reshape(-1))
        slab = host.to(dev, non_blocking=True)
        for t, o in zip(ts, offs):
            t.data = slab[o:o + t.numel()].view(t.shape)
        torch.cuda.current_stream(dev).synchronize()      # the pinned staging buffer is released when this returns
    return nn.Module.cuda(module, device)


class EfficientNet(nn.Module):
    cuda = packed_cuda

    def __init__(self, in_channels=6):
        super().__init__()
        if in_channels != arch.IN_C:
            raise ValueError('the MI355X path implements the 6-channel {observed, rendered} backbone only')
        self._conv_stem = nn.Conv2d(in_channels, arch.STEM_C, 3, stride=2, bias=False)
        self._bn0 = nn.BatchNorm2d(arch.STEM_C, momentum=0.01, eps=arch.BN_EPS)
        self._blocks = nn.ModuleList([_Block(*b) for b in arch.B3_BLOCKS])
        self._conv_head = nn.Conv2d(arch.B3_BLOCKS[-1][4], arch.HEAD_C, 1, bias=False)
        self._bn1 = nn.BatchNorm2d(arch.HEAD_C, momentum=0.01, eps=arch.BN_EPS)
        self.n_features = arch.HEAD_C
        self.n_inputs = in_channels

    @classmethod
    def from_name(cls, model_name, override_params=None, in_channels=6):
        if model_name != 'efficientnet-b3':
            raise ValueError('only efficientnet-b3 is implemented (the backbone of every released CosyPose model)')
        return cls(in_channels=in_channels)

    def extract_features(self, inputs):
        return self.forward(inputs)

    def forward(self, inputs):
        """(B,6,H,W) fp32 NCHW -> (B,1536,h,w) fp32, via a standalone engine (no pose head)."""
        pool = getattr(self, '_engines', None)
        if pool is None:
            pool = self.__dict__['_engines'] = EnginePool(self, None)
        return pool.current(inputs.device).features(inputs)


def flat_params(backbone, pose_fc):
    """Flat fp32 host blob in the order include/cosyhip.h documents."""
    parts = []

    def bn(m):
        parts.extend([m.weight, m.bias, m.running_mean, m.running_var])
    parts.append(backbone._conv_stem.weight); bn(backbone._bn0)
    for blk, (k, s, e, cin, cout) in zip(backbone._blocks, arch.B3_BLOCKS):
        if e != 1:
            parts.append(blk._expand_conv.weight); bn(blk._bn0)
        parts.append(blk._depthwise_conv.weight); bn(blk._bn1)
        parts.extend([blk._se_reduce.weight, blk._se_reduce.bias, blk._se_expand.weight, blk._se_expand.bias])
        parts.append(blk._project_conv.weight); bn(blk._bn2)
    parts.append(backbone._conv_head.weight); bn(backbone._bn1)
    if pose_fc is not None:
        parts.extend([pose_fc.weight, pose_fc.bias])
    else:
        parts.extend([torch.zeros(arch.N_POSE, arch.HEAD_C), torch.zeros(arch.N_POSE)])
    blob = torch.cat([p.detach().reshape(-1).to('cpu', torch.float32) for p in parts]).contiguous()
    assert blob.numel() == arch.param_count()
    return blob, parts


def check_crop_size(H, W):
    """cosy_effnet_b3_create's shape rule, raised on the Python side with a message a caller can act on."""
    if not (128 <= H <= 1024 and 128 <= W <= 1024 and H % 16 == 0 and W % 16 == 0):
        raise ValueError(f'crop size {H}x{W} not supported by the MI355X backbone: both sides must be multiples of 16 in [128, 1024] '
                         f'(e.g. 240x320 or 256x256; pad or resize, e.g. to {max(128, min(1024, -(-H // 16) * 16))}x{max(128, min(1024, -(-W // 16) * 16))})')


class EnginePool:
    """One NetEngine per HIP stream: an engine's activations and workspaces are private to the forward that is running on
    it, so forwards issued on different streams (CoarseRefinePosePredictor's concurrent chunks, or a caller's own streams)
    each get their own and may overlap on the device.  Keyed by the raw stream handle (0 = the default stream).
    Bounded: at most COSY_ENGINE_POOL_MAX (default 8) engines are kept, the least recently used one is released first
    (a caller that creates streams without end would otherwise pin one set of weights + workspaces per dead stream)."""

    def __init__(self, backbone, pose_fc):
        import collections
        import os
        self.backbone, self.pose_fc = backbone, pose_fc
        self.engines = collections.OrderedDict()
        self.max_engines = max(1, int(os.environ.get('COSY_ENGINE_POOL_MAX', '8')))

    def current(self, device=None):
        key = (torch.device(device).index if device is not None else None, torch.cuda.current_stream(device).cuda_stream)
        eng = self.engines.get(key)
        if eng is None:
            while len(self.engines) >= self.max_engines:
                _, old = self.engines.popitem(last=False)
                torch.cuda.synchronize()          # the evicted engine's stream may still be running on its workspaces
                old.release()
            eng = self.engines[key] = NetEngine(self.backbone, self.pose_fc)
        else:
            self.engines.move_to_end(key)
        return eng

    def release(self):
        for eng in self.engines.values():
            eng.release()
        self.engines.clear()


class NetEngine:
    """Owns the cosy_net_t for one (backbone, pose_fc) pair; rebuilt lazily when the
    weights, the crop size, the compute dtype or the required batch capacity change."""

    def __init__(self, backbone, pose_fc):
        self.backbone, self.pose_fc = backbone, pose_fc
        self.handle = None
        self.key = None
        self.capacity = 0

    def _weights_version(self):
        """(storage, in-place version) of every tensor the engine was built from.  Read from the cached name tables (_modwatch: walking the
        backbone's ~300 submodules in front of EVERY model call was half of a step's host time, and GPU idle whenever the host is not ahead)."""
        from . import _modwatch
        params, buffers = _modwatch.named_tensors(self.backbone)
        ts = list(params.values()) + list(buffers.values())
        if self.pose_fc is not None:
            ts += list(_modwatch.named_tensors(self.pose_fc)[0].values())
        return tuple((t.data_ptr(), t._version) for t in ts)

    def ensure(self, B, H, W, dtype, device):
        if self.backbone.training:
            raise CosyHipError('the inference engine runs eval-mode BatchNorm: call .eval(); the train-mode forward '
                               '(cosypose_amd.train_engine) needs gradients enabled')
        check_crop_size(H, W)
        key = (H, W, _DTYPES[dtype], device.index, self._weights_version())
        if self.handle is not None and key == self.key and B <= self.capacity:
            return self.handle
        self.release()
        cap = max(B, self.capacity if key[:4] == (self.key or (None,) * 4)[:4] else 0, 16)
        cap = 1 << (cap - 1).bit_length()
        blob, _ = flat_params(self.backbone, self.pose_fc)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(lib().cosy_effnet_b3_create(blob.data_ptr(), blob.numel(), _DTYPES[dtype], H, W, cap, ctypes.byref(h)))
        self.handle, self.key, self.capacity = h, key, cap
        return h

    def release(self):
        if self.handle is not None:
            torch.cuda.synchronize()
            lib().cosy_effnet_b3_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def features(self, x, dtype='fp32'):
        require_device(x)
        x = x.detach().float().contiguous()
        B, C, H, W = x.shape
        h = self.ensure(B, H, W, dtype, x.device)
        pose = torch.empty(B, arch.N_POSE, device=x.device)
        check(lib().cosy_effnet_b3_set_input_nchw(h, ptr(x), B, stream()))
        check(lib().cosy_effnet_b3_forward(h, B, None, ptr(pose), None, stream()))
        fh, fw = arch.feature_hw(H, W)
        out = torch.empty(B, arch.HEAD_C, fh, fw, device=x.device)
        check(lib().cosy_effnet_b3_features_nchw(h, B, ptr(out), stream()))
        return out
