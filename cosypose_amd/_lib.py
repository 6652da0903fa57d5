"""ctypes binding of libcosyhip.so (the C ABI declared in include/cosyhip.h).

There is no fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os

from .build import LIB

COSY_F32, COSY_BF16, COSY_F16 = 0, 1, 2
_lib = None

_c = ctypes
_P, _I, _F, _SZ, _L = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t, _c.c_long

_SIGNATURES = {
    'cosy_version': ([], _I),
    'cosy_last_error': ([], _c.c_char_p),
    'cosy_effnet_b3_param_count': ([], _c.c_long),
    'cosy_effnet_b3_out_hw': ([_I, _I, _c.POINTER(_I), _c.POINTER(_I)], _I),
    'cosy_effnet_b3_create': ([_P, _SZ, _I, _I, _I, _I, _c.POINTER(_P)], _I),
    'cosy_effnet_b3_destroy': ([_P], _I),
    'cosy_effnet_b3_workspace_bytes': ([_P], _SZ),
    'cosy_effnet_b3_set_input_nchw': ([_P, _P, _I, _P], _I),
    'cosy_effnet_b3_features_nchw': ([_P, _I, _P, _P], _I),
    'cosy_effnet_b3_set_probe': ([_P, _I, _P], _I),
    'cosy_effnet_b3_block_info': ([_P, _I, _c.POINTER(_I)], _I),
    'cosy_effnet_b3_set_profiling': ([_P, _I], _I),
    'cosy_effnet_b3_profile_read': ([_P, _P, _I, _c.POINTER(_I)], _I),
    'cosy_frames_to_nhwc4': ([_P, _P, _I, _I, _I, _P], _I),
    'cosy_frames_u8_to_nhwc4': ([_P, _P, _I, _I, _I, _P], _I),
    'cosy_crop_pack': ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    'cosy_effnet_b3_forward': ([_P, _I, _P, _P, _P, _P], _I),
    'cosy_crop_geometry': ([_P, _P, _P, _P, _P, _I, _I, _F, _I, _I, _I, _I, _F, _P, _P, _P, _P], _I),
    'cosy_roi_align': ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P], _I),
    'cosy_pose_update': ([_P, _P, _P, _I, _P, _P], _I),
    'cosy_tco_init_from_boxes': ([_P, _P, _P, _I, _F, _P, _P], _I),
    'cosy_tco_init_zup_autodepth': ([_P, _P, _P, _P, _P, _I, _I, _P, _P], _I),
    'cosy_scatter_argmin': ([_P, _P, _I, _I, _P, _P], _I),
    'cosy_expand_ids_for_symmetry': ([_P, _I, _P, _P, _P, _P], _I),
    'cosy_symmetric_distance': ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P], _I),
    'cosy_loss_co_symmetric': ([_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P], _I),
    'cosy_loss_refiner_disentangled': ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P], _I),
    'cosy_dists_add': ([_P, _P, _P, _P, _I, _I, _I, _P, _P], _I),
    'cosy_render_scratch_bytes': ([_I, _I, _I, _I], _c.c_size_t),
    'cosy_render_meshes': ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P], _I),
    'cosy_render_meshes_ex': ([_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P], _I),
    'cosy_render_crop_pack': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P], _I),
    'cosy_render_crop_pack_to': ([_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P], _I),
    'cosy_train_workspace_bytes': ([], _c.c_size_t),
    'cosy_crop_pack_to': ([_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    'cosy_crop_pack_to_ws': ([_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P], _I),
    'cosy_crop_pack_workspace_bytes': ([_I, _I, _I], _SZ),
    'cosy_bn_train_stats': ([_P, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P], _I),
    'cosy_bn_train_apply': ([_P, _P, _P, _P, _P, _L, _I, _I, _P, _I, _P, _P, _P], _I),
    'cosy_bn_train_backward': ([_P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P], _I),
    'cosy_bn_train_backward_gated': ([_P, _P, _P, _F, _P, _P, _P, _P, _P, _L, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P], _I),
    'cosy_dw_train_forward': ([_P, _P, _I, _I, _I, _I, _I, _I, _P, _P], _I),
    'cosy_dw_train_backward_data': ([_P, _P, _I, _I, _I, _I, _I, _I, _P, _P], _I),
    'cosy_dw_train_backward_data_add': ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P], _I),
    'cosy_dw_train_backward_weight': ([_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P], _I),
    'cosy_dw_train_backward_weight_ex': ([_P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P], _I),
    'cosy_wgrad_tall_supported': ([_L, _I, _I], _I),
    'cosy_wgrad_tall': ([_P, _P, _L, _I, _I, _P, _P, _P], _I),
    'cosy_wgrad': ([_P, _P, _L, _I, _I, _P, _P, _P], _I),
    'cosy_train_gemm': ([_P, _P, _I, _L, _I, _I, _P, _P, _P, _P], _I),
    'cosy_train_pack_plan': ([_I, _P, _P, _P, _P, _P, _P, _P], _I),
    'cosy_train_pack_all': ([_P, _I, _c.c_longlong, _P, _P], _I),
    'cosy_train_gemm_packed': ([_P, _P, _c.c_longlong, _L, _I, _I, _P, _P, _P], _I),
    'cosy_stem_im2col_ld': ([_P, _I, _I, _I, _I, _P, _P], _I),
    'cosy_rows_mean': ([_P, _I, _I, _I, _P, _P, _P], _I),
    'cosy_rows_mean_bn': ([_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P], _I),
    'cosy_rows_dot_bn': ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P], _I),
    'cosy_bn_train_apply_gated': ([_P, _P, _P, _P, _P, _L, _I, _I, _P, _I, _P, _P], _I),
    'cosy_rows_dot': ([_P, _P, _I, _I, _I, _P, _P, _P], _I),
    'cosy_rows_scale': ([_P, _P, _P, _F, _I, _I, _I, _P, _P], _I),
    'cosy_rows_broadcast': ([_P, _F, _I, _I, _I, _P, _P], _I),
    'cosy_act_forward': ([_P, _L, _I, _P, _P], _I),
    'cosy_se_train_forward': ([_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P], _I),
    'cosy_se_train_backward': ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P], _I),
    'cosy_fc_small_forward': ([_P, _P, _P, _I, _I, _I, _P, _P], _I),
    'cosy_fc_small_backward': ([_P, _P, _P, _I, _I, _I, _P, _P, _P, _P], _I),
    'cosy_act_backward': ([_P, _P, _L, _I, _P, _P], _I),
    'cosy_stem_im2col': ([_P, _I, _I, _I, _P, _P], _I),
    'cosy_loss_refiner_disentangled_backward': ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P], _I),
    'cosy_grad_norm_clip': ([_P, _L, _F, _P, _P, _P], _I),
    'cosy_adam_step': ([_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P, _P], _I),
}
EXPORTS = tuple(_SIGNATURES)


class MeshSet(ctypes.Structure):       # cosy_mesh_t
    _fields_ = [('verts', _P), ('colors', _P), ('normals', _P), ('uvs', _P), ('tex', _P), ('faces', _P), ('n_faces', _P),
                ('V', _I), ('F', _I), ('TH', _I), ('TW', _I)]


class Shade(ctypes.Structure):         # cosy_shade_t
    _fields_ = [('ambient', _F), ('diffuse', _F), ('specular', _F), ('shininess', _F), ('light', _F * 3),
                ('light_frame', _I), ('smooth', _I), ('quantize', _I)]


class ProfRec(ctypes.Structure):
    _fields_ = [('name', _c.c_char * 64), ('layer', _I), ('n', _I), ('ms_avg', _F), ('ms_min', _F),
                ('bytes', _c.c_double), ('flops', _c.c_double), ('cbytes', _c.c_double)]


def profile_read(handle):
    """-> list of dicts, one per launch slot of the backbone schedule (see cosy_prof_rec_t)."""
    recs = (ProfRec * 1100)()
    n = _I(0)
    check(lib().cosy_effnet_b3_profile_read(handle, recs, 1100, ctypes.byref(n)))
    return [dict(name=r.name.decode(), layer=r.layer, n=r.n, ms_avg=r.ms_avg, ms_min=r.ms_min, bytes=r.bytes, flops=r.flops, cbytes=r.cbytes)
            for r in recs[:n.value]]


class CosyHipError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        path = LIB
        if os.environ.get('COSY_TUNE_LIB'):       # experiments only: the -DCOSY_TUNE build that reads COSY_* knobs (or, for A/B
            from .build import TUNE_LIB as path   # runs inside one gpurun call, another build of it given by its path)
            if os.environ['COSY_TUNE_LIB'].endswith('.so'):
                path = os.environ['COSY_TUNE_LIB']
        if not os.path.exists(path):
            raise CosyHipError(f'{path} not found: build it with `python -m cosypose_amd.build` '
                               '(or __graft_entry__.build()); cosypose_amd has no CPU / eager fallback')
        if path == LIB:
            # the wave kernels keep values in registers hipcc is not told about; only a library whose ISA was checked at build time
            # (cosypose_amd/build.py stamps the verdict with the library's hash) is loaded -- never a lazily rebuilt, unchecked one
            from .build import stamp_matches, ISA_STAMP
            if not stamp_matches(path):
                raise CosyHipError(f'{path} has no matching clean wave-kernel ISA stamp ({ISA_STAMP}): it was not produced by '
                                   '`python -m cosypose_amd.build` / __graft_entry__.build() as it stands -- rebuild')
        l = ctypes.CDLL(path)
        for name, (args, res) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = res
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise CosyHipError(f'libcosyhip error {rc}: {lib().cosy_last_error().decode()}')


def ptr(t):
    """Device pointer of a contiguous torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CosyHipError('cosypose_amd runs on a ROCm device only (got a CPU tensor); there is no CPU fallback')


import collections as _collections

_id_cache = _collections.OrderedDict()       # (device index, shape, bytes) -> int32 device tensor
_ID_CACHE_ENTRIES, _ID_CACHE_MAX_BYTES = 64, 64 << 10


def host_to_device(values, device, dtype=None):
    """numpy array / CPU tensor -> device tensor through page-locked memory and a non-blocking copy: the host does not wait for the
    kernels already queued on the stream (torch's .to(device) of pageable memory does)."""
    import numpy as np
    import torch
    host = values if isinstance(values, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(values))
    if dtype is not None:
        host = host.to(dtype)
    if torch.device(device).type == 'cpu' or host.numel() == 0 or host.device.type != 'cpu':
        return host.to(device)
    return host.pin_memory().to(device, non_blocking=True)


def ints_to_device(values, device):
    """int32 device tensor from host ids (list / numpy / CPU tensor) without draining the stream: the ids go through
    pinned memory and a non-blocking copy, so the host keeps running ahead of the GPU (a pageable H2D copy would wait
    for every kernel already queued on the stream).  Device tensors are only cast."""
    import numpy as np
    import torch
    if values is None:
        return None
    if isinstance(values, torch.Tensor) and values.device.type != 'cpu':
        return values.to(device=device, dtype=torch.int32).contiguous()
    arr = np.ascontiguousarray(np.asarray(values.cpu() if isinstance(values, torch.Tensor) else values), dtype=np.int32)
    host = torch.as_tensor(arr)
    if torch.device(device).type == 'cpu' or host.numel() == 0:
        return host.to(device)
    # Small id arrays (labels -> object rows, frame ids) repeat from call to call -- the refiner's four iterations and every step of a
    # served workload hand over the same ones: keyed by CONTENT, the device copy is reused and no copy packet enters the compute stream
    # (each one stalls it for a DMA round trip).  Read-only by contract: callers pass these to kernels as inputs.
    if arr.nbytes <= _ID_CACHE_MAX_BYTES:
        key = (torch.device(device).index, arr.shape, arr.tobytes())
        hit = _id_cache.get(key)
        if hit is not None:
            _id_cache.move_to_end(key)
            dev_t, ready, home = hit
            # the upload was enqueued on the stream that first asked for this content: another stream (a concurrent chunk with the same
            # ids) must not read it before it has landed, and the caching allocator -- which would hand the block back to the HOME stream's
            # pool the moment the entry is evicted -- must know that this stream reads it too
            cur = torch.cuda.current_stream(device)
            if not ready.query():
                cur.wait_event(ready)
            if cur.cuda_stream != home:
                dev_t.record_stream(cur)
            return dev_t
        dev_t = host.pin_memory().to(device, non_blocking=True)
        ready = torch.cuda.Event()
        cur = torch.cuda.current_stream(device)
        ready.record(cur)
        _id_cache[key] = (dev_t, ready, cur.cuda_stream)
        while len(_id_cache) > _ID_CACHE_ENTRIES:
            _id_cache.popitem(last=False)
        return dev_t
    return host.pin_memory().to(device, non_blocking=True)
