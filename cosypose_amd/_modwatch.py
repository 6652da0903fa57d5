"""Cached (name, tensor) tables of a module tree.

Walking ~300 submodules for parameters() / buffers() costs 0.4-1 ms of host time; the inference engine did it in front of every model
call (is the engine still built from these weights?) and the training step in front of every backbone launch -- host time during which
the GPU has nothing queued whenever the host is not already ahead.  The tensor OBJECTS of a module tree only change through a
registration on some module (torch's global registration hooks bump one counter) or through Module._apply (.cuda() / .to() / .float():
swaps buffer objects without registering -- caught by identity sentinels on the first and last buffer and the last parameter).
Module._apply itself is wrapped to bump the counter too (a partial conversion of a submodule swaps tensors the sentinels do not cover).
In-place updates, `.data` re-pointing and load_state_dict keep the objects, and are seen by whoever reads data_ptr() / _version
from the cached objects."""
import torch

_epoch = [0]


def _bump(*args):
    _epoch[0] += 1


for _hook in ('register_module_parameter_registration_hook', 'register_module_buffer_registration_hook',
              'register_module_module_registration_hook'):
    getattr(torch.nn.modules.module, _hook)(_bump)


# Module._apply (.cuda() / .to() / .float() / .half() on ANY module of a tree) swaps parameter / buffer objects without registering anything; a
# partial one -- model.backbone._blocks[k].float() -- changes tensors in the middle of the cached tables that the first / last sentinels
# below cannot see.  The method is wrapped once, process-wide, to bump the same counter (cheap: it runs on explicit conversions only).
_orig_apply = torch.nn.Module._apply
if not getattr(_orig_apply, '_cosy_modwatch', False):
    def _apply(self, fn, *args, **kwargs):
        _bump()
        out = _orig_apply(self, fn, *args, **kwargs)
        _bump()
        return out
    _apply._cosy_modwatch = True
    torch.nn.Module._apply = _apply


def registration_epoch():
    return _epoch[0]


def named_tensors(module, slot='_cosy_named_tensors'):
    """(OrderedDict name -> Parameter, OrderedDict name -> buffer) of `module`, cached on the module until a registration happens anywhere
    or a sentinel no longer is the module's own tensor."""
    cached = module.__dict__.get(slot)
    if cached is not None and cached[0] == _epoch[0]:
        _, params, buffers = cached
        ok = True
        if buffers:
            first, last = next(iter(buffers)), next(reversed(buffers))
            ok = module.get_buffer(first) is buffers[first] and module.get_buffer(last) is buffers[last]
        if ok and params:
            last = next(reversed(params))
            ok = module.get_parameter(last) is params[last]
        if ok:
            return params, buffers
    params, buffers = dict(module.named_parameters()), dict(module.named_buffers())
    module.__dict__[slot] = (_epoch[0], params, buffers)
    return params, buffers
