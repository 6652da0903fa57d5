"""Symmetric pose distances, same surface as the reference's cosypose/lib3d/symmetric_distances.py:8-57.

The reference expands every (sample, symmetry) pair on the host (cosypose_cext.expand_ids_for_symmetry), gathers
points and symmetries per pair, computes one distance per pair, copies the distances to the CPU and runs a C++
segmented argmin (scatter_argmin).  Here one HIP kernel per call walks the symmetries of each sample directly
(libcosyhip.so: cosy_symmetric_distance): no expansion, no host round trip, same tie rule (strict <, first wins).
"""
import numpy as np
import torch

from . import lib3d
from ._lib import lib, check, ptr, stream, require_device, ints_to_device


def expand_ids_for_symmetry(labels, n_symmetries):
    """(ids_expand, sym_ids) int32 host arrays: for n, for k < n_symmetries[labels[n]] -> (n, k)
    (cosypose_cext.cpp:247-259).  Host inputs, host outputs, as in the reference."""
    n = np.fromiter((n_symmetries[l] for l in labels), dtype=np.int64, count=len(labels))
    ids_expand = np.repeat(np.arange(len(labels), dtype=np.int32), n)
    starts = np.repeat(np.cumsum(n) - n, n)
    sym_ids = (np.arange(int(n.sum()), dtype=np.int64) - starts).astype(np.int32)
    return ids_expand, sym_ids


def expand_ids_for_symmetry_device(n_sym_item):
    """Device variant: n_sym_item (B,) int32 on the device -> (ids_expand, sym_ids) int32 device tensors.
    The output length sum(n_sym_item) is read back once to size them."""
    require_device(n_sym_item)
    n = n_sym_item.to(torch.int32).contiguous()
    M = int(n.sum().item())
    a = torch.empty(M, dtype=torch.int32, device=n.device)
    b = torch.empty(M, dtype=torch.int32, device=n.device)
    check(lib().cosy_expand_ids_for_symmetry(ptr(n), n.shape[0], ptr(a), ptr(b), None, stream()))
    return a, b


def scatter_argmin(dists, ids_expand):
    return lib3d.scatter_argmin(dists, ids_expand)


def _n_sym_table(mesh_db, device):
    n = np.fromiter((mesh_db.infos[l]['n_sym'] for l in mesh_db.labels), dtype=np.int32, count=len(mesh_db.labels))
    return ints_to_device(n, device)


def _symmetric_distance(T1, T2, labels, mesh_db, mode):
    bsz = T1.shape[0]
    assert T1.shape == (bsz, 4, 4)
    assert T2.shape == (bsz, 4, 4)
    assert len(labels) == bsz
    if bsz == 0:
        return torch.empty(0, dtype=T1.dtype, device=T1.device), None
    require_device(T1, T2, mesh_db.points, mesh_db.symmetries)
    dev = T1.device
    T1c, T2c = T1.detach().float().contiguous(), T2.detach().float().contiguous()
    pts = mesh_db.points.detach().float().contiguous()
    sym = mesh_db.symmetries.detach().float().contiguous()
    obj = mesh_db.object_ids(labels, dev)
    n_sym = _n_sym_table(mesh_db, dev)
    min_dists = torch.empty(bsz, device=dev)
    best = torch.empty(bsz, dtype=torch.int32, device=dev)
    S12 = torch.empty(bsz, 4, 4, device=dev)
    check(lib().cosy_symmetric_distance(ptr(T1c), ptr(T2c), ptr(obj), ptr(pts), ptr(sym), ptr(n_sym), bsz, pts.shape[1],
                                        sym.shape[1], mode, ptr(min_dists), ptr(best), ptr(S12), stream()))
    return min_dists, S12


def symmetric_distance_batched(T1, T2, labels, mesh_db):
    """min over the object's symmetries S of mean_p ||T1 S p - T2 p||, and that S (reference :19-36)."""
    return _symmetric_distance(T1, T2, labels, mesh_db, 0)


def symmetric_distance_batched_fast(T1, T2, labels, mesh_db):
    """Same with the best symmetry chosen by the mean SQUARED distance over the padded table (reference :39-57)."""
    return _symmetric_distance(T1, T2, labels, mesh_db, 1)
