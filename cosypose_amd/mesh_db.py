"""Batched mesh database: the `mesh_db` object PosePredictor consumes.

Same surface as the reference's BatchedMeshes / Meshes
(cosypose/lib3d/rigid_mesh_database.py:59-94): `select(labels)` ->
`.sample_points(n, deterministic=True)`, `.points`, `.symmetries`, `.labels`,
`.label_to_id`, `.infos`, `.n_sym_mapping`.

MI355X-first addition: the 2000 point ids drawn by
sample_points(2000, deterministic=True) are the same for every call and every object
(np.random.RandomState(0).choice(Nmax, 2000, replace=False), cosypose/lib3d/mesh_ops.py:31-41),
so the sub-sampled table (n_obj, 2000, 3) is built ONCE on the device (`point_table`)
and kernels index it by object id instead of gathering B x Nmax x 3 floats per iteration.
Loading meshes from .ply (MeshDataBase.from_object_ds, needs trimesh) is out of scope.
"""
import numpy as np
import torch

from .tensor_collection import TensorCollection


def deterministic_point_ids(n_max, n_points):
    return np.random.RandomState(0).choice(n_max, size=n_points, replace=False)


def sample_points(points, n_points, deterministic=False):
    assert points.dim() == 3
    assert n_points <= points.shape[1]
    rng = np.random.RandomState(0) if deterministic else np.random
    from ._lib import host_to_device
    ids = host_to_device(rng.choice(points.shape[1], size=n_points, replace=False), points.device)      # no wait for the queued kernels
    return torch.index_select(points, 1, ids)


class Meshes(TensorCollection):
    def __init__(self, infos, labels, points, symmetries):
        super().__init__()
        self.infos = infos
        self.labels = np.asarray(labels)
        self.register_tensor('points', points)
        self.register_tensor('symmetries', symmetries)

    def select_labels(self, labels):
        raise NotImplementedError

    def sample_points(self, n_points, deterministic=False):
        return sample_points(self.points, n_points, deterministic=deterministic)


class BatchedMeshes(TensorCollection):
    def __init__(self, infos, labels, points, symmetries):
        super().__init__()
        self.infos = infos
        self.label_to_id = {label: n for n, label in enumerate(labels)}
        self.labels = np.asarray(labels)
        self.register_tensor('points', points)
        self.register_tensor('symmetries', symmetries)
        self.__dict__['_tables'] = {}

    @property
    def n_sym_mapping(self):
        return {label: obj['n_sym'] for label, obj in self.infos.items()}

    def select(self, labels):
        ids = [self.label_to_id[l] for l in labels]
        # rows gathered with device-resident ids (cached upload, _lib.ints_to_device): indexing with the Python list would copy it to the
        # device synchronously, i.e. stop the host until every kernel queued so far has run -- once per training step
        rows = self.object_ids(labels).long() if self.points.is_cuda and len(ids) else torch.as_tensor(ids, dtype=torch.long, device=self.points.device)
        return Meshes(infos=[self.infos[l] for l in labels], labels=self.labels[ids],
                      points=torch.index_select(self.points, 0, rows), symmetries=torch.index_select(self.symmetries, 0, rows))

    # ---- device-side fast path -------------------------------------------------------------
    def object_ids(self, labels, device=None):
        """int32 row indices of `labels` in the point table."""
        from ._lib import ints_to_device
        ids = np.fromiter((self.label_to_id[l] for l in labels), dtype=np.int32, count=len(labels))
        return ints_to_device(ids, device if device is not None else self.points.device)

    def point_table(self, n_points=2000):
        """(n_obj, n_points, 3) fp32 contiguous on the points' device: points[:, deterministic ids]."""
        pts = self.points
        key = (n_points, pts.device, pts.data_ptr(), pts._version)
        tab = self._tables.get(key)
        if tab is None:
            ids = torch.as_tensor(deterministic_point_ids(pts.shape[1], n_points)).to(pts.device)
            tab = torch.index_select(pts, 1, ids).float().contiguous()
            self._tables.clear()
            self._tables[key] = tab
        return tab

    def to(self, torch_attr):
        super().to(torch_attr)
        self._tables.clear()
        return self
