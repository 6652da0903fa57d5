"""I/O containers of the drop-in API: same surface as the reference's
cosypose/utils/tensor_collection.py:7-174 (TensorCollection, PandasTensorCollection,
concatenate), so detections go in and predictions come out in the shape its callers
(cosypose/evaluation/pred_runner/bop_predictions.py:108-112) expect.

The one behavioural difference: `gather_distributed` collects ranks with a
torch.distributed all-gather (RCCL on MI355X) instead of files on a shared disk
(reference :142-163); see cosypose_amd/distributed.py.
"""
import numpy as np
import pandas as pd
import torch


class TensorCollection:
    """Named tensors sharing their first dimension; attribute access and fancy indexing.

    The tensors live in one dict (`.tensors`); every name in it is also readable / assignable as an attribute, any other
    attribute is an ordinary instance attribute.  Casting / moving methods work in place and return self, as the reference's do."""

    def __init__(self, **kwargs):
        object.__setattr__(self, '_tensors', {})
        for name, tensor in kwargs.items():
            self.register_tensor(name, tensor)

    # ---- the tensor table
    def register_tensor(self, name, tensor):
        self._tensors[name] = tensor

    def delete_tensor(self, name):
        self._tensors.pop(name)

    @property
    def tensors(self):
        return self._tensors

    @property
    def device(self):
        for t in self._tensors.values():
            return t.device
        raise StopIteration('empty collection has no device')

    # ---- attribute protocol: tensor names shadow nothing, they are looked up only when normal lookup fails
    def __getattr__(self, name):
        table = self.__dict__.get('_tensors')
        if table is not None and name in table:
            return table[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        table = self.__dict__.get('_tensors')
        if table is None:
            raise ValueError(f'{type(self).__name__}.__init__ has not run: cannot set {name!r}')
        if name in table:
            table[name] = value
        else:
            object.__setattr__(self, name, value)

    def __getitem__(self, ids):
        ids = _as_slice(ids)
        return TensorCollection(**{k: v[ids] for k, v in self._tensors.items()})

    def __repr__(self):
        body = ''.join(f'    {k}: {t.shape} {t.dtype} {t.device},\n' for k, t in self._tensors.items())
        return f'{type(self).__name__}(\n{body})'

    # ---- pickling: the tensor table only
    def __getstate__(self):
        return {'tensors': self.tensors}

    def __setstate__(self, state):
        TensorCollection.__init__(self, **state['tensors'])

    # ---- in-place casts / moves
    def _apply(self, fn):
        for k in list(self._tensors):
            self._tensors[k] = fn(self._tensors[k])
        return self

    def to(self, torch_attr):
        return self._apply(lambda t: t.to(torch_attr))

    def cuda(self):
        return self._apply(lambda t: t.to('cuda'))

    def cpu(self):
        return self._apply(lambda t: t.to('cpu'))

    def float(self):
        return self._apply(lambda t: t.to(torch.float32))

    def double(self):
        return self._apply(lambda t: t.to(torch.float64))

    def half(self):
        return self._apply(lambda t: t.to(torch.float16))

    def clone(self):
        return TensorCollection(**{k: v.clone() for k, v in self._tensors.items()})


class PandasTensorCollection(TensorCollection):
    """TensorCollection + a pandas DataFrame `infos` with one row per element."""

    def __init__(self, infos, **tensors):
        super().__init__(**tensors)
        self.infos = infos.reset_index(drop=True)
        self.meta = dict()

    def merge_df(self, df, *args, **kwargs):
        infos = self.infos.merge(df, how='left', *args, **kwargs)
        assert len(infos) == len(self.infos)
        assert (infos.index == self.infos.index).all()
        return PandasTensorCollection(infos=infos, **self.tensors)

    def clone(self):
        return PandasTensorCollection(self.infos.copy(), **super().clone().tensors)

    def __repr__(self):
        return super().__repr__()[:-1] + '-' * 40 + f'\n    infos:\n{self.infos!r}\n)'

    def __getitem__(self, ids):
        ids = _as_slice(ids)
        infos = self.infos.iloc[ids].reset_index(drop=True)
        return PandasTensorCollection(infos, **super().__getitem__(ids).tensors)

    def __len__(self):
        return len(self.infos)

    def gather_distributed(self, tmp_dir=None):
        """All ranks' collections concatenated in rank order (on every rank).  `tmp_dir` is
        accepted for signature compatibility and unused: no files, one collective."""
        from .distributed import gather_collection
        return gather_collection(self)

    # ---- pickling: tensors + the frame + the free-form meta dict
    def __getstate__(self):
        return dict(tensors=self.tensors, infos=self.infos, meta=self.meta)

    def __setstate__(self, state):
        PandasTensorCollection.__init__(self, state['infos'], **state['tensors'])
        self.meta = state.get('meta', {})


def _as_slice(ids):
    """A run of consecutive row ids (the chunks of batched_model_predictions) as a slice: tensor views instead of
    an index upload + one gather per field."""
    if isinstance(ids, np.ndarray) and ids.ndim == 1 and ids.dtype.kind in 'iu' and len(ids) > 0:
        first = int(ids[0])
        if first >= 0 and int(ids[-1]) - first + 1 == len(ids) and (len(ids) < 2 or bool(np.all(np.diff(ids) == 1))):
            return slice(first, first + len(ids))
    return ids


def concatenate(datas):
    """Row-wise concatenation; empty collections are dropped (reference :7-20)."""
    datas = [d for d in datas if len(d) > 0]
    if len(datas) == 0:
        return PandasTensorCollection(infos=pd.DataFrame())
    assert all(d.__class__ is datas[0].__class__ for d in datas)
    if len(datas) == 1:
        # one chunk (bsz_objects >= the number of objects): the tensors are handed on as they are (SHARED with the input, not copied row
        # by row); the infos get the fresh 0..n-1 index every concatenation has (reference :17: reset_index(drop=True))
        return PandasTensorCollection(infos=datas[0].infos.reset_index(drop=True), **datas[0].tensors)
    infos = pd.concat([d.infos for d in datas], axis=0, sort=False).reset_index(drop=True)
    tensors = {k: torch.cat([getattr(d, k) for d in datas], dim=0) for k in datas[0].tensors.keys()}
    return PandasTensorCollection(infos=infos, **tensors)
