"""One process per GPU; candidates shard across ranks; refined poses come back with ONE
all-gather (RCCL over xGMI on MI355X, gloo on CPU for tests).

Replaces the reference's file-system gather (cosypose/utils/tensor_collection.py:142-163:
rank>0 torch.save to a shared tmp dir, barrier, rank 0 torch.load + concatenate) and its
scene-level sharding (cosypose/datasets/samplers.py:20-34).  Each candidate crop is independent
across the whole coarse->refiner chain, so there is no data-path collective inside the loop.

The payload is tiny (<= 2048 candidates x 132 B), i.e. latency-bound: counts and rows travel in
a single padded all_gather (row 0 of each rank's slab carries its row count), never a ring of
point-to-point sends.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def local_device_index():
    """The GPU this rank binds: LOCAL_RANK, wrapped over the visible devices (so that a rehearsal with more ranks than
    GPUs -- COSY_DIST_BACKEND=gloo, all ranks on one device -- exercises the same code path as the real launch)."""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return int(os.environ.get('LOCAL_RANK', '0')) % n if n else 0


def init_distributed_mode(backend=None):
    """torchrun-style env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  One visible GPU per rank:
    the rank binds LOCAL_RANK's device (reference: cosypose/utils/distributed.py:55-69 uses SLURM vars + file store).
    COSY_DIST_BACKEND overrides the backend (RCCL refuses two ranks on one GPU; gloo does not)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 or dist.is_initialized():
        return get_rank(), get_world_size()
    backend = os.environ.get('COSY_DIST_BACKEND', backend)
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if torch.cuda.is_available():
        torch.cuda.set_device(local_device_index())
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend=backend, init_method='env://')
    return get_rank(), get_world_size()


def shard_range(n, rank=None, world_size=None):
    """Contiguous split of range(n) in rank order (same sizes as np.array_split): concatenating the
    ranks' results in rank order reproduces detection order, like gather_distributed's rank-0-first concat."""
    rank = get_rank() if rank is None else rank
    world_size = get_world_size() if world_size is None else world_size
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def balanced_assignment(costs, world_size):
    """Greedy size-balanced sharding for load-imbalanced mixes (config 3): returns, per rank, the sorted
    candidate indices; cost = e.g. frame pixels per candidate."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind='stable')
    loads = np.zeros(world_size); out = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(loads))
        out[r].append(int(i)); loads[r] += costs[i]
    return [np.sort(np.asarray(o, dtype=np.int64)) for o in out]


def all_gather_rows(local, max_rows=None):
    """local (n_r, ...) -> (sum n_r, ...) on every rank, rank order.  Variable n_r allowed.
    One collective when `max_rows` (an upper bound valid on every rank) is given; otherwise the bound is
    agreed with one extra tiny all_reduce(MAX)."""
    world = get_world_size()
    if world == 1:
        return local
    tail = tuple(local.shape[1:])
    width = int(np.prod(tail)) if tail else 1
    n = local.shape[0]
    if max_rows is None:
        m = torch.tensor([n], device=local.device, dtype=torch.int64)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        max_rows = int(m.item())
    assert n <= max_rows
    slab = torch.zeros(max_rows + 1, max(width, 1), device=local.device, dtype=torch.float32)
    slab[0, 0] = float(n)                      # exact for n < 2**24
    if n:
        slab[1:n + 1] = local.reshape(n, width).to(torch.float32)
    out = torch.empty(world * (max_rows + 1), max(width, 1), device=local.device, dtype=torch.float32)
    dist.all_gather_into_tensor(out, slab)
    out = out.view(world, max_rows + 1, -1)
    counts = out[:, 0, 0].round().to(torch.int64).tolist()
    rows = [out[r, 1:1 + c] for r, c in enumerate(counts)]
    return torch.cat(rows, 0).reshape((sum(counts),) + tail).to(local.dtype)


def gather_collection(coll):
    """PandasTensorCollection on each rank -> concatenation in rank order on every rank.  Tensors go through
    all_gather_rows; the small `infos` DataFrames through all_gather_object (host side)."""
    from . import tensor_collection as tc
    world = get_world_size()
    if world == 1:
        return tc.concatenate([coll])
    tensors = {k: all_gather_rows(v) for k, v in coll.tensors.items()}
    infos = [None] * world
    dist.all_gather_object(infos, coll.infos)
    import pandas as pd
    infos = pd.concat([i for i in infos if len(i) > 0], axis=0, sort=False).reset_index(drop=True) if any(
        len(i) for i in infos) else pd.DataFrame()
    return tc.PandasTensorCollection(infos=infos, **tensors)
