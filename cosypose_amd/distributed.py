"""One process per GPU; candidates shard across ranks; results come back with ONE all-gather (RCCL over xGMI on MI355X,
gloo on CPU for tests).

Replaces the reference's scene-level sharding (cosypose/datasets/samplers.py:20-34: seeded permutation + np.array_split by
rank) and its file-system gather (cosypose/utils/tensor_collection.py:142-163: rank>0 torch.save to a shared tmp dir,
barrier, rank 0 torch.load + concatenate).  Each candidate crop is independent across the whole coarse->refiner chain, so
there is no data-path collective inside the loop; the only exchange is the gather of the results.

The payload is tiny (196 B per candidate and iteration: <= 2048 candidates x 5 iterations = 2 MB), i.e. latency-bound:
everything a call produces travels in ONE padded `all_gather_into_tensor` of BYTES (so integer and float64 fields survive exactly), never a ring of point-to-point sends.
Shard sizes follow from (number of detections, world size, balance rule) alone, so no count exchange is needed on the
sharded-predictor path.
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


def init_distributed_mode(backend=None, force=None):
    """torchrun-style env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  One visible GPU per rank:
    the rank binds LOCAL_RANK's device (reference: cosypose/utils/distributed.py:55-69 uses SLURM vars + file store).
    COSY_DIST_BACKEND overrides the backend (RCCL refuses two ranks on one GPU; gloo does not).
    A single process normally needs no process group; force=True (or COSY_FORCE_DIST=1) creates a 1-rank group anyway,
    so that the collectives of this module run through RCCL on a 1-GPU box exactly as they do on 8."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if force is None:
        force = os.environ.get('COSY_FORCE_DIST', '0') == '1'
    if dist.is_initialized() or (world <= 1 and not force):
        return get_rank(), get_world_size()
    if world <= 1:
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
        os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 2000))
    backend = os.environ.get('COSY_DIST_BACKEND', backend)
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if torch.cuda.is_available():
        torch.cuda.set_device(local_device_index())
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend=backend, init_method='env://')
    return get_rank(), get_world_size()


def self_launch(n_ranks, argv=None):
    """`python bench.py --gpus N` without a launcher: when WORLD_SIZE is unset and N > 1 the script re-executes itself as N
    ranks under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1 at a free port) and returns the launcher's
    exit code; stdout / stderr pass through, so rank 0's single JSON line is the only line on stdout.  Returns None when
    nothing has to be launched (N <= 1, or this process already is a rank of a launched job).
    Replaces the reference's SLURM launch (cosypose/utils/distributed.py:55-69, job-runner/); rank order stays candidate
    order (cosypose/utils/tensor_collection.py:142-163)."""
    import socket
    import subprocess
    import sys
    if n_ranks <= 1 or 'WORLD_SIZE' in os.environ:
        return None
    argv = list(sys.argv if argv is None else argv)
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    backend = os.environ.get('COSY_DIST_BACKEND', 'nccl')
    if backend == 'nccl' and n_dev < n_ranks:
        raise SystemExit(f'--gpus {n_ranks}: {n_dev} GPU(s) visible and RCCL refuses several ranks on one device '
                         f'(rehearse with COSY_DIST_BACKEND=gloo, which time-slices them on the visible device(s))')
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: the only form the host driver supports
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_ranks}', '--master-addr', '127.0.0.1',
           '--master-port', str(port)] + argv
    return subprocess.call(cmd, env=env)


def process_group_info():
    """What the bench lines record about the process group: backend, world size and the collective library's version
    (torch.cuda.nccl.version() is RCCL's on ROCm), so that a record shows RCCL saw N ranks."""
    info = dict(backend=None, world_size=get_world_size(), rccl_version=None)
    if dist.is_available() and dist.is_initialized():
        info['backend'] = dist.get_backend()
        if info['backend'] == 'nccl':
            try:
                info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:      # noqa: BLE001 -- the version is a label, never a reason to fail a run
                info['rccl_version'] = f'unavailable ({type(e).__name__})'
    return info


# ---------------------------------------------------------------------------------------------
# partitioning
# ---------------------------------------------------------------------------------------------
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


def plan_shards(n, world_size, balance='contiguous', costs=None, counts=None):
    """Candidate indices of every rank (list of sorted int64 arrays), a pure function of its arguments -- every rank
    computes the same plan, so shard sizes never have to be communicated.
      'contiguous': np.array_split sizes in detection order (the default; rank order == detection order);
      'cost'      : greedy balance of `costs` (one per candidate; default 1 = equal counts);
      'counts'    : contiguous runs of the given per-rank sizes (a pre-sharded, possibly skewed mix as in config 3)."""
    if balance == 'contiguous':
        return [np.arange(*shard_range(n, r, world_size), dtype=np.int64) for r in range(world_size)]
    if balance == 'cost':
        costs = np.ones(n) if costs is None else np.asarray(costs, dtype=np.float64)
        assert len(costs) == n
        return balanced_assignment(costs, world_size)
    if balance == 'counts':
        counts = [int(c) for c in counts]
        assert len(counts) == world_size and sum(counts) == n, (counts, n)
        edges = np.concatenate([[0], np.cumsum(counts)])
        return [np.arange(edges[r], edges[r + 1], dtype=np.int64) for r in range(world_size)]
    raise ValueError(f'unknown balance rule {balance!r}')


# ---------------------------------------------------------------------------------------------
# the collective: padded all-gather of byte rows
# ---------------------------------------------------------------------------------------------
def _as_byte_rows(t):
    """(n, ...) tensor of any dtype -> (n, bytes per row) uint8 view/copy; well defined for n == 0 too."""
    n = t.shape[0]
    wb = int(np.prod(t.shape[1:], dtype=np.int64)) * t.element_size()
    if n == 0 or wb == 0:
        return torch.empty(n, wb, dtype=torch.uint8, device=t.device)
    return t.contiguous().reshape(n, -1).view(torch.uint8).reshape(n, wb)


def all_gather_rows(local, max_rows=None, counts=None):
    """local (n_r, ...) -> (sum n_r, ...) on every rank, rank order, in `local`'s own dtype (the rows travel as bytes).
    ONE collective when the per-rank `counts` are known a priori (sharded predictor) or when `max_rows`, an upper bound
    valid on every rank, is given (then each slab's first 8 bytes carry its row count as int64); otherwise the bound is
    agreed with one extra tiny all_reduce(MAX)."""
    world = get_world_size()
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local
    tail = tuple(local.shape[1:])
    rows = _as_byte_rows(local)
    n, wb = rows.shape
    if wb == 0:             # nothing to move (a row schema without fields): no collective
        total = int(sum(counts)) if counts is not None else None
        assert total is not None, 'all_gather_rows: zero-width rows need the per-rank counts'
        return local.new_zeros((total,) + tail)
    if counts is not None:
        assert counts[get_rank()] == n
        max_rows = max(max(counts), 1)
    elif max_rows is None:
        m = torch.tensor([n], device=local.device, dtype=torch.int64)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        max_rows = max(int(m.item()), 1)
    assert n <= max_rows
    head = 8 if counts is None else 0
    # the slab's padding is never read back (only counts[r] rows of rank r are), so it is not cleared
    slab = torch.empty(head + max_rows * wb, device=local.device, dtype=torch.uint8)
    if head:
        slab[:8] = torch.tensor([n], dtype=torch.int64).view(torch.uint8).to(local.device, non_blocking=True)
    if n:
        slab[head:head + n * wb] = rows.reshape(-1)
    out = torch.empty(world * slab.numel(), device=local.device, dtype=torch.uint8)
    if local.is_cuda and dist.get_backend() == 'gloo':
        # rehearsal only (several ranks on one GPU): gloo stages device tensors through the host; entering it with a deep
        # queue of kernels from 3+ processes time-slicing one device was measured at seconds per call.  RCCL is stream-ordered.
        torch.cuda.current_stream().synchronize()
    dist.all_gather_into_tensor(out, slab)
    out = out.view(world, -1)
    if counts is None:
        counts = out[:, :8].contiguous().view(torch.int64).reshape(world).tolist()
    parts = [out[r, head:head + c * wb] for r, c in enumerate(counts)]
    flat = torch.cat(parts) if parts else out.new_zeros(0)
    total = int(sum(counts))
    return flat.view(local.dtype).reshape((total,) + tail) if total else local.new_zeros((0,) + tail)


class RowPacker:
    """Several per-candidate tensors <-> one (n, row_bytes) uint8 matrix: the payload of the single all-gather.
    `schema` = [(name, trailing shape, dtype)], identical on every rank (an empty rank packs zero rows of it)."""

    def __init__(self, schema):
        self.schema = [(name, tuple(shape), dtype) for name, shape, dtype in schema]
        self.widths = [int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size() for _, shape, dtype in self.schema]
        self.row_bytes = int(sum(self.widths))

    @classmethod
    def of(cls, tensors):
        return cls([(k, tuple(v.shape[1:]), v.dtype) for k, v in tensors.items()])

    def pack(self, tensors, n, device):
        out = torch.empty(n, self.row_bytes, dtype=torch.uint8, device=device)
        off = 0
        for (name, shape, dtype), wb in zip(self.schema, self.widths):
            if n:
                t = tensors[name]
                assert tuple(t.shape) == (n,) + shape and t.dtype == dtype, (name, t.shape, t.dtype)
                out[:, off:off + wb] = _as_byte_rows(t)
            off += wb
        return out

    def unpack(self, rows):
        n = rows.shape[0]
        out, off = {}, 0
        for (name, shape, dtype), wb in zip(self.schema, self.widths):
            out[name] = rows[:, off:off + wb].contiguous().view(dtype).reshape((n,) + shape)
            off += wb
        return out


def gather_collection(coll):
    """PandasTensorCollection on each rank -> concatenation in rank order on every rank (what the reference's
    gather_distributed returns on rank 0).  Two steps: the small `infos` DataFrames and the tensor schema travel through
    one all_gather_object (host side; this also tells every rank every count), then ALL tensors of the collection in one
    packed byte all-gather.  A rank with zero predictions -- tc.concatenate([]) has no tensors at all -- takes the schema
    from the other ranks and contributes zero rows, so the collectives always match."""
    from . import tensor_collection as tc
    import pandas as pd
    world = get_world_size()
    if world == 1:
        return tc.concatenate([coll])
    schema = [(k, tuple(v.shape[1:]), v.dtype) for k, v in coll.tensors.items()]
    device = coll.device if coll.tensors else None
    meta = [None] * world
    dist.all_gather_object(meta, (coll.infos, schema))
    infos_all = [m[0] for m in meta]
    counts = [len(i) for i in infos_all]
    schemas = [m[1] for m, c in zip(meta, counts) if c > 0 and m[1]]
    infos = (pd.concat([i for i in infos_all if len(i) > 0], axis=0, sort=False).reset_index(drop=True)
             if any(counts) else pd.DataFrame())
    if not schemas:
        return tc.PandasTensorCollection(infos=infos)
    assert all(s == schemas[0] for s in schemas), 'ranks disagree on the tensors of the collection'
    packer = RowPacker(schemas[0])
    if device is None:      # an empty rank: any device the backend accepts
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    rows = packer.pack(coll.tensors, len(coll.infos), device)
    full = all_gather_rows(rows, counts=counts)
    return tc.PandasTensorCollection(infos=infos, **packer.unpack(full))


# ---------------------------------------------------------------------------------------------
# the sharded driver of the hot path (BASELINE configs[1..3])
# ---------------------------------------------------------------------------------------------
def _subset_frames(images, K, im_ids):
    """Only the frames this shard references, with the shard's frame ids renumbered (the per-rank frame conversion and
    HBM footprint then scale with the shard, not with the whole job)."""
    uniq, inv = np.unique(np.asarray(im_ids, dtype=np.int64), return_inverse=True)
    if len(uniq) == images.shape[0]:
        return images, K, np.asarray(im_ids)
    sel = torch.as_tensor(uniq, device=images.device)
    return images[sel], K[sel], inv.astype(np.asarray(im_ids).dtype)


def _trace_begin():
    """COSY_SHARD_TRACE=1: wall-clock split of a sharded call on stderr (synchronises the device: diagnostics only)"""
    if not os.environ.get('COSY_SHARD_TRACE'):
        return None
    import time
    torch.cuda.synchronize()
    return time.perf_counter()


def _trace_mark(t0, what, rank):
    if t0 is None:
        return None
    import sys
    import time
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f'[shard trace] rank {rank}: {what} {1e3 * (t1 - t0):.1f} ms', file=sys.stderr, flush=True)
    return t1


def run_shard(predictor, images, K, table, ids, **kwargs):
    """predictor.get_predictions on the candidates `ids` of the global table (`detections` or `data_TCO_init`)."""
    sub = table[np.asarray(ids, dtype=np.int64)]
    if len(ids):
        imgs, Ks, local_im = _subset_frames(images, K, sub.infos['batch_im_id'].values)
        infos = sub.infos.copy()
        infos['batch_im_id'] = local_im
        sub = type(sub)(infos, **sub.tensors)
    else:
        imgs, Ks = images[:0], K[:0]
    key = 'data_TCO_init' if 'poses' in sub.tensors else 'detections'
    return predictor.get_predictions(imgs, Ks, **{key: sub}, **kwargs)


def get_predictions_sharded(predictor, images, K, detections=None, data_TCO_init=None, n_coarse_iterations=1,
                            n_refiner_iterations=1, balance='contiguous', costs=None, counts=None, rank=None, world_size=None,
                            gather_rows=None):
    """CoarseRefinePosePredictor.get_predictions over ALL ranks: every rank passes the same global table (detections or
    data_TCO_init), runs its shard, and receives the full result -- (final collection, {'stage/iteration=k': collection})
    in the ORIGINAL candidate order, identical (bit for bit) to what a single process computes, because every candidate
    is independent of its batch.  Exactly ONE collective per call: a padded byte all-gather of
    [poses | poses_input | K_crop | boxes_rend | boxes_crop] x every stage/iteration (196 B per candidate and iteration).
    `infos` never travel: they are rebuilt from the global table.

    balance: 'contiguous' (detection order), 'cost' (greedy on `costs`, e.g. crops per frame size) or 'counts' (given
    per-rank run lengths: a pre-sharded skewed mix).  `gather_rows(local_rows, counts)` can replace the collective
    (tests simulate N ranks in one process with it)."""
    from . import tensor_collection as tc
    table = detections if data_TCO_init is None else data_TCO_init
    assert table is not None
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world_size is None else world_size
    n = len(table)
    plan = plan_shards(n, world, balance, costs=costs, counts=counts)
    shard_counts = [len(p) for p in plan]
    kw = dict(n_coarse_iterations=n_coarse_iterations, n_refiner_iterations=n_refiner_iterations)
    _t0 = _trace_begin()
    final, preds = run_shard(predictor, images, K, table, plan[rank], **kw)
    _t1 = _trace_mark(_t0, 'run_shard', rank)
    # what a call produces is known from the arguments alone (an empty rank has no collections to inspect)
    keys = ([f'coarse/iteration={i}' for i in range(1, n_coarse_iterations + 1)] if data_TCO_init is None else []) + \
           [f'refiner/iteration={i}' for i in range(1, n_refiner_iterations + 1)]
    if not keys:            # external coarse poses and no refiner iteration: nothing was computed, nothing to exchange
        return data_TCO_init, {'external_coarse': data_TCO_init}
    fields = (('poses', (4, 4)), ('poses_input', (4, 4)), ('K_crop', (3, 3)), ('boxes_rend', (4,)), ('boxes_crop', (4,)))
    packer = RowPacker([(f'{k}|{f}', shape, torch.float32) for k in keys for f, shape in fields])
    device = images.device
    mine = len(plan[rank])
    local = {f'{k}|{f}': getattr(preds[k], f) for k in keys for f, _ in fields} if mine else {}
    rows = packer.pack(local, mine, device)
    if gather_rows is not None:
        full = gather_rows(rows, shard_counts)
    else:
        full = all_gather_rows(rows, counts=shard_counts)
    _trace_mark(_t1, 'pack+all_gather', rank)
    # rank order -> original candidate order
    order = np.concatenate(plan) if n else np.zeros(0, np.int64)
    if n and not np.array_equal(order, np.arange(n)):
        inv = np.empty(n, np.int64); inv[order] = np.arange(n)
        full = full[torch.as_tensor(inv, device=full.device)]
    got = packer.unpack(full)
    out = {}
    for k in keys:
        out[k] = tc.PandasTensorCollection(table.infos, **{f: got[f'{k}|{f}'] for f, _ in fields}) if n else \
            tc.PandasTensorCollection(infos=table.infos)
    if data_TCO_init is not None:
        out['external_coarse'] = data_TCO_init
    last = keys[-1] if keys else 'external_coarse'
    return out[last], out


def get_predictions_sharded_scenes(predictor, scenes, n_coarse_iterations=1, n_refiner_iterations=1, balance='contiguous', costs=None,
                                   counts=None, rank=None, world_size=None, gather_rows=None):
    """The sharded driver for a MIX of frame sizes (BASELINE configs[3]: seven datasets with different cameras): `scenes` is
    a list of (images (N_g,3,h_g,w_g), K (N_g,3,3), table_g) -- one entry per frame size, `table_g` a detection table or a
    data_TCO_init table whose batch_im_id index that group's frames.  The candidates of all groups form ONE global list
    (group after group) that is partitioned by `plan_shards`; a rank runs its candidates group by group and the refined
    poses of everything come back with ONE byte all-gather.  Returns (poses (D,4,4) in global candidate order, the plan)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world_size is None else world_size
    sizes = [len(t) for _, _, t in scenes]
    n = int(sum(sizes))
    plan = plan_shards(n, world, balance, costs=costs, counts=counts)
    mine = plan[rank]
    edges = np.concatenate([[0], np.cumsum(sizes)])
    kw = dict(n_coarse_iterations=n_coarse_iterations, n_refiner_iterations=n_refiner_iterations)
    device = scenes[0][0].device
    parts = []
    for g, (images, K, table) in enumerate(scenes):
        ids = mine[(mine >= edges[g]) & (mine < edges[g + 1])] - edges[g]
        if len(ids) == 0:
            continue
        final, _ = run_shard(predictor, images, K, table, ids, **kw)
        parts.append(final.poses.reshape(len(ids), 16))
    local = torch.cat(parts) if parts else torch.zeros(0, 16, device=device)
    shard_counts = [len(p) for p in plan]
    rows = _as_byte_rows(local.float())
    full = gather_rows(rows, shard_counts) if gather_rows is not None else all_gather_rows(rows, counts=shard_counts)
    order = np.concatenate(plan) if n else np.zeros(0, np.int64)
    poses = full.contiguous().view(torch.float32).reshape(-1, 4, 4)
    if n and not np.array_equal(order, np.arange(n)):
        inv = np.empty(n, np.int64); inv[order] = np.arange(n)
        poses = poses[torch.as_tensor(inv, device=poses.device)]
    return poses, plan
