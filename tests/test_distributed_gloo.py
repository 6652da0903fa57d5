"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding + single-collective
gather used by bench.py / gather_distributed (RCCL on the GPU box, gloo here)."""
import os
import sys
import pathlib
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = pathlib.Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import pandas as pd
    from cosypose_amd.distributed import init_distributed_mode, shard_range, all_gather_rows, get_rank, get_world_size
    from cosypose_amd import tensor_collection as tc
    init_distributed_mode('gloo')
    assert (get_rank(), get_world_size()) == (rank, world)
    D = 11                                               # ragged: 6 + 5
    poses = torch.arange(D * 16, dtype=torch.float32).reshape(D, 4, 4)
    s, e = shard_range(D)
    got = all_gather_rows(poses[s:e], max_rows=6)        # one collective, bound known a priori
    ok = torch.equal(got, poses)
    skew = poses[:3] if rank == 0 else poses[3:]         # load-imbalanced shares, bound agreed on the fly
    ok &= torch.equal(all_gather_rows(skew), poses)
    ok &= all_gather_rows(poses[:0] if rank == 0 else poses).shape[0] == D   # an empty rank
    infos = pd.DataFrame(dict(label=[f'o{i}' for i in range(s, e)], batch_im_id=list(range(s, e))))
    coll = tc.PandasTensorCollection(infos, poses=poses[s:e], boxes_crop=torch.ones(e - s, 4) * rank)
    full = coll.gather_distributed(tmp_dir=None)
    ok &= list(full.infos['label']) == [f'o{i}' for i in range(D)] and torch.equal(full.poses, poses)
    ok &= full.boxes_crop[:, 0].tolist() == [0.0] * 6 + [1.0] * 5
    # training: DDP's gradient averaging as one all-reduce of the flat gradient buffer
    from cosypose_amd.train_engine import allreduce_gradients
    gflat = torch.full((1000,), float(rank + 1))
    allreduce_gradients(gflat)
    ok &= bool(torch.allclose(gflat, torch.full((1000,), 1.5)))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
