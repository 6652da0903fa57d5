"""Worker of test_self_launch_spawns_its_ranks: the launch logic of bench.py / bench_train.py (`python script.py --gpus N` with no launcher around
it) on CPU -- the script re-executes itself as N ranks through cosypose_amd.distributed.self_launch, every rank joins a gloo group, the ranks'
ids are all-gathered and rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from cosypose_amd.distributed import self_launch, init_distributed_mode, all_gather_rows, process_group_info
    rc = self_launch(args.gpus)
    if rc is not None:
        raise SystemExit(rc)
    rank, world = init_distributed_mode('gloo')
    assert world == args.gpus, (world, args.gpus)
    rows = torch.full((rank + 1, 2), float(rank))           # ragged shares: rank r contributes r + 1 rows
    got = all_gather_rows(rows, counts=[r + 1 for r in range(world)])
    if rank == 0:
        print(json.dumps(dict(n_gpus=world, rows=got[:, 0].tolist(), process_group=process_group_info(), launched='WORLD_SIZE' in os.environ)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
