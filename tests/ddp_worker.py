"""Worker of test_ddp_two_ranks_on_one_gpu: h_pose through a DistributedDataParallel wrapper (as train_pose.py:246 wraps the
model), each rank on its own batch; reports a checksum of every gradient after backward."""
import argparse
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch


def main(rank, world, port, q):
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
        import torch.distributed as dist
        from cosypose_amd import synthetic as syn, pose_forward_loss as pfl
        from cosypose_amd.pose_models_cfg import create_model_refiner
        from cosypose_amd.mesh_db import BatchedMeshes
        dist.init_process_group('gloo', init_method='env://')
        torch.cuda.set_device(0)
        n_obj = 21
        labels = np.array([f'obj_{i:06d}' for i in range(1, n_obj + 1)])
        pts = syn.make_mesh_points(7, n_obj, 2500)
        infos = {l: dict(label=l, n_points=2500, n_sym=1) for l in labels}
        mesh_db = BatchedMeshes(infos, labels, torch.from_numpy(pts), torch.eye(4).reshape(1, 1, 4, 4).repeat(n_obj, 1, 1, 1)).float().cuda()

        class R:
            def render(self, obj_infos, TCO, K, resolution):
                return torch.from_numpy(syn.make_renders(900 + rank, len(obj_infos), *resolution)).cuda()
        cfg = argparse.Namespace(backbone_str='efficientnet-b3', n_pose_dims=9, init_method='v0', n_points_loss=600, loss_disentangled=True)
        model = create_model_refiner(cfg, R(), mesh_db)
        model.load_state_dict({k: torch.as_tensor(v) for k, v in syn.golden_state_dict(0).items()}, strict=True)
        model = model.cuda().train()
        model.drop_connect_rate = 0.0
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0) if world > 1 else model
        B = 2
        frames, K, TCO, obj = syn.make_training_batch(61 + rank, B)
        rs = np.random.RandomState(5 + rank)
        xy = rs.uniform(150, 300, (B, 2)); wh = rs.uniform(80, 160, (B, 2))
        data = types.SimpleNamespace(images=torch.from_numpy(frames), K=torch.from_numpy(K), TCO=torch.from_numpy(TCO),
                                     objects=[dict(name=l) for l in labels[obj]],
                                     bboxes=torch.from_numpy(np.concatenate([xy, xy + wh], 1).astype(np.float32)))

        class M:
            def add(self, v): pass
        np.random.seed(7)
        loss = pfl.h_pose(model=ddp, mesh_db=mesh_db, data=data, meters=defaultdict(M), cfg=cfg, n_iterations=1, input_generator='fixed')
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().double().cpu().numpy() for n, p in model.named_parameters()}
        report = {n: (float(g.sum()), float(np.abs(g).sum())) for n, g in grads.items()}
        # train_loop(DDP(model)) with its default optimizer (FlatAdam): two steps on rank-specific batches.  FlatAdam's direct gradient path
        # writes into its flat buffer and returns None per parameter -- under DDP that would skip the averaging hooks; train_loop must build
        # it with direct_grads=False, and the ranks must end with identical weights (round 4's advisor finding).
        if world > 1:
            from cosypose_amd import training, train_engine
            tcfg = argparse.Namespace(**vars(cfg), lr=3e-4, weight_decay=0.0, n_epochs_warmup=0, lr_epoch_decay=10, clip_grad_norm=0.5, n_iterations=1)
            for p_ in model.parameters():
                p_.grad = None
            hist = training.train_loop(ddp, mesh_db, tcfg, lambda e: [data, data], n_epochs=1, prefetch=False, lazy_meters=False)
            opt = model.__dict__['_cosy_flat_adam']
            assert isinstance(opt, train_engine.FlatAdam) and not opt.direct_grads
            try:
                training.train_loop(ddp, mesh_db, tcfg, lambda e: [data], n_epochs=1, optimizer=train_engine.FlatAdam(model, direct_grads=True))
                report['__refused__'] = (0.0, 0.0)
            except ValueError:
                report['__refused__'] = (1.0, 1.0)
            # bucketed all-reduce launched from inside the backward (FlatAdam.overlap_allreduce) == one all-reduce behind it, bit for bit at 2 ranks
            flat = []
            for overlap in (False, True):
                m2 = create_model_refiner(cfg, R(), mesh_db)
                m2.load_state_dict({k: torch.as_tensor(v) for k, v in syn.golden_state_dict(0).items()}, strict=True)
                m2 = m2.cuda().train()
                m2.drop_connect_rate = 0.0
                o2 = train_engine.FlatAdam(m2, overlap_allreduce=overlap, bucket_bytes=4 << 20)
                o2.zero_grad()
                np.random.seed(7)
                l2 = pfl.h_pose(model=m2, mesh_db=mesh_db, data=data, meters=defaultdict(M), cfg=cfg, n_iterations=1, input_generator='fixed')
                l2.backward()
                assert o2.reduced == overlap
                train_engine.allreduce_gradients(o2)          # the plain path reduces here; the overlapped one has nothing left to do
                torch.cuda.synchronize()
                flat.append(o2.grad.clone())
            report['__overlap_equal__'] = (1.0 if torch.equal(flat[0], flat[1]) else 0.0, float(flat[1].abs().sum()))
            torch.cuda.synchronize()
            w = torch.cat([p_.detach().reshape(-1) for p_ in model.parameters()]).double()
            report['__weights__'] = (float(w.sum()), float(w.abs().sum()))
            report['__loss__'] = (float(hist[0]), 1.0)
        q.put((rank, float(loss.item()), report))
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        traceback.print_exc()
        q.put((rank, None, None))
        raise
