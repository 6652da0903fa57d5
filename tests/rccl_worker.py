"""Worker of test_rccl_single_rank_collectives: a 1-rank process group on the `nccl` backend (= RCCL on ROCm), so that the
module's collectives -- the padded byte all-gather of the sharded predictor and the gradient all-reduce -- execute as RCCL
kernels on the one GPU of the test box exactly as they do on 8 (reference: cosypose/utils/tensor_collection.py:142-163,
cosypose/utils/distributed.py:55-69).  Prints RCCL_OK on success."""
import argparse
import os
import sys

import numpy as np
import torch


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import pandas as pd
    import torch.distributed as dist
    from cosypose_amd import distributed as D, synthetic as syn, tensor_collection as tc, train_engine
    from cosypose_amd.mesh_db import BatchedMeshes
    from cosypose_amd.pose_models_cfg import create_model_pose, check_update_config
    from cosypose_amd.pose_predictor import CoarseRefinePosePredictor
    os.environ['COSY_FORCE_DIST'] = '1'
    os.environ.pop('COSY_DIST_BACKEND', None)
    rank, world = D.init_distributed_mode()
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == 'nccl', (rank, world, dist.get_backend())
    dev = torch.device('cuda', 0)
    # 1. the byte all-gather: a-priori counts (no header) and the max_rows form (8-byte count header), mixed dtypes as bytes
    rows = torch.arange(7 * 196, dtype=torch.int64, device=dev).reshape(7, 196).to(torch.uint8)
    assert torch.equal(D.all_gather_rows(rows, counts=[7]), rows)
    assert torch.equal(D.all_gather_rows(rows, max_rows=16), rows)
    f = torch.randn(5, 4, 4, device=dev, dtype=torch.float64)
    assert torch.equal(D.all_gather_rows(f), f)
    assert D.all_gather_rows(f[:0], counts=[0]).shape == (0, 4, 4)
    # 2. the gradient all-reduce (config 4's DDP averaging) through RCCL
    g = torch.randn(10_711_145, device=dev)
    want = g.clone()
    train_engine.allreduce_gradients(g, force=True)
    assert torch.equal(g, want)
    # 3. the sharded predictor path end to end: same result as the plain call, bit for bit
    n_obj = 21
    labels = np.array([f'obj_{i:06d}' for i in range(1, n_obj + 1)])
    pts = syn.make_mesh_points(7, n_obj, 2500)
    mesh_db = BatchedMeshes({l: dict(label=l, n_points=2500, n_sym=1) for l in labels}, labels, torch.from_numpy(pts),
                            torch.eye(4).reshape(1, 1, 4, 4).repeat(n_obj, 1, 1, 1)).float().cuda()

    class R:
        def __init__(self): self.calls = 0
        def render(self, obj_infos, TCO, K, resolution):
            r = syn.make_renders(500 + self.calls, len(obj_infos), *resolution); self.calls += 1
            return torch.from_numpy(r).cuda()
    cfg = check_update_config(argparse.Namespace(backbone_str='efficientnet-b3', n_pose_dims=9, init_method='v0'))
    model = create_model_pose(cfg, R(), mesh_db)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in syn.golden_state_dict(0).items()}, strict=False)
    model.cfg = cfg
    model = model.cuda().eval()
    model.compute_dtype = 'fp16'
    D_, h, w = 11, 480, 640
    obj, im, boxes = syn.make_detections(3, D_, 2, n_obj, h, w)
    det = tc.PandasTensorCollection(infos=pd.DataFrame(dict(label=labels[obj], batch_im_id=im, score=1.0)), bboxes=torch.from_numpy(boxes).cuda())
    frames = torch.from_numpy(syn.make_frames(4, 2, h, w)).cuda(); K = torch.from_numpy(syn.make_K(2, h, w)).cuda()
    pred = CoarseRefinePosePredictor(coarse_model=model, refiner_model=model, bsz_objects=8)
    model.renderer = R()
    f1, a1 = pred.get_predictions(frames, K, detections=det, n_coarse_iterations=1, n_refiner_iterations=2)
    model.renderer = R()
    f2, a2 = D.get_predictions_sharded(pred, frames, K, detections=det, n_coarse_iterations=1, n_refiner_iterations=2)
    assert set(a1) == set(a2)
    for k in a1:
        for fld in ('poses', 'poses_input', 'K_crop', 'boxes_rend', 'boxes_crop'):
            assert torch.equal(getattr(a1[k], fld), getattr(a2[k], fld)), (k, fld)
    assert torch.equal(f1.poses, f2.poses)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print('RCCL_OK')


if __name__ == '__main__':
    main()
