"""h_pose: forward + loss of one training batch, same surface as the reference's
cosypose/training/pose_forward_loss.py:17-84 (the function train_pose.py:317-331 differentiates).

`data` carries images (B,3,h,w) uint8, K, TCO (ground truth), objects [{'name': label}], bboxes; `cfg` carries
n_points_loss, loss_disentangled, n_pose_dims, init_method.  The returned loss is a scalar attached to the model's
parameters through the HIP training engine (cosypose_amd.train_engine), so `loss.backward()` works as in the reference.
Supported: the disentangled loss with 9-d pose outputs (every released CosyPose model); the ablation losses
(quaternion outputs, plain ADD-L1) are not built.
"""
import numpy as np
import torch

from . import lib3d, train_engine


def cast(obj):
    return obj.cuda(non_blocking=True)


def add_noise(TCO, euler_deg_std=(15, 15, 15), trans_std=(0.01, 0.01, 0.05)):
    """Random rigid perturbation of the poses (cosypose/lib3d/transform_ops.py:35-51): R <- R R_noise(euler 'sxyz'),
    t <- t + N(0, trans_std).  Host-side numpy RNG, drawn in the reference's order."""
    TCO_out = TCO.clone()
    bsz = TCO.shape[0]
    eul = np.concatenate([np.random.normal(loc=0, scale=s, size=bsz)[:, None] for s in euler_deg_std], axis=1) * np.pi / 180
    R = np.zeros((bsz, 3, 3), np.float64)
    for b, (ai, aj, ak) in enumerate(eul):          # transforms3d.euler.euler2mat(ai, aj, ak), default 'sxyz'
        ci, si, cj, sj, ck, sk = np.cos(ai), np.sin(ai), np.cos(aj), np.sin(aj), np.cos(ak), np.sin(ak)
        R[b] = [[cj * ck, sj * si * ck - ci * sk, sj * ci * ck + si * sk],
                [cj * sk, sj * si * sk + ci * ck, sj * ci * sk - si * ck],
                [-sj, cj * si, cj * ci]]
    R = torch.as_tensor(R, dtype=TCO.dtype).to(TCO.device)
    t = np.concatenate([np.random.normal(loc=0, scale=s, size=bsz)[:, None] for s in trans_std], axis=1)
    t = torch.as_tensor(t, dtype=TCO.dtype).to(TCO.device)
    TCO_out[:, :3, :3] = TCO_out[:, :3, :3] @ R
    TCO_out[:, :3, 3] += t
    return TCO_out


def _initial_poses(input_generator, cfg, TCO_possible_gt, bboxes, points, K):
    """TCO_init of the training batch for the reference's three input generators (pose_forward_loss.py:32-43)."""
    if input_generator == 'fixed':
        return lib3d.TCO_init_from_boxes(z_range=(1.0, 1.0), boxes=bboxes, K=K)
    if input_generator == 'gt+noise':
        return add_noise(TCO_possible_gt[:, 0], euler_deg_std=[15, 15, 15], trans_std=[0.01, 0.01, 0.05])
    if input_generator == 'fixed+trans_noise':
        assert cfg.init_method == 'z-up+auto-depth'
        rows = torch.arange(bboxes.shape[0], dtype=torch.int32, device=bboxes.device)      # points are already per sample
        TCO_init = lib3d.TCO_init_from_boxes_zup_autodepth(bboxes, points, rows, K)
        return add_noise(TCO_init, euler_deg_std=[0, 0, 0], trans_std=[0.01, 0.01, 0.05])
    raise ValueError('Unknown input generator', input_generator)


def h_pose(model, mesh_db, data, meters, cfg, n_iterations=1, input_generator='fixed'):
    if not (cfg.loss_disentangled and cfg.n_pose_dims == 9):
        raise ValueError('only the disentangled loss on 9-d pose outputs is built (cfg.loss_disentangled, n_pose_dims=9)')
    # batch -> device (uint8 frames to [0,1] floats as in the reference)
    # the reference: images = cast(data.images).float() / 255. (pose_forward_loss.py:24).  uint8 frames go to the model as they are: its frame
    # conversion kernel computes value / 255.f itself (the same arithmetic, without two fp32 passes over 59 MB of frames)
    images = cast(data.images)
    if images.dtype != torch.uint8:
        images = images.float() / 255.
    K, TCO_gt, bboxes = cast(data.K).float(), cast(data.TCO).float(), cast(data.bboxes).float()
    labels = np.array([obj['name'] for obj in data.objects])

    meshes = mesh_db.select(labels)
    points = meshes.sample_points(cfg.n_points_loss, deterministic=False)      # global numpy RNG, as in the reference
    TCO_possible_gt = TCO_gt.unsqueeze(1) @ meshes.symmetries                  # every symmetric copy of the ground truth
    TCO_init = _initial_poses(input_generator, cfg, TCO_possible_gt, bboxes, points, K)

    # Through the wrapper when the model is DistributedDataParallel (as train_pose.py:246 wraps it): DDP.forward is what
    # arms the reducer for this backward -- calling model.module directly would skip the gradient all-reduce and the
    # buffer broadcast without any error.  The network is ONE autograd node whose inputs are the leaf parameters, so the
    # reducer's per-parameter hooks fire as usual.
    outputs = model(images=images, K=K, labels=labels, TCO=TCO_init, n_iterations=n_iterations)

    per_iteration = []
    for n in range(1, n_iterations + 1):
        it = outputs[f'iteration={n}']
        loss_n = train_engine.loss_refiner_CO_disentangled(TCO_possible_gt=TCO_possible_gt, TCO_input=it['TCO_input'],
                                                           refiner_outputs=it['model_outputs']['pose'], K_crop=it['K_crop'],
                                                           points=points)
        per_iteration.append(loss_n)

    loss = torch.cat(per_iteration).mean()
    # the meters take the same numbers as in the reference (pose_forward_loss.py:74-83: .item() per iteration, then twice on the total).  Plain
    # meters: read back with ONE device synchronisation instead of n_iterations + 2.  Meters with a `defer` method (training.LazyMeters): no
    # synchronisation at all -- the values are added when their copy has arrived, and the host goes straight on to enqueue the backward pass.
    values = torch.stack([l.mean() for l in per_iteration] + [loss.detach()])
    names = [(f'loss_TCO-iter={n}',) for n in range(1, n_iterations + 1)] + [('loss_TCO', 'loss_total')]
    if hasattr(meters, 'defer') and values.is_cuda:
        meters.defer(names, values)
    else:
        for keys, v in zip(names, values.tolist()):
            for k in keys:
                meters[k].add(v)
    return loss
