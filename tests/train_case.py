"""The training batch behind tests/golden/reference_golden_train.npz, rebuilt from seeds (inputs are not stored in
the fixture): network input x (crop + render), K_crop, TCO_input, possible ground truths and loss points."""
import numpy as np

from cosypose_amd import synthetic as syn


def build(oracle, golden_train, mesh_table, B=4, render_size=(240, 320)):
    frames, K, TCO_gt, obj = syn.make_training_batch(61, B)
    images = frames.astype(np.float32) / np.float32(255.)
    mesh_points = syn.make_mesh_points(7, 21, 2500)
    TCO_init = oracle.tco_init_from_boxes(golden_train['tr_bboxes'], K, z=1.0)
    out = oracle.pose_predictor_forward(images, K, obj, TCO_init, mesh_table, None, lambda n, T, Kc: syn.make_renders(900 + n, B, *render_size),
                                        n_iterations=1, render_size=render_size, backbone=lambda x: (None, np.zeros((B, 9), np.float32)))
    it = out['iteration=1']
    rend = syn.make_renders(900, B, *render_size)
    x = np.concatenate([it['images_crop'], rend], 1)
    points = mesh_points[obj][:, golden_train['tr_point_ids']]
    gt = TCO_gt[:, None].copy()                                  # symmetries = identity (n_sym = 1)
    return dict(frames=frames, images=images, K=K, TCO_gt=TCO_gt, obj=obj, bboxes=golden_train['tr_bboxes'], TCO_init=TCO_init,
                x=x, K_crop=it['K_crop'], TCO_input=it['TCO_input'], points=points, gt=gt, mesh_points=mesh_points)
