"""CoarseRefinePosePredictor: the driver of the hot path, with the interface of the reference's
cosypose/integrated/pose_predictor.py:14-107 -- batched_model_predictions, make_TCO_init, get_predictions -- returning the
same PandasTensorCollections under the same keys ('coarse/iteration=k', 'refiner/iteration=k', 'external_coarse').

MI355X-first differences, invisible to callers: detections are processed in chunks of `bsz_objects` in their given order as
in the reference (:30-33), but the frames are handed over ONCE and indexed per object on the device (the reference
replicates them with images[im_ids], :41); consecutive-row chunks are tensor views; the default of 64 objects per chunk is
kept for drop-in parity and can be raised (288 GB of HBM holds thousands of crops in flight).
"""
import numpy as np
import torch

from . import lib3d
from . import tensor_collection as tc

# per-iteration tensors of PosePredictor.forward's output that travel in the result collections: field -> output key
_ITERATION_FIELDS = (('poses', 'TCO_output'), ('poses_input', 'TCO_input'), ('K_crop', 'K_crop'),
                     ('boxes_rend', 'boxes_rend'), ('boxes_crop', 'boxes_crop'))


def _iteration_key(n):
    return f'iteration={n}'


class CoarseRefinePosePredictor(torch.nn.Module):
    def __init__(self, coarse_model=None, refiner_model=None, bsz_objects=64):
        super().__init__()
        self.coarse_model = coarse_model
        self.refiner_model = refiner_model
        self.bsz_objects = bsz_objects
        self.eval()

    @torch.no_grad()
    def batched_model_predictions(self, model, images, K, obj_data, n_iterations=1):
        """Run `model` over obj_data (infos[label, batch_im_id], poses) chunk by chunk -> {'iteration=k': collection}."""
        per_iteration = {_iteration_key(n): [] for n in range(1, n_iterations + 1)}
        n_objects = len(obj_data)
        for first in range(0, n_objects, self.bsz_objects):
            chunk = obj_data[np.arange(first, min(first + self.bsz_objects, n_objects))]
            outputs = model(images=images, K=K, TCO=chunk.poses, n_iterations=n_iterations,
                            labels=chunk.infos['label'].values, im_ids=chunk.infos['batch_im_id'].values)
            for key, collected in per_iteration.items():
                fields = {name: outputs[key][source] for name, source in _ITERATION_FIELDS}
                collected.append(tc.PandasTensorCollection(chunk.infos, **fields))
        return {key: tc.concatenate(parts) for key, parts in per_iteration.items()}

    def make_TCO_init(self, detections, K):
        """Initial poses from 2D boxes: 'v0' (identity rotation at 1 m) or 'z-up+auto-depth' (cosypose_ops.py:121-173)."""
        im_ids = detections.infos['batch_im_id'].values
        if self.coarse_model.cfg.init_method == 'z-up+auto-depth':
            mesh_db = self.coarse_model.mesh_db
            obj_ids = mesh_db.object_ids(detections.infos['label'], detections.bboxes.device)
            TCO_init = lib3d.TCO_init_from_boxes_zup_autodepth(detections.bboxes, mesh_db.point_table(2000), obj_ids, K, im_ids=im_ids)
        else:
            TCO_init = lib3d.TCO_init_from_boxes(z_range=(1.0, 1.0), boxes=detections.bboxes, K=K, im_ids=im_ids)
        return tc.PandasTensorCollection(infos=detections.infos, poses=TCO_init)

    def get_predictions(self, images, K, detections=None, data_TCO_init=None,
                        n_coarse_iterations=1, n_refiner_iterations=1):
        preds = dict()

        def run_stage(stage, model, start, n_iterations):
            out = self.batched_model_predictions(model, images, K, start, n_iterations=n_iterations)
            for n in range(1, n_iterations + 1):
                preds[f'{stage}/{_iteration_key(n)}'] = out[_iteration_key(n)]
            return out[_iteration_key(n_iterations)]

        if data_TCO_init is None:
            # coarse estimate from detections
            assert detections is not None
            assert self.coarse_model is not None
            assert n_coarse_iterations > 0
            data_TCO = run_stage('coarse', self.coarse_model, self.make_TCO_init(detections, K), n_coarse_iterations)
        else:
            # externally provided coarse poses
            assert n_coarse_iterations == 0
            data_TCO = preds['external_coarse'] = data_TCO_init

        if n_refiner_iterations >= 1:
            assert self.refiner_model is not None
            data_TCO = run_stage('refiner', self.refiner_model, data_TCO, n_refiner_iterations)
        return data_TCO, preds
