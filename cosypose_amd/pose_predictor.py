"""CoarseRefinePosePredictor: the driver of the hot path, same surface as the reference's
cosypose/integrated/pose_predictor.py:14-107 (batched_model_predictions, make_TCO_init,
get_predictions), returning the same PandasTensorCollections under the same keys
('coarse/iteration=k', 'refiner/iteration=k', 'external_coarse').

MI355X-first: detections are chunked by `bsz_objects` in order as in the reference (:30-33),
but frames are passed once and indexed per object on the device instead of being replicated
with images[im_ids] (:41); the default of 64 objects per chunk is kept for drop-in parity and
can be raised (288 GB of HBM holds thousands of crops in flight).
"""
from collections import defaultdict

import numpy as np
import torch

from . import lib3d
from . import tensor_collection as tc


class CoarseRefinePosePredictor(torch.nn.Module):
    def __init__(self, coarse_model=None, refiner_model=None, bsz_objects=64):
        super().__init__()
        self.coarse_model = coarse_model
        self.refiner_model = refiner_model
        self.bsz_objects = bsz_objects
        self.eval()

    @torch.no_grad()
    def batched_model_predictions(self, model, images, K, obj_data, n_iterations=1):
        preds = defaultdict(list)
        n = len(obj_data)
        for start in range(0, n, self.bsz_objects):
            batch_ids = np.arange(start, min(start + self.bsz_objects, n))
            obj_inputs = obj_data[batch_ids]
            labels = obj_inputs.infos['label'].values
            im_ids = obj_inputs.infos['batch_im_id'].values
            outputs = model(images=images, K=K, TCO=obj_inputs.poses, n_iterations=n_iterations, labels=labels,
                            im_ids=im_ids)
            for it in range(1, n_iterations + 1):
                iter_outputs = outputs[f'iteration={it}']
                preds[f'iteration={it}'].append(tc.PandasTensorCollection(
                    obj_inputs.infos,
                    poses=iter_outputs['TCO_output'],
                    poses_input=iter_outputs['TCO_input'],
                    K_crop=iter_outputs['K_crop'],
                    boxes_rend=iter_outputs['boxes_rend'],
                    boxes_crop=iter_outputs['boxes_crop']))
        return {k: tc.concatenate(v) for k, v in preds.items()}

    def make_TCO_init(self, detections, K):
        im_ids = detections.infos['batch_im_id'].values
        boxes = detections.bboxes
        if self.coarse_model.cfg.init_method == 'z-up+auto-depth':
            mesh_db = self.coarse_model.mesh_db
            obj_ids = mesh_db.object_ids(detections.infos['label'], boxes.device)
            TCO_init = lib3d.TCO_init_from_boxes_zup_autodepth(boxes, mesh_db.point_table(2000), obj_ids, K, im_ids=im_ids)
        else:
            TCO_init = lib3d.TCO_init_from_boxes(z_range=(1.0, 1.0), boxes=boxes, K=K, im_ids=im_ids)
        return tc.PandasTensorCollection(infos=detections.infos, poses=TCO_init)

    def get_predictions(self, images, K, detections=None, data_TCO_init=None,
                        n_coarse_iterations=1, n_refiner_iterations=1):
        preds = dict()
        if data_TCO_init is None:
            assert detections is not None
            assert self.coarse_model is not None
            assert n_coarse_iterations > 0
            data_TCO_init = self.make_TCO_init(detections, K)
            coarse_preds = self.batched_model_predictions(self.coarse_model, images, K, data_TCO_init,
                                                          n_iterations=n_coarse_iterations)
            for n in range(1, n_coarse_iterations + 1):
                preds[f'coarse/iteration={n}'] = coarse_preds[f'iteration={n}']
            data_TCO = coarse_preds[f'iteration={n_coarse_iterations}']
        else:
            assert n_coarse_iterations == 0
            data_TCO = data_TCO_init
            preds['external_coarse'] = data_TCO

        if n_refiner_iterations >= 1:
            assert self.refiner_model is not None
            refiner_preds = self.batched_model_predictions(self.refiner_model, images, K, data_TCO,
                                                           n_iterations=n_refiner_iterations)
            for n in range(1, n_refiner_iterations + 1):
                preds[f'refiner/iteration={n}'] = refiner_preds[f'iteration={n}']
            data_TCO = refiner_preds[f'iteration={n_refiner_iterations}']
        return data_TCO, preds
