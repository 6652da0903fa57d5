#!/usr/bin/env python3
"""Fixtures for the data formats either side of the hot path, produced by running the REFERENCE's own code in
place (build container only; the GPU box sees the .npz):
  * Detector.get_detections' post-processing (cosypose/integrated/detector.py:36-72) on a fake Mask R-CNN output:
    score threshold, one_instance_per_class, masks, the empty case;
  * read_csv_candidates (cosypose/scripts/run_custom_scenario.py:44-58) on a literal BOP19 csv;
  * tc_to_csv (:26-41): the rows the reference hands to bop_toolkit_lib.inout.save_bop_results (third-party, absent:
    replaced by a recorder, so what is pinned is the reference's part -- field names, m -> mm, obj_id parsing).
Stubs: the modules run_custom_scenario.py imports but these two functions never touch (datasets, multiview predictor,
PyBullet renderer, visualisation, bop_toolkit) and cosypose.config; `Tensor.cuda` is the identity (no GPU here).
Writes tests/golden/reference_golden_io.npz.
"""
import sys
import types
import pathlib

import numpy as np
import pandas as pd
import torch

HERE = pathlib.Path(__file__).resolve().parent
REF = pathlib.Path('/root/reference')
sys.dont_write_bytecode = True

CSV_TEXT = """scene_id,im_id,obj_id,score,R,t,time
48,1,5,0.87,0.36 0.48 -0.8 -0.8 0.6 0.0 0.48 0.64 0.6,12.5 -30.25 812.0,-1
48,1,21,0.5,1.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 1.0,0.0 0.0 1000.0,0.25
7,103,1,0.125,0.0 -1.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0,-100.5 200.0 654.321,-1
"""


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    recorded = {}
    for n in ('pinocchio', 'eigenpy', 'transforms3d', 'transforms3d.euler', 'trimesh', 'simplejson', 'torchnet', 'colorama'):
        stub(n)
    sys.modules['eigenpy'].switchToNumpyArray = lambda: None
    stub('cosypose.config', PROJECT_DIR=REF, LOCAL_DATA_DIR=pathlib.Path('/tmp'), DEBUG_DATA_DIR=pathlib.Path('/tmp'),
         BOP_TOOLKIT_DIR=pathlib.Path('/tmp/no_bop_toolkit'))
    stub('cosypose.datasets.bop_object_datasets', BOPObjectDataset=None)
    stub('cosypose.integrated.multiview_predictor', MultiviewScenePredictor=None)
    stub('cosypose.rendering.bullet_scene_renderer', BulletSceneRenderer=None)
    stub('cosypose.visualization.multiview', make_cosypose_plots=None, make_scene_renderings=None, nms3d=None)
    stub('cosypose.lib3d.rigid_mesh_database', MeshDataBase=None)
    inout = stub('bop_toolkit_lib.inout', save_bop_results=lambda path, preds: recorded.update(path=path, preds=preds))
    stub('bop_toolkit_lib', inout=inout)
    sys.path.insert(0, str(REF))
    torch.Tensor.cuda = lambda self, *a, **k: self

    import cosypose.utils.tensor_collection as rtc
    from cosypose.integrated.detector import Detector
    from cosypose.scripts import run_custom_scenario as rcs

    out = {}
    # ---- detector post-processing
    rs = np.random.RandomState(5)
    label_to_cat = {f'obj_{i:06d}': i for i in range(1, 8)}
    n_per_image = [4, 0, 5]
    h, w = 6, 8
    raw = []
    for n in n_per_image:
        xy = rs.uniform(0, 300, (n, 2)); wh = rs.uniform(20, 120, (n, 2))
        raw.append(dict(boxes=torch.tensor(np.concatenate([xy, xy + wh], 1), dtype=torch.float32),
                        labels=torch.tensor(rs.randint(1, 5, n), dtype=torch.int64),
                        scores=torch.tensor(np.round(rs.uniform(0.05, 0.99, n), 3), dtype=torch.float32),
                        masks=torch.tensor(rs.uniform(0, 1, (n, 1, h, w)), dtype=torch.float32)))
    for i, r in enumerate(raw):
        for k, v in r.items():
            out[f'det_in{i}_{k}'] = v.numpy()
    out['det_label_names'] = np.array(sorted(label_to_cat))
    out['det_label_ids'] = np.array([label_to_cat[k] for k in sorted(label_to_cat)])

    class FakeModel:
        config = types.SimpleNamespace(label_to_category_id=label_to_cat)

        def eval(self):
            return self

        def __call__(self, images):
            return [{k: v.clone() for k, v in r.items()} for r in raw]
    det = Detector(FakeModel())
    images = torch.zeros(len(raw), 3, h, w)
    cases = dict(plain={}, th=dict(detection_th=0.5), one=dict(one_instance_per_class=True),
                 masks=dict(output_masks=True, mask_th=0.6, detection_th=0.3),
                 th_one=dict(detection_th=0.2, one_instance_per_class=True))
    for name, kw in cases.items():
        o = det.get_detections(images, **kw)
        out[f'det_{name}_columns'] = np.array(list(o.infos.columns))
        for c in o.infos.columns:
            out[f'det_{name}_info_{c}'] = np.asarray(o.infos[c].values if c != 'label' else o.infos[c].values.astype(str))
        out[f'det_{name}_bboxes'] = o.bboxes.numpy()
        if 'masks' in o.tensors:
            out[f'det_{name}_masks'] = o.masks.numpy()
    raw_backup, raw[:] = list(raw), [dict(boxes=torch.zeros(0, 4), labels=torch.zeros(0, dtype=torch.int64), scores=torch.zeros(0),
                                          masks=torch.zeros(0, 1, h, w)) for _ in raw]
    o = det.get_detections(images, output_masks=True)
    out['det_empty_columns'] = np.array(list(o.infos.columns)); out['det_empty_n'] = np.array(len(o))
    out['det_empty_bboxes_shape'] = np.array(o.bboxes.shape)
    raw[:] = raw_backup

    # ---- BOP csv reader on a literal file, and the rows of the writer
    p = pathlib.Path('/tmp/_golden_io.csv'); p.write_text(CSV_TEXT)
    cand = rcs.read_csv_candidates(p)
    out['csv_text'] = np.array(CSV_TEXT)
    out['csv_columns'] = np.array(list(cand.infos.columns))
    for c in cand.infos.columns:
        out[f'csv_info_{c}'] = np.asarray(cand.infos[c].values if c != 'label' else cand.infos[c].values.astype(str))
    out['csv_poses'] = cand.poses.numpy()
    infos = pd.DataFrame(dict(label=['obj_000005', 'obj_000021', 'obj_000001'], score=[0.87, 0.5, 0.125], scene_id=[48, 48, 7],
                              view_id=[1, 1, 103]))
    poses = cand.poses.clone()
    rcs.tc_to_csv(rtc.PandasTensorCollection(infos=infos, poses=poses), '/tmp/_unused.csv')
    rows = recorded['preds']
    out['rows_keys'] = np.array(sorted(rows[0]))
    for k in ('scene_id', 'im_id', 'obj_id', 'score', 'time'):
        out[f'rows_{k}'] = np.array([float(r[k]) for r in rows])
    out['rows_t'] = np.stack([np.asarray(r['t'], np.float64) for r in rows])
    out['rows_R'] = np.stack([np.asarray(r['R'], np.float64) for r in rows])
    np.savez_compressed(HERE / 'reference_golden_io.npz', **out)
    print('wrote reference_golden_io.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
