"""Data formats either side of the hot path: detector output -> candidate collection, refined poses ->
BOP result files and back.

* `make_detections`: the post-processing contract of Detector.get_detections (cosypose/integrated/detector.py:19-72)
  after the Mask R-CNN itself (out of scope): per-image boxes/labels/scores -> PandasTensorCollection(infos[batch_im_id,
  label, score], bboxes (D,4) float on the device [, masks]), score threshold, one_instance_per_class.
* `tc_to_csv` / `read_csv_candidates`: cosypose/scripts/run_custom_scenario.py:26-58 (poses in metres <-> BOP rows with `t` in
  millimetres and row-major `R`).  The writer restates bop_toolkit_lib.inout.save_bop_results (third-party, BOP19 format:
  header `scene_id,im_id,obj_id,score,R,t,time`, R and t as space-separated floats) -- that dependency is not in the
  reference tree, so the file format is pinned by the round trip through the reference's own reader logic only.
"""
import numpy as np
import pandas as pd
import torch

from . import tensor_collection as tc


def make_detections(per_image, device='cuda', detection_th=None, one_instance_per_class=False, output_masks=False, mask_th=0.8):
    """per_image: list (one entry per frame, in batch order) of dicts with 'boxes' (n,4) xyxy px, 'labels' (n,) str,
    'scores' (n,) [, 'masks' (n,1,h,w) probabilities].  Same output as Detector.get_detections."""
    infos, bboxes, masks = [], [], []
    for n, out in enumerate(per_image):
        for obj_id in range(len(out['boxes'])):
            infos.append(dict(batch_im_id=n, label=out['labels'][obj_id], score=float(out['scores'][obj_id])))
            bboxes.append(torch.as_tensor(out['boxes'][obj_id]))
            if output_masks:
                masks.append(torch.as_tensor(out['masks'][obj_id, 0]) > mask_th)
    if len(bboxes) > 0:
        bboxes = torch.stack(bboxes).to(device).float()
        if output_masks:
            masks = torch.stack(masks).to(device)
    else:
        infos = dict(score=[], label=[], batch_im_id=[])
        bboxes = torch.empty(0, 4, device=device).float()
    outputs = tc.PandasTensorCollection(infos=pd.DataFrame(infos), bboxes=bboxes)
    if output_masks and len(outputs) > 0:
        outputs.register_tensor('masks', masks)
    if detection_th is not None:
        keep = np.where(outputs.infos['score'] > detection_th)[0]
        outputs = outputs[keep]
    if one_instance_per_class:
        infos = outputs.infos
        infos['det_idx'] = np.arange(len(infos))
        keep_ids = infos.sort_values('score', ascending=False).drop_duplicates('label')['det_idx'].values
        outputs = outputs[keep_ids]
        outputs.infos = outputs.infos.drop('det_idx', axis=1)
    return outputs


def save_bop_results(path, results):
    """BOP19 result file (bop_toolkit_lib.inout.save_bop_results): one row per estimate."""
    lines = ['scene_id,im_id,obj_id,score,R,t,time']
    for res in results:
        R = ' '.join(map(str, np.asarray(res['R'], dtype=np.float64).flatten().tolist()))
        t = ' '.join(map(str, np.asarray(res['t'], dtype=np.float64).flatten().tolist()))
        lines.append(f"{res['scene_id']},{res['im_id']},{res['obj_id']},{res['score']},{R},{t},{res.get('time', -1)}")
    with open(path, 'w') as f:
        f.write('\n'.join(lines))


def tc_to_csv(predictions, csv_path):
    """PandasTensorCollection(infos[label 'obj_%06d', score, scene_id, view_id], poses (D,4,4) metres) -> BOP csv."""
    poses = predictions.poses.detach().cpu().numpy()      # ONE device->host copy for the whole collection
    preds = []
    for n in range(len(predictions)):
        row = predictions.infos.iloc[n]
        preds.append(dict(scene_id=row.scene_id, im_id=row.view_id, obj_id=int(row.label.split('_')[-1]), score=row.score,
                          t=poses[n, :3, -1] * 1e3, R=poses[n, :3, :3], time=-1.0))
    save_bop_results(csv_path, preds)


def read_csv_candidates(csv_path):
    df = pd.read_csv(csv_path)
    infos = df.loc[:, ['im_id', 'scene_id', 'score', 'obj_id']]
    infos['obj_id'] = infos['obj_id'].apply(lambda x: f'obj_{x:06d}')
    infos = infos.rename(dict(im_id='view_id', obj_id='label'), axis=1)
    R = np.stack(df['R'].apply(lambda x: list(map(float, x.split(' '))))).reshape(-1, 3, 3)
    t = np.stack(df['t'].apply(lambda x: list(map(float, x.split(' '))))).reshape(-1, 3) * 1e-3
    TCO = torch.eye(4, dtype=torch.float).unsqueeze(0).repeat(len(R), 1, 1)
    TCO[:, :3, :3] = torch.tensor(R, dtype=torch.float)
    TCO[:, :3, -1] = torch.tensor(t, dtype=torch.float)
    return tc.PandasTensorCollection(poses=TCO, infos=infos)
