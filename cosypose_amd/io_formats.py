"""Data formats either side of the hot path, written from their contracts and pinned by fixtures the
reference's own code produced (tests/golden/generate_golden_io.py -> reference_golden_io.npz).

Detector output -> candidates.  Contract of Detector.get_detections after the Mask R-CNN itself
(cosypose/integrated/detector.py:36-72; the network is out of scope): one row per detection in image order, `infos`
columns [batch_im_id, label, score], `bboxes` (D,4) float32 xyxy on the device, optional boolean `masks` (probability >
mask_th); `detection_th` keeps scores strictly above it; `one_instance_per_class` keeps the best-scoring detection of
every label, listed by descending score.

Poses <-> BOP result files.  Contract of cosypose/scripts/run_custom_scenario.py:26-58: BOP19 rows
`scene_id,im_id,obj_id,score,R,t,time` with R row-major, t in millimetres, both space separated (the writer itself is
bop_toolkit_lib.inout.save_bop_results, third-party); the reader returns infos [view_id, scene_id, score, label] with
label 'obj_%06d' and poses in metres.
"""

import numpy as np
import pandas as pd
import torch

from . import tensor_collection as tc

BOP19_HEADER = 'scene_id,im_id,obj_id,score,R,t,time'


def _host(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)


def make_detections(per_image, device='cuda', detection_th=None, one_instance_per_class=False, output_masks=False, mask_th=0.8):
    """per_image: one dict per frame, in batch order: 'boxes' (n,4) xyxy px, 'labels' (n,) label strings, 'scores' (n,)
    [, 'masks' (n,1,h,w) probabilities] -> PandasTensorCollection as Detector.get_detections returns it."""
    counts = [len(out['boxes']) for out in per_image]
    total = int(sum(counts))
    frame_of = np.repeat(np.arange(len(per_image)), counts)
    labels = np.array([str(l) for out in per_image for l in list(out['labels'])], dtype=object)
    scores = np.concatenate([_host(out['scores']).astype(np.float64).reshape(-1) for out in per_image]) if total else np.zeros(0)
    if total:
        boxes = torch.cat([torch.as_tensor(out['boxes']).reshape(-1, 4) for out in per_image]).to(device=device, dtype=torch.float32)
        table = pd.DataFrame({'batch_im_id': frame_of, 'label': labels, 'score': scores})
    else:
        boxes = torch.empty(0, 4, device=device, dtype=torch.float32)
        table = pd.DataFrame({'score': [], 'label': [], 'batch_im_id': []})      # the reference's column order for "nothing found"
    keep = np.arange(total)
    if detection_th is not None:
        keep = keep[scores[keep] > detection_th]
    if one_instance_per_class:
        # best score first (stable: the earlier detection wins a tie), then the first occurrence of every label
        ranked = keep[np.argsort(-scores[keep], kind='stable')]
        _, first = np.unique(labels[ranked].astype(str), return_index=True)
        keep = ranked[np.sort(first)]
    fields = {'bboxes': boxes[torch.as_tensor(keep, device=boxes.device)] if total else boxes}
    if output_masks and total:
        probs = torch.cat([torch.as_tensor(out['masks'])[:, 0] for out in per_image if len(out['boxes'])])
        fields['masks'] = (probs > mask_th).to(device)[torch.as_tensor(keep, device=device)]
    return tc.PandasTensorCollection(infos=table.iloc[keep].reset_index(drop=True) if total else table, **fields)


def bop19_rows(predictions):
    """The estimates of a collection (infos: label 'obj_%06d', score, scene_id, view_id; poses (D,4,4) in metres) as BOP19
    fields: what run_custom_scenario.tc_to_csv hands to the BOP toolkit's writer."""
    T = predictions.poses.detach().to('cpu', torch.float32)                   # ONE device -> host copy
    R, t_mm = T[:, :3, :3].numpy(), (T[:, :3, 3] * 1e3).numpy()               # metres -> millimetres in float32, as the reference
    info = predictions.infos
    obj = [int(str(l).rsplit('_', 1)[-1]) for l in info['label']]
    return [dict(scene_id=info['scene_id'].iloc[i], im_id=info['view_id'].iloc[i], obj_id=obj[i], score=info['score'].iloc[i],
                 R=R[i], t=t_mm[i], time=-1.0) for i in range(len(info))]


def save_bop_results(path, rows):
    """BOP19 result file, one line per estimate (format of bop_toolkit_lib.inout.save_bop_results)."""
    fmt = lambda v: ' '.join(repr(float(x)) for x in np.asarray(v, np.float64).ravel())
    body = [f"{r['scene_id']},{r['im_id']},{r['obj_id']},{r['score']},{fmt(r['R'])},{fmt(r['t'])},{r.get('time', -1)}" for r in rows]
    with open(path, 'w') as f:
        f.write('\n'.join([BOP19_HEADER] + body))


def tc_to_csv(predictions, csv_path):
    save_bop_results(csv_path, bop19_rows(predictions))


def read_csv_candidates(csv_path):
    """BOP19 csv -> PandasTensorCollection(infos [view_id, scene_id, score, label], poses (D,4,4) float32 in metres)."""
    df = pd.read_csv(csv_path)
    vec = lambda col, n: np.array([np.array(s.split(), dtype=np.float64) for s in df[col]], dtype=np.float64).reshape(len(df), n)
    poses = np.tile(np.eye(4), (len(df), 1, 1))
    poses[:, :3, :3] = vec('R', 9).reshape(-1, 3, 3)
    poses[:, :3, 3] = vec('t', 3) * 1e-3
    infos = pd.DataFrame({'view_id': df['im_id'], 'scene_id': df['scene_id'], 'score': df['score'],
                          'label': ['obj_%06d' % int(o) for o in df['obj_id']]})
    return tc.PandasTensorCollection(infos=infos, poses=torch.as_tensor(poses, dtype=torch.float32))
