"""CoarseRefinePosePredictor: the driver of the hot path, with the interface of the reference's
cosypose/integrated/pose_predictor.py:14-107 -- batched_model_predictions, make_TCO_init, get_predictions -- returning the
same PandasTensorCollections under the same keys ('coarse/iteration=k', 'refiner/iteration=k', 'external_coarse').

MI355X-first differences, invisible to callers: detections are processed in chunks of `bsz_objects` in their given order as
in the reference (:30-33), but the frames are handed over ONCE and indexed per object on the device (the reference
replicates them with images[im_ids], :41); consecutive-row chunks are tensor views; the default of 64 objects per chunk is
kept for drop-in parity and can be raised (288 GB of HBM holds thousands of crops in flight).  With `n_streams` > 1 the
chunks run CONCURRENTLY, round-robin on that many HIP streams, each chunk through coarse and refiner on its own stream
(every stream has its own engine: activations and workspaces are per stream, cosypose_amd/efficientnet.py EnginePool): a
forward is ~100 dependent kernel launches, and the ramp-up / drain of each overlaps with another chunk's kernels instead of
idling the chip.  HBM cost: every stream that runs a model owns a full engine of it (weights 21 MB in a 16-bit type + workspaces
sized for the largest chunk it has seen: ~10 MB per crop of 256x256), i.e. (n_streams + 1) x 2 engines for a coarse + refiner pair;
EnginePool keeps at most COSY_ENGINE_POOL_MAX (default 8) engines per model and evicts the least recently used.  The chunks are independent (the kernels are batch-invariant), so the results are bit-identical to the
sequential order.  Keep n_streams <= 3: ROCm maps streams onto 4 hardware queues, the caller's stream holds one.
"""
import numpy as np
import torch

from . import lib3d
from . import tensor_collection as tc

# per-iteration tensors of PosePredictor.forward's output that travel in the result collections: field -> output key
_ITERATION_FIELDS = (('poses', 'TCO_output'), ('poses_input', 'TCO_input'), ('K_crop', 'K_crop'),
                     ('boxes_rend', 'boxes_rend'), ('boxes_crop', 'boxes_crop'))


_ITERATION_SHAPES = (('poses', (4, 4)), ('poses_input', (4, 4)), ('K_crop', (3, 3)), ('boxes_rend', (4,)), ('boxes_crop', (4,)))


import weakref

_ACCEPTS = weakref.WeakKeyDictionary()       # forward function -> {keyword: bool}, see CoarseRefinePosePredictor._accepts (weak: closures / instance-level forwards die with their owners)


def _iteration_key(n):
    return f'iteration={n}'


def _iteration_collection(infos, outputs):
    return tc.PandasTensorCollection(infos, **{name: outputs[source] for name, source in _ITERATION_FIELDS})


class CoarseRefinePosePredictor(torch.nn.Module):
    def __init__(self, coarse_model=None, refiner_model=None, bsz_objects=64, n_streams=1):
        super().__init__()
        self.coarse_model = coarse_model
        self.refiner_model = refiner_model
        self.bsz_objects = bsz_objects
        self.n_streams = n_streams
        self.__dict__['_lanes'] = {}
        self.eval()

    def _lane_streams(self, device, n):
        """n side streams of `device` (created once, reused by every call)"""
        lanes = self._lanes.setdefault(torch.device(device).index, [])
        while len(lanes) < n:
            lanes.append(torch.cuda.Stream(device=device))
        return lanes[:n]

    @staticmethod
    def _accepts(model, name):
        """does model.forward take the keyword `name` (this package's extensions `frames_nhwc4=` / `out=`)?  A coarse / refiner pair may mix a
        cosypose_amd PosePredictor with any module that has the reference's forward signature."""
        import inspect
        fwd = getattr(model, 'forward', None)
        if fwd is None:
            return False
        fn = getattr(fwd, '__func__', fwd)                   # per forward FUNCTION: get_predictions asks on every call
        try:
            table = _ACCEPTS.setdefault(fn, {})
        except TypeError:                                    # not weak-referenceable (a builtin / C callable): not cached
            table = {}
        hit = table.get(name)
        if hit is None:
            try:
                params = inspect.signature(fwd).parameters
                hit = name in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())
            except (TypeError, ValueError):
                hit = False
            table[name] = hit
        return hit

    @staticmethod
    def _frames4(model, images):
        """the frames' interleaved copy, made ONCE per call for every model / chunk that crops from them (models without the hook: None)"""
        make = getattr(model, 'frames_to_nhwc4', None)
        return make(images) if make is not None and images.is_cuda else None

    @torch.no_grad()
    def batched_model_predictions(self, model, images, K, obj_data, n_iterations=1, frames_nhwc4=None):
        """Run `model` over obj_data (infos[label, batch_im_id], poses) chunk by chunk -> {'iteration=k': collection}."""
        per_iteration = {_iteration_key(n): [] for n in range(1, n_iterations + 1)}
        n_objects = len(obj_data)
        extra = {}
        if (n_objects > self.bsz_objects or frames_nhwc4 is not None) and self._accepts(model, 'frames_nhwc4'):      # several chunks share one conversion of the frames
            f4 = frames_nhwc4 if frames_nhwc4 is not None else self._frames4(model, images)
            if f4 is not None:
                extra = dict(frames_nhwc4=f4)
        for first in range(0, n_objects, self.bsz_objects):
            chunk = obj_data[np.arange(first, min(first + self.bsz_objects, n_objects))]
            outputs = model(images=images, K=K, TCO=chunk.poses, n_iterations=n_iterations,
                            labels=chunk.infos['label'].values, im_ids=chunk.infos['batch_im_id'].values, **extra)
            for key, collected in per_iteration.items():
                collected.append(_iteration_collection(chunk.infos, outputs[key]))
        return {key: tc.concatenate(parts) for key, parts in per_iteration.items()}

    @torch.no_grad()
    def _concurrent_predictions(self, images, K, start, stages):
        """The chunks of `start` each run ALL `stages` = [(name, model, n_iterations)] on their own HIP stream (round-robin over
        n_streams side streams; enqueued stage by stage): a chunk's refiner input is its own coarse output, so nothing crosses streams until the one
        join at the end.  Every chunk writes its rows straight into the full-size result tensors (allocated up front on the
        caller's stream), so the caller's stream has nothing to do behind the join: no per-key concatenation on the critical
        path between two calls.  Returns {'stage/iteration=k': collection}."""
        n_objects = len(start)
        firsts = range(0, n_objects, self.bsz_objects)
        device = start.poses.device
        lanes = self._lane_streams(device, min(int(self.n_streams), len(firsts)))
        main = torch.cuda.current_stream(device)
        keys = [f'{name}/{_iteration_key(n)}' for name, _, n_it in stages for n in range(1, n_it + 1)]
        # Every chunk's kernels write their outputs STRAIGHT into the chunk's rows of the batch-wide result tensors (PosePredictor.forward
        # `out=`): no per-field copy launches on the lanes (5 fields x 5 iterations x 2 chunks of ~4 us each broke up every chunk's kernel
        # chain).  An iteration's input poses ARE the previous iteration's outputs (or the initial poses): `poses_input` shares that tensor,
        # as the sequential path's collections do.
        made = {key: {name: torch.empty((n_objects,) + shape, device=device) for name, shape in _ITERATION_SHAPES if name != 'poses_input'}
                for key in keys}
        full, prev = {}, start.poses
        for key in keys:
            full[key] = dict(made[key], poses_input=prev)
            prev = made[key]['poses']
        # the chunks are cut on `main` BEFORE the lanes wait for it, and must be views: a gather enqueued on `main` behind the wait
        # would be read by a lane unsynchronised
        chunks = [start[np.arange(first, min(first + self.bsz_objects, n_objects))] for first in firsts]
        base = start.poses.untyped_storage().data_ptr()
        assert all(c.poses.untyped_storage().data_ptr() == base for c in chunks), 'chunks of consecutive rows must be tensor views'
        frames4 = self._frames4(stages[0][1], images) if stages else None      # on `main`, once for every chunk and both models
        extra = dict(frames_nhwc4=frames4) if frames4 is not None else {}
        for lane in lanes:
            lane.wait_stream(main)             # frames, K, the initial poses and the result buffers are ready on `main`
        _OUT = (('TCO_output', 'poses'), ('K_crop', 'K_crop'), ('boxes_rend', 'boxes_rend'), ('boxes_crop', 'boxes_crop'))
        # Host order: STAGE by stage across the chunks (every chunk's coarse call, then every chunk's refiner call), not chunk by chunk: on an idle device
        # (one call, or the first step behind a synchronisation) the second lane's kernels arrive behind ONE forward's worth of launches of the first lane
        # instead of five: the first step behind a synchronisation 23.9-24.3 instead of 24.9-26.6 ms (steady steps 21.9-22.0 either way; same call,
        # driver command: 58.05 vs 58.00 k).  The same four model calls per step, every lane sees its own kernels in the same order: bit-identical results.
        ids = [(chunk.infos['label'].values, chunk.infos['batch_im_id'].values) for chunk in chunks]
        poses = [chunk.poses for chunk in chunks]
        for name, model, n_it in stages:
            for i, first in enumerate(firsts):
                last = min(first + self.bsz_objects, n_objects)
                with torch.cuda.stream(lanes[i % len(lanes)]):
                    dst = {n: {src: full[f'{name}/{_iteration_key(n)}'][field][first:last] for src, field in _OUT} for n in range(1, n_it + 1)}
                    outputs = model(images=images, K=K, TCO=poses[i], n_iterations=n_it, labels=ids[i][0], im_ids=ids[i][1], out=dst, **extra)
                    for n in range(1, n_it + 1):       # the model wrote into the rows it was given
                        assert outputs[_iteration_key(n)]['TCO_output'].data_ptr() == dst[n]['TCO_output'].data_ptr()
                    poses[i] = outputs[_iteration_key(n_it)]['TCO_output']
        for lane in lanes:
            main.wait_stream(lane)
        # every result collection owns its infos (as after the sequential path's concatenation): a column added to one is not seen by the others
        return {key: tc.PandasTensorCollection(start.infos.reset_index(drop=True).copy(), **full[key]) for key in keys}

    def _streams_usable(self):
        """False when a model's renderer declares that it must not share the device with launches on other streams (rasterizer.HipBatchRenderer:
        `concurrent_streams_safe = False`, profiles/r04_raster_streams.txt): the chunks then run one after the other on the caller's stream."""
        for model in (self.coarse_model, self.refiner_model):
            r = getattr(model, 'renderer', None)
            while r is not None:
                if getattr(r, 'concurrent_streams_safe', True) is False:
                    return False
                r = getattr(r, 'r', None) if hasattr(r, 'r') and r is not getattr(r, 'r') else None        # a thin wrapper around a renderer
        return True

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
        assert detections is not None or data_TCO_init is not None, 'get_predictions needs detections or data_TCO_init'
        n_objects = len(detections if data_TCO_init is None else data_TCO_init)
        stage_models = [m for m, n in ((self.coarse_model, n_coarse_iterations if data_TCO_init is None else 0), (self.refiner_model, n_refiner_iterations)) if n > 0]
        lanes_ok = all(self._accepts(m, 'out') and self._accepts(m, 'frames_nhwc4') for m in stage_models if m is not None)    # foreign models: sequential path
        if self.n_streams > 1 and n_objects > self.bsz_objects and self._streams_usable() and lanes_ok:
            if data_TCO_init is None:
                assert detections is not None and self.coarse_model is not None and n_coarse_iterations > 0
                start = self.make_TCO_init(detections, K)
                stages = [('coarse', self.coarse_model, n_coarse_iterations)]
            else:
                assert n_coarse_iterations == 0
                start = preds['external_coarse'] = data_TCO_init
                stages = []
            if n_refiner_iterations >= 1:
                assert self.refiner_model is not None
                stages.append(('refiner', self.refiner_model, n_refiner_iterations))
            preds.update(self._concurrent_predictions(images, K, start, stages))
            return (preds[f'{stages[-1][0]}/{_iteration_key(stages[-1][2])}'] if stages else start), preds

        shared = {}

        def run_stage(stage, model, start, n_iterations):
            if 'f4' not in shared:                 # one conversion of the frames for the coarse and the refiner model and all their chunks
                shared['f4'] = next((self._frames4(m, images) for m in stage_models if m is not None and hasattr(m, 'frames_to_nhwc4')), None)
            out = self.batched_model_predictions(model, images, K, start, n_iterations=n_iterations, frames_nhwc4=shared['f4'])
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
