#!/usr/bin/env python3
"""Headline benchmark: refined pose-iterations/sec of the render-and-compare hot path on MI355X.

A "step" = one pass of the hot path over one batch of synthetic input (BASELINE.json configs[1]):
D=256 detections over 16 frames, 21 objects, coarse 1 + refiner 4 iterations = 1280 pose-iterations per
GPU per step, 256x256 crops, bf16 backbone, geometry/pose update fp32.  Everything (frames, intrinsics,
detections, mesh table, weights, the synthetic renderer's images) is resident in HBM before the timed region.
With --gpus N each rank runs its own 256 detections (weak scaling) and the refined poses are all-gathered
over RCCL once per step.

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel of the backbone, timed live with
HIP events recorded on the launch stream after every launch (cosy_effnet_b3_set_profiling) in a second pass over
the same steps right after the timed region (an event per kernel costs ~6 % of throughput, so not inside it);
`cpu_baseline` is the CPU oracle (a port of the reference's PyTorch-CPU arithmetic) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3}
ALGO_MB_PER_POSE_ITER = {('bf16', 256): 10.7, ('fp16', 256): 10.7, ('fp32', 256): 21.3, ('bf16', 240): 12.4, ('fp16', 240): 12.4, ('fp32', 240): 24.9}


class SyntheticRenderer:
    """Stand-in for renderer.render (out of scope): returns pre-generated device images."""

    def __init__(self, renders):
        self.renders, self.i = renders, 0

    def render(self, obj_infos, TCO, K, resolution):
        r = self.renders[self.i % len(self.renders)]
        self.i += 1
        return r[:len(obj_infos)]


def build_model(seed, mesh_db, render_size, dtype, renderer):
    import argparse as ap
    import torch
    from cosypose_amd import synthetic as syn
    from cosypose_amd.pose_models_cfg import create_model_pose, check_update_config
    cfg = check_update_config(ap.Namespace(backbone_str='efficientnet-b3', n_pose_dims=9))
    m = create_model_pose(cfg, renderer, mesh_db)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.golden_state_dict(seed).items()}, strict=False)
    m.cfg = cfg
    m.render_size = render_size
    m.compute_dtype = dtype
    return m.cuda().eval()


def cpu_baseline(crop, n_det=32):
    """The oracle (torch-CPU port of the reference arithmetic + C geometry/roi_align) on a bounded sample."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import torch
    import cosy_oracle as O
    from cosypose_amd import synthetic as syn
    # the stock torch-CPU convolutions stop scaling at ~16 threads on the GPU box's 256-core host
    # (measured: 1 thread 10.3, 16 threads 17.3, 64 threads 10.6, 128 threads 3.7 crops/s at B=16)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    O.set_threads(cores)
    H, W = crop
    h, w = (512, 512) if H == W else (480, 640)
    sd = syn.golden_state_dict(0)
    tr = O.TorchRef(sd)
    pts = syn.make_mesh_points(7, 21, 2500)[:, np.random.RandomState(0).choice(2500, 2000, replace=False)]
    obj, im, boxes = syn.make_detections(1, n_det, 2, 21, h, w)
    frames = syn.make_frames(2, 2, h, w)[im]
    K = syn.make_K(n_det, h, w)
    TCO = O.tco_init_from_boxes(boxes, K)
    rend = syn.make_renders(3, n_det, H, W)
    run = lambda n, b: O.pose_predictor_forward(frames[:b], K[:b], obj[:b], TCO[:b], pts, None, lambda i, t, k: rend[:b],
                                                 n, (H, W), backbone=tr.net_forward)
    run(1, 2)  # warm-up
    t0 = time.time()
    run(2, n_det)  # coarse 1 + refiner 1 worth of iterations
    dt = time.time() - t0
    return dict(value=round(2 * n_det / dt, 3), unit='pose-iterations/s', cores=cores, kind='port',
                sample=f'{n_det} detections x 2 iterations ({2 * n_det} pose-iterations), {H}x{W} crops, fp32, '
                       f'torch-CPU backbone + C geometry/roi_align oracle, {cores} threads, {dt:.1f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--crop', default='256x256', help='HxW of the crops: 256x256 (metric) or 240x320 (reference native)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16', 'fp32'])
    ap.add_argument('--detections', type=int, default=256, help='detections per GPU per step')
    ap.add_argument('--bsz-objects', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='do not record per-kernel HIP events in the timed region')
    ap.add_argument('--layers', action='store_true', help='print the per-launch table to stderr')
    ap.add_argument('--renderer', default='pregenerated', choices=['pregenerated', 'hip'],
                    help="renderer.render source: pre-generated device images (default: the reference's renderer is outside the "
                         "path) or the on-device HIP rasteriser (SURVEY 8f-1) rendering every crop in every iteration")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import pandas as pd
    from cosypose_amd import synthetic as syn, _lib
    from cosypose_amd import tensor_collection as tc
    from cosypose_amd.mesh_db import BatchedMeshes
    from cosypose_amd.pose_predictor import CoarseRefinePosePredictor
    from cosypose_amd.distributed import init_distributed_mode, all_gather_rows, local_device_index

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    rank, world = init_distributed_mode('nccl')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    torch.cuda.set_device(local_device_index())
    H, W = (int(v) for v in args.crop.split('x'))
    h, w = (512, 512) if H == W else (480, 640)   # square frames for square crops
    D, n_frames, n_obj = args.detections, 16, 21
    seed = 1 + rank                                  # BASELINE config index 1; each rank its own candidates

    labels = np.array([f'obj_{i:06d}' for i in range(1, n_obj + 1)])
    pts = syn.make_mesh_points(7, n_obj, 2500)
    infos = {l: dict(label=l, n_points=2500, n_sym=1) for l in labels}
    mesh_db = BatchedMeshes(infos, labels, torch.from_numpy(pts), torch.eye(4).reshape(1, 1, 4, 4).repeat(n_obj, 1, 1, 1)).float().cuda()
    frames = torch.from_numpy(syn.make_frames(seed, n_frames, h, w)).cuda()
    K = torch.from_numpy(syn.make_K(n_frames, h, w)).cuda()
    obj, im, boxes = syn.make_detections(seed + 10, D, n_frames, n_obj, h, w)
    det = tc.PandasTensorCollection(infos=pd.DataFrame(dict(label=labels[obj], batch_im_id=im, score=1.0)),
                                    bboxes=torch.from_numpy(boxes).cuda())
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    renders = [torch.rand(min(D, args.bsz_objects), 3, H, W, device='cuda', generator=g) for _ in range(5)]
    renderer = SyntheticRenderer(renders)
    if args.renderer == 'hip':
        from cosypose_amd.rasterizer import RenderMeshes, HipBatchRenderer
        mv, mf, mc = syn.make_render_meshes(7, n_obj, n_lat=48, n_lon=64)        # 6,016 triangles per object
        renderer = HipBatchRenderer(RenderMeshes(labels, mv, mf, mc).cuda())
    coarse = build_model(0, mesh_db, (H, W), args.dtype, renderer)
    refiner = build_model(1, mesh_db, (H, W), args.dtype, renderer)
    predictor = CoarseRefinePosePredictor(coarse_model=coarse, refiner_model=refiner, bsz_objects=args.bsz_objects)
    n_coarse, n_refine = 1, 4
    iters_per_step = D * (n_coarse + n_refine)

    def step():
        final, _ = predictor.get_predictions(frames, K, detections=det, n_coarse_iterations=n_coarse, n_refiner_iterations=n_refine)
        poses = final.poses
        if world > 1:
            poses = all_gather_rows(poses, max_rows=D)   # ONE collective: refined poses of all ranks, rank order
        return poses

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    assert torch.isfinite(out).all(), 'non-finite refined poses'
    assert out.shape == (world * D, 4, 4)
    nets = [coarse._net(min(D, args.bsz_objects), frames.device), refiner._net(min(D, args.bsz_objects), frames.device)]
    profile = not args.no_profile
    sync()
    t0 = time.perf_counter()
    host_ms = []
    for _ in range(args.steps):
        th = time.perf_counter()
        step()
        host_ms.append((time.perf_counter() - th) * 1e3)   # host enqueue time of the step (the GPU runs behind)
    sync()
    dt = time.perf_counter() - t0
    if args.layers and rank == 0:
        print('host enqueue ms/step: ' + ' '.join(f'{v:.2f}' for v in host_ms), file=sys.stderr)
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # Per-kernel timing for `roofline`: the SAME steps again with a HIP event recorded on the launch stream after every
    # backbone launch.  Kept out of the timed region on purpose: one event per kernel serialises back-to-back
    # launches and costs ~6 % of the headline value (measured 34.4k vs 32.3k pose-iter/s).
    prof_steps = min(args.steps, 4)
    if profile:
        for n_ in nets:
            _lib.check(_lib.lib().cosy_effnet_b3_set_profiling(n_, 1))
        for _ in range(prof_steps):
            step()
        sync()

    roofline = None
    if profile and rank == 0:
        recs = []
        for n_ in nets:
            recs += _lib.profile_read(n_)
            _lib.check(_lib.lib().cosy_effnet_b3_set_profiling(n_, 0))
        kinds = {}
        for r in recs:
            k = kinds.setdefault(r['name'], dict(ms=0.0, bytes=0.0, flops=0.0, n=0))
            k['ms'] += r['ms_avg'] * r['n']; k['bytes'] += r['bytes'] * r['n']; k['flops'] += r['flops'] * r['n']; k['n'] += r['n']
        total_ms = sum(k['ms'] for k in kinds.values())
        if args.layers:
            per = {}
            for r in recs:   # a chunked segment launches the same layer several times per forward: aggregate
                k = per.setdefault((r['layer'], r['name']), dict(ms=0.0, bytes=0.0, flops=0.0, n=0))
                k['ms'] += r['ms_avg'] * r['n']; k['bytes'] += r['bytes'] * r['n']; k['flops'] += r['flops'] * r['n']; k['n'] += r['n']
            nfw = prof_steps * (n_coarse + n_refine)
            for (layer, name), k in per.items():
                print(f"{layer:3d} {name:34s} n={k['n']:4d} {k['ms'] / nfw * 1e3:9.1f} us/fwd  {k['bytes'] / k['ms'] / 1e6:8.1f} GB/s "
                      f"{k['flops'] / k['ms'] / 1e9:8.1f} TFLOP/s", file=sys.stderr)
            for name, k in sorted(kinds.items(), key=lambda kv: -kv[1]['ms']):
                print(f"{name:34s} {100 * k['ms'] / total_ms:5.1f}%  avg {k['ms'] / k['n'] * 1e3:8.1f} us  {k['bytes'] / k['ms'] / 1e6:8.1f} GB/s "
                      f"{k['flops'] / k['ms'] / 1e9:8.1f} TFLOP/s", file=sys.stderr)
        name, k = max(kinds.items(), key=lambda kv: kv[1]['ms'])
        intensity = k['flops'] / k['bytes']
        balance = MFMA_PEAK_TFLOPS[args.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)
        if intensity < balance:
            ach = k['bytes'] / k['ms'] / 1e6
            roofline = dict(bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4))
        else:
            ach = k['flops'] / k['ms'] / 1e9
            roofline = dict(bound='mfma', achieved=round(ach, 1), peak=MFMA_PEAK_TFLOPS[args.dtype], unit='TFLOP/s',
                            frac=round(ach / MFMA_PEAK_TFLOPS[args.dtype], 4))
        traffic = None   # HBM bytes per launch from the committed PMC passes of the same command (profiles/collect.sh)
        tfile = os.path.join(REPO, 'profiles', 'r01_pmc_traffic.json')
        if os.path.exists(tfile) and args.crop == '256x256' and args.dtype == 'bf16' and D == 256:
            t = json.load(open(tfile)).get(name)
            if t:
                traffic = round(t['read_bytes'] + t['write_bytes'])
        roofline.update(traffic=traffic, kernel=name, launches_timed=k['n'], avg_launch_us=round(k['ms'] / k['n'] * 1e3, 2),
                        algorithmic_bytes_per_launch=round(k['bytes'] / k['n']), algorithmic_flops_per_launch=round(k['flops'] / k['n']),
                        share_of_backbone_time=round(k['ms'] / total_ms, 4),
                        backbone_ms_per_forward=round(total_ms / (prof_steps * (n_coarse + n_refine)), 3))

    if rank == 0:
        value = world * iters_per_step * args.steps / dt
        mb = ALGO_MB_PER_POSE_ITER.get((args.dtype, H))
        line = {
            'metric': 'refined pose-iterations/sec (256x256 crops, n_iter=1+4)' if H == W else f'refined pose-iterations/sec ({H}x{W} crops, n_iter=1+4)',
            'value': round(value, 1), 'unit': 'pose-iterations/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[1]: {D} detections/GPU over {n_frames} frames {h}x{w}, {n_obj} objects, '
                                   f'coarse {n_coarse} + refiner {n_refine} iterations, {H}x{W} crops, ' +
                                   ('synthetic on-device renders' if args.renderer == 'pregenerated' else
                                    'renders by the on-device HIP rasteriser (6k-triangle meshes) inside the loop'),
                       'pose_iterations_per_step_per_gpu': iters_per_step, 'bsz_objects': args.bsz_objects,
                       'parallelism': f'candidate-sharded x{world}, 1 all-gather of refined poses per step'},
            'roofline': roofline,
            'path_hbm_frac': round(value / world * mb * 1e6 / (HBM_PEAK_GBS * 1e9), 5) if mb else None,
            'cpu_baseline': None,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline((H, W))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
