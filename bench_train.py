#!/usr/bin/env python3
"""Training-step timing (BASELINE configs[4], SURVEY 8d "Config 4"): 64 crops/GPU from 480x640 uint8 frames,
240x320 crops, fp32 -- h_pose forward (train-mode BatchNorm, drop_connect), disentangled loss, backward, gradient
all-reduce over RCCL, clip 0.5, Adam.  NOT the headline metric (bench.py is); prints one JSON line with the step time,
its split and the all-reduce time.

    python bench_train.py [--gpus N] [--steps K] [--warmup W] [--batch 64]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench_train.py --gpus N
"""
import argparse
import json
import os
import sys
import time
import types
from collections import defaultdict

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


HBM_PEAK_GBS = 8000.0


def bn_tensor_bytes(B, H, W):
    """fp32 bytes of every tensor a train-mode BatchNorm of EfficientNet-B3 normalises, for B crops of H x W (stem, bn0 / bn1 / bn2 of the 26
    blocks, head): the unit the BatchNorm kernels stream several times per step"""
    from cosypose_amd import arch
    h, w = arch.conv_out(H, 3, 2), arch.conv_out(W, 3, 2)
    n = float(h * w * 40)
    for (k, s, e, cin, cout) in arch.B3_BLOCKS:
        ho, wo = arch.conv_out(h, k, s), arch.conv_out(w, k, s)
        if e != 1:
            n += h * w * cin * e
        n += ho * wo * cin * e + ho * wo * cout
        h, w = ho, wo
    n += h * w * 1536
    return 4.0 * B * n


def bn_roofline(torch, run_step, B, H, W, profile_on=True):
    """roofline object of the step's dominant kernel family: the train-mode BatchNorm kernels (statistics, apply + Swish, backward reduce,
    backward apply: cosypose_amd/csrc/kernels_train.hip), HBM-bound.  Their device time comes from ONE torch.profiler step behind the timed
    region; algorithmic bytes = 8 passes over the normalised tensors (forward: statistics read, apply read + write; backward: the reduce
    reads dy and x, the apply reads dy and x and writes dx) -- the gated forms read the (B, C) gate rows besides, which is noise."""
    from collections import defaultdict
    from torch.profiler import profile, ProfilerActivity
    if not profile_on:
        run_step()
        return None
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run_step()
    fam, total = defaultdict(lambda: [0.0, 0]), 0.0
    for e in prof.key_averages():
        t = getattr(e, 'self_device_time_total', None)
        if t is None:
            t = e.self_cuda_time_total
        total += t / 1e3
        k = e.key
        name = ('batchnorm (bn_stats / bn_apply / bn_bwd_reduce / bn_bwd_apply)' if 'bn_' in k else 'depthwise (dw_fwd / dw_bwd_data / dw_bwd_weight)' if 'dw_' in k else
                '1x1-conv GEMMs (pw_gemm_dma_kernel<float>)' if 'pw_gemm' in k or 'pw_pack' in k else 'weight gradients (wgrad)' if 'wgrad' in k else
                'per-sample reductions / squeeze-excite' if 'rows_' in k or 'se_' in k else 'other')
        fam[name][0] += t / 1e3; fam[name][1] += e.count
    bn_ms, bn_n = fam['batchnorm (bn_stats / bn_apply / bn_bwd_reduce / bn_bwd_apply)']
    alg = 8.0 * bn_tensor_bytes(B, H, W)
    ach = alg / (bn_ms * 1e-3) / 1e9 if bn_ms else 0.0
    return dict(bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                kernel='bn_* kernels of kernels_train.hip (family: %d launches per step)' % bn_n, family_ms_per_step=round(bn_ms, 2),
                algorithmic_bytes_per_step=int(alg), launches_per_step=bn_n, device_ms_per_step=round(total, 2),
                families_ms={k: round(v[0], 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])},
                source='one torch.profiler step behind the timed region; bytes = 8 passes over the fp32 tensors the BatchNorms normalise')


def cpu_baseline_train(H, W, B=8, repeats=3):
    """the oracle's training step (torch-CPU autograd over the functional restatement of the reference's network + disentangled loss) on a
    bounded sample: B crops, forward + backward, warm-up then the median of `repeats`, 16 threads"""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import torch
    import cosy_oracle as O
    from cosypose_amd import synthetic as syn
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    rs = np.random.RandomState(0)
    x = rs.rand(B, 6, H, W).astype(np.float32)
    TCO = syn.make_TCO(1, B)
    gt = TCO[:, None].copy()                       # (B, n_sym = 1, 4, 4) ground truths
    gt[:, 0, :3, 3] += rs.normal(0, 0.02, (B, 3)).astype(np.float32)
    K = syn.make_K(B, H, W)
    pts = (rs.uniform(-1, 1, (B, 2600, 3)) * rs.uniform(0.03, 0.12, (B, 1, 3))).astype(np.float32)
    ts = []
    for i in range(repeats + 1):
        ref = O.TorchRef(syn.golden_state_dict(1))
        t0 = time.time()
        ref.train_forward_backward(x, gt, TCO, K, pts)
        ts.append(time.time() - t0)
    v = B / float(np.median(ts[1:]))
    return dict(value=round(v, 3), unit='crops/s (forward + backward)', cores=threads, kind='port',
                sample=f'{B} crops of {H}x{W}, fp32, train-mode forward + disentangled loss + backward of the torch-CPU oracle, warm-up then median of {repeats}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--kernels', action='store_true', help='per-kernel table from torch.profiler to stderr')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='skip the profiled step (no roofline object)')
    ap.add_argument('--upload', choices=('prefetch', 'in_step'), default='prefetch',
                    help="prefetch: batch i+1 is uploaded on a copy stream while step i runs (training.DevicePrefetcher, what train_loop does); "
                         "in_step: h_pose's .cuda() uploads the batch at the top of its own step (the reference's order)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bench import SyntheticRenderer, build_model
    import itertools
    from cosypose_amd import synthetic as syn, train_engine, pose_forward_loss as pfl
    from cosypose_amd.training import DevicePrefetcher, LazyMeters
    from cosypose_amd.mesh_db import BatchedMeshes
    from cosypose_amd.distributed import init_distributed_mode, local_device_index, self_launch, process_group_info

    if not torch.cuda.is_available():
        raise SystemExit('bench_train.py needs an MI355X (no CPU fallback)')
    rc = self_launch(args.gpus)          # `python bench_train.py --gpus N` launches its own N ranks
    if rc is not None:
        raise SystemExit(rc)
    rank, world = init_distributed_mode('nccl')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree')
    torch.cuda.set_device(local_device_index())
    B, n_obj, h, w, H, W = args.batch, 21, 480, 640, 240, 320
    labels = np.array([f'obj_{i:06d}' for i in range(1, n_obj + 1)])
    pts = syn.make_mesh_points(7, n_obj, 2600)
    infos = {l: dict(label=l, n_points=2600, n_sym=1) for l in labels}
    mesh_db = BatchedMeshes(infos, labels, torch.from_numpy(pts), torch.eye(4).reshape(1, 1, 4, 4).repeat(n_obj, 1, 1, 1)).float().cuda()
    frames, K, TCO, obj = syn.make_training_batch(100 + rank, B, n_obj, h, w)
    g = torch.Generator(device='cuda'); g.manual_seed(1 + rank)
    renderer = SyntheticRenderer([torch.rand(B, 3, H, W, device='cuda', generator=g) for _ in range(3)])
    model = build_model(1, mesh_db, (H, W), 'fp32', renderer).train()
    # 2D boxes of the projected ground-truth models (what the dataset provides)
    P = torch.from_numpy(pts[obj]).cuda()
    Kc, Tc = torch.from_numpy(K).cuda(), torch.from_numpy(TCO).cuda()
    cam = (Tc[:, :3, :3].unsqueeze(1) @ P.unsqueeze(-1)).squeeze(-1) + Tc[:, :3, 3].unsqueeze(1)
    uv = (Kc.unsqueeze(1) @ cam.unsqueeze(-1)).squeeze(-1)
    uv = uv[..., :2] / uv[..., 2:]
    bboxes = torch.cat([uv.min(1)[0], uv.max(1)[0]], 1)
    # the batch arrives as the reference's DataLoader delivers it (train_pose.py:242: pin_memory=True): page-locked host tensors.  EVERY step
    # uploads one batch (59 MB of frames): with --upload prefetch (default, what training.train_loop does) the upload of step i+1's batch runs on a
    # copy stream beside step i's kernels; with --upload in_step h_pose's non-blocking .cuda() uploads it at the top of its own step
    pin = lambda t: t.contiguous().pin_memory()
    data = types.SimpleNamespace(images=pin(torch.from_numpy(frames)), K=pin(torch.from_numpy(K)), TCO=pin(torch.from_numpy(TCO)),
                                 objects=[dict(name=l) for l in labels[obj]], bboxes=pin(bboxes.cpu()))
    cfg = argparse.Namespace(n_points_loss=2600, loss_disentangled=True, n_pose_dims=9, init_method='v0')
    opt = train_engine.FlatAdam(model, lr=3e-4, clip_grad_norm=0.5, overlap_allreduce=None)     # None: buckets from inside the backward when there is more than one rank
    meters = LazyMeters()          # as train_loop: the loss values are read back without stopping the host (no .item() in the middle of the step)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    split = defaultdict(float)

    feeds = {'in_step': itertools.repeat(data), 'prefetch': iter(DevicePrefetcher(itertools.repeat(data)))}
    mode = [args.upload]

    def step(timed):
        e = [ev() for _ in range(5)]
        opt.zero_grad()
        batch = next(feeds[mode[0]])
        e[0].record()
        loss = pfl.h_pose(model=model, mesh_db=mesh_db, data=batch, meters=meters, cfg=cfg, n_iterations=1, input_generator='fixed')
        e[1].record()
        loss.backward()
        e[2].record()
        train_engine.allreduce_gradients(opt)       # no-op when the backward launched its own bucketed all-reduce (FlatAdam.overlap_allreduce: on with > 1 rank)
        e[3].record()
        opt.step()
        e[4].record()
        if timed:
            torch.cuda.synchronize()
            for k, a, b in (('forward_loss', 0, 1), ('backward', 1, 2), ('allreduce', 2, 3), ('clip_adam', 3, 4)):
                split[k] += e[a].elapsed_time(e[b])
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step(False)
    assert torch.isfinite(loss).all()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    meters.flush()
    assert meters['loss_total'].n == args.warmup + args.steps
    for _ in range(min(args.steps, 3)):
        step(True)
    nsp = min(args.steps, 3)
    if args.kernels and rank == 0:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step(False); torch.cuda.synchronize()
        ka = prof.key_averages()
        print(ka.table(sort_by='cuda_time_total', row_limit=30, max_name_column_width=70), file=sys.stderr)
        fam = defaultdict(float)
        gemms = []
        for e in ka:
            t = getattr(e, 'self_device_time_total', None)
            if t is None:
                t = e.self_cuda_time_total
            key = ('rocBLAS GEMM' if e.key.startswith('Cijk') else 'cosy HIP kernels' if 'cosy::' in e.key else
                   'torch elementwise/reduce' if 'at::native' in e.key else 'copies' if 'Memcpy' in e.key or 'copyBuffer' in e.key else 'other')
            fam[key] += t / 1e3
            if e.key.startswith('Cijk'):
                gemms.append((t / 1e3, e.count, e.key[:90]))
        print('time by family (ms/step): ' + ', '.join(f'{k} {v:.2f}' for k, v in sorted(fam.items(), key=lambda kv: -kv[1])), file=sys.stderr)
        for t, n, k in sorted(gemms, reverse=True)[:8]:
            print(f'  gemm {t:8.2f} ms  x{n:3d}  {k}', file=sys.stderr)
    roofline = cpu_base = None
    if not args.no_profile:          # every rank runs the profiled step (it holds the gradient all-reduce); rank 0's profile makes the object
        roofline = bn_roofline(torch, lambda: (step(False), torch.cuda.synchronize()), B, H, W, profile_on=rank == 0)
    if rank == 0 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline_train(H, W)
    if rank == 0:
        ms = 1e3 * dt / args.steps
        print(json.dumps({
            'metric': 'training step time (refiner, 240x320 crops, fp32)', 'value': round(ms, 2), 'unit': 'ms/step', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': False, 'scaling': 'weak', 'dtype': 'f32', 'data': 'synthetic',
            'crops_per_s': round(world * B * args.steps / dt, 1),
            'config': {'workload': f'BASELINE configs[4]: {B} crops/GPU from {h}x{w} uint8 frames, {H}x{W} crops, h_pose forward + disentangled '
                                   f'loss + backward + gradient all-reduce ({opt.grad.numel() * 4 / 1e6:.1f} MB fp32) + clip 0.5 + Adam',
                       'upload': {'mode': args.upload, 'bytes_per_step': int(data.images.numel())},
                       'host_syncs_per_step': 0,
                       'gradient_allreduce': 'bucketed (8 MB), launched from inside the backward' if world > 1 else 'none (1 rank)',
                       'n_points_loss': 2600, 'drop_connect_rate': model.drop_connect_rate, 'process_group': process_group_info()},
            'split_ms': {k: round(v / nsp, 2) for k, v in split.items()},
            'peak_memory_gb': round(torch.cuda.max_memory_allocated() / 1e9, 2),
            'roofline': roofline, 'cpu_baseline': cpu_base,
        }))


if __name__ == '__main__':
    main()
