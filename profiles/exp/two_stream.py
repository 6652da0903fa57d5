"""Experiment: the headline workload (256 detections, coarse 1 + refiner 4, 256x256, fp16) as NS independent slices of the
detections, each on its own HIP stream with its own engines, against the single-stream run.  Do the tails of the ~100
dependent kernels of a forward overlap with the other slice's kernels?
    python profiles/exp/two_stream.py [NS ...]
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import pandas as pd
import bench
from cosypose_amd import synthetic as syn
from cosypose_amd import tensor_collection as tc
from cosypose_amd.mesh_db import BatchedMeshes
from cosypose_amd.pose_predictor import CoarseRefinePosePredictor


def main():
    mode = os.environ.get('MODE', 'free')          # free: streams never join; join: joined on the main stream every step; stagger: join + delayed starts
    stagger_us = float(os.environ.get('STAGGER_US', '1200'))
    ns_list = [int(v) for v in sys.argv[1:]] or [1, 2, 4]
    H = W = 256; D = 256; n_obj = 21; dtype = 'fp16'
    labels = np.array([f'obj_{i:06d}' for i in range(1, n_obj + 1)])
    pts = syn.make_mesh_points(7, n_obj, 2500)
    infos = {l: dict(label=l, n_points=2500, n_sym=1) for l in labels}
    mesh_db = BatchedMeshes(infos, labels, torch.from_numpy(pts), torch.eye(4).reshape(1, 1, 4, 4).repeat(n_obj, 1, 1, 1)).float().cuda()
    frames, K, det = bench.make_scene(syn, torch, tc, pd, labels, 1, D, 16, 512, 512, n_obj)
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    renders = [torch.rand(D, 3, H, W, device='cuda', generator=g) for _ in range(5)]
    ref = None
    for ns in ns_list:
        per = D // ns
        preds, streams, dets = [], [], []
        shared = os.environ.get('SHARED', '0')
        for s in range(ns):
            if s == 0 or shared == '0':
                rend = bench.SyntheticRenderer([r[s * per:(s + 1) * per] for r in renders])
            if s == 0 or shared in '01':
                coarse = bench.build_model(0, mesh_db, (H, W), dtype, rend)
                refiner = bench.build_model(1, mesh_db, (H, W), dtype, rend)
            preds.append(CoarseRefinePosePredictor(coarse_model=coarse, refiner_model=refiner, bsz_objects=per))
            streams.append(torch.cuda.Stream())
            dets.append(det[np.arange(s * per, (s + 1) * per)])

        main = torch.cuda.current_stream()
        clock = 100e6   # torch.cuda._sleep spins on wall_clock64 (100 MHz constant clock on gfx9)? calibrated below

        if mode == 'stagger':      # calibrate _sleep
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(1000); e0.record(); torch.cuda._sleep(10_000_000); e1.record(); e1.synchronize()
            clock = 10_000_000 / (e0.elapsed_time(e1) * 1e-3)
            print(f'_sleep clock {clock / 1e6:.1f} MHz', flush=True)

        def step():
            outs = []
            for s in range(ns):
                if mode != 'free':
                    streams[s].wait_stream(main)
                with torch.cuda.stream(streams[s]):
                    if mode == 'stagger' and s:
                        torch.cuda._sleep(int(s * stagger_us * 1e-6 * clock))
                    final, _ = preds[s].get_predictions(frames, K, detections=dets[s], n_coarse_iterations=1, n_refiner_iterations=4)
                    outs.append(final.poses)
            if mode != 'free':
                for s in range(ns):
                    main.wait_stream(streams[s])
            return outs

        torch.cuda.synchronize()
        for _ in range(3):
            outs = step()
        torch.cuda.synchronize()
        poses = torch.cat(outs)
        if ref is None:
            ref = poses
        same = bool((poses == ref).all())
        import gc; gc.collect(); gc.freeze()
        t0 = time.perf_counter()
        n = 12
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f'{mode} slices {ns} x {per}: {dt * 1e3:.3f} ms/step  {D * 5 / dt:,.0f} pose-iter/s  bit-identical to 1 slice: {same}', flush=True)
        del preds


if __name__ == '__main__':
    main()
