"""s_memtime timeline of wave-front jobs (mbconv_wave_kernel), experiment build only.

    COSY_TUNE_LIB=1 python profiles/exp/wave_timeline.py [--cmid 816] [--crops 256] [--dtype fp16] [--crop 256x256]

The tune build's wave kernel parks the shader clock in its LDS block at fixed points of a job (kernels_wave.hip: WAVE_STAMP) for wave 0 of
every `stride`-th workgroup of the launches whose Cmid matches; the stamps of one forward are read back here and averaged over the recorded
jobs: prologue, and per input row the five segments  wait for fragments | copy + expansion (MFMA, BN, SiLU, halo) | issue next loads +
stores | tap rows + finished output rows.  Times in shader cycles and ns (2.4 GHz).  One forward is run first without stamping (warm
caches), then one with.  Only the LAST matching launch of the forward keeps its stamps (several blocks share a Cmid: 14-17)."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cmid', type=int, default=816)
    ap.add_argument('--crops', type=int, default=256)
    ap.add_argument('--dtype', default='fp16')
    ap.add_argument('--crop', default='256x256')
    ap.add_argument('--stride', type=int, default=37)
    ap.add_argument('--slots', type=int, default=64)
    a = ap.parse_args()
    assert os.environ.get('COSY_TUNE_LIB'), 'needs the experiment build: COSY_TUNE_LIB=1'
    from cosypose_amd import synthetic as syn
    from cosypose_amd._lib import lib, check, ptr, stream
    from cosypose_amd.pose_models_cfg import create_model_pose, check_update_config
    H, W = (int(v) for v in a.crop.split('x'))
    cfg = check_update_config(argparse.Namespace(backbone_str='efficientnet-b3', n_pose_dims=9))
    m = create_model_pose(cfg, None, None)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.golden_state_dict(0).items()}, strict=False)
    m = m.cuda().eval()
    m.compute_dtype = a.dtype; m.render_size = (H, W)
    B = a.crops
    x = torch.rand(B, 6, H, W, device='cuda')
    h = m._net(B, x.device)
    pose = torch.empty(B, 9, device='cuda')
    check(lib().cosy_effnet_b3_set_input_nchw(h, ptr(x), B, stream()))

    def fwd():
        check(lib().cosy_effnet_b3_forward(h, B, None, ptr(pose), None, stream()))
        torch.cuda.synchronize()
    fwd(); fwd()
    buf = torch.zeros(a.slots, 8, dtype=torch.int64, device='cuda')
    os.environ.update(COSY_WAVE_STAMP_PTR=hex(buf.data_ptr()), COSY_WAVE_STAMP_CMID=str(a.cmid), COSY_WAVE_STAMP_STRIDE=str(a.stride),
                      COSY_WAVE_STAMP_SLOTS=str(a.slots))
    fwd()
    for k in ('COSY_WAVE_STAMP_PTR', 'COSY_WAVE_STAMP_CMID'):
        os.environ.pop(k)
    st = buf.cpu().numpy().astype(np.float64)
    st = st[st[:, 0] >= 3]
    print(f'wave-front timeline, Cmid = {a.cmid}, {B} crops of {H}x{W}, {a.dtype}: {len(st)} recorded jobs (wave 0 of every {a.stride}-th workgroup; the LAST launch with this Cmid)')
    if not len(st):
        print('no stamps recorded (no launch with that Cmid, or not the -DCOSY_WAVE_STAMPS build)')
        return
    ghz = 2.4
    n, t0, t1, tf, te, rsum, rmin, rmax = st.T
    if os.environ.get('COSY_WAVE_PHASES'):      # phase experiment build (profiles/exp/wave_phases.diff): words 6 / 7 hold three phase sums instead of min / max
        pa, pb, pc = (rmin.astype(np.uint64) & 0xffffffff).astype(float), (rmin.astype(np.uint64) >> 32).astype(float), (rmax.astype(np.uint64) & 0xffffffff).astype(float)
        rows = n.astype(float)
        print(f'phases per row (cycles, mean over the recorded jobs): wait + expansion MFMAs {np.mean(pa / rows):.0f} | BN0 + SiLU + convert + permutation + halo operands {np.mean(pb / rows):.0f} | '
              f'next loads + tap MFMAs {np.mean(pc / rows):.0f} | row total {np.mean(rsum / (n - 1)):.0f} (the rest: BN1 + SiLU + sums + convert + permutation + transposition + store + loop)')
        return
    tot, pro, first = te - t0, t1 - t0, tf - t1
    last = te - (tf + rsum)                     # last row start -> job end: the last row, the last store, the squeeze tree
    mean_row = rsum / (n - 1)
    print(f'rows per job {n.mean():.1f}; job {tot.mean():.0f} cycles = {tot.mean() / ghz / 1e3:.2f} us (min {tot.min() / ghz / 1e3:.2f}, max {tot.max() / ghz / 1e3:.2f})')
    print(f'  prologue (parameters + weight fragments + first row requested -> staged)   {pro.mean():7.0f} cycles = {pro.mean() / ghz:6.0f} ns = {100 * pro.mean() / tot.mean():4.1f} % of the job')
    print(f'  row start -> row start, mean over the rows of a job                        {mean_row.mean():7.0f} cycles (fastest job {mean_row.min():.0f}, slowest {mean_row.max():.0f}); '
          f'shortest single row {rmin.min():.0f}, median of the jobs\' shortest {np.median(rmin):.0f}, longest single row {rmax.max():.0f}, median of the jobs\' longest {np.median(rmax):.0f}')
    print(f'  last row + last store + squeeze tree                                       {last.mean():7.0f} cycles = {100 * last.mean() / tot.mean():4.1f} % of the job')
    print(f'  launch: first recorded start -> last recorded end {(te.max() - t0.min()) / ghz / 1e3:.1f} us; starts spread over {(t0.max() - t0.min()) / ghz / 1e3:.1f} us, ends over {(te.max() - te.min()) / ghz / 1e3:.1f} us')
    order = np.argsort(t0)
    print('  jobs by start time: start us / duration us / mean row cycles / longest row cycles')
    for j in order[::max(1, len(order) // 16)]:
        print(f'   {(t0[j] - t0.min()) / ghz / 1e3:7.1f} {tot[j] / ghz / 1e3:7.2f} {mean_row[j]:8.0f} {rmax[j]:8.0f}')


if __name__ == '__main__':
    main()
