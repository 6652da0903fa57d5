"""Round 6: the crop + pack launch alone on the bench's crop geometry (256 crops of HxW from 16 frames of 512^2; boxes = the synthetic
detections grown by deepim's 1.4 to the crop's aspect).  usage: python profiles/exp/crop_bench.py [H W]
With COSY_TUNE_LIB=1 the knob COSY_CROP_TILED=0 selects the per-pixel kernel on the same tap tables (round 5's kernel)."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from cosypose_amd import synthetic as syn
from cosypose_amd._lib import lib, check, ptr, stream, COSY_F16
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 256)
B, N, h, w = 256, 16, 512, 512
_, im, det = syn.make_detections(3, B, N, 21, h, w)
cx, cy = (det[:, 0] + det[:, 2]) / 2, (det[:, 1] + det[:, 3]) / 2
xd, yd = (det[:, 2] - det[:, 0]) / 2, (det[:, 3] - det[:, 1]) / 2
r = W / H
bw, bh = np.maximum(xd, yd * r) * 2 * 1.4, np.maximum(xd / r, yd) * 2 * 1.4
boxes = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).astype(np.float32)
frames4 = torch.rand(N, h, w, 4, device='cuda')
renders = torch.rand(B, 3, H, W, device='cuda')
im_d, boxes_d = torch.tensor(im, dtype=torch.int32, device='cuda'), torch.tensor(boxes, device='cuda')
x = torch.empty(B, H, W, 8, dtype=torch.float16, device='cuda')
ws = torch.empty(lib().cosy_crop_pack_workspace_bytes(B, H, W), dtype=torch.uint8, device='cuda')
def run():
    check(lib().cosy_crop_pack_to_ws(ptr(x), COSY_F16, ptr(frames4), ptr(im_d), ptr(boxes_d), ptr(renders), B, N, h, w, H, W, ptr(ws), stream()))
for _ in range(5): run()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(json.dumps({'crop': [H, W], 'tiled': os.environ.get('COSY_CROP_TILED', '1'), 'us_per_launch_pair': round(min(ts), 1), 'all': [round(t, 1) for t in ts],
                  'bin_mean': round(float((bw / W).mean()), 3), 'bin_max': round(float((bw / W).max()), 3), 'checksum': float(x.float().sum())}))
