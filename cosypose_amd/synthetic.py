"""Seeded synthetic inputs for parity tests and bench.py.

Everything is generated with numpy's legacy ``RandomState`` (bit-stable across
numpy versions and machines), so the build container and the GPU box produce
identical tensors from a seed and golden fixtures only need to hold outputs.
"""
import numpy as np

# (k, s, expand, cin, cout) per MBConv block of EfficientNet-B3 with 6 input
# channels, as instantiated by EfficientNet.from_name('efficientnet-b3', in_channels=6)
# (reference cosypose/training/pose_models_cfg.py:23; block strings
# cosypose/models/efficientnet_utils.py:259-264).  Derived, not copied:
# see cosypose_amd.efficientnet.b3_block_table().
from .arch import B3_BLOCKS, STEM_C, HEAD_C, IN_C, N_POSE


def state_dict_shapes(prefix='backbone.'):
    """Ordered {key: shape} of the reference PosePredictor state_dict (sans num_batches_tracked)."""
    out = {}

    def bn(p, c):
        for s in ('weight', 'bias', 'running_mean', 'running_var'):
            out[f'{p}.{s}'] = (c,)
    out[prefix + '_conv_stem.weight'] = (STEM_C, IN_C, 3, 3)
    bn(prefix + '_bn0', STEM_C)
    for i, (k, s, e, cin, cout) in enumerate(B3_BLOCKS):
        p = f'{prefix}_blocks.{i}.'
        cmid = cin * e
        cse = max(1, int(cin * 0.25))
        if e != 1:
            out[p + '_expand_conv.weight'] = (cmid, cin, 1, 1)
            bn(p + '_bn0', cmid)
        out[p + '_depthwise_conv.weight'] = (cmid, 1, k, k)
        bn(p + '_bn1', cmid)
        out[p + '_se_reduce.weight'] = (cse, cmid, 1, 1)
        out[p + '_se_reduce.bias'] = (cse,)
        out[p + '_se_expand.weight'] = (cmid, cse, 1, 1)
        out[p + '_se_expand.bias'] = (cmid,)
        out[p + '_project_conv.weight'] = (cout, cmid, 1, 1)
        bn(p + '_bn2', cout)
    out[prefix + '_conv_head.weight'] = (HEAD_C, 384, 1, 1)
    bn(prefix + '_bn1', HEAD_C)
    out['pose_fc.weight'] = (N_POSE, HEAD_C)
    out['pose_fc.bias'] = (N_POSE,)
    return out


def golden_state_dict(seed=0):
    """Well-conditioned random weights (SURVEY 8c "Golden weights"): variance-preserving
    convs, randomised BN statistics, and a pose head with small weights and bias
    [1,0,0, 0,1,0, 0,0,1] so that dR ~ I and (vx,vy,vz) ~ (0,0,1): poses stay finite
    over many iterations.  Returns {key: np.float32 array}."""
    rs = np.random.RandomState(seed)
    sd = {}
    for k, shp in state_dict_shapes().items():
        if k.endswith('running_var'):
            v = rs.uniform(0.5, 1.5, shp)
        elif k.endswith('running_mean'):
            v = rs.normal(0, 0.1, shp)
        elif '_bn' in k and k.endswith('.weight'):
            v = rs.uniform(0.7, 1.3, shp)
        elif '_bn' in k and k.endswith('.bias'):
            v = rs.normal(0, 0.1, shp)
        elif k == 'pose_fc.weight':
            v = rs.normal(0, 1.0 / np.sqrt(HEAD_C), shp) * 1e-2
        elif k == 'pose_fc.bias':
            v = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float64)
        elif k.endswith('.bias'):
            v = rs.normal(0, 0.1, shp)
        else:  # conv weight (O, I/groups, kh, kw)
            fan_in = shp[1] * shp[2] * shp[3]
            v = rs.normal(0, np.sqrt(1.8 / fan_in), shp)
        sd[k] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def make_frames(seed, n, h, w):
    """(n,3,h,w) uniform [0,1) fp32 frames."""
    return np.random.RandomState(seed).random_sample((n, 3, h, w)).astype(np.float32)


def make_K(n, h, w):
    """fx=fy=1066.8*(w/640), cx=w/2-7, cy=h/2+1.3 (YCB-V-like)."""
    K = np.zeros((n, 3, 3), np.float32)
    K[:, 0, 0] = K[:, 1, 1] = 1066.8 * (w / 640.0)
    K[:, 0, 2] = w / 2.0 - 7.0
    K[:, 1, 2] = h / 2.0 + 1.3
    K[:, 2, 2] = 1.0
    return K


def make_mesh_points(seed, n_obj, n_pts=2500):
    """(n_obj, n_pts, 3) points uniform in a box of half-extent U(0.03,0.12) m per axis."""
    rs = np.random.RandomState(seed)
    ext = rs.uniform(0.03, 0.12, (n_obj, 1, 3))
    pts = rs.uniform(-1, 1, (n_obj, n_pts, 3)) * ext
    return pts.astype(np.float32)


def make_detections(seed, n_det, n_frames, n_obj, h, w):
    """label ids (n_det,), batch_im_id (n_det,), bboxes (n_det,4): centre in the central
    60% of the frame, side U(60,220) px."""
    rs = np.random.RandomState(seed)
    obj = rs.randint(0, n_obj, n_det)
    im = rs.randint(0, n_frames, n_det)
    cx = rs.uniform(0.2 * w, 0.8 * w, n_det); cy = rs.uniform(0.2 * h, 0.8 * h, n_det)
    sw = rs.uniform(60, 220, n_det) * (min(h, w) / 480.0); sh = rs.uniform(60, 220, n_det) * (min(h, w) / 480.0)
    boxes = np.stack([cx - sw / 2, cy - sh / 2, cx + sw / 2, cy + sh / 2], 1).astype(np.float32)
    return obj.astype(np.int64), im.astype(np.int64), boxes


def make_TCO(seed, n, z_range=(0.6, 1.4), xy=0.15):
    """(n,4,4) random well-conditioned object poses in front of the camera."""
    rs = np.random.RandomState(seed)
    q = rs.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    w_, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w_), 2 * (x * z + y * w_),
                  2 * (x * y + z * w_), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w_),
                  2 * (x * z - y * w_), 2 * (y * z + x * w_), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
    T = np.tile(np.eye(4), (n, 1, 1))
    T[:, :3, :3] = R
    T[:, 0, 3] = rs.uniform(-xy, xy, n); T[:, 1, 3] = rs.uniform(-xy, xy, n); T[:, 2, 3] = rs.uniform(*z_range, n)
    return T.astype(np.float32)


def make_renders(seed, n, H, W):
    """Deterministic stand-in for renderer.render: (n,3,H,W) in [0,1)."""
    return np.random.RandomState(seed).random_sample((n, 3, H, W)).astype(np.float32)


def make_training_batch(seed, B, n_obj=21, h=480, w=640):
    """One synthetic training batch: uint8 frames (B,3,h,w), K (B,3,3), ground-truth
    poses (B,4,4) and object ids (B,)."""
    frames = (make_frames(seed, B, h, w) * 255).astype(np.uint8)
    K = make_K(B, h, w)
    TCO = make_TCO(seed + 1, B)
    obj = np.random.RandomState(seed + 2).randint(0, n_obj, B).astype(np.int32)
    return frames, K, TCO, obj


def make_render_meshes(seed, n_obj, n_lat=24, n_lon=32):
    """Closed triangle meshes for the rasteriser: bumpy ellipsoids with the half-extents make_mesh_points(seed, ...) draws
    (so that geometry points and render meshes describe the same objects) and smooth vertex colours.
    -> verts list [(V,3)], faces list [(F,3) int32], colors list [(V,3)]"""
    rs = np.random.RandomState(seed)
    ext = rs.uniform(0.03, 0.12, (n_obj, 1, 3))
    rs = np.random.RandomState(seed + 1000)
    lat = np.linspace(0, np.pi, n_lat + 1)[1:-1]
    lon = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    verts_l, faces_l, colors_l = [], [], []
    ring = lambda i: 1 + i * n_lon
    faces = []
    for j in range(n_lon):
        faces.append((0, ring(0) + j, ring(0) + (j + 1) % n_lon))
    for i in range(len(lat) - 1):
        for j in range(n_lon):
            a, b = ring(i) + j, ring(i) + (j + 1) % n_lon
            c, d = ring(i + 1) + j, ring(i + 1) + (j + 1) % n_lon
            faces += [(a, c, b), (b, c, d)]
    south = 1 + len(lat) * n_lon
    for j in range(n_lon):
        faces.append((south, ring(len(lat) - 1) + (j + 1) % n_lon, ring(len(lat) - 1) + j))
    faces = np.asarray(faces, np.int32)
    for o in range(n_obj):
        dirs = [np.array([0, 0, 1.0])]
        for t in lat:
            for p in lon:
                dirs.append(np.array([np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t)]))
        dirs.append(np.array([0, 0, -1.0]))
        dirs = np.asarray(dirs)
        f = rs.uniform(1, 3, 3); ph = rs.uniform(0, 6.28, 3)
        bump = 1 + 0.15 * np.sin(f[0] * dirs[:, 0] * 3 + ph[0]) * np.cos(f[1] * dirs[:, 1] * 3 + ph[1]) + 0.1 * np.sin(f[2] * dirs[:, 2] * 4 + ph[2])
        verts_l.append((dirs * bump[:, None] * ext[o]).astype(np.float32))
        base = rs.uniform(0.2, 0.9, 3)
        colors_l.append(np.clip(base + 0.3 * dirs * rs.uniform(-1, 1, 3), 0, 1).astype(np.float32))
        faces_l.append(faces)
    return verts_l, faces_l, colors_l
