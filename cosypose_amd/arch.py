"""EfficientNet-B3 (6 input channels) architecture table, derived from the scaling rule.

The reference instantiates ``EfficientNet.from_name('efficientnet-b3', in_channels=6)``
(cosypose/training/pose_models_cfg.py:23).  The layer table follows from the base
EfficientNet-B0 stage list (cosypose/models/efficientnet_utils.py:259-264) scaled by
width 1.2 / depth 1.4 (:169) with filters rounded to multiples of 8 (:60-72) and
repeats rounded up (:75-80).  The same table is hard-coded in csrc/effnet_arch.h
and oracle/cosy_oracle.c; tests assert the three agree.
"""
import math

IN_C = 6
N_POSE = 9
BN_EPS = 1e-3
_WIDTH, _DEPTH, _DIVISOR = 1.2, 1.4, 8
_STATIC_IMAGE_SIZE = 300  # padding is fixed at construction from this, not from the input

# base stages: (kernel, repeats, in, out, expand, stride)
_BASE = [(3, 1, 32, 16, 1, 1), (3, 2, 16, 24, 6, 2), (5, 2, 24, 40, 6, 2), (3, 3, 40, 80, 6, 2),
         (5, 3, 80, 112, 6, 1), (5, 4, 112, 192, 6, 2), (3, 1, 192, 320, 6, 1)]


def _round_filters(f):
    f *= _WIDTH
    new = max(_DIVISOR, int(f + _DIVISOR / 2) // _DIVISOR * _DIVISOR)
    if new < 0.9 * f:
        new += _DIVISOR
    return int(new)


def b3_block_table():
    blocks = []
    for k, r, i, o, e, s in _BASE:
        i, o, r = _round_filters(i), _round_filters(o), int(math.ceil(_DEPTH * r))
        blocks.append((k, s, e, i, o))
        blocks += [(k, 1, e, o, o)] * (r - 1)
    return blocks


B3_BLOCKS = b3_block_table()          # 26 x (k, s, expand, cin, cout)
STEM_C = _round_filters(32)           # 40
HEAD_C = _round_filters(1280)         # 1536
STAGE_END = (1, 4, 7, 12, 17, 23, 25)  # last block index of each of the 7 stages


def se_channels(cin):
    """Squeeze width comes from the block's *input* filters (efficientnet.py:61)."""
    return max(1, int(cin * 0.25))


def static_pad(k, s):
    """(lo, hi) zero padding of Conv2dStaticSamePadding(image_size=300), efficientnet_utils.py:130-141."""
    out = math.ceil(_STATIC_IMAGE_SIZE / s)
    tot = max((out - 1) * s + (k - 1) + 1 - _STATIC_IMAGE_SIZE, 0)
    return tot // 2, tot - tot // 2


def conv_out(n, k, s):
    lo, hi = static_pad(k, s)
    return (n + lo + hi - k) // s + 1


def feature_hw(H, W):
    h, w = conv_out(H, 3, 2), conv_out(W, 3, 2)
    for k, s, *_ in B3_BLOCKS:
        h, w = conv_out(h, k, s), conv_out(w, k, s)
    return h, w


def param_count():
    n = STEM_C * IN_C * 9 + 4 * STEM_C
    for k, s, e, cin, cout in B3_BLOCKS:
        cmid, cse = cin * e, se_channels(cin)
        if e != 1:
            n += cmid * cin + 4 * cmid
        n += cmid * k * k + 4 * cmid + cse * cmid + cse + cmid * cse + cmid + cout * cmid + 4 * cout
    return n + HEAD_C * 384 + 4 * HEAD_C + N_POSE * HEAD_C + N_POSE
