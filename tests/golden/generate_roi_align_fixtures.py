#!/usr/bin/env python3
"""Hand-derivable fixtures for roi_align with torchvision-0.4.2 semantics (the op behind
cosypose/lib3d/cropping.py:64-75: `torchvision.ops.roi_align(images, boxes, output_size, sampling_ratio=4)`, spatial
scale 1, legacy "aligned=False" pixel model).  torchvision is not installed here and the reference has no vectors, so
the expected values below do NOT come from any restatement of the 2-D algorithm in this repository: they follow from
three facts about that algorithm that can be checked by hand, applied to images of the form  I[c, y, x] = gy[c, y] * gx[c, x].

  (F1) an output bin is the plain mean of its sampling_ratio x sampling_ratio bilinear samples placed at
           y_s = y1 + (ph + (iy + 0.5) / S) * bin_h,   bin_h = max(y2 - y1, 1) / out_h     (same for x; S = 4);
       note max(roi, 1): a box thinner than one pixel is sampled as if it were one pixel wide;
  (F2) a sample outside [-1, n] on either axis contributes 0; inside, the coordinate is clamped into [0, n - 1]
       (s <= 0 -> 0; floor(s) >= n - 1 -> n - 1) and the four neighbours are blended bilinearly;
  (F3) bilinear blending and the validity test are both separable, so for a separable image every sample is
           phi_y(y_s) * phi_x(x_s),   phi(s) = 0 if s < -1 or s > n else lerp(g, clamp(s, 0, n - 1)),
       and the mean over the S x S grid of a bin factorises into (mean over iy of phi_y) * (mean over ix of phi_x).

So the expected output is an outer product of two 1-D means -- ten lines of 1-D numpy (`phi`, `axis_means`), no 2-D gather.
Cases: affine ramps (closed form: the value at the bin centre), a box thinner than a pixel (the max(roi,1) rule), one-hot
rows / columns at the borders (the [-1, n] window and the clamp at n - 1), boxes partly and fully outside the image.
Writes tests/golden/roi_align_handmade.npz.
"""
import os
import numpy as np

S = 4


def phi(g, s):
    """1-D sample of the 0.4.2 bilinear rule on the row/column profile g at coordinate s (float64)."""
    n = len(g)
    if s < -1.0 or s > n:
        return 0.0
    s = max(s, 0.0)
    lo = int(s)
    if lo >= n - 1:
        return float(g[n - 1])
    f = s - lo
    return float(g[lo] * (1.0 - f) + g[lo + 1] * f)


def axis_means(g, a, b, n_out):
    size = max(b - a, 1.0)
    bin_ = size / n_out
    return np.array([np.mean([phi(g, a + (o + (i + 0.5) / S) * bin_) for i in range(S)]) for o in range(n_out)])


def expected(gy, gx, rois, out_hw):
    """gy (N,C,h), gx (N,C,w): image n, channel c is the outer product gy[n,c] (x) gx[n,c]."""
    oh, ow = out_hw
    out = np.zeros((len(rois), gy.shape[1], oh, ow))
    for r, (n, x1, y1, x2, y2) in enumerate(rois):
        for c in range(gy.shape[1]):
            out[r, c] = np.outer(axis_means(gy[int(n), c], y1, y2, oh), axis_means(gx[int(n), c], x1, x2, ow))
    return out


def main():
    h, w, N, C = 9, 11, 2, 3
    rs = np.random.RandomState(0)
    gy = np.zeros((N, C, h)); gx = np.zeros((N, C, w))
    # image 0: channel 0 = x ramp (gy = 1), channel 1 = y ramp (gx = 1), channel 2 = product of two random profiles
    gy[0, 0] = 1.0; gx[0, 0] = 0.5 * np.arange(w) + 2.0
    gy[0, 1] = -0.25 * np.arange(h) + 3.0; gx[0, 1] = 1.0
    gy[0, 2] = rs.uniform(-1, 1, h); gx[0, 2] = rs.uniform(-1, 1, w)
    # image 1: one-hot profiles at the borders (first / last row and column) and one in the interior
    gy[1, 0, h - 1] = 1.0; gx[1, 0] = 1.0
    gy[1, 1] = 1.0; gx[1, 1, 0] = 1.0
    gy[1, 2, 4] = 1.0; gx[1, 2, w - 1] = 1.0
    images = np.einsum('nch,ncw->nchw', gy, gx).astype(np.float32)
    rois = np.array([
        [0, 1.0, 1.0, 9.0, 7.0],          # inside: affine channels = value at the bin centre
        [0, 2.25, 3.5, 2.75, 3.6],        # thinner than a pixel in both axes: max(roi, 1)
        [0, -3.0, -2.0, 6.0, 5.0],        # hangs over the top-left corner: [-1, 0] clamps to 0, beyond -1 -> 0
        [1, 4.0, 5.0, 14.0, 12.0],        # hangs over the bottom-right corner: (n-1, n] clamps to n-1, beyond n -> 0
        [1, -0.9, 7.2, 0.7, 9.9],         # narrow box straddling the left border and the last row
        [1, 20.0, 20.0, 30.0, 30.0],      # fully outside -> zeros
        [0, 0.0, 0.0, 11.0, 9.0],         # the whole image extent
    ], np.float64)
    out_hw = (6, 8)
    exp = expected(gy, gx, rois, out_hw)
    # the closed forms the docstring promises, checked here
    n, x1, y1, x2, y2 = rois[0]
    cx = x1 + (np.arange(out_hw[1]) + 0.5) * (x2 - x1) / out_hw[1]
    cy = y1 + (np.arange(out_hw[0]) + 0.5) * (y2 - y1) / out_hw[0]
    assert np.allclose(exp[0, 0], np.tile(0.5 * cx + 2.0, (out_hw[0], 1)))
    assert np.allclose(exp[0, 1], np.tile((-0.25 * cy + 3.0)[:, None], (1, out_hw[1])))
    cx1 = 2.25 + (np.arange(out_hw[1]) + 0.5) / out_hw[1]          # width forced to 1
    assert np.allclose(exp[1, 0], np.tile(0.5 * cx1 + 2.0, (out_hw[0], 1)))
    assert np.all(exp[5] == 0.0)
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, 'roi_align_handmade.npz'), images=images, rois=rois.astype(np.float32),
                        expected=exp.astype(np.float32), out_hw=np.array(out_hw))
    print('wrote roi_align_handmade.npz', images.shape, rois.shape, exp.shape)


if __name__ == '__main__':
    main()
