"""ADD / ADD-S point distances, same surface as the reference's cosypose/lib3d/distances.py:5-21
(HIP: cosy_dists_add; the (B,P,P,3) intermediate of dists_add_symmetric never exists)."""
import torch

from ._lib import lib, check, ptr, stream, require_device


def _dists(TXO_pred, TXO_gt, points, symmetric):
    bsz, n_pts = points.shape[:2]
    assert TXO_pred.shape == (bsz, 4, 4) and TXO_gt.shape == (bsz, 4, 4) and points.shape == (bsz, n_pts, 3)
    out = torch.empty(bsz, n_pts, 3, device=points.device)
    if bsz == 0:
        return out
    require_device(TXO_pred, TXO_gt, points)
    p, g, pts = (t.detach().float().contiguous() for t in (TXO_pred, TXO_gt, points))
    check(lib().cosy_dists_add(ptr(p), ptr(g), ptr(pts), None, bsz, n_pts, int(symmetric), ptr(out), stream()))
    return out


def dists_add(TXO_pred, TXO_gt, points):
    """gt points - predicted points, (B,P,3)"""
    return _dists(TXO_pred, TXO_gt, points, False)


def dists_add_symmetric(TXO_pred, TXO_gt, points):
    """every gt point minus its NEAREST predicted point, (B,P,3)"""
    return _dists(TXO_pred, TXO_gt, points, True)
