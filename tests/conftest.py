import os
import sys
import pathlib

import numpy as np
import pytest

REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / 'oracle'))  # test infrastructure: the parity oracle


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    return dict(np.load(REPO / 'tests' / 'golden' / 'reference_golden.npz', allow_pickle=False))


@pytest.fixture(scope='session')
def golden_dist():
    """distance-op fixtures (symmetric distances, loss argmin, ADD/ADD-S) generated from the reference's Python"""
    return dict(np.load(REPO / 'tests' / 'golden' / 'reference_golden_dist.npz', allow_pickle=False))


@pytest.fixture(scope='session')
def golden_train():
    """one reference training step (loss, pose outputs, gradients, BN running statistics, Adam update)"""
    return dict(np.load(REPO / 'tests' / 'golden' / 'reference_golden_train.npz', allow_pickle=False))


@pytest.fixture(scope='session')
def oracle():
    import cosy_oracle
    cosy_oracle.build()
    return cosy_oracle


@pytest.fixture(scope='session')
def golden_sd():
    from cosypose_amd import synthetic
    return synthetic.golden_state_dict(0)


@pytest.fixture(scope='session')
def mesh_table(golden):
    """(21, 2000, 3): the mesh DB of the golden generator after sample_points(2000, deterministic=True)."""
    from cosypose_amd import synthetic
    pts = synthetic.make_mesh_points(7, 21, 2500)
    return pts[:, golden['sample_ids_2500']]


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def pose_errors(got, want):
    """Per-parameter-group deviation of (B,4,4) poses, each pose judged on its own scale (a max-norm over the whole batch
    hides a centimetre-scale translation behind a metre-scale one):
      rotation    : max over poses of max |R - R'|           (entries of a rotation are O(1): absolute == relative)
      translation : max over poses of max |t - t'| / max |t'|  (relative to that pose's own translation)."""
    a = np.asarray(got, np.float64).reshape(-1, 4, 4); b = np.asarray(want, np.float64).reshape(-1, 4, 4)
    rot = float(np.abs(a[:, :3, :3] - b[:, :3, :3]).max()) if len(a) else 0.0
    tn = np.maximum(np.abs(b[:, :3, 3]).max(axis=1), 1e-12)
    tr = float((np.abs(a[:, :3, 3] - b[:, :3, 3]).max(axis=1) / tn).max()) if len(a) else 0.0
    return rot, tr


def rows_rel_err(got, want):
    """max over rows (crops) of max|a - b| / max|b| of that row: per-crop relative error of K_crop / boxes."""
    a = np.asarray(got, np.float64).reshape(len(got), -1); b = np.asarray(want, np.float64).reshape(len(want), -1)
    return float((np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-12)).max()) if len(a) else 0.0
