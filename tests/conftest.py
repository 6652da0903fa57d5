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
