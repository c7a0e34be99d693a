"""pytest configuration: the ``gpu`` marker and shared helpers.

``-m "not gpu"`` tests run on any host (oracle vs golden vectors, host logic,
ABI surface).  ``-m gpu`` tests need an MI355X and call the HIP kernels through
the C ABI; they compare against the CPU oracle in ``oracle/`` (the checker only).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (runs the HIP kernels)')


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this environment')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle module, with its C half built on demand."""
    lib = os.path.join(ROOT, 'oracle', 'liboracle.so')
    if not os.path.exists(lib):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    import gcn_oracle
    return gcn_oracle


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session', params=['cora', 'citeseer', 'pubmed_graph'])
def graph_fx(request):
    return request.param, load_golden(request.param)


@pytest.fixture(scope='session', params=['cora', 'citeseer'])
def dataset_fx(request):
    return request.param, load_golden(request.param)
