import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "slow: multi-second full-size CPU case")
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def tiny_weights():
    return {k: torch.from_numpy(v) for k, v in load_golden('tiny_weights.npz').items()}


@pytest.fixture(scope='session')
def tiny_forward():
    return load_golden('tiny_forward.npz')


@pytest.fixture(scope='session')
def tiny_train():
    return load_golden('tiny_train.npz')


@pytest.fixture(scope='session')
def tiny_decode():
    return load_golden('tiny_decode.npz')


@pytest.fixture(scope='session')
def nano224_golden():
    return load_golden('nano224.npz')
