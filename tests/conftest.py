import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    return np.load(os.path.join(GOLDEN_DIR, 'golden.npz'))


@pytest.fixture(scope='session')
def g1():
    with open(os.path.join(GOLDEN_DIR, 'g1_parser.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def g5_names():
    with open(os.path.join(GOLDEN_DIR, 'g5_param_names.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_extra():
    return np.load(os.path.join(GOLDEN_DIR, 'golden_extra.npz'))


@pytest.fixture(scope='session')
def golden_r2():
    return np.load(os.path.join(GOLDEN_DIR, 'golden_r2.npz'))


@pytest.fixture(scope='session')
def r2_names():
    with open(os.path.join(GOLDEN_DIR, 'g_r2_names.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def g12():
    with open(os.path.join(GOLDEN_DIR, 'g12_checkpoints.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_r3():
    return np.load(os.path.join(GOLDEN_DIR, 'golden_r3.npz'))


@pytest.fixture(scope='session')
def r3_names():
    with open(os.path.join(GOLDEN_DIR, 'g_r3_names.json')) as f:
        return json.load(f)
