import json
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


def load_golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def load_schema(name):
    with open(os.path.join(GOLD, f'schema_{name}.json')) as f:
        return OrderedDict((k, (tuple(s), d)) for k, s, d in json.load(f))


def rms(a):
    a = np.abs(np.asarray(a)).astype(np.float64)
    return float(np.sqrt(np.mean(a * a)))


@pytest.fixture(scope='session')
def golden():
    return load_golden


@pytest.fixture(scope='session')
def schema():
    return load_schema
