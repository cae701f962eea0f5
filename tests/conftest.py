import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _decode(x):
    if isinstance(x, dict):
        if '__nd__' in x:
            return np.array(x['__nd__'], dtype=np.dtype(x['dtype']))
        return {k: _decode(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_decode(v) for v in x]
    return x


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return _decode(json.load(f))


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as _o
    _o.build()
    return _o
