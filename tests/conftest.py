import ast
import json
import os
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a fresh checkout has no built artefacts (they are git-ignored): build the HIP library and the oracle once
    if not os.path.exists(os.path.join(ROOT, 'nemoflux_amd', 'libnemoflux_amd.so')):
        import __graft_entry__
        __graft_entry__.build()


def load_cases():
    with open(os.path.join(GOLDEN, 'cases.json')) as f:
        return json.load(f)


def load_golden(name):
    return numpy.load(os.path.join(GOLDEN, name + '.npz'))


def transect_xyz(points_str):
    xy = numpy.array(ast.literal_eval(points_str), dtype=numpy.float64)
    xyz = numpy.zeros((xy.shape[0], 3), numpy.float64)
    xyz[:, :2] = xy
    return xyz


@pytest.fixture(scope='session')
def oracle():
    import nf_oracle
    nf_oracle.build()
    return nf_oracle


@pytest.fixture(scope='session')
def cases():
    return load_cases()


FULL_CASES = ['c1_x', 'singular', 'cossin36', 'rot36_zt', 'def36_zt', 'sv36_land']
