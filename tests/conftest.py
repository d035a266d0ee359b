import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(tag):
    """(Problem, dict of reference outputs) from tests/golden/ref_<tag>.npz."""
    from joxsz_amd.problem import Problem
    z = np.load(os.path.join(GOLDEN, 'ref_%s.npz' % tag))
    pb = Problem.from_dict(z).validate()
    ref = {k: z[k] for k in z.files if not k.startswith('pb_')}
    return pb, ref


@pytest.fixture(scope='session', params=['tiny', 'bundled'])
def golden(request):
    return load_golden(request.param)


@pytest.fixture(scope='session')
def golden_tiny():
    return load_golden('tiny')


@pytest.fixture(scope='session')
def golden_bundled():
    return load_golden('bundled')


@pytest.fixture(scope='session', params=['tiny_integ', 'tiny_integ_even'])
def golden_integ(request):
    """The reference run with calc_integ = True (joxsz_funcs.py:480-487): odd and even numbers of Simpson samples."""
    return load_golden(request.param)


@pytest.fixture
def legacy_forms(monkeypatch):
    """The contracted forms of rounds 3-4 (low-rank / full, with their sub-grids and the truncation guard): kept for one round behind
    JOXSZ_MIX_FORM=legacy beside the exact form that replaced them as the default."""
    monkeypatch.setenv('JOXSZ_MIX_FORM', 'legacy')
    return monkeypatch
