import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_env():
    """The synthetic environment the golden fixtures were generated on."""
    import numpy as np
    from tropical_cyclone_risk_amd import synthetic
    meta = np.load(os.path.join(GOLDEN, 'tracks_NA.npz'))
    return synthetic.make_env(shape=str(meta['meta_env_shape']), seed=int(meta['meta_env_seed']),
                              zero_cov_patch=bool(meta['meta_env_zero_cov_patch']))


@pytest.fixture(scope='session')
def built_lib():
    from tropical_cyclone_risk_amd import build
    return build.build()
