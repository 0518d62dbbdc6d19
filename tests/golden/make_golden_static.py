#!/usr/bin/env python3
"""Golden tracks on the reference's REAL static-field shape, from the reference's own code (build container only):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_static.py

`intensity/data/land.nc` is int8 0 / 1 on a 0.125-degree grid (lon 0 .. 359.875, lat -89.875 .. 90); `geo.read_land` hands that
int8 array — cropped by `transform_global_field` — to RectBivariateSpline(kx=1, ky=1) (intensity/geo.py:23-34), and
`_get_over_land` tests the interpolated value `== 1` (coupled_fast.py:35-38).  `synthetic.make_env(static_res=0.125)` builds planes
of that grid and type (int8 land, int16 whole-metre bathymetry; the reference's bathymetry.nc is not shipped), and the harness
(ref_harness.build_coupled_fast) passes them to the reference exactly as geo.py would.  Output: tracks_NA_res0125.npz — inputs,
the reference's tracks, env winds, vmax, accept flags and its `land == 1` decision at every dydt call.  Data only.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from tests.golden import make_golden as G            # noqa: E402
from tests.golden import ref_harness as H            # noqa: E402
from tropical_cyclone_risk_amd import synthetic      # noqa: E402


def main():
    ref = H.import_reference()
    kw = dict(G.ENV_KW, static_res=0.125, bathy_kind='i16')
    env = synthetic.make_env(**kw)
    assert env.land.dtype == np.int8 and env.land.shape == (1440, 2880)
    warnings.simplefilter('ignore')
    d = G.run_set(ref, env, 'NA', 260, 404, per_class=5)
    meta = dict(G.META, env_static_res=0.125, env_bathy_kind='i16')
    np.savez_compressed(os.path.join(HERE, 'tracks_NA_res0125.npz'), **d, **{'meta_' + k: v for k, v in meta.items()})
    print('%d tracks, %d bytes' % (len(d['n_valid']), os.path.getsize(os.path.join(HERE, 'tracks_NA_res0125.npz'))))


if __name__ == '__main__':
    main()
