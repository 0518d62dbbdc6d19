#!/usr/bin/env python3
"""Run the reference's own `scripts/generate_land_masks.generate_land_masks()` (lines 11-110) in the build container
and commit what it writes, bit-packed, as tests/golden/masks.npz.  Needs /root/reference:

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_masks.py

Neither xarray nor global_land_mask is installable here, and the script uses them only as containers / a data source:
  * `xr.DataArray(data=..., dims=..., coords=...)`, `xr.Dataset(data_vars=...)`, `.to_netcdf(fn)`: a recording stub that
    keeps `.data`, supports the `~` and `&` the script applies to `land`, and captures {file name: {variable: array}};
  * `global_land_mask.globe.is_land(lat_grid, lon_grid)`: the analytic planet of tests/golden/planted_land.py.
Every line of geometry — the 0.25-degree grid, `TC_Basin.transform_lon_r`, the boxes, the two staircases, `& ~land`,
the |lat| > 50 cut — is the reference's, executed unmodified.  The script writes into `./land/`, so it runs in a temporary
directory.  Data only: the nine arrays it handed to `to_netcdf`, and the coordinates it attached to them.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from tests.golden import planted_land               # noqa: E402
from tests.golden import ref_harness as H           # noqa: E402

WRITTEN = {}


class DataArray:
    __array_ufunc__ = None          # ndarray & DataArray -> DataArray.__rand__, as with the real class

    def __init__(self, data=None, dims=None, coords=None):
        self.data, self.dims, self.coords = np.asarray(data), dims, dict(coords or {})

    def __invert__(self):
        return DataArray(~self.data, self.dims, self.coords)

    def __and__(self, other):
        return np.asarray(getattr(other, 'data', other)) & self.data

    __rand__ = __and__


class Dataset:
    def __init__(self, data_vars=None):
        self.data_vars = dict(data_vars)

    def to_netcdf(self, fn):
        WRITTEN[os.path.basename(fn)] = {k: (np.array(v.data), {c: np.array(x) for c, x in v.coords.items()}, v.dims)
                                         for k, v in self.data_vars.items()}


def main():
    H.import_reference()                                  # stubs xarray / dask / cftime / global_land_mask, sets sys.path
    xr = sys.modules['xarray']
    xr.DataArray, xr.Dataset = DataArray, Dataset
    glm = sys.modules['global_land_mask']
    glm.globe = types.SimpleNamespace(is_land=planted_land.is_land)
    from scripts import generate_land_masks as G          # the reference's module
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            G.generate_land_masks()
        finally:
            os.chdir(cwd)
    names = ['land', 'NA', 'EP', 'WP', 'NI', 'SI', 'AU', 'SP', 'GL']
    assert sorted(WRITTEN) == sorted(n + '.nc' for n in names), sorted(WRITTEN)
    out = {}
    for n in names:
        (var, (data, coords, dims)), = WRITTEN[n + '.nc'].items()
        assert var == ('land' if n == 'land' else 'basin') and tuple(dims) == ('lat', 'lon') and data.shape == (721, 1440)
        out['mask_' + n] = np.packbits(data.astype(bool), axis=1)
        out['lon_' + n], out['lat_' + n] = coords['lon'], coords['lat']
        print('%-4s %7d points set   lon[0] = %g' % (n, int(data.sum()), coords['lon'][0]))
    np.savez_compressed(os.path.join(HERE, 'masks.npz'), **out)


if __name__ == '__main__':
    main()
