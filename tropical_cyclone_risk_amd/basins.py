"""Basin boxes and global→basin field cropping.

Host-side mirror of the reference's ``TC_Basin`` (`util/basins.py:11-107`):
same public names (``get_bounds``, ``in_basin``, ``transform_global_field``)
and the same semantics, because the cropped grid decides where bilinear lookups
clamp and therefore is part of the numerical contract of the hot path.
"""
import numpy as np

from . import namelist

_EPS = 1e-5

def basin_ids(nl=None):
    """Order used for per-basin tables: sorted ids without 'GL' (compute.py:87), of `nl` (default: the package namelist,
    read now — a later `namelist.load()` is seen)."""
    return tuple(sorted(k for k in (nl or namelist).basin_bounds if k != 'GL'))


BASIN_IDS = basin_ids()      # the seven reference basins; the device tables (TCR_N_BASINS) are laid out in this order


def _parse_bound(token):
    """'260E' -> 260.0, '45S' -> -45.0 (basins.py:22-26)."""
    val = float(token[:-1])
    return -val if token[-1] in 'WS' else val


class TC_Basin:
    def __init__(self, basin_id, nl=None):
        """`nl`: the namelist whose `basin_bounds` define the box (default: the package namelist)."""
        bounds = (nl or namelist).basin_bounds
        if basin_id.upper() not in bounds:
            raise ValueError('Basin ID is not valid. See list of valid basins.')
        self.basin_id = basin_id
        self.basin_bounds = bounds[basin_id]

    def get_bounds(self):
        """(lon_min, lat_min, lon_max, lat_max) in degrees."""
        return tuple(_parse_bound(b) for b in self.basin_bounds)

    def in_basin(self, clon, clat, dx):
        """Strictly inside the box shrunk by dx degrees (basins.py:32-37)."""
        x0, y0, x1, y1 = self.get_bounds()
        return bool((x0 + dx) < clon < (x1 - dx) and (y0 + dx) < clat < (y1 - dx))

    # -- longitude convention helpers ---------------------------------------
    @staticmethod
    def _to_pm180(lon, field):
        east = lon >= (180 - _EPS)
        return (np.concatenate((lon[east] - 360, lon[~east])),
                np.concatenate((field[:, east], field[:, ~east]), axis=1))

    @staticmethod
    def _to_0_360(lon, field):
        west = lon < -_EPS
        return (np.concatenate((lon[~west], lon[west] + 360)),
                np.concatenate((field[:, ~west], field[:, west]), axis=1))

    def transform_global_field(self, lon, lat, field):
        """Rotate lon to the basin's sign convention, crop to box ±1e-5.

        field is [lat, lon]; returns (lon_b, lat_b, field_b) (basins.py:57-75).
        """
        lon = np.asarray(lon)
        lat = np.asarray(lat)
        field = np.asarray(field)
        x0, y0, x1, y1 = self.get_bounds()
        if lon[0] >= -_EPS and (x0 < 0 or x1 < 0):
            lon, field = self._to_pm180(lon, field)
        elif (lon < 0).any() and x0 >= 0:
            lon, field = self._to_0_360(lon, field)
        keep_x = (lon <= x1 + _EPS) & (lon >= x0 - _EPS)
        keep_y = (lat >= y0 - _EPS) & (lat <= y1 + _EPS)
        return lon[keep_x], lat[keep_y], field[keep_y][:, keep_x]
