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

    def _crop_plan(self, lon, lat):
        """Column / row selection of `transform_global_field` for one pair of axes: (lon_b, lat_b, ix, iy, sx, sy) with ix / iy
        the source columns / rows in output order and sx / sy the equivalent slices when they are contiguous runs (else None).
        Cached per axes: a year's 216 planes share two or three grids."""
        cache = self.__dict__.setdefault('_plans', [])
        for c_lon, c_lat, plan, o_lon, o_lat in cache:           # the very same axis objects: no comparison at all
            if o_lon is lon and o_lat is lat:
                return plan
        o_lon, o_lat = lon, lat
        lon = np.asarray(lon)
        lat = np.asarray(lat)
        for c_lon, c_lat, plan, _, _ in cache:
            if c_lon.shape == lon.shape and c_lat.shape == lat.shape and np.array_equal(c_lon, lon) and np.array_equal(c_lat, lat):
                return plan
        x0, y0, x1, y1 = self.get_bounds()
        ix = np.arange(lon.size)
        rot = lon
        if lon[0] >= -_EPS and (x0 < 0 or x1 < 0):
            east = lon >= (180 - _EPS)                                 # _to_pm180
            ix = np.concatenate((ix[east], ix[~east]))
            rot = np.concatenate((lon[east] - 360, lon[~east]))
        elif (lon < 0).any() and x0 >= 0:
            west = lon < -_EPS                                         # _to_0_360
            ix = np.concatenate((ix[~west], ix[west]))
            rot = np.concatenate((lon[~west], lon[west] + 360))
        keep_x = (rot <= x1 + _EPS) & (rot >= x0 - _EPS)
        keep_y = (lat >= y0 - _EPS) & (lat <= y1 + _EPS)
        ix, iy = ix[keep_x], np.nonzero(keep_y)[0]

        def as_slice(i):
            return slice(int(i[0]), int(i[-1]) + 1) if i.size and np.array_equal(i, np.arange(i[0], i[0] + i.size)) else None
        plan = (rot[keep_x], lat[keep_y], ix, iy, as_slice(ix), as_slice(iy))
        cache.append((lon.copy(), lat.copy(), plan, o_lon, o_lat))
        if len(cache) > 16:
            del cache[0]
        return plan

    def transform_global_field(self, lon, lat, field):
        """Rotate lon to the basin's sign convention, crop to box ±1e-5.

        field is [lat, lon]; returns (lon_b, lat_b, field_b) (basins.py:57-75).  The selection is computed once per pair of
        axes; a plane whose selection is a contiguous block comes back as a view of `field` (no copy).
        """
        plan = self._crop_plan(lon, lat)
        return plan[0], plan[1], self._apply_plan(plan, field)

    @staticmethod
    def _apply_plan(plan, field):
        """The selection of `_crop_plan` applied to one [lat, lon] plane."""
        field = np.asarray(field)
        _, _, ix, iy, sx, sy = plan
        if sx is not None and sy is not None:
            return field[sy, sx]
        if sy is not None:
            return field[sy][:, ix]
        return field[iy][:, ix]
