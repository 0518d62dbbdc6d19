"""Land / basin mask generation: the reference's `scripts/generate_land_masks.py:11-110`.

The reference builds a 0.25-degree global land mask with the `global_land_mask` package (a 1-km GLOBE lookup,
not installable here) and then cuts the seven basin masks and the global one out of it with pure geometry:
longitude / latitude boxes, staircase boundaries between the Atlantic and the eastern Pacific, `& ~land`.
This module restates that geometry line by line; the land source is pluggable:

  * `global_land_mask.globe.is_land` when that package is importable (the reference's own source), else
  * the land mask the reference ships for the intensity model, `intensity/data/land.nc` (0.125 degrees, read
    through `fields._Dataset` / `hdf5lite`), sampled at the nearest cell of every 0.25-degree grid point.  Coastlines
    then follow that file instead of GLOBE — a documented substitution of the *input*, not of the algorithm.

Outputs, as the reference writes them into `./land/`: `land.nc` (variable `land`) and `<B>.nc` for
B in NA, EP, WP, NI, SI, AU, SP, GL (variable `basin`), all on lat = linspace(-90, 90, 721), lon = 0 .. 359.75.
`util/compute.run_tracks` reads them back through `mat.interp2_fx` (compute.py:87-97), here `fields.load_year_env`.

One reference quirk is fixed and named: its "already generated?" test lists `'SI.nc' 'AU.nc'` without the comma
(generate_land_masks.py:15-16), so it looks for `SI.ncAU.nc` and regenerates on every run; here existing files are kept
unless `force=True`.
"""
import os

import numpy as np

BASIN_FILES = ('NA', 'EP', 'WP', 'NI', 'SI', 'AU', 'SP', 'GL')


def mask_grid():
    """generate_land_masks.py:24-26,30-32: 0.25 degrees; longitudes rotated from [-180, 180) to [0, 360)."""
    lat = np.linspace(-90, 90, 721)
    lon = np.linspace(-180, 180, 1441)[:-1]
    west = lon < -1e-5                                   # TC_Basin.transform_lon_r (util/basins.py:103-107)
    lon_gl = np.hstack((lon[~west], lon[west] + 360))
    return lon_gl, lat


def land_from_file(land_file, lon, lat):
    """is_land at the grid points from a [lat, lon] 0/1 land file on a regular global grid (nearest cell)."""
    from .fields import _Dataset
    ds = _Dataset(land_file)
    flon, flat = np.asarray(ds['lon'], dtype=np.float64), np.asarray(ds['lat'], dtype=np.float64)
    land = np.asarray(ds['land'])
    if flat[0] > flat[1]:
        flat, land = flat[::-1], land[::-1]
    flon = flon % 360.0
    order = np.argsort(flon)
    flon, land = flon[order], land[:, order]
    dlon, dlat = flon[1] - flon[0], flat[1] - flat[0]
    i = np.rint((np.asarray(lon) % 360.0 - flon[0]) / dlon).astype(int) % flon.size
    j = np.clip(np.rint((np.asarray(lat) - flat[0]) / dlat).astype(int), 0, flat.size - 1)
    return land[np.ix_(j, i)] > 0.5


def basin_masks(land_gl, lon_gl, lat_gl):
    """generate_land_masks.py:43-110 on a [lat, lon] boolean land mask over lon 0..360: dict id -> boolean mask."""
    LON, LAT = np.meshgrid(lon_gl, lat_gl)
    sea = ~np.asarray(land_gl, dtype=bool)
    out = {}
    # Atlantic: the box 255-360E, 0-60N cut along a staircase through Central America (:43-55)
    lat_box_na = [0, 9, 10, 14, 18]
    lon_box_na = [285, 278, 276, 271, 262]
    box = (LON >= 255) & (LON <= 360) & (LAT >= 0) & (LAT <= 60)
    stair = np.zeros(box.shape, bool)
    for la, lo in zip(lat_box_na, lon_box_na):
        stair |= (LAT >= la) & (LON >= lo) & sea
    out['NA'] = box & stair
    # eastern Pacific: everything to the west of the Atlantic staircase (:57-70)
    lat_box_ep = [7.5, 8.8, 9, 10, 15, 18, 60]
    lon_box_ep = [295, 282, 277, 276.5, 276, 271, 262]
    box = (LON >= 180) & (LON <= 290) & (LAT >= 0) & (LAT <= 60)
    stair = np.zeros(box.shape, bool)
    for la, lo in zip(lat_box_ep, lon_box_ep):
        stair |= (LAT <= la) & (LON <= lo) & sea
    out['EP'] = box & stair
    # plain boxes over the sea (:72-105)
    for b, (x0, x1, y0, y1) in dict(WP=(100, 180, 0, 60), NI=(30, 100, 0, 49), SI=(10, 100, -45, 0),
                                    AU=(100, 170, -45, 0), SP=(170, 260, -45, 0)).items():
        out[b] = (LON >= x0) & (LON <= x1) & (LAT >= y0) & (LAT <= y1) & sea
    # global: the sea equatorward of 50 degrees (:107-110)
    gl = sea.copy()
    gl[np.abs(LAT) > 50] = False
    out['GL'] = gl
    return out


def _write(fn, name, lon, lat, arr):
    try:
        import xarray as xr
        xr.Dataset({name: xr.DataArray(np.asarray(arr, bool), dims=['lat', 'lon'], coords=dict(lon=lon, lat=lat))}).to_netcdf(fn)
        return
    except ImportError:
        pass
    from scipy.io import netcdf_file
    with netcdf_file(fn, 'w', version=2) as f:
        f.createDimension('lat', len(lat)); f.createDimension('lon', len(lon))
        v = f.createVariable('lat', 'd', ('lat',)); v[:] = lat
        v = f.createVariable('lon', 'd', ('lon',)); v[:] = lon
        v = f.createVariable(name, 'b', ('lat', 'lon')); v[:] = np.asarray(arr, dtype=np.int8)
        v.dtype_hint = 'bool'


def generate_land_masks(out_dir='land', land_file=None, is_land=None, force=False, verbose=True):
    """Write land.nc and the eight basin files into `out_dir`; returns (lon, lat, land, masks).

    is_land(lat_grid, lon_grid) -> bool array (the reference's `globe.is_land` signature, longitudes in
    [-180, 180)); default: `global_land_mask` if installed, else `land_file` (default: the package's
    `intensity/data/land.nc`, i.e. where the reference keeps it)."""
    lon_gl, lat = mask_grid()
    fns = ['land.nc'] + ['%s.nc' % b for b in BASIN_FILES]
    if not force and all(os.path.exists(os.path.join(out_dir, f)) for f in fns):
        return None
    if verbose:
        print('Generating land masks...')
    os.makedirs(out_dir, exist_ok=True)
    if is_land is None:
        try:
            from global_land_mask import globe
            is_land = globe.is_land
        except ImportError:
            if land_file is None:
                from . import namelist
                land_file = os.path.join(namelist.src_directory, 'intensity', 'data', 'land.nc')
            if not os.path.exists(land_file):
                raise FileNotFoundError('no land source: install global_land_mask or provide %s (the reference ships it as '
                                        'intensity/data/land.nc)' % land_file)
            is_land = None
    if is_land is not None:
        lon_pm = np.linspace(-180, 180, 1441)[:-1]
        LONG, LATG = np.meshgrid(lon_pm, lat)
        land_pm = np.asarray(is_land(LATG, LONG), dtype=bool)
        west = lon_pm < -1e-5
        land_gl = np.concatenate((land_pm[:, ~west], land_pm[:, west]), axis=1)        # transform_lon_r
    else:
        land_gl = land_from_file(land_file, lon_gl, lat)
    masks = basin_masks(land_gl, lon_gl, lat)
    _write(os.path.join(out_dir, 'land.nc'), 'land', lon_gl, lat, land_gl)
    for b in BASIN_FILES:
        _write(os.path.join(out_dir, '%s.nc' % b), 'basin', lon_gl, lat, masks[b])
    return lon_gl, lat, land_gl, masks
