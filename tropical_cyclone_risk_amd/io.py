"""Track-file writer and environment loader (host plumbing around the hot path).

Output schema = the reference's (`util/compute.py:250-264`, README "Model Output"):
``lon_trks, lat_trks, u250_trks, v250_trks, u850_trks, v850_trks, v_trks, m_trks,
vmax_trks [n_trk, time]``, ``tc_month, tc_basins, tc_years [n_trk]``,
``seeds_per_month [year, basin, month]`` with coords ``n_trk, time, year, basin, month``.
NetCDF-4 through xarray when it is installed (as the reference writes it); otherwise
NetCDF-3 classic through ``scipy.io.netcdf_file`` — same variables, dimensions and
coordinates, readable by ``xarray.open_dataset`` (neither the build nor the bench container
has xarray / netCDF4 / h5py; SciPy is part of the image).
"""
import os

import numpy as np

from .basins import BASIN_IDS


def _try_xarray():
    try:
        import xarray as xr
        return xr
    except Exception:
        return None


def get_fn_tracks(b, nl, out_dir=None, ext='nc'):
    """`compute.py:40-46`."""
    base = out_dir or '%s/%s' % (nl.output_directory, nl.exp_name)
    return '%s/tracks_%s_%s_%d%02d_%d%02d.%s' % (base, b.basin_id, nl.exp_prefix, nl.start_year,
                                                 nl.start_month, nl.end_year, nl.end_month, ext)


def fn_tracks_duplicates(fn_trk):
    """`compute.py:52-58`: never overwrite — append _e<k>."""
    stem, ext = os.path.splitext(fn_trk)
    out, k = fn_trk, 0
    while os.path.exists(out):
        out = '%s_e%d%s' % (stem, k, ext)
        k += 1
    return out


def assemble(out, years, nl):
    """Concatenate the per-year 9-tuples into the output variables (`compute.py:233-248`)."""
    cat = lambda i: np.concatenate([x[i] for x in out], axis=0)
    env = cat(5)
    total_time_s = nl.total_track_time_days * 24 * 60 * 60
    n_steps_output = int(total_time_s / nl.output_interval_s) + 1
    data = dict(lon_trks=cat(0), lat_trks=cat(1), u250_trks=env[:, :, 0], v250_trks=env[:, :, 1],
                u850_trks=env[:, :, 2], v850_trks=env[:, :, 3], v_trks=cat(2), m_trks=cat(3),
                vmax_trks=cat(4), tc_month=cat(6), tc_basins=cat(7),
                tc_years=np.concatenate([[yr] * out[i][0].shape[0] for i, yr in enumerate(years)]),
                seeds_per_month=np.array([x[8] for x in out]))
    coords = dict(n_trk=np.arange(data['lon_trks'].shape[0]), time=np.linspace(0, total_time_s, n_steps_output),
                  year=np.array(years), basin=np.array(BASIN_IDS), month=np.arange(1, 13))
    return data, coords


TWO_D = ['lon_trks', 'lat_trks', 'u250_trks', 'v250_trks', 'u850_trks', 'v850_trks', 'v_trks', 'm_trks', 'vmax_trks']


def _write_netcdf3(fn, data, coords, skip=()):
    """The reference's Dataset (compute.py:250-264) in NetCDF-3 classic.  Strings (tc_basins,
    basin) become fixed-width char arrays with a trailing `string2` dimension, which is how
    xarray itself encodes them in this format.

    `fn` may be an open binary file.  Variables named in `skip` get their header entry and their place in the file but
    no data (the file position just moves on): `TrackFileWriter` fills them row block by row block.  Returns
    {name: byte offset of the variable's data} for the skipped ones."""
    from scipy.io import netcdf_file

    class _File(netcdf_file):
        def _write_var_data(self, name):
            if name not in skip:
                return netcdf_file._write_var_data(self, name)
            var = self.variables[name]
            begin = self.fp.tell()                      # what netcdf_file._write_var_data does, minus the data
            self.fp.seek(var._begin)
            self._pack_begin(begin)
            offsets[name] = begin
            self.fp.seek(begin + var._vsize)
    offsets = {}
    with _File(fn, 'w', version=2) as f:
        for dim in ('n_trk', 'time', 'year', 'basin', 'month'):
            f.createDimension(dim, len(coords[dim]))
        f.createDimension('string2', 2)

        def put(name, arr, dims, dtype=None):
            if name in skip:                            # (the variable's array is allocated but never touched)
                return f.createVariable(name, dtype or 'd', dims)
            arr = np.asarray(arr)
            v = f.createVariable(name, dtype or arr.dtype.char, dims)
            v[:] = arr
            return v
        put('n_trk', coords['n_trk'].astype(np.int32), ('n_trk',))
        put('time', coords['time'].astype(np.float64), ('time',))
        put('year', coords['year'].astype(np.int32), ('year',))
        put('month', coords['month'].astype(np.int32), ('month',))
        chars = lambda a: np.array([list(str(x).ljust(2)[:2]) for x in a], dtype='S1').reshape(len(a), 2)
        put('basin', chars(coords['basin']), ('basin', 'string2'), 'c')
        for k in TWO_D:
            put(k, None if k in skip else data[k].astype(np.float64), ('n_trk', 'time'))
        put('tc_month', data['tc_month'].astype(np.float64), ('n_trk',))
        put('tc_years', data['tc_years'].astype(np.int32), ('n_trk',))
        put('tc_basins', chars(data['tc_basins']), ('n_trk', 'string2'), 'c')
        put('seeds_per_month', data['seeds_per_month'].astype(np.float64), ('year', 'basin', 'month'))
    return offsets


_STREAMING_OK = None


def _streaming_supported():
    """`_write_netcdf3(skip=...)` leans on scipy internals (netcdf_file._write_var_data / _pack_begin, netcdf_variable._begin /
    _vsize: stable since scipy 0.9, but private — ADVICE r4).  Checked ONCE on a scratch file: a two-track layout written with
    its row variables skipped and then filled by hand must be byte for byte the file written whole; if scipy has changed
    underneath, the streaming writer is not used and `TrackFileWriter` falls back to `write_tracks` at close."""
    global _STREAMING_OK
    if _STREAMING_OK is None:
        import io as _io
        try:
            n, ns = 2, 3
            coords = dict(n_trk=np.arange(n), time=np.linspace(0, 2.0, ns), year=np.array([2000]), basin=np.array(BASIN_IDS), month=np.arange(1, 13))
            data = {k: np.arange(n * ns, dtype=np.float64).reshape(n, ns) + j for j, k in enumerate(TWO_D)}
            data.update(tc_month=np.ones(n), tc_years=np.full(n, 2000, np.int32), tc_basins=np.array(['NA'] * n),
                        seeds_per_month=np.ones((1, len(BASIN_IDS), 12)))
            whole, holes = _io.BytesIO(), _io.BytesIO()
            _write_netcdf3(_NoClose(whole), data, coords)
            off = _write_netcdf3(_NoClose(holes), data, coords, skip=frozenset(TWO_D))
            for k in TWO_D:
                holes.seek(off[k])
                holes.write(np.asarray(data[k], dtype='>f8').tobytes())
            _STREAMING_OK = sorted(off) == sorted(TWO_D) and holes.getvalue() == whole.getvalue()
        except Exception:
            _STREAMING_OK = False
    return _STREAMING_OK


class TrackFileWriter:
    """The track file of `write_tracks`, written while the years are still being computed.

    `run_downscaling` hands over every year's 9-tuple as it completes (`put`, any order, any thread); a background thread
    converts the rows to the file's big-endian doubles and writes them at their final place, so that when the last year
    arrives only the small per-track variables and the header remain (`close`).  Every year yields exactly
    `tracks_per_year` tracks (`run_tracks` returns when the quota is full), which fixes the layout in advance.  The file is
    byte for byte the one `write_tracks` produces (tests/test_host_units.py).  With xarray installed the file is NetCDF-4
    like the reference's and is written at `close` through `write_tracks`."""

    def __init__(self, years, b, nl, out_dir=None):
        import queue
        import threading
        self.years, self.b, self.nl, self.out_dir = list(years), b, nl, out_dir
        self.per_year = int(nl.tracks_per_year)
        self.out = [None] * len(self.years)
        self.streaming = _try_xarray() is None and self.per_year > 0 and _streaming_supported()
        self.fn, self._err = None, None
        if not self.streaming:
            return
        self.fn = fn_tracks_duplicates(get_fn_tracks(b, nl, out_dir, 'nc'))
        os.makedirs(os.path.dirname(self.fn), exist_ok=True)
        total_time_s = nl.total_track_time_days * 24 * 60 * 60
        self.ns = int(total_time_s / nl.output_interval_s) + 1
        n = self.per_year * len(self.years)
        self.coords = dict(n_trk=np.arange(n), time=np.linspace(0, total_time_s, self.ns), year=np.array(self.years),
                           basin=np.array(BASIN_IDS), month=np.arange(1, 13))
        self.f = open(self.fn, 'w+b')
        # the header (and with it every variable's place) does not depend on the values: a first pass with placeholders
        blank = dict(tc_month=np.zeros(n), tc_years=np.zeros(n, np.int32), tc_basins=np.array(['  '] * n),
                     seeds_per_month=np.zeros((len(self.years), len(BASIN_IDS), 12)))
        self.offsets = self._header(blank)
        self.q = queue.Queue()
        self.thread = threading.Thread(target=self._run, name='track-file-writer', daemon=True)
        self.thread.start()

    def _header(self, small):
        self.f.seek(0)
        off = _write_netcdf3(_NoClose(self.f), small, self.coords, skip=frozenset(TWO_D))
        return off

    def _run(self):
        try:
            while True:
                item = self.q.get()
                if item is None:
                    return
                i, t9 = item
                env = t9[5]
                planes = dict(lon_trks=t9[0], lat_trks=t9[1], u250_trks=env[:, :, 0], v250_trks=env[:, :, 1], u850_trks=env[:, :, 2],
                              v850_trks=env[:, :, 3], v_trks=t9[2], m_trks=t9[3], vmax_trks=t9[4])
                for k in TWO_D:
                    rows = np.asarray(planes[k], dtype='>f8')
                    if rows.shape != (self.per_year, self.ns):
                        raise ValueError('year %d: %s has shape %s, expected %s' % (self.years[i], k, rows.shape, (self.per_year, self.ns)))
                    self.f.seek(self.offsets[k] + i * self.per_year * self.ns * 8)
                    self.f.write(rows.tobytes())
        except Exception as e:                      # surfaces in close()
            self._err = e

    def put(self, year_index, tuple9):
        self.out[year_index] = tuple9
        if self.streaming:
            self.q.put((year_index, tuple9))

    def abort(self):
        """Stop the writer thread and remove the partial file (a year failed)."""
        if not self.streaming:
            return
        self.q.put(None)
        self.thread.join()
        self.f.close()
        try:
            os.remove(self.fn)
        except OSError:
            pass

    def close(self):
        """Finish the file; returns its name."""
        if not self.streaming:
            return write_tracks(self.out, self.years, self.b, self.nl, self.out_dir)
        self.q.put(None)
        self.thread.join()
        if self._err is not None:
            self.f.close()
            raise self._err
        cat = lambda j: np.concatenate([x[j] for x in self.out], axis=0)
        small = dict(tc_month=cat(6), tc_basins=cat(7),
                     tc_years=np.concatenate([[yr] * self.out[i][0].shape[0] for i, yr in enumerate(self.years)]),
                     seeds_per_month=np.array([x[8] for x in self.out]))
        off = self._header(small)
        assert off == self.offsets
        self.f.close()
        return self.fn


class _NoClose:
    """File proxy whose close() only flushes: scipy's netcdf_file closes the file object it is given (and looks at
    `.closed` again when it is garbage-collected, to decide whether to write once more)."""

    def __init__(self, f):
        self._f = f
        self.closed = False

    def __getattr__(self, k):
        return getattr(self._f, k)

    def close(self):
        self._f.flush()
        self.closed = True


def read_tracks(fn):
    """Read a track file written by write_tracks (either flavour) into plain NumPy arrays."""
    xr = _try_xarray()
    if xr is not None:
        ds = xr.open_dataset(fn)
        return {k: np.asarray(ds[k]) for k in list(ds.data_vars) + list(ds.coords)}
    from scipy.io import netcdf_file
    out = {}
    with netcdf_file(fn, 'r', mmap=False) as f:
        for k, v in f.variables.items():
            a = np.array(v[:])
            if a.dtype.kind == 'S' and a.ndim == 2:
                a = np.array([b''.join(r).decode().strip() for r in a])
            out[k] = a
    return out


def write_tracks(out, years, b, nl, out_dir=None):
    data, coords = assemble(out, years, nl)
    xr = _try_xarray()
    fn = fn_tracks_duplicates(get_fn_tracks(b, nl, out_dir, 'nc'))
    os.makedirs(os.path.dirname(fn), exist_ok=True)
    if xr is not None:
        dv = {k: (['n_trk', 'time'], data[k]) for k in TWO_D}
        dv.update({k: (['n_trk'], data[k]) for k in ('tc_month', 'tc_basins', 'tc_years')})
        dv['seeds_per_month'] = (['year', 'basin', 'month'], data['seeds_per_month'])
        xr.Dataset(data_vars=dv, coords={k: list(v) if k in ('basin',) else v for k, v in coords.items()}).to_netcdf(fn, mode='w')
    else:
        _write_netcdf3(fn, data, coords)
    return fn


def load_env(nl, year=None, files=None):
    """Environment for run_downscaling.  ``dataset_type = 'SYNTHETIC'`` selects the regenerated
    ERA5-shaped fields; anything else reads the reference's preprocessed files for ``year``
    (`fields.load_year_env`: thermo_*.nc, env_wnd_*.nc, climatologies, land/bathymetry, basin masks)."""
    from . import synthetic
    if getattr(nl, 'dataset_type', '').upper() == 'SYNTHETIC':
        return synthetic.make_env('era5', seed=getattr(nl, 'gpu_experiment_seed', 20250614))
    from . import fields
    return fields.FileEnvironment(nl, files) if year is None else fields.load_year_env(year, nl, files)
