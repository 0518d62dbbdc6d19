"""Land / basin mask generation (scripts/generate_land_masks.py:11-110).

Pinned to the reference's own output: tests/golden/masks.npz holds the nine arrays the reference's
`generate_land_masks()` handed to `to_netcdf` when it was run, unmodified, in the build container on the analytic
planet of tests/golden/planted_land.py (tests/golden/make_golden_masks.py: xarray replaced by a recording container
stub, `globe.is_land` by the planted function) — `test_masks_equal_the_references_own_output` compares exactly.
The hand-derived known-answer tests below it stay as readable documentation of the geometry, plus a run on the land
mask the reference ships (tests/golden/ref_land.nc, a copy of its intensity/data/land.nc)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _at(mask, lon, lat, lo, la):
    i, j = np.argmin(np.abs(lon - lo)), np.argmin(np.abs(lat - la))
    assert lon[i] == lo and lat[j] == la, 'pick grid points'
    return bool(mask[j, i])


def test_masks_equal_the_references_own_output(tmp_path):
    """Exact equality, all 721 x 1440 points of land.nc and the eight basin files, with what the reference's own
    generate_land_masks() (scripts/generate_land_masks.py:23-110) wrote for the same `is_land`; also through the
    files this module writes and `fields._Dataset` reads back (the path compute.py:87-97 takes)."""
    from tests.golden import planted_land
    from tropical_cyclone_risk_amd import fields, masks
    g = np.load(os.path.join(GOLDEN, 'masks.npz'))
    want = {n: np.unpackbits(g['mask_' + n], axis=1)[:, :1440].astype(bool) for n in ('land',) + masks.BASIN_FILES}
    assert 0.2 < want['land'].mean() < 0.35 and all(want[b].sum() > 30000 for b in masks.BASIN_FILES)
    lon, lat, land, m = masks.generate_land_masks(str(tmp_path / 'land'), is_land=planted_land.is_land, verbose=False)
    assert np.array_equal(land, want['land'])
    for b in masks.BASIN_FILES:
        assert np.array_equal(m[b], want[b]), b
        assert np.array_equal(lon, g['lon_' + b]) and np.array_equal(lat, g['lat_' + b]), b      # bit for bit
        ds = fields._Dataset(str(tmp_path / 'land' / ('%s.nc' % b)))
        assert np.array_equal(np.asarray(ds['basin']) > 0.5, want[b]) and np.array_equal(ds['lon'], g['lon_' + b])
    # the planted planet exercises what the hand tests cannot: land inside both staircases, on box corners, on the 50-degree cut
    LON, LAT = np.meshgrid(lon, lat)
    stairs = (LON >= 258) & (LON <= 296) & (LAT >= 0) & (LAT <= 20)
    assert (want['land'] & stairs).sum() > 500 and (want['NA'] & stairs).sum() > 500 and (want['EP'] & stairs).sum() > 500
    sea = masks.basin_masks(np.zeros_like(land), lon, lat)
    assert 0 < (want['NA'] & want['EP']).sum() < (sea['NA'] & sea['EP']).sum()       # the isthmus removes part of the overlap of the two staircases
    # one reference quirk, recorded: it attaches the UNROTATED longitudes (-180 .. 179.75) to the rotated land array of
    # land/land.nc (generate_land_masks.py:34-35, `coords = dict(lon=lon, ...)`); nothing reads that file back
    # (compute.py:87-97 opens only the basin files), and this module writes the rotated axis
    assert g['lon_land'][0] == -180.0 and lon[0] == 0.0


def test_grid_is_the_references():
    from tropical_cyclone_risk_amd import masks
    lon, lat = masks.mask_grid()
    assert lon.shape == (1440,) and lat.shape == (721,)
    assert lon[0] == 0.0 and lon[-1] == 359.75 and lon[720] == 180.0 and np.all(np.diff(lon) == 0.25)     # -180 -> 180 sits after 179.75
    assert lat[0] == -90.0 and lat[-1] == 90.0 and lat[360] == 0.0


def test_box_corners_and_staircases_all_sea():
    from tropical_cyclone_risk_amd import masks
    lon, lat = masks.mask_grid()
    m = masks.basin_masks(np.zeros((721, 1440), bool), lon, lat)
    at = lambda b, lo, la: _at(m[b], lon, lat, lo, la)
    # Atlantic staircase (:44-51): lat >= 0 needs lon >= 285, >= 9 -> 278, >= 10 -> 276, >= 14 -> 271, >= 18 -> 262
    for la, lo in ((0, 285), (9, 278), (10, 276), (14, 271), (18, 262), (60, 262), (30, 359.75)):
        assert at('NA', lo, la), (la, lo)
    for la, lo in ((0, 284.75), (8.75, 278), (9, 277.75), (9.75, 276), (13.75, 271), (17.75, 262), (18, 261.75),
                   (60.25, 300), (-0.25, 300), (30, 254.75)):
        assert not at('NA', lo, la), (la, lo)
    # eastern Pacific staircase (:58-66): lat <= 7.5 up to 290E (the box ends there), <= 8.75 -> 282, <= 9 -> 277,
    # <= 10 -> 276.5, <= 15 -> 276, <= 18 -> 271, <= 60 -> 262
    for la, lo in ((7.5, 290), (8.75, 282), (9, 277), (10, 276.5), (15, 276), (18, 271), (60, 262), (0, 180), (60, 180)):
        assert at('EP', lo, la), (la, lo)
    for la, lo in ((7.75, 290), (7.5, 290.25), (9, 282), (9.25, 277), (10.25, 276.5), (15.25, 276), (18.25, 271),
                   (60, 262.25), (60.25, 200), (-0.25, 200), (30, 179.75)):
        assert not at('EP', lo, la), (la, lo)
    # on an all-sea planet the two staircases overlap where South / Central America would be (e.g. 5N 287E is inside
    # both `lat >= 0 & lon >= 285` and `lat <= 7.5 & lon <= 295`); the reference relies on the land there
    assert at('NA', 287, 5) and at('EP', 287, 5) and not at('EP', 287, 8) and at('NA', 287, 8)
    # plain boxes (:72-105), corners inclusive
    boxes = dict(WP=(100, 180, 0, 60), NI=(30, 100, 0, 49), SI=(10, 100, -45, 0), AU=(100, 170, -45, 0), SP=(170, 260, -45, 0))
    for b, (x0, x1, y0, y1) in boxes.items():
        for lo, la in ((x0, y0), (x1, y0), (x0, y1), (x1, y1)):
            assert at(b, lo, la), (b, lo, la)
        for lo, la in ((x0 - 0.25, y0), (x1 + 0.25, y1), (x0, y0 - 0.25), (x1, y1 + 0.25)):
            assert not at(b, lo, la), (b, lo, la)
        assert m[b].sum() == (round((x1 - x0) / 0.25) + 1) * (round((y1 - y0) / 0.25) + 1)
    # global (:107-110): everything equatorward of 50 degrees, inclusive
    assert at('GL', 0, 50) and at('GL', 200, -50) and not at('GL', 0, 50.25) and not at('GL', 0, -50.25)
    assert m['GL'].sum() == 1440 * 401


def test_land_is_cut_out_of_every_mask():
    from tropical_cyclone_risk_amd import masks
    lon, lat = masks.mask_grid()
    land = np.zeros((721, 1440), bool)
    LON, LAT = np.meshgrid(lon, lat)
    land[(np.abs(LAT) <= 40) & (((LON >= 120) & (LON <= 130)) | ((LON >= 300) & (LON <= 310)) | ((LON >= 60) & (LON <= 70)) |
                                ((LON >= 200) & (LON <= 210)))] = True
    m = masks.basin_masks(land, lon, lat)
    sea = masks.basin_masks(np.zeros_like(land), lon, lat)
    for b in masks.BASIN_FILES:
        assert not (m[b] & land).any(), b
        assert np.array_equal(m[b], sea[b] & ~land), b


def test_masks_from_the_shipped_land_file(tmp_path):
    from tropical_cyclone_risk_amd import fields, masks
    out = masks.generate_land_masks(str(tmp_path / 'land'), land_file=os.path.join(GOLDEN, 'ref_land.nc'), verbose=False)
    lon, lat, land, m = out
    assert 0.25 < land.mean() < 0.45
    at = lambda b, lo, la: _at(m[b], lon, lat, lo, la)
    assert at('NA', 270, 25) and not at('EP', 270, 25)                 # Gulf of Mexico
    assert at('EP', 250, 15) and not at('NA', 250, 15)                 # off Mexico's Pacific coast
    assert at('WP', 135, 20) and at('NI', 88, 15) and at('SI', 75, -15) and at('AU', 115, -15) and at('SP', 190, -15)
    assert not at('GL', 20, 10) and not at('GL', 300, -10) and at('GL', 330, 30)      # Africa, Amazonia, mid-Atlantic
    assert not m['GL'][np.abs(lat) > 50].any()
    for b in masks.BASIN_FILES:
        assert not (m[b] & land).any()
    # the files hold exactly these masks in the schema compute.py:87-97 reads, and a second call keeps them
    for b in masks.BASIN_FILES:
        ds = fields._Dataset(str(tmp_path / 'land' / ('%s.nc' % b)))
        assert np.array_equal(np.asarray(ds['basin']) > 0.5, m[b]) and np.array_equal(ds['lon'], lon) and np.array_equal(ds['lat'], lat)
    assert np.array_equal(np.asarray(fields._Dataset(str(tmp_path / 'land' / 'land.nc'))['land']) > 0.5, land)
    assert masks.generate_land_masks(str(tmp_path / 'land'), land_file=os.path.join(GOLDEN, 'ref_land.nc'), verbose=False) is None
    # a caller-supplied is_land with the reference's signature (globe.is_land(lat, lon), longitudes in [-180, 180))
    out2 = masks.generate_land_masks(str(tmp_path / 'land2'), is_land=lambda la, lo: (lo > -10) & (lo < 0) & (np.abs(la) < 30), verbose=False)
    assert out2[2][360, 1420] and not out2[2][360, 20] and not out2[3]['NA'][400, 1420]
