"""Analytic "planet" handed to the reference's mask generator in place of global_land_mask.globe.is_land
(tests/golden/make_golden_masks.py) and to masks.generate_land_masks in tests/test_masks.py: same signature,
`is_land(lat, lon)` with longitudes in [-180, 180).  Data-free so that both sides evaluate the same function."""
import numpy as np


def is_land(lat, lon):
    lat = np.asarray(lat, dtype=np.float64)
    lon = np.asarray(lon, dtype=np.float64)
    lam, phi = np.deg2rad(lon), np.deg2rad(lat)
    f = (0.6 * np.sin(3 * lam + 1.0) * np.cos(2 * phi) + 0.5 * np.cos(5 * lam) * np.sin(3 * phi + 0.5)
         + 0.4 * np.sin(7 * lam - 2 * phi))
    land = f > 0.35
    # an isthmus across the Atlantic / eastern-Pacific staircases (262-295E, 0-20N), so that `& ~land` matters there
    lon360 = np.where(lon < 0, lon + 360.0, lon)
    land |= (lon360 >= 258) & (lon360 <= 296) & (np.abs(lat - (8.0 + 0.45 * (280.0 - lon360))) < 1.6)
    # islands on box corners and on the |lat| = 50 cut of the global mask
    for lo, la, r in ((100.0, 0.0, 1.1), (180.0, 30.0, 0.9), (170.0, -45.0, 1.3), (30.0, 49.0, 0.8), (10.0, -20.0, 1.0),
                      (260.0, -10.0, 0.7), (330.0, 50.0, 1.2), (45.0, -50.0, 1.2)):
        land |= (lon360 - lo) ** 2 + (lat - la) ** 2 < r * r
    return land
