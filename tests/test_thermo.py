"""PI / chi / RH preprocessing: oracle vs the reference's golden vectors (CPU),
HIP kernels vs oracle and golden (GPU)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def cases():
    return np.load(os.path.join(GOLDEN, 'thermo_cases.npz'))


@pytest.fixture(scope='module')
def table():
    return np.load(os.path.join(GOLDEN, 'entropy_table.npz'))


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_oracle_matches_reference(cases, table, tag):
    from oracle import thermo_oracle as to
    tb = to.Table(table['p'], table['s'], table['T'])
    p, sst, psl, T, r = (cases[tag + '_' + k] for k in ('p', 'sst', 'psl', 'T', 'r'))
    k_mid = int(cases[tag + '_k_mid'])
    pi, chi, rh = to.column_fields(tb, float(cases['Ck_over_Cd']), p, sst, psl, T, r, k_mid)
    # same formulas, same libm: the scalar restatement reproduces the vectorised reference to rounding
    np.testing.assert_allclose(pi, cases[tag + '_PI'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(chi, cases[tag + '_chi'], rtol=1e-12, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(rh, cases[tag + '_rh_mid'], rtol=1e-13, atol=0, equal_nan=True)
    for idx in [(0, 0), (1, 1), (2, 2), (5, 5), (7, 3)]:
        aux = to.potential_intensity(tb, 1.0, float(sst[idx]), float(psl[idx]), p, T[(slice(None),) + idx], r[(slice(None),) + idx])[1]
        np.testing.assert_allclose(aux['p_lcl'], cases[tag + '_pLCL'][idx], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(aux['s_ns'], cases[tag + '_s_ns'][idx], rtol=1e-13, equal_nan=True)
        np.testing.assert_allclose(aux['ss'], cases[tag + '_ss'][idx], rtol=1e-13, equal_nan=True)


def _gpu_fields(eng, table, cases, tag):
    from tropical_cyclone_risk_amd import preprocess as pp
    pp.stage_entropy_table(eng, table['p'], table['s'], table['T'])
    p, sst, psl, T, r = (cases[tag + '_' + k] for k in ('p', 'sst', 'psl', 'T', 'r'))
    k_mid = int(cases[tag + '_k_mid'])
    pi = pp.potential_intensity(eng, sst, psl, p, T, r)
    chi, rh = pp.chi_rh(eng, sst, psl, T[k_mid], r[k_mid], float(p[k_mid]))
    return pi, chi, rh


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_kernels_match_reference_golden(cases, table, built_lib, tag):
    """k_potential_intensity / k_chi_rh against the reference's own outputs.  Tolerance: 1e-9 relative
    (device libm vs glibc in exp/log/pow, own Lambert W); zeros and NaN-handling cases must agree exactly."""
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine('GL', device=0)
    pi, chi, rh = _gpu_fields(eng, table, cases, tag)
    eng.close()
    ref = cases[tag + '_PI']
    assert np.array_equal(pi == 0, ref == 0)
    err = np.abs(pi - ref) / np.maximum(ref, 1.0)
    print('PI: max rel err %.2e, p99 %.2e' % (err.max(), np.percentile(err, 99)))
    assert err.max() < 1e-9
    np.testing.assert_allclose(chi, cases[tag + '_chi'], rtol=1e-9, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(rh, cases[tag + '_rh_mid'], rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.gpu
def test_compute_thermo_host_mirror(cases, table, built_lib):
    """calc_thermo.compute_thermo's array handling: level order, hPa, mid-level pick, chi clip."""
    from tropical_cyclone_risk_amd import preprocess as pp
    from tropical_cyclone_risk_amd.engine import TCEngine
    eng = TCEngine('GL', device=0)
    pi, chi, rh = _gpu_fields(eng, table, cases, 'b')
    p, sst, psl, T, r = (cases['b_' + k] for k in ('p', 'sst', 'psl', 'T', 'r'))
    v2, c2, r2 = pp.compute_thermo(eng, sst, psl, (p / 100)[::-1], 'hPa', T[::-1], r[::-1])     # top-down, hPa
    # (p / 100) * 100 is not p bit for bit, so agreement is to rounding, not exact
    np.testing.assert_allclose(v2, pi, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(r2, rh, rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(c2, np.minimum(np.maximum(chi, 0), 10), rtol=1e-9, atol=1e-12, equal_nan=True)
    assert np.nanmin(c2) >= 0 and np.nanmax(c2) <= 10
    with pytest.raises(Exception, match='lowest'):
        dp = lambda a: np.ascontiguousarray(a).ctypes.data_as(pp._lib.DP)
        out = np.empty(sst.shape)
        eng._ck(eng.L.tcr_potential_intensity_host(eng.h, sst.size, len(p), dp(p[::-1].copy()), dp(sst), dp(psl), dp(T), dp(r), 1.0, dp(out)))
    eng2 = TCEngine('GL', device=0)
    with pytest.raises(Exception, match='entropy table'):
        pp.potential_intensity(eng2, sst, psl, p, T, r)
    eng.close(); eng2.close()
