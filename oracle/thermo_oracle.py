"""TEST INFRASTRUCTURE — CPU restatement of the reference's potential-intensity / chi / RH preprocessing
(SURVEY §8 f-3), one column at a time with scalar arithmetic.

Follows thermo/thermo.py: sat_thermo (:29-38), conv_q_to_rh (:41-46), s_unsat (:49-61), s_sat (:64-75),
sat_deficit (:92-104), get_LCL (:107-126, Romps 2017 through scipy.special.lambertw branch -1),
calc_T_rho (:129-134) and CAPE_PI_vectorized (:266-412) for select_thermo = 1, select_interp = 2
(namelist.py:59-60), plus the chi clip and mid-level pick of calc_thermo.compute_thermo (:55-74).

Pinned: tests/test_thermo.py checks it against tests/golden/thermo_cases.npz, which
tests/golden/make_golden_thermo.py produced by running the reference's own functions.
Only tests/ may import this module.
"""
import math

import numpy as np
from scipy.special import lambertw

T_trip, e_trip = 273.16, 611.65
Rd, Rv = 287.04, 461.5
cv = 718
cp = cv + Rd
eps = Rd / Rv
L0 = 2.555e6
NAN = float('nan')


def sat_thermo(T, p):
    """(:29-38) Bolton; a NaN temperature gives es = 0."""
    if T != T:
        es = 0.0
    else:
        tc = T - 273
        with np.errstate(all='ignore'):
            es = 610.94 * math.exp(min(float(np.float64(17.625 * tc) / np.float64(tc + 243.04)), 10))
    with np.errstate(all='ignore'):
        rs = float(np.float64(Rd / Rv * es) / np.float64(p - es))
    return es, rs


def _log(x):
    with np.errstate(all='ignore'):
        return float(np.log(np.float64(x)))


def s_unsat(T, p, r):
    es, rs = sat_thermo(T, p)
    with np.errstate(all='ignore'):
        rh = float(np.maximum(np.float64(r) / rs * (1 + rs / eps) / (1 + r / eps), 0))
        return float(cp * _log(T) - Rd * _log(p - es * rh) + np.float64(L0 * r) / T - r * Rv * _log(rh))


def s_sat(T, p):
    es, rs = sat_thermo(T, p)
    T = T if T != T else max(T, 1e-4)
    with np.errstate(all='ignore'):
        arg = p - es
        arg = arg if arg != arg else max(arg, 1e-4)
        return float(cp * _log(T) - Rd * _log(arg) + np.float64(L0 * rs) / T)


def get_lcl(p, T, r, rh):
    E0v, cvv, cvl = 2.3740e6, 1418, 4119
    cpv = cvv + Rv
    with np.errstate(all='ignore'):
        q = np.float64(r) / (1 + r)
        Rm = (1 - q) * Rd + q * Rv
        cpm = (1 - q) * cp + q * cpv
        a = cpm / Rm + (cvl - cpv) / Rv
        b = -(E0v - (cvv - cvl) * T_trip) / (Rv * np.float64(T))
        c = b / a
        w = lambertw(np.float64(rh) ** (1 / a) * c * np.exp(c), -1).real
        T_lcl = c * T / w
        return float(p * (T_lcl / T) ** (cpm / Rm))


def t_rho(T, rv):
    with np.errstate(all='ignore'):
        return float(np.float64(T) * (1 + rv / eps) / (1 + np.float64(rv)))


class Table:
    """RectBivariateSpline(p_look, s_look, T, kx=1, ky=1).ev: clamp, interval search, bilinear (fpbisp)."""

    def __init__(self, p, s, T):
        self.p, self.s, self.T = np.asarray(p, float), np.asarray(s, float), np.asarray(T, float)

    @staticmethod
    def _cell(x, arg):
        if arg != arg:
            return None
        arg = min(max(arg, x[0]), x[-1])
        i = int(np.searchsorted(x, arg, side='right')) - 1
        i = min(max(i, 0), len(x) - 2)
        f = 1.0 / (x[i + 1] - x[i])
        return i, f * (x[i + 1] - arg), f * (arg - x[i])

    def ev(self, p, s):
        cp_, cs_ = self._cell(self.p, p), self._cell(self.s, s)
        if cp_ is None or cs_ is None:
            return NAN
        i, a0, a1 = cp_
        j, b0, b1 = cs_
        T = self.T
        sp = 0.0
        sp = sp + T[i, j] * a0 * b0
        sp = sp + T[i, j + 1] * a0 * b1
        sp = sp + T[i + 1, j] * a1 * b0
        sp = sp + T[i + 1, j + 1] * a1 * b1
        return float(sp)


def potential_intensity(table, cecd, sst, p_surf, p_env, T_env, r_env):
    """CAPE_PI_vectorized (:266-412) for one column; p_env lowest level first."""
    L = len(p_env)
    T_ns, r_ns, p_ns = float(T_env[0]), float(r_env[0]), float(p_env[0])
    ess, rs = sat_thermo(sst, p_surf)
    with np.errstate(all='ignore'):
        rh = float(np.float64(r_ns) / rs * (1 + rs / eps) / (1 + r_ns / eps))
    s_ns = s_unsat(T_ns, p_ns, r_ns)
    ss = s_sat(sst, p_surf)
    lnp = [math.log(p) for p in p_env]
    dlnp = [lnp[k + 1] - lnp[k] for k in range(L - 1)] + [(2 * lnp[-1] - lnp[-2]) - lnp[-1]]
    p_lcl = get_lcl(p_ns, T_ns, r_ns, rh)
    i_cond = L - 1
    for k in range(L):
        if p_lcl > p_env[k]:
            i_cond = k
            break
    tre, tra, trs = [], [], []
    for k in range(L):
        tre.append(t_rho(float(T_env[k]), float(r_env[k])))
        if k < i_cond:
            with np.errstate(all='ignore'):
                Ta = float(T_ns * np.power(np.float64(p_env[k]) / p_ns, Rd / cp))
            ra = r_ns
        else:
            Ta = table.ev(p_env[k], s_ns)
            ra = sat_thermo(Ta, p_env[k])[1]
        tra.append(t_rho(Ta, ra))
        Ts = table.ev(p_env[k], ss)
        trs.append(t_rho(Ts, sat_thermo(Ts, p_env[k])[1]))

    def last_ge(tr):
        idx = [k for k in range(L) if tr[k] >= tre[k]]
        return idx[-1] if idx else L - 1              # argmax of an all-False column is 0 -> L-1 (:353)

    def outflow(tr, k_out, T_out0):
        if k_out >= L - 1:
            return T_out0, 0.0
        k = k_out
        with np.errstate(all='ignore'):
            dT1, dT2 = np.float64(tr[k] - tre[k]), np.float64(tr[k + 1] - tre[k + 1])
            p_out = (p_env[k] * dT2 - p_env[k + 1] * dT1) / (dT2 - dT1)
            T_out = (T_env[k] * (p_out - p_env[k + 1]) + T_env[k + 1] * (p_env[k] - p_out)) / (p_env[k] - p_env[k + 1])
            add = Rd * dT1 * (p_env[k] - p_out) / (p_env[k] + p_out)
        return float(T_out), float(add)

    a_out, s_out = last_ge(tra), last_ge(trs)
    _, add_a = outflow(tra, a_out, 0.0)
    T_out_s, add_s = outflow(trs, s_out, NAN)
    cape = capes = 0.0
    for k in range(L):
        if k <= a_out:
            cape += Rd * (tra[k] - tre[k]) * -dlnp[k]
        if k <= s_out:
            capes += Rd * (trs[k] - tre[k]) * -dlnp[k]
    cape += add_a
    capes += add_s
    cape = cape if cape != cape else max(cape, 0.0)
    cape = 0.0 if cape != cape else cape
    with np.errstate(all='ignore'):
        x = cecd * float(np.float64(sst) / np.float64(T_out_s)) * (capes - cape)
        x = x if x != x else max(x, 0.0)
        pi = math.sqrt(x) if x == x else NAN
    return (0.0 if pi != pi else pi), dict(p_lcl=p_lcl, s_ns=s_ns, ss=ss, i_cond=i_cond, a_out=a_out, s_out=s_out)


def sat_deficit(sst, ps, T, pm, rv):
    sp, sps, spss = s_unsat(T, pm, rv), s_sat(T, pm), s_sat(sst, ps)
    with np.errstate(all='ignore'):
        return float(np.float64(sps - sp) / np.float64(spss - sps))


def conv_q_to_rh(T, q, p):
    es, rs = sat_thermo(T, p)
    with np.errstate(all='ignore'):
        qs = np.float64(rs) / (1 + rs)
        return float(np.minimum(np.maximum(np.float64(q) / qs, 1e-5), 1))


def column_fields(table, cecd, p_env, sst, psl, T, r, k_mid):
    """PI, chi (unclipped), rh_mid for [L, ...] soundings; loops over the trailing axes."""
    shape = sst.shape
    pi, chi, rh = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for idx in np.ndindex(shape):
        col = (slice(None),) + idx
        pi[idx] = potential_intensity(table, cecd, float(sst[idx]), float(psl[idx]), p_env, T[col], r[col])[0]
        chi[idx] = sat_deficit(float(sst[idx]), float(psl[idx]), float(T[(k_mid,) + idx]), float(p_env[k_mid]), float(r[(k_mid,) + idx]))
        rh[idx] = conv_q_to_rh(float(T[(k_mid,) + idx]), float(r[(k_mid,) + idx]), float(p_env[k_mid]))
    return pi, chi, rh
