// Potential intensity, saturation deficit and mid-level relative humidity: the
// thermodynamic preprocessing whose outputs (vmax, chi, rh_mid) the hot path consumes.
//
// Reference: thermo/thermo.py — sat_thermo (:29-38), conv_q_to_rh (:41-46), s_unsat (:49-61), s_sat
// (:64-75), sat_deficit (:92-104), get_LCL (:107-126), calc_T_rho (:129-134), CAPE_PI_vectorized
// (:266-412) for select_thermo = 1 (pseudoadiabatic), select_interp = 2 (temperature from the
// entropy table), the configuration of namelist.py:59-60.
//
// One thread per column.  The reference makes several masked sweeps over [level, lat, lon] arrays; per
// column all of it folds into ONE pass over the levels with a few carried scalars:
//   * the parcel is dry-adiabatic below the first level above the LCL and follows the moist adiabat
//     (bilinear table lookup at its entropy) from there on;
//   * "last level where the parcel is at least as dense-warm as the environment" (the flipped argmax,
//     :353-354) is the last hit of a running test, and the CAPE sum up to it (:390-395) is the running
//     sum sampled at every hit — the same additions in the same order;
//   * the outflow interpolation (:363-388) between that level and the next is evaluated for every hit
//     when the next level arrives and the last one is kept.
// Soundings are [level][point] planes, so a wave reads 512 contiguous bytes per level and variable;
// the 200 x 200 table (320 kB) stays L2-resident.  Bound: HBM reads of the soundings,
// 2 x 8 B per level and point, against ~150 flops + two table lookups + three exp per level.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tcr {
namespace th {

constexpr double T_trip = 273.16, Rd = 287.04, Rv = 461.5, cvd = 718.0, cp = cvd + Rd, eps = Rd / Rv, L0 = 2.555e6;

struct Table {
    int np, ns;
    const double *p, *s, *T;     // knots (ascending) and T[np][ns]
};

// thermo.py:29-38 (Bolton); a NaN temperature has es = 0 there (masked assignment into zeros)
__device__ __forceinline__ void sat_thermo(double T, double p, double &es, double &rs)
{
    const double tc = T - 273;
    const double q = (17.625 * tc) / (tc + 243.04);
    es = (T != T) ? 0.0 : 610.94 * exp((q < 10) ? q : ((q != q) ? q : 10.0));      // np.minimum propagates NaN
    rs = (Rd / Rv * es) / (p - es);
}

__device__ __forceinline__ double np_maximum(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
__device__ __forceinline__ double np_minimum(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }

// s_unsat (:49-61), pseudoadiabatic
__device__ __forceinline__ double s_unsat(double T, double p, double r)
{
    double es, rs;
    sat_thermo(T, p, es, rs);
    const double rh = np_maximum(r / rs * (1 + rs / eps) / (1 + r / eps), 0.0);
    return cp * log(T) - Rd * log(p - es * rh) + L0 * r / T - r * Rv * log(rh);
}

// s_sat (:64-75)
__device__ __forceinline__ double s_sat(double T, double p)
{
    double es, rs;
    sat_thermo(T, p, es, rs);
    T = np_maximum(T, 1e-4);
    return cp * log(T) - Rd * log(np_maximum(p - es, 1e-4)) + L0 * rs / T;
}

// Real branch -1 of the Lambert W function on [-1/e, 0): what scipy.special.lambertw(z, -1).real
// returns there.  Halley iterations from the branch-point series / the asymptotic form.
__device__ __forceinline__ double lambertw_m1(double z)
{
    if (z != z) return z;
    if (z == 0.0) return -INFINITY;
    const double em1 = 0.36787944117144233;
    if (z > 0.0 || z < -em1) return __longlong_as_double(0x7ff8000000000000LL);   // complex there (reference keeps the real part)
    double w;
    if (z < -0.25) {
        const double pp = -sqrt(2.0 * (2.718281828459045 * z + 1.0));
        w = -1.0 + pp - pp * pp / 3.0 + 11.0 / 72.0 * pp * pp * pp;
    } else {
        const double l1 = log(-z), l2 = log(-l1);
        w = l1 - l2 + l2 / l1;
    }
    for (int it = 0; it < 40; ++it) {
        const double ew = exp(w), f = w * ew - z;
        const double dw = f / (ew * (w + 1.0) - (w + 2.0) * f / (2.0 * w + 2.0));
        if (!(dw == dw)) break;                      // at the branch point itself (w = -1)
        w -= dw;
        if (fabs(dw) <= 1e-16 * fabs(w)) break;
    }
    return w;
}

// get_LCL (:107-126), Romps (2017)
__device__ __forceinline__ double get_lcl(double p, double T, double r, double rh)
{
    const double E0v = 2.3740e6, cvv = 1418, cvl = 4119, cpv = cvv + Rv;
    const double q = r / (1 + r);
    const double Rm = (1 - q) * Rd + q * Rv;
    const double cpm = (1 - q) * cp + q * cpv;
    const double a = cpm / Rm + (cvl - cpv) / Rv;
    const double b = -(E0v - (cvv - cvl) * T_trip) / (Rv * T);
    const double c = b / a;
    const double T_lcl = c * T / lambertw_m1(pow(rh, 1 / a) * c * exp(c));
    return p * pow(T_lcl / T, cpm / Rm);
}

// calc_T_rho (:129-134), select_thermo = 1
__device__ __forceinline__ double t_rho(double T, double rv) { return T * (1 + rv / eps) / (1 + rv); }

// RectBivariateSpline(p_look, s_look, T, kx = 1, ky = 1).ev = fpbisp/fpbspl: clamp, interval, (c*hx)*hy
__device__ __forceinline__ void tab_cell(const double *__restrict__ x, int n, double arg, int &i, double &w0, double &w1)
{
    arg = (arg < x[0]) ? x[0] : arg;
    arg = (arg > x[n - 1]) ? x[n - 1] : arg;
    int g = (int)((arg - x[0]) / (x[n - 1] - x[0]) * (double)(n - 1));
    g = g < 0 ? 0 : (g > n - 2 ? n - 2 : g);
    while (g < n - 2 && arg >= x[g + 1]) ++g;
    while (g > 0 && arg < x[g]) --g;
    const double f = 1.0 / (x[g + 1] - x[g]);
    i = g;
    w0 = 0.0 + f * (x[g + 1] - arg);
    w1 = f * (arg - x[g]);
}

__device__ __forceinline__ double tab_ev(const Table &t, double p, double s)
{
    if (p != p || s != s) return __longlong_as_double(0x7ff8000000000000LL);
    int i, j;
    double a0, a1, b0, b1;
    tab_cell(t.p, t.np, p, i, a0, a1);
    tab_cell(t.s, t.ns, s, j, b0, b1);
    const double *r0 = t.T + (size_t)i * t.ns + j, *r1 = r0 + t.ns;
    double sp = 0.0;
    sp = sp + r0[0] * a0 * b0;
    sp = sp + r0[1] * a0 * b1;
    sp = sp + r1[0] * a1 * b0;
    sp = sp + r1[1] * a1 * b1;
    return sp;
}

struct PiArgs {
    Table tab;
    int64_t n_points;
    int n_lev;
    const double *p_env;         // [n_lev] Pa, lowest level (highest pressure) first
    const double *sst, *psl;     // [n_points]
    const double *T_env, *r_env; // [n_lev][n_points]
    double cecd;                 // Ck / Cd
    double *pi;                  // [n_points]
};

__global__ __launch_bounds__(256) void k_potential_intensity(PiArgs a)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.n_points) return;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const int L = a.n_lev;
    const size_t N = (size_t)a.n_points;
    const double sst = a.sst[c], p_surf = a.psl[c];
    const double T_ns = a.T_env[c], r_ns = a.r_env[c], p_ns = a.p_env[0];
    double ess, rs;
    sat_thermo(sst, p_surf, ess, rs);
    const double rh = r_ns / rs * (1 + rs / eps) / (1 + r_ns / eps);
    const double s_ns = s_unsat(T_ns, p_ns, r_ns);
    const double ss = s_sat(sst, p_surf);
    const double p_lcl = get_lcl(p_ns, T_ns, r_ns, rh);

    bool moist = false;                         // past the first level with p_LCL > p (or at the top level)
    double cape_run = 0.0, capes_run = 0.0;     // running sums of the buoyancy integrals
    double cape = 0.0, capes = 0.0;             // ... sampled at the last level where the parcel is not denser
    bool any_a = false, any_s = false;
    bool hit_a_prev = false, hit_s_prev = false;
    double tra_prev = 0, trs_prev = 0, tre_prev = 0, Te_prev = 0, p_prev = 0;
    double add_a = 0.0, add_s = 0.0, T_out_s = nan;
    double lnp = log(p_ns), lnp_prev = lnp;
    for (int k = 0; k < L; ++k) {
        const double p = a.p_env[k];
        const double Te = a.T_env[(size_t)k * N + c], re = a.r_env[(size_t)k * N + c];
        // dlnp = diff(lnp, append = 2 lnp[-1] - lnp[-2])  (:302)
        const double lnp_next = (k + 1 < L) ? log(a.p_env[k + 1]) : (2 * lnp - lnp_prev);
        const double dlnp = lnp_next - lnp;
        const double tre = t_rho(Te, re);
        moist = moist || (p_lcl > p) || (k == L - 1);
        double Ta, ra;
        if (!moist) { Ta = T_ns * pow(p / p_ns, Rd / cp); ra = r_ns; }
        else { double es_; Ta = tab_ev(a.tab, p, s_ns); sat_thermo(Ta, p, es_, ra); }
        const double tra = t_rho(Ta, ra);
        double es2, rsp;
        const double Ts = tab_ev(a.tab, p, ss);
        sat_thermo(Ts, p, es2, rsp);
        const double trs = t_rho(Ts, rsp);
        // outflow between the previous level and this one, kept if the previous level was a hit (:363-388)
        if (hit_s_prev) {
            const double dT1 = trs_prev - tre_prev, dT2 = trs - tre;
            const double p_out = (p_prev * dT2 - p * dT1) / (dT2 - dT1);
            T_out_s = (Te_prev * (p_out - p) + Te * (p_prev - p_out)) / (p_prev - p);
            add_s = Rd * dT1 * (p_prev - p_out) / (p_prev + p_out);
        }
        if (hit_a_prev) {
            const double dT1 = tra_prev - tre_prev, dT2 = tra - tre;
            const double p_out = (p_prev * dT2 - p * dT1) / (dT2 - dT1);
            add_a = Rd * dT1 * (p_prev - p_out) / (p_prev + p_out);
        }
        cape_run += Rd * (tra - tre) * -dlnp;
        capes_run += Rd * (trs - tre) * -dlnp;
        hit_a_prev = tra >= tre;
        hit_s_prev = trs >= tre;
        if (hit_a_prev) { cape = cape_run; any_a = true; }
        if (hit_s_prev) { capes = capes_run; any_s = true; }
        tra_prev = tra; trs_prev = trs; tre_prev = tre; Te_prev = Te; p_prev = p;
        lnp_prev = lnp; lnp = lnp_next;
    }
    // a hit at the top level has no level above it: nothing interpolated (the reference's loop stops at L-2)
    if (hit_a_prev || !any_a) { add_a = 0.0; cape = cape_run; }
    if (hit_s_prev || !any_s) { add_s = 0.0; T_out_s = nan; capes = capes_run; }
    cape += add_a;
    capes += add_s;
    cape = np_maximum(cape, 0.0);
    cape = (cape != cape) ? 0.0 : cape;
    const double x = np_maximum(a.cecd * (sst / T_out_s) * (capes - cape), 0.0);
    const double pi = sqrt(x);
    a.pi[c] = (pi != pi) ? 0.0 : pi;
}

// sat_deficit (:92-104) and conv_q_to_rh (:41-46) at the mid level
__global__ __launch_bounds__(256) void k_chi_rh(int64_t n, const double *__restrict__ sst, const double *__restrict__ psl,
                                                const double *__restrict__ T_mid, const double *__restrict__ q_mid,
                                                double p_mid, double *__restrict__ chi, double *__restrict__ rh)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const double T = T_mid[c], q = q_mid[c];
    const double sp = s_unsat(T, p_mid, q), sps = s_sat(T, p_mid), spss = s_sat(sst[c], psl[c]);
    chi[c] = (sps - sp) / (spss - sps);
    double es, rs;
    sat_thermo(T, p_mid, es, rs);
    const double qs = rs / (1 + rs);
    rh[c] = np_minimum(np_maximum(q / qs, 1e-5), 1.0);
}

}  // namespace th
}  // namespace tcr
