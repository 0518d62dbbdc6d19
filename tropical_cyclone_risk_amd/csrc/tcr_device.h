// Device-side building blocks of the storm integrator (gfx950, fp64, no MFMA).
//
// Everything here restates reference semantics for one storm on one lane; the
// kernels in tcr_kernels.hip decide how lanes map to storms.  Arithmetic that
// feeds discrete decisions (bilinear weights and sums, `land == 1`) keeps the
// exact operation order of FITPACK's fpbisp/fpbspl with kx=ky=1 — which is what
// RectBivariateSpline(kx=1,ky=1).ev executes at intensity/coupled_fast.py:37-57,126
// and track/bam_track.py:100-103 — and the translation unit is compiled with
// -ffp-contract=off so no FMA is fused into it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tcrisk_hip.h"

namespace tcr {

constexpr double kPi = 3.141592653589793;
constexpr int kWindStride = 16;     // 14 fields + 2 pad  -> one 128-B line per grid point
constexpr int kThermoStride = 4;    // vpot, chi, mld, strat -> 32 B per grid point
constexpr int kStaticStride = 2;    // land, bathy -> 16 B per grid point
constexpr int kStepRec = 24;         // accepted-step record: t_old, h, t_new, -, y_old[4], Q[4][4]

// Rectilinear grid: knots + per-cell reciprocal widths (host-computed 1.0/(x[i+1]-x[i]),
// the same IEEE division fpbspl.f performs) + a uniform-grid guess for the cell index.
struct DevAxis {
    int n;
    int affine;              // knots are bitwise x0 + i*dx and every 1/(x[i+1]-x[i]) == rdx (host-verified)
    const double *x, *rx;    // knots, per-cell reciprocal widths (general path)
    double x0, xn, dx, rdx;  // first/last knot; affine step and its reciprocal
    double inv_step;         // (n-1)/(xn-x0): cell-index guess
};

struct DevGrid {
    int nlon, nlat;
    DevAxis ax, ay;
};

// One month slot.  Field-interleaved ("AoS at the grid point") so that a
// bilinear corner is one or two contiguous cache lines instead of 14 planes.
struct DevSlot {
    const double *wind;      // [nlat_w][nlon_w][16]: mean0..3, cov(0,0),(1,0),(1,1),(2,0)...(3,3), pad, pad
    const double *thermo;    // [nlat_t][nlon_t][4] : vpot, chi, mld, strat
    const double *rh;        // [nlat_t][nlon_t]
};

struct DevFields {
    DevGrid wg, tg, hg, mg;
    const DevSlot *slots;    // device array
    const double *stat;      // [nlat_h][nlon_h][2]: land, bathy
    const uint8_t *run_mask; // [nlat_m][nlon_m]
    const uint8_t *basin_masks;   // [7][nlat_m][nlon_m]
};

struct Cell {
    int i;
    double w0, w1;
};

// fpbisp.f (clamp + interval search) and fpbspl.f (k = 1 weights).
// Affine axes (ERA5's 1 deg / 0.25 deg grids, CMIP regular grids) need no memory
// traffic at all: the host has verified that x0 + i*dx reproduces every knot
// bit for bit and that 1/(x[i+1]-x[i]) is the same double in every cell.
__device__ __forceinline__ Cell locate(const DevAxis &A, double arg)
{
    if (arg < A.x0) arg = A.x0;
    if (arg > A.xn) arg = A.xn;
    const int n = A.n;
    int i = (int)((arg - A.x0) * A.inv_step);
    i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
    Cell c;
    if (A.affine) {
        double xl = A.x0 + (double)i * A.dx, xr = A.x0 + (double)(i + 1) * A.dx;
        if (i < n - 2 && arg >= xr) { ++i; xl = xr; xr = A.x0 + (double)(i + 1) * A.dx; }
        else if (i > 0 && arg < xl) { --i; xr = xl; xl = A.x0 + (double)i * A.dx; }
        c.i = i;
        c.w0 = 0.0 + A.rdx * (xr - arg);
        c.w1 = A.rdx * (arg - xl);
    } else {
        const double *__restrict__ x = A.x;
        while (i < n - 2 && arg >= x[i + 1]) ++i;
        while (i > 0 && arg < x[i]) --i;
        const double f = A.rx[i];
        c.i = i;
        c.w0 = 0.0 + f * (x[i + 1] - arg);
        c.w1 = f * (arg - x[i]);
    }
    return c;
}

// Bilinear sum of NF interleaved fields in fpbisp.f's order:
// (x0,y0), (x0,y1), (x1,y0), (x1,y1), each term (c*hx)*hy.
template <int NF, int STRIDE>
__device__ __forceinline__ void bilinear(const double *__restrict__ base, int nlon, const Cell &cx,
                                         const Cell &cy, double (&out)[NF])
{
    const double *p00 = base + ((size_t)cy.i * nlon + cx.i) * STRIDE;
    const double *p01 = p00 + (size_t)nlon * STRIDE;
    const double2 *q00 = reinterpret_cast<const double2 *>(p00);
    const double2 *q01 = reinterpret_cast<const double2 *>(p01);
    const double2 *q10 = reinterpret_cast<const double2 *>(p00 + STRIDE);
    const double2 *q11 = reinterpret_cast<const double2 *>(p01 + STRIDE);
    constexpr int NV = (NF + 1) / 2;
    double2 c00[NV], c01[NV], c10[NV], c11[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { c00[k] = q00[k]; c01[k] = q01[k]; c10[k] = q10[k]; c11[k] = q11[k]; }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const double a = (f & 1) ? c00[f >> 1].y : c00[f >> 1].x;
        const double b = (f & 1) ? c01[f >> 1].y : c01[f >> 1].x;
        const double c = (f & 1) ? c10[f >> 1].y : c10[f >> 1].x;
        const double d = (f & 1) ? c11[f >> 1].y : c11[f >> 1].x;
        double sp = 0.0;
        sp = sp + a * cx.w0 * cy.w0;
        sp = sp + b * cx.w0 * cy.w1;
        sp = sp + c * cx.w1 * cy.w0;
        sp = sp + d * cx.w1 * cy.w1;
        out[f] = sp;
    }
}

__device__ __forceinline__ double ts_at(const tcr_params &P, int i)
{
    // np.linspace(0, total_time, n_steps)[i]
    return (i == P.n_steps - 1) ? P.total_time : (double)i * (P.total_time / (double)(P.n_steps - 1));
}

// interp1d(t_s, Fs, axis=1)(t): scipy/interpolate/_interpolate.py:457-486.
// fs is this storm's table laid out [n_steps][4].
__device__ __forceinline__ void fs_at(const tcr_params &P, const double *__restrict__ fs, double t,
                                      double (&F)[4])
{
    const int ns = P.n_steps;
    const double step = P.total_time / (double)(ns - 1);
    int idx = (int)ceil(t / step);
    idx = idx < 0 ? 0 : (idx > ns - 1 ? ns - 1 : idx);
    while (idx > 0 && ts_at(P, idx - 1) >= t) --idx;          // searchsorted(..., side='left')
    while (idx < ns - 1 && ts_at(P, idx) < t) ++idx;
    idx = idx < 1 ? 1 : idx;
    const int lo = idx - 1;
    const double x_lo = ts_at(P, lo), x_hi = ts_at(P, idx);
    const double2 *q = reinterpret_cast<const double2 *>(fs + (size_t)lo * 4);
    const double2 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3];
    const double dx = x_hi - x_lo, dt = t - x_lo;
    F[0] = (b0.x - a0.x) / dx * dt + a0.x;
    F[1] = (b0.y - a0.y) / dx * dt + a0.y;
    F[2] = (b1.x - a1.x) / dx * dt + a1.x;
    F[3] = (b1.y - a1.y) / dx * dt + a1.y;
}

// track/bam_track.py:116-128: mean + chol(cov) · F(t).  Cholesky as LAPACK dpotrf('L')
// unblocked: ajj = a_jj - dot; fail if ajj <= 0; sub-column scaled by 1/ajj.
__device__ __forceinline__ void env_winds(const tcr_params &P, const DevFields &D, const DevSlot &S,
                                          const double *__restrict__ fs, double lon, double lat,
                                          double t, double (&w)[4])
{
    if (lon != lon || t != t) { w[0] = w[1] = w[2] = w[3] = 0.0; return; }
    const Cell cx = locate(D.wg.ax, lon);
    const Cell cy = locate(D.wg.ay, lat);
    double q[14];
    bilinear<14, kWindStride>(S.wind, D.wg.nlon, cx, cy, q);
    double F[4];
    fs_at(P, fs, t, F);
    // packed lower triangle: q[4]=a00 q[5]=a10 q[6]=a11 q[7]=a20 q[8]=a21 q[9]=a22 q[10]=a30 q[11]=a31 q[12]=a32 q[13]=a33
    bool ok = true;
    double l00, l10, l20, l30, l11, l21, l31, l22, l32, l33;
    {
        double ajj = q[4] - 0.0;
        ok = ok && (ajj > 0.0);
        l00 = sqrt(ajj);
        const double r = 1.0 / l00;
        l10 = (q[5] - 0.0) * r; l20 = (q[7] - 0.0) * r; l30 = (q[10] - 0.0) * r;
    }
    {
        double ajj = q[6] - (0.0 + l10 * l10);
        ok = ok && (ajj > 0.0);
        l11 = sqrt(ajj);
        const double r = 1.0 / l11;
        l21 = (q[8] - (0.0 + l20 * l10)) * r;
        l31 = (q[11] - (0.0 + l30 * l10)) * r;
    }
    {
        double ajj = q[9] - ((0.0 + l20 * l20) + l21 * l21);
        ok = ok && (ajj > 0.0);
        l22 = sqrt(ajj);
        const double r = 1.0 / l22;
        l32 = (q[12] - ((0.0 + l30 * l20) + l31 * l21)) * r;
    }
    {
        double ajj = q[13] - (((0.0 + l30 * l30) + l31 * l31) + l32 * l32);
        ok = ok && (ajj > 0.0);
        l33 = sqrt(ajj);
    }
    if (!ok) { w[0] = w[1] = w[2] = w[3] = 0.0; return; }     // LinAlgError branch (bam_track.py:124-126)
    w[0] = q[0] + ((((0.0 + l00 * F[0]) + 0.0 * F[1]) + 0.0 * F[2]) + 0.0 * F[3]);
    w[1] = q[1] + ((((0.0 + l10 * F[0]) + l11 * F[1]) + 0.0 * F[2]) + 0.0 * F[3]);
    w[2] = q[2] + ((((0.0 + l20 * F[0]) + l21 * F[1]) + l22 * F[2]) + 0.0 * F[3]);
    w[3] = q[3] + ((((0.0 + l30 * F[0]) + l31 * F[1]) + l32 * F[2]) + l33 * F[3]);
}

__device__ __forceinline__ double sign_of(double x) { return (double)((x > 0) - (x < 0)); }
__device__ __forceinline__ double np_min(double a, double b) { return (a != a) ? a : (a < b ? a : b); }
__device__ __forceinline__ double np_max(double a, double b) { return (a != a) ? a : (a > b ? a : b); }

// coupled_fast.py:183-192
__device__ __forceinline__ void steering(const tcr_params &P, double v, double (&c)[2])
{
    if (!P.coupled_track) { c[0] = P.steering_coefs[0]; c[1] = P.steering_coefs[1]; return; }
    double a0 = (v * 1.94384) * P.m_alpha[0] + P.y_alpha[0];
    double a1 = (v * 1.94384) * P.m_alpha[1] + P.y_alpha[1];
    a0 = np_max(np_min(a0, P.alpha_max[0]), P.alpha_min[0]);
    a1 = np_max(np_min(a1, P.alpha_max[1]), P.alpha_min[1]);
    if (a0 != a0 || a1 != a1) { a0 = P.y_alpha[0]; a1 = P.y_alpha[1]; }
    c[0] = a0; c[1] = a1;
}

struct Rhs {
    double d[4];     // d lon/dt, d lat/dt, dv/dt, dm/dt
    double alpha;    // ocean feedback (probe only)
    double shear, vpot, chi;   // what the ventilation gate needs (coupled_fast.py:238-244)
};

// coupled_fast.py:196-207 (dydt) given the raw env winds at (lon, lat, t):
// _step_bam_track (bam_track.py:131-144), _dvdt (:141-150), _calc_alpha/_calc_z (:65-94),
// _dmdt (:175-180).  `shear` is the gate's S of the *raw* winds (coupled_fast.py:238).
__device__ __forceinline__ Rhs rhs_from_winds(const tcr_params &P, const DevFields &D, const DevSlot &S,
                                              double h_bl, double lon, double lat, double v, double m,
                                              const double (&w_raw)[4])
{
    Rhs r;
    double c[2], w[4], vb0, vb1;
    steering(P, v, c);
    {
        const double du = w_raw[0] - w_raw[2], dw = w_raw[1] - w_raw[3];
        r.shear = sqrt(du * du + dw * dw);
    }
    if (fabs(lat) >= 80) {
        vb0 = vb1 = 0.0; w[0] = w[1] = w[2] = w[3] = 0.0;
    } else {
        w[0] = w_raw[0]; w[1] = w_raw[1]; w[2] = w_raw[2]; w[3] = w_raw[3];
        const double cl = cos(lat * (kPi / 180.0));                        // np.deg2rad
        vb0 = (w[0] * c[0] + w[2] * c[1]) + P.u_beta * cl;
        vb1 = (w[1] * c[0] + w[3] * c[1]) + (sign_of(lat) * P.v_beta) * cl;
    }
    r.d[0] = vb0 / P.earth_R * 180. / kPi / cos(lat * kPi / 180.);
    r.d[1] = vb1 / P.earth_R * 180. / kPi;

    // thermo grid: vpot, chi, mld, strat; hi-res grid: land, bathy
    const Cell tx = locate(D.tg.ax, lon);
    const Cell ty = locate(D.tg.ay, lat);
    double th[4];
    bilinear<4, kThermoStride>(S.thermo, D.tg.nlon, tx, ty, th);
    const Cell hx = locate(D.hg.ax, lon);
    const Cell hy = locate(D.hg.ay, lat);
    double lb[2];
    bilinear<2, kStaticStride>(D.stat, D.hg.nlon, hx, hy, lb);
    const double vp = (lb[0] == 1.0) ? 0.0 : th[0];                        // coupled_fast.py:35-58
    const double h_m = th[2], gam = th[3], bathy = lb[1];
    double al;
    if (bathy >= 0 || -h_m <= bathy || gam == 0) {
        al = 1.0;
    } else {
        const double uT = sqrt(vb0 * vb0 + vb1 * vb1);
        const double z = 0.01 * pow(gam, -0.4) * h_m * uT * vp / v;
        const double zc = np_min(np_max(z, 0.0), 100.0);
        al = 1 - 0.87 * exp(-zc);
    }
    r.alpha = al;
    const double beta = 1 - P.epsilon - P.kappa;
    const double gamma = P.epsilon + al * P.kappa;
    const double m3 = m * m * m;
    double dv = 0.5 * P.Ck / h_bl * (al * beta * (vp * vp) * m3 - (1 - gamma * m3) * (v * v));
    if (dv != dv) dv = 0.0;
    const double du = w[0] - w[2], dw = w[1] - w[3];
    const double venti = sqrt(du * du + dw * dw) * th[1];
    r.vpot = vp; r.chi = th[1];
    r.d[2] = dv;
    r.d[3] = 0.5 * P.Ck / h_bl * ((1 - m) * v - venti * m);
    return r;
}

__device__ __forceinline__ Rhs rhs_eval(const tcr_params &P, const DevFields &D, const DevSlot &S,
                                        const double *__restrict__ fs, double h_bl, double t,
                                        double lon, double lat, double v, double m)
{
    double w[4];
    env_winds(P, D, S, fs, lon, lat, t, w);
    return rhs_from_winds(P, D, S, h_bl, lon, lat, v, m, w);
}

// coupled_fast.py:246-256 with util/basins.py:32-37 (dx = 1); always >= 0
__device__ __forceinline__ double event_fn(const tcr_params &P, double lon, double lat, double v)
{
    const bool inside = (P.box[0] + 1 < lon) && (lon < P.box[2] - 1) &&
                        (P.box[1] + 1 < lat) && (lat < P.box[3] - 1);
    if (!inside) return 0.0;
    if (fabs(lat) <= 2) return 0.0;
    const double g = v - P.v_dissipate;
    return g > 0 ? g : 0.0;
}

}  // namespace tcr
