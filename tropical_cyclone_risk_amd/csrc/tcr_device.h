// Device-side building blocks of the storm integrator (gfx950, fp64, no MFMA).
//
// Everything here restates reference semantics for one storm on one lane; the
// kernels in tcr_kernels.hip decide how lanes map to storms.  Arithmetic that
// feeds discrete decisions (bilinear weights and sums, `land == 1`) keeps the
// exact operation order of FITPACK's fpbisp/fpbspl with kx=ky=1 — which is what
// RectBivariateSpline(kx=1,ky=1).ev executes at intensity/coupled_fast.py:37-57,126
// and track/bam_track.py:100-103 — and the translation unit is compiled with
// -ffp-contract=off so no FMA is fused into it.
//
// Performance shape of one evaluation of fun(t, y) (the unit everything is made of):
// all six cell searches are pure ALU on affine grids, then the 44 16-byte gathers
// (28 wind, 4 forcing table, 8 thermo, 4 land/bathymetry) are independent of each
// other and are issued back to back — ONE memory round trip per evaluation — and
// the rest is straight-line fp64 math with selects instead of branches, so the
// lanes of a wave never diverge inside an evaluation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/tcrisk_hip.h"

// Read of an evaluation constant (LDS).  Kept as a hook: making it a volatile access (reload at
// every use instead of letting the compiler hoist the ~70 constants into VGPRs) was measured and
// made register allocation worse, not better.
#define RD(x) (x)

namespace tcr {

constexpr double kPi = 3.141592653589793;
constexpr int kWindStride = 16;     // 14 fields + 2 pad  -> one 128-B line per grid point
constexpr int kThermoStride = 4;    // vpot, chi, mld, strat -> 32 B per grid point
constexpr int kStaticStride = 2;    // land, bathy -> 16 B per grid point
constexpr int kStepRec = 36;        // accepted-step record: t_old, h, t_new, -, y_old[4], K[7][4]

// One axis of a rectilinear grid: knots + per-cell reciprocal widths (host-computed
// 1.0/(x[i+1]-x[i]), the IEEE division fpbspl.f performs) + a uniform guess for the cell.
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
    const double *rh;        // [nlat_r][nlon_r] on the uncropped thermo grid (seeding only)
};

struct DevFields {
    DevGrid wg, tg, hg, mg, rg;   // wind, thermo, static hi-res, basin masks, (uncropped) rh grid
    const DevSlot *slots;    // device array
    int n_slots;
    const double *stat;      // [nlat_h][nlon_h][2]: land, bathy
    const uint8_t *run_mask; // [nlat_m][nlon_m]
    const uint8_t *basin_masks;   // [7][nlat_m][nlon_m]
    int all_affine;          // wind, thermo and static axes are all affine
};

// Constants one evaluation reads.  Kernels copy them into LDS once per workgroup so
// they neither occupy ~150 SGPRs for the whole kernel (spilling) nor get reloaded
// from the kernarg segment in the hot loop.
struct EvalK {
    DevAxis wx, wy, tx, ty, hx, hy;
    const double *stat;
    double earth_R, Ck, epsilon, kappa, u_beta, v_beta;
    double y_alpha[2], m_alpha[2], alpha_max[2], alpha_min[2], steering_coefs[2];
    double total_time, tstep, inv_tstep;
    int n_steps, coupled_track;
};

__host__ __device__ inline void make_eval_k(const tcr_params &P, const DevFields &D, EvalK &K)
{
    K.wx = D.wg.ax; K.wy = D.wg.ay; K.tx = D.tg.ax; K.ty = D.tg.ay; K.hx = D.hg.ax; K.hy = D.hg.ay;
    K.stat = D.stat;
    K.earth_R = P.earth_R; K.Ck = P.Ck; K.epsilon = P.epsilon; K.kappa = P.kappa;
    K.u_beta = P.u_beta; K.v_beta = P.v_beta;
    for (int i = 0; i < 2; ++i) {
        K.y_alpha[i] = P.y_alpha[i]; K.m_alpha[i] = P.m_alpha[i]; K.alpha_max[i] = P.alpha_max[i];
        K.alpha_min[i] = P.alpha_min[i]; K.steering_coefs[i] = P.steering_coefs[i];
    }
    K.total_time = P.total_time;
    K.tstep = P.total_time / (double)(P.n_steps - 1);
    K.inv_tstep = (double)(P.n_steps - 1) / P.total_time;
    K.n_steps = P.n_steps; K.coupled_track = P.coupled_track;
}

struct Cell {
    int i;
    double w0, w1;
};

// 16-byte load from *global* memory.  Pointers that travelled through structs / LDS are generic,
// and a generic (flat_load) access also counts against lgkmcnt, so every LDS wait would stall on
// the outstanding field gathers; telling the compiler the address space gives global_load.
typedef double tcr_dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ldg16(const double *p)
{
    const tcr_dbl2 v = *(const __attribute__((address_space(1))) tcr_dbl2 *)(p);
    return make_double2(v.x, v.y);
}

// fpbisp.f (clamp + interval search) and fpbspl.f (k = 1 weights).
// Affine axes (ERA5's 1 deg / 0.25 deg grids, CMIP regular grids) need no memory
// traffic at all: the host has verified that x0 + i*dx reproduces every knot
// bit for bit and that 1/(x[i+1]-x[i]) is the same double in every cell.
template <bool AFFINE>
__device__ __forceinline__ Cell locate_t(const DevAxis &A, double arg)
{
    const double ax0 = RD(A.x0), axn = RD(A.xn);
    arg = (arg < ax0) ? ax0 : arg;
    arg = (arg > axn) ? axn : arg;
    const int n = RD(A.n);
    int i = (int)((arg - ax0) * RD(A.inv_step));
    i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
    Cell c;
    if (AFFINE) {
        const double dx = RD(A.dx), x0 = ax0, rdx = RD(A.rdx);
        const double xl = x0 + (double)i * dx, xr = x0 + (double)(i + 1) * dx;
        const bool up = (i < n - 2) && (arg >= xr);
        const bool dn = !up && (i > 0) && (arg < xl);
        i += up ? 1 : (dn ? -1 : 0);
        const double xl2 = up ? xr : (dn ? x0 + (double)i * dx : xl);
        const double xr2 = up ? x0 + (double)(i + 1) * dx : (dn ? xl : xr);
        c.i = i;
        c.w0 = 0.0 + rdx * (xr2 - arg);
        c.w1 = rdx * (arg - xl2);
    } else {
        const double *__restrict__ x = RD(A.x);
        while (i < n - 2 && arg >= x[i + 1]) ++i;
        while (i > 0 && arg < x[i]) --i;
        const double f = RD(A.rx)[i];
        c.i = i;
        c.w0 = 0.0 + f * (x[i + 1] - arg);
        c.w1 = f * (arg - x[i]);
    }
    return c;
}

__device__ __forceinline__ Cell locate(const DevAxis &A, double arg)
{
    return RD(A.affine) ? locate_t<true>(A, arg) : locate_t<false>(A, arg);
}

// The four corners of NF interleaved fields as 16-byte gathers ...
template <int NF>
struct Corners {
    static constexpr int NV = (NF + 1) / 2;
    double2 c00[NV], c01[NV], c10[NV], c11[NV];
};

template <int NF, int STRIDE>
__device__ __forceinline__ void gather(const double *__restrict__ base, int nlon, const Cell &cx,
                                       const Cell &cy, Corners<NF> &C)
{
    const double *p00 = base + ((size_t)cy.i * nlon + cx.i) * STRIDE;
    const double *p01 = p00 + (size_t)nlon * STRIDE;
#pragma unroll
    for (int k = 0; k < Corners<NF>::NV; ++k) {
        C.c00[k] = ldg16(p00 + 2 * k); C.c01[k] = ldg16(p01 + 2 * k);
        C.c10[k] = ldg16(p00 + STRIDE + 2 * k); C.c11[k] = ldg16(p01 + STRIDE + 2 * k);
    }
}

// ... and their bilinear sums in fpbisp.f's order: (x0,y0), (x0,y1), (x1,y0), (x1,y1),
// each term (c*hx)*hy.
template <int NF>
__device__ __forceinline__ void blend(const Corners<NF> &C, const Cell &cx, const Cell &cy, double (&out)[NF])
{
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const double a = (f & 1) ? C.c00[f >> 1].y : C.c00[f >> 1].x;
        const double b = (f & 1) ? C.c01[f >> 1].y : C.c01[f >> 1].x;
        const double c = (f & 1) ? C.c10[f >> 1].y : C.c10[f >> 1].x;
        const double d = (f & 1) ? C.c11[f >> 1].y : C.c11[f >> 1].x;
        double sp = 0.0;
        sp = sp + a * cx.w0 * cy.w0;
        sp = sp + b * cx.w0 * cy.w1;
        sp = sp + c * cx.w1 * cy.w0;
        sp = sp + d * cx.w1 * cy.w1;
        out[f] = sp;
    }
}

template <int NF, int STRIDE>
__device__ __forceinline__ void bilinear(const double *__restrict__ base, int nlon, const Cell &cx,
                                         const Cell &cy, double (&out)[NF])
{
    Corners<NF> C;
    gather<NF, STRIDE>(base, nlon, cx, cy, C);
    blend<NF>(C, cx, cy, out);
}

__device__ __forceinline__ double ts_at(const tcr_params &P, int i)
{
    // np.linspace(0, total_time, n_steps)[i]
    return (i == P.n_steps - 1) ? P.total_time : (double)i * (P.total_time / (double)(P.n_steps - 1));
}

__device__ __forceinline__ double ts_k(const EvalK &K, int i)
{
    return (i == RD(K.n_steps) - 1) ? RD(K.total_time) : (double)i * RD(K.tstep);
}

// interp1d(t_s, Fs, axis=1)(t) (scipy/interpolate/_interpolate.py:457-486), split into the
// bracket search (searchsorted side='left', clipped to [1, n-1]; branch-free: the guess from
// t * (1/step) is at most one off) and the blend, so the 64-byte gather sits with the others.
struct FsBracket {
    int lo;
    double x_lo, dx;
};

__device__ __forceinline__ FsBracket fs_bracket(const EvalK &K, double t)
{
    const int ns = RD(K.n_steps);
    int idx = (int)ceil(t * RD(K.inv_tstep));
    idx = idx < 0 ? 0 : (idx > ns - 1 ? ns - 1 : idx);
    const bool dn = (idx > 0) && (ts_k(K, idx - 1) >= t);
    const bool up = !dn && (idx < ns - 1) && (ts_k(K, idx) < t);
    idx += up ? 1 : (dn ? -1 : 0);
    idx = idx < 1 ? 1 : idx;
    FsBracket b;
    b.lo = idx - 1;
    b.x_lo = ts_k(K, idx - 1);
    b.dx = ts_k(K, idx) - b.x_lo;
    return b;
}

struct FsPair {
    double2 a0, a1, b0, b1;
};

__device__ __forceinline__ void fs_gather(const double *__restrict__ fs, const FsBracket &b, FsPair &p)
{
    const double *q = fs + (size_t)b.lo * 4;
    p.a0 = ldg16(q); p.a1 = ldg16(q + 2); p.b0 = ldg16(q + 4); p.b1 = ldg16(q + 6);
}

__device__ __forceinline__ void fs_blend(const FsPair &p, const FsBracket &b, double t, double (&F)[4])
{
    const double dt = t - b.x_lo;
    F[0] = (p.b0.x - p.a0.x) / b.dx * dt + p.a0.x;
    F[1] = (p.b0.y - p.a0.y) / b.dx * dt + p.a0.y;
    F[2] = (p.b1.x - p.a1.x) / b.dx * dt + p.a1.x;
    F[3] = (p.b1.y - p.a1.y) / b.dx * dt + p.a1.y;
}

// track/bam_track.py:116-128 after the 14 lookups: mean + chol(cov) · F(t).  Cholesky as
// LAPACK dpotrf('L') unblocked: ajj = a_jj - dot; fail if ajj <= 0 (-> zero winds, the
// LinAlgError branch :124-126); sub-column scaled by 1/ajj.  NaN lon/t -> zeros (:117-118).
__device__ __forceinline__ void winds_from_lookups(const double (&q)[14], const double (&F)[4], double lon,
                                                   double t, double (&w)[4])
{
    // packed lower triangle: q[4]=a00 q[5]=a10 q[6]=a11 q[7]=a20 q[8]=a21 q[9]=a22 q[10]=a30 q[11]=a31 q[12]=a32 q[13]=a33
    bool ok = !(lon != lon) && !(t != t);
    double l00, l10, l20, l30, l11, l21, l31, l22, l32, l33;
    {
        const double ajj = q[4] - 0.0;
        ok = ok && (ajj > 0.0);
        l00 = sqrt(ajj);
        const double r = 1.0 / l00;
        l10 = (q[5] - 0.0) * r; l20 = (q[7] - 0.0) * r; l30 = (q[10] - 0.0) * r;
    }
    {
        const double ajj = q[6] - (0.0 + l10 * l10);
        ok = ok && (ajj > 0.0);
        l11 = sqrt(ajj);
        const double r = 1.0 / l11;
        l21 = (q[8] - (0.0 + l20 * l10)) * r;
        l31 = (q[11] - (0.0 + l30 * l10)) * r;
    }
    {
        const double ajj = q[9] - ((0.0 + l20 * l20) + l21 * l21);
        ok = ok && (ajj > 0.0);
        l22 = sqrt(ajj);
        const double r = 1.0 / l22;
        l32 = (q[12] - ((0.0 + l30 * l20) + l31 * l21)) * r;
    }
    {
        const double ajj = q[13] - (((0.0 + l30 * l30) + l31 * l31) + l32 * l32);
        ok = ok && (ajj > 0.0);
        l33 = sqrt(ajj);
    }
    const double w0 = q[0] + ((((0.0 + l00 * F[0]) + 0.0 * F[1]) + 0.0 * F[2]) + 0.0 * F[3]);
    const double w1 = q[1] + ((((0.0 + l10 * F[0]) + l11 * F[1]) + 0.0 * F[2]) + 0.0 * F[3]);
    const double w2 = q[2] + ((((0.0 + l20 * F[0]) + l21 * F[1]) + l22 * F[2]) + 0.0 * F[3]);
    const double w3 = q[3] + ((((0.0 + l30 * F[0]) + l31 * F[1]) + l32 * F[2]) + l33 * F[3]);
    w[0] = ok ? w0 : 0.0; w[1] = ok ? w1 : 0.0; w[2] = ok ? w2 : 0.0; w[3] = ok ? w3 : 0.0;
}

// _env_winds(lon, lat, t) on its own (output samples, probes)
template <bool AFFINE>
__device__ __forceinline__ void env_winds(const EvalK &K, const DevSlot &S, const double *__restrict__ fs,
                                          double lon, double lat, double t, double (&w)[4])
{
    const Cell cx = locate_t<AFFINE>(K.wx, lon);
    const Cell cy = locate_t<AFFINE>(K.wy, lat);
    const FsBracket fb = fs_bracket(K, t);
    Corners<14> CW;
    FsPair fp;
    gather<14, kWindStride>(S.wind, RD(K.wx.n), cx, cy, CW);
    fs_gather(fs, fb, fp);
    double q[14], F[4];
    blend<14>(CW, cx, cy, q);
    fs_blend(fp, fb, t, F);
    winds_from_lookups(q, F, lon, t, w);
}

__device__ __forceinline__ double sign_of(double x) { return (double)((x > 0) - (x < 0)); }
__device__ __forceinline__ double np_min(double a, double b) { return (a != a) ? a : (a < b ? a : b); }
__device__ __forceinline__ double np_max(double a, double b) { return (a != a) ? a : (a > b ? a : b); }

struct Rhs {
    double d[4];     // d lon/dt, d lat/dt, dv/dt, dm/dt
    double w[4];     // raw env winds at the point (what _env_winds returns)
    double alpha;    // ocean feedback (probe only)
    double shear, vpot, chi;   // what the ventilation gate needs (coupled_fast.py:238-244)
    int dec;         // decision probe (tests only; dead code elsewhere): bit0 `land == 1`, bit1 PI != 0, bit2 |land - 1| <= 1e-12
};

// Everything of dydt after the lookups: steering, beta-advection, _dvdt, ocean feedback, _dmdt.
__device__ __forceinline__ void rhs_tail(const EvalK &K, double h_bl, double lat, double v, double m,
                                         const double (&th)[4], const double (&lb)[2], Rhs &r)
{
    {
        const double du = r.w[0] - r.w[2], dw = r.w[1] - r.w[3];
        r.shear = sqrt(du * du + dw * dw);
    }
    // steering coefficients
    double c0, c1;
    {
        double a0 = (v * 1.94384) * RD(K.m_alpha[0]) + RD(K.y_alpha[0]);
        double a1 = (v * 1.94384) * RD(K.m_alpha[1]) + RD(K.y_alpha[1]);
        a0 = np_max(np_min(a0, RD(K.alpha_max[0])), RD(K.alpha_min[0]));
        a1 = np_max(np_min(a1, RD(K.alpha_max[1])), RD(K.alpha_min[1]));
        const bool bad = (a0 != a0) || (a1 != a1);
        a0 = bad ? RD(K.y_alpha[0]) : a0;
        a1 = bad ? RD(K.y_alpha[1]) : a1;
        c0 = RD(K.coupled_track) ? a0 : RD(K.steering_coefs[0]);
        c1 = RD(K.coupled_track) ? a1 : RD(K.steering_coefs[1]);
    }
    // beta-advection; |lat| >= 80 -> zero motion and zero winds (bam_track.py:134-135)
    const bool polar = fabs(lat) >= 80;
    const double w0 = polar ? 0.0 : r.w[0], w1 = polar ? 0.0 : r.w[1];
    const double w2 = polar ? 0.0 : r.w[2], w3 = polar ? 0.0 : r.w[3];
    const double cl = cos(lat * (kPi / 180.0));                            // np.deg2rad
    double vb0 = (w0 * c0 + w2 * c1) + RD(K.u_beta) * cl;
    double vb1 = (w1 * c0 + w3 * c1) + (sign_of(lat) * RD(K.v_beta)) * cl;
    vb0 = polar ? 0.0 : vb0;
    vb1 = polar ? 0.0 : vb1;
    r.d[0] = vb0 / RD(K.earth_R) * 180. / kPi / cos(lat * kPi / 180.);
    r.d[1] = vb1 / RD(K.earth_R) * 180. / kPi;
    // intensity
    const double vp = (lb[0] == 1.0) ? 0.0 : th[0];                        // coupled_fast.py:35-58
    r.dec = (lb[0] == 1.0 ? 1 : 0) | (th[0] != 0.0 ? 2 : 0) | (fabs(lb[0] - 1.0) <= 1e-12 ? 4 : 0);
    const double h_m = th[2], gam = th[3], bathy = lb[1];
    const bool no_mix = (bathy >= 0) || (-h_m <= bathy) || (gam == 0);
    const double uT = sqrt(vb0 * vb0 + vb1 * vb1);
    const double z = 0.01 * pow(gam, -0.4) * h_m * uT * vp / v;
    const double zc = np_min(np_max(z, 0.0), 100.0);
    const double al = no_mix ? 1.0 : 1 - 0.87 * exp(-zc);
    r.alpha = al;
    const double beta = 1 - RD(K.epsilon) - RD(K.kappa);
    const double gamma = RD(K.epsilon) + al * RD(K.kappa);
    const double m3 = m * m * m;
    const double dv = 0.5 * RD(K.Ck) / h_bl * (al * beta * (vp * vp) * m3 - (1 - gamma * m3) * (v * v));
    const double du = w0 - w2, dw = w1 - w3;
    const double venti = sqrt(du * du + dw * dw) * th[1];
    r.vpot = vp; r.chi = th[1];
    r.d[2] = (dv != dv) ? 0.0 : dv;
    r.d[3] = 0.5 * RD(K.Ck) / h_bl * ((1 - m) * v - venti * m);
}

// fun(t, y) = Coupled_FAST.dydt (coupled_fast.py:196-207): _calc_steering_coefs (:183-192),
// _step_bam_track (bam_track.py:131-144) on _env_winds (:116-128), _dvdt (:141-150) with
// _get_current_vpot (:54-58), _calc_alpha/_calc_z (:65-94), _dmdt (:175-180).
template <bool AFFINE>
__device__ __forceinline__ Rhs rhs_eval(const EvalK &K, const DevSlot &S, const double *__restrict__ fs,
                                        double h_bl, double t, double lon, double lat, double v, double m)
{
    // ---- address generation: pure ALU on affine grids
    const Cell wx = locate_t<AFFINE>(K.wx, lon), wy = locate_t<AFFINE>(K.wy, lat);
    const Cell tx = locate_t<AFFINE>(K.tx, lon), ty = locate_t<AFFINE>(K.ty, lat);
    const Cell hx = locate_t<AFFINE>(K.hx, lon), hy = locate_t<AFFINE>(K.hy, lat);
    const FsBracket fb = fs_bracket(K, t);
    // ---- one round of independent 16-byte gathers
    Corners<14> CW;
    Corners<4> CT;
    Corners<2> CH;
    FsPair fp;
    gather<14, kWindStride>(S.wind, RD(K.wx.n), wx, wy, CW);
    fs_gather(fs, fb, fp);
    gather<4, kThermoStride>(S.thermo, RD(K.tx.n), tx, ty, CT);
    gather<2, kStaticStride>(RD(K.stat), RD(K.hx.n), hx, hy, CH);
    // ---- straight-line math
    Rhs r;
    double q[14], F[4], th[4], lb[2];
    blend<14>(CW, wx, wy, q);
    fs_blend(fp, fb, t, F);
    winds_from_lookups(q, F, lon, t, r.w);
    blend<4>(CT, tx, ty, th);
    blend<2>(CH, hx, hy, lb);
    rhs_tail(K, h_bl, lat, v, m, th, lb, r);
    return r;
}

// Per-lane corner cache for the sequential integrator.  Consecutive RK stage points of a storm
// mostly fall into the same grid cell (a stage moves a storm by a fraction of a degree), and the
// cost of a gather was measured to be per lane-request: keeping the last cell's corner data in
// registers and re-gathering (exec-masked) only when a lane's cell changes cuts the requests per
// evaluation from 44 to ~15 without changing a single value.  The cache registers are the load
// destinations the direct path needs anyway.
struct CornerCache {
    int wi, wj;                      // cached wind cell (-1 = empty)
    Corners<14> CW;
};

__device__ __forceinline__ void cache_reset(CornerCache &C) { C.wi = C.wj = -1; }

template <bool AFFINE>
__device__ __forceinline__ Rhs rhs_eval_cached(CornerCache &C, const EvalK &K, const DevSlot &S,
                                               const double *__restrict__ fs, double h_bl, double t,
                                               double lon, double lat, double v, double m)
{
    const Cell wx = locate_t<AFFINE>(K.wx, lon), wy = locate_t<AFFINE>(K.wy, lat);
    const Cell tx = locate_t<AFFINE>(K.tx, lon), ty = locate_t<AFFINE>(K.ty, lat);
    const Cell hx = locate_t<AFFINE>(K.hx, lon), hy = locate_t<AFFINE>(K.hy, lat);
    const FsBracket fb = fs_bracket(K, t);
    FsPair fp;
    fs_gather(fs, fb, fp);
    if (wx.i != C.wi || wy.i != C.wj) {
        gather<14, kWindStride>(S.wind, RD(K.wx.n), wx, wy, C.CW);
        C.wi = wx.i; C.wj = wy.i;
    }
    Corners<4> CT;
    Corners<2> CH;
    gather<4, kThermoStride>(S.thermo, RD(K.tx.n), tx, ty, CT);
    gather<2, kStaticStride>(RD(K.stat), RD(K.hx.n), hx, hy, CH);
    Rhs r;
    double q[14], F[4], th[4], lb[2];
    blend<14>(C.CW, wx, wy, q);
    fs_blend(fp, fb, t, F);
    winds_from_lookups(q, F, lon, t, r.w);
    blend<4>(CT, tx, ty, th);
    blend<2>(CH, hx, hy, lb);
    rhs_tail(K, h_bl, lat, v, m, th, lb, r);
    return r;
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
