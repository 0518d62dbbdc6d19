// Device-side building blocks of the storm integrator (gfx950, no MFMA).
//
// Everything here restates reference semantics for one storm on one lane; the
// kernels in tcr_kernels.hip decide how lanes map to storms.  All of it is templated
// on the arithmetic / storage type R:
//   R = double  the reference's own precision (it computes in IEEE fp64 throughout).  Arithmetic
//               that feeds discrete decisions (bilinear weights and sums, `land == 1`) keeps the
//               exact operation order of FITPACK's fpbisp/fpbspl with kx=ky=1 — which is what
//               RectBivariateSpline(kx=1,ky=1).ev executes at intensity/coupled_fast.py:37-57,126
//               and track/bam_track.py:100-103 — and the translation unit is compiled with
//               -ffp-contract=off so no FMA is fused into it.  Every R(...) cast below is the
//               identity for this instantiation.
//   R = float   BASELINE config 5 ("fp32 intensity ODE with tolerance study"): fp32 fields, state and
//               RHS arithmetic in the same operation order; *time* (t, h, the output grid, the
//               forcing-table bracket) and the step-size controller stay fp64 (tcr_kernels.hip).
//
// Performance shape of one evaluation of fun(t, y) (the unit everything is made of):
// all six cell searches are pure ALU on affine grids, then the gathers (fp64: 28 wind, 4 forcing
// table, 8 thermo, 4 land/bathymetry 16-byte loads — 2 narrow ones with the exact narrow static storage, StaticLookup —; fp32: 16 + 2 + 4 16-byte and 4 8-byte) are
// independent of each other and are issued back to back — ONE memory round trip per evaluation —
// and the rest is straight-line math with selects instead of branches, so the
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

// Arithmetic policy of the hot path (round 6; the A / B behind every line is in DESIGN.md section 9 and profiles/r06_ab_variants.txt).
// The translation unit is compiled with -ffp-contract=off: unless a function opts in below, every multiply and add is rounded
// on its own, in the reference's operation order.
//   exact (never fused, never substituted)
//                         everything a discrete decision of the reference is taken on, or that two kernels must reproduce
//                         bit for bit: the cell search and bilinear weights (locate_t), every FITPACK bilinear sum (`land == 1`,
//                         coupled_fast.py:35-38; `PI != 0`, `t_strat == 0`, `-h_m <= bathymetry`), the output grid (ts_k, fs_bracket),
//                         dense output (dense_at and the integrator's in-flight 2-day sample, which k_screen must reproduce),
//                         seeding, wind statistics, thermodynamics;
//   same value, fewer instructions (on by default)
//                         TCR_FAST_DIV / TCR_FAST_SQRT: division and square root of fun(t, y) without the range-scaling instructions
//                         (qdiv_nz, qsqrt: the compiler's own Newton sequences; identical results for the operands of this path);
//                         0.5 Ck / h_bl once per storm instead of once per evaluation (ck_over_h: the same expression);
//   libm-class substitutions, each within 2 ulp of the function it replaces (on by default)
//                         TCR_FAST_POW: t_strat ** -0.4 and err ** -0.2 as a Newton iteration for a ** (-1/5) (inv_fifth_root);
//                         TCR_FAST_COS: cos(lat pi / 180) as a polynomial on [-pi / 2, pi / 2] (cos_lat);
//                         TCR_SHARE_COS: dlon/dt divides by the beta drift's cos(deg2rad(lat)) instead of a second cosine of
//                         lat * pi / 180, and multiplies by the host-computed 180 / (pi R) (rhs_track);
//   fused multiply-adds   TCR_FUSE_RK (on): the RK45 step — stage inputs, y_new, error norm, initial-step norms (k_integrate);
//                         TCR_FUSE_RHS (off): the continuous arithmetic inside fun(t, y) — wind blend, forcing blend, Cholesky,
//                         beta-advection, _dvdt / _dmdt.  Measured: with it the isolated chain is 3-8 % shorter and the pipelined
//                         step and the small batches do not move, the parity distribution of a 10 000-storm ensemble does not move
//                         either, but a second storm of the 48 curated NA golden tracks crosses 1e-9 where the tier allows one —
//                         so it stays off.
// TCR_FUSE=0 switches both fusion knobs off at once; every knob at 0 is the arithmetic of rounds 1-5.
#ifndef TCR_FUSE
#define TCR_FUSE 1
#endif
#ifndef TCR_FAST_DIV
#define TCR_FAST_DIV 1
#endif
#ifndef TCR_FAST_SQRT
#define TCR_FAST_SQRT 1
#endif
#ifndef TCR_SHARE_COS
#define TCR_SHARE_COS 1
#endif
#ifndef TCR_FAST_POW
#define TCR_FAST_POW 1
#endif
#ifndef TCR_FAST_COS
#define TCR_FAST_COS 1
#endif
#ifndef TCR_FUSE_RHS
#define TCR_FUSE_RHS 0
#endif
#ifndef TCR_FUSE_RK
#define TCR_FUSE_RK TCR_FUSE
#endif
#if TCR_FUSE_RHS
#define TCR_FP_FUSE _Pragma("clang fp contract(fast)")
#else
#define TCR_FP_FUSE
#endif
#if TCR_FUSE_RK
#define TCR_FP_FUSE_RK _Pragma("clang fp contract(fast)")
#else
#define TCR_FP_FUSE_RK
#endif
#define TCR_FP_EXACT _Pragma("clang fp contract(off)")

namespace tcr {

constexpr double kPi = 3.141592653589793;
constexpr int kWindStride = 16;     // 14 fields + 2 pad  -> one 128-B (fp64) / 64-B (fp32) line per grid point
constexpr int kThermoStride = 4;    // vpot, chi, mld, strat
constexpr int kStaticStride = 2;    // land, bathy
enum StaticMode { kStatF64 = 0, kStatF64Split = 1, kStatPack16 = 2, kStatU8F32 = 3, kStatPack64 = 4 };
constexpr int kPack16Bias = 16384;  // kStatPack16: stored = (bathymetry + bias) * 2 + land
constexpr int kStepHdr = 4;         // accepted-step record: doubles t_old, h, t_new, - ; then R y_old[4], K[7][4]
constexpr int kStepBody = 32;

// doubles per accepted-step record (fp64: 36, the round-1 layout; fp32: 20)
template <typename R> constexpr int step_rec_doubles() { return kStepHdr + kStepBody * (int)sizeof(R) / 8; }

// One axis of a rectilinear grid: knots + per-cell reciprocal widths (host-computed
// 1.0/(x[i+1]-x[i]), the IEEE division fpbspl.f performs) + a uniform guess for the cell.
template <typename R>
struct AxisT {
    int n;
    int affine;              // knots are bitwise x0 + i*dx and every 1/(x[i+1]-x[i]) == rdx (host-verified, in R)
    const R *x, *rx;         // knots, per-cell reciprocal widths (general path)
    R x0, xn, dx, rdx;       // first/last knot; affine step and its reciprocal
    R inv_step;              // (n-1)/(xn-x0): cell-index guess
};
using DevAxis = AxisT<double>;

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
    const float *wind32, *thermo32;   // fp32 copies of the same layouts (NULL until an fp32 entry point is used)
};

template <typename R> __host__ __device__ inline const R *slot_wind(const DevSlot &S);
template <> __host__ __device__ inline const double *slot_wind<double>(const DevSlot &S) { return S.wind; }
template <> __host__ __device__ inline const float *slot_wind<float>(const DevSlot &S) { return S.wind32; }
template <typename R> __host__ __device__ inline const R *slot_thermo(const DevSlot &S);
template <> __host__ __device__ inline const double *slot_thermo<double>(const DevSlot &S) { return S.thermo; }
template <> __host__ __device__ inline const float *slot_thermo<float>(const DevSlot &S) { return S.thermo32; }

struct DevFields {
    DevGrid wg, tg, hg, mg, rg;   // wind, thermo, static hi-res (land [+ bathymetry]), basin masks, (uncropped) rh grid
    const DevSlot *slots;    // device array
    int n_slots;
    const uint8_t *mask_bits;     // [nlat_m][nlon_m]: bit b < 7 = basin mask b (sorted ids), bit 7 = the run basin's mask
    int all_affine;          // wind, thermo and static axes are all affine
};

// Constants one evaluation reads.  Kernels copy them into LDS once per workgroup so
// they neither occupy ~150 SGPRs for the whole kernel (spilling) nor get reloaded
// from the kernarg segment in the hot loop.
template <typename R>
struct EvalKT {
    AxisT<R> wx, wy, tx, ty, hx, hy;
    // land and bathymetry are two independent interpolators in the reference (intensity/geo.py:9-34).  How they are stored
    // is the kernels' static mode SM (StaticLookup below):
    //   kStatF64      one grid, `stat` = [lat][lon][2] R (land, bathymetry interleaved), one 16-byte gather per corner;
    //   kStatF64Split `stat` = the land plane (R) on (hx, hy), `bathy` its own plane (R) on (bx, by);
    //   kStatPack16   one grid, `stat` = [lat][lon] uint16 = ((int)bathymetry + 16384) * 2 + land: exact when land is 0 / 1 and
    //                 the bathymetry is whole metres in [-16384, 16383] (ETOPO / GEBCO); the two lon-adjacent corners of a
    //                 row are ONE 4-byte gather;
    //   kStatPack64   one grid, `stat` = [lat][lon] 8 bytes: the bathymetry as float32, then the land value as a byte (3 pad bytes):
    //                 exact when land is in 0 .. 255 and every bathymetry value round-trips through float32; a row's two corners
    //                 are ONE 16-byte gather;
    //   kStatU8F32    two grids: `stat` = the land plane as uint8 (values 0 .. 255, the reference's int8 land.nc) on (hx, hy), `bathy` a
    //                 float plane on (bx, by) (exact when every value round-trips through float32); a row's two corners are
    //                 one 2-byte and one 8-byte gather.
    // The narrow forms hold exactly the values the fp64 planes would: the kernels widen them to R before FITPACK's arithmetic.
    AxisT<R> bx, by;
    const void *stat;
    const void *bathy;
    R earth_R, Ck, epsilon, kappa, u_beta, v_beta;
    R deg_per_m;             // 1 / earth_R * 180 / pi (rhs_track)
    R y_alpha[2], m_alpha[2], alpha_max[2], alpha_min[2], steering_coefs[2];
    double total_time, tstep, inv_tstep;        // time stays fp64 in both instantiations
    int n_steps, coupled_track;
    int tw_same;             // the thermo grid IS the wind grid (ERA5: both 1 degree): one cell search serves both
    int hb_same;             // land and bathymetry share one grid (bx / by == hx / hy): kStatU8F32 searches the cell once
};
using EvalK = EvalKT<double>;

// Everything but the axes and the static pointer (those come from the staged grids: fp64 straight from
// DevFields, fp32 from the float knot copies the host keeps — tcr_abi.hip).
template <typename R>
__host__ __device__ inline void eval_k_scalars(const tcr_params &P, EvalKT<R> &K)
{
    K.earth_R = (R)P.earth_R; K.Ck = (R)P.Ck; K.epsilon = (R)P.epsilon; K.kappa = (R)P.kappa;
    K.u_beta = (R)P.u_beta; K.v_beta = (R)P.v_beta;
    K.deg_per_m = (R)(1.0 / P.earth_R * 180. / kPi);
    for (int i = 0; i < 2; ++i) {
        K.y_alpha[i] = (R)P.y_alpha[i]; K.m_alpha[i] = (R)P.m_alpha[i]; K.alpha_max[i] = (R)P.alpha_max[i];
        K.alpha_min[i] = (R)P.alpha_min[i]; K.steering_coefs[i] = (R)P.steering_coefs[i];
    }
    K.total_time = P.total_time;
    K.tstep = P.total_time / (double)(P.n_steps - 1);
    K.inv_tstep = (double)(P.n_steps - 1) / P.total_time;
    K.n_steps = P.n_steps; K.coupled_track = P.coupled_track;
}

template <typename R>
struct CellT {
    int i;
    R w0, w1;
};
using Cell = CellT<double>;

// Vector loads from *global* memory.  Pointers that travelled through structs / LDS are generic,
// and a generic (flat_load) access also counts against lgkmcnt, so every LDS wait would stall on
// the outstanding field gathers; telling the compiler the address space gives global_load.
template <typename R, int L> struct VecT { typedef R type __attribute__((ext_vector_type(L))); };

template <typename R, int L>
__device__ __forceinline__ typename VecT<R, L>::type ldg(const R *p)
{
    typedef typename VecT<R, L>::type V;
    return *(const __attribute__((address_space(1))) V *)(p);
}

// Division and square root of the continuous arithmetic of fun(t, y) (see the policy above).  IEEE fp64 division on gfx950 is v_div_scale x 2,
// v_rcp_f64, five fmas, a mul, v_div_fmas, v_div_fixup (12 VALU instructions, 10 of them one dependent chain); the scale / fmas /
// fixup instructions only serve operands whose quotient leaves the normal range, and ocml's sqrt spends 8 of its 22 on the same.
//   qdiv_nz(a, b)   a / b for a divisor that is finite, non-zero and normal (grid steps, cos(lat) equatorward of 80 deg, h_bl, the
//                   error scale atol + rtol |y|, standard deviations of a positive-definite covariance) and a finite or NaN
//                   dividend: v_rcp_f64, two Newton steps, q = a r, one residual correction — LLVM's own sequence without the
//                   range scaling, within 1 ulp of a / b.  Divisions whose divisor can be zero (`/ v`) stay IEEE.
//   qsqrt(a)        v_rsq_f64, one coupled Newton step and two residual corrections (LLVM's fp64 sqrt without the denormal scaling),
//                   zero and +inf passed through; negative and NaN give NaN.  Arguments are sums of squares of speeds in m/s and
//                   variances of winds: never denormal.
//   qsqrt_pos(a)    the same without the zero / inf select, for an argument already tested > 0 whose result is discarded otherwise
//                   (the Cholesky pivots).
template <typename R> __device__ __forceinline__ R qdiv_nz(R a, R b) { return a / b; }
template <typename R> __device__ __forceinline__ R qsqrt(R a) { return sqrt(a); }
template <typename R> __device__ __forceinline__ R qsqrt_pos(R a) { return sqrt(a); }
#if TCR_FAST_DIV
template <> __device__ __forceinline__ double qdiv_nz<double>(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = a * r;
    const double res = __builtin_fma(-b, q, a);
    return __builtin_fma(res, r, q);
}
#endif
#if TCR_FAST_SQRT
template <> __device__ __forceinline__ double qsqrt_pos<double>(double a)
{
    const double y = __builtin_amdgcn_rsq(a);
    const double g0 = a * y, h0 = 0.5 * y;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, a);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, a);
    return __builtin_fma(d1, h1, g2);
}
template <> __device__ __forceinline__ double qsqrt<double>(double a)
{
    const double g = qsqrt_pos<double>(a);
    return (a == 0.0 || a == __builtin_inf()) ? a : g;      // rsq(0) = inf, rsq(inf) = 0: the products above are NaN
}
#endif

// The two powers of the path — t_strat ** -0.4 (coupled_fast.py:91, every evaluation) and err ** -0.2 (rk.py:160, every attempt) —
// are both a ** (-1/5): of t_strat ** 2 and of err.  ocml's pow is a general double-double log / exp (~220 VALU instructions);
// a ** (-1/5) is the root of f(y) = y ** -5 - a, whose Newton step y <- y + 0.2 y (1 - a y ** 5) needs six multiply-adds and
// converges quadratically: seeded with the fp32 hardware exp2(-0.2 log2(a)) (relative error ~1e-6), two steps reach fp64 round-off
// (tests/test_gpu_parity.py::test_integrator_arithmetic_helpers_against_numpy: against 80-bit pow with the reference's double
// exponents, 400 000 arguments over twenty decades; ocml / glibc: 0.6 ulp).  Domain: a within the fp32
// range (t_strat between 1e-19 and 1e19 K / 100 m); a <= 0, inf and NaN give NaN or the limit value like pow, except a = 0 -> NaN
// (not inf) — callers select on `t_strat == 0` / `err == 0` before they use the value, as the reference does.
template <typename R> __device__ __forceinline__ R inv_fifth_root(R a) { return pow(a, R(-0.2)); }
template <typename R> __device__ __forceinline__ R strat_pow(R g) { return pow(g, R(-0.4)); }
#if TCR_FAST_POW
template <> __device__ __forceinline__ double inv_fifth_root<double>(double a)
{
    const float L = __builtin_amdgcn_logf((float)a);                          // v_log_f32 is log2, v_exp_f32 is exp2
    double y = (double)__builtin_amdgcn_exp2f(-0.2f * L);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y, y4 = y2 * y2, y5 = y4 * y;
        const double r = __builtin_fma(-a, y5, 1.0);
        y = __builtin_fma(y * 0.2, r, y);
    }
    // What the reference raises to is the DOUBLE -0.2 = -(1/5 + 1.11e-17) (and -0.4 = -(2/5 + 2.22e-17) of a = t_strat ** 2: the same
    // factor per unit of log2 a): a ** (-0.2_double) = a ** (-1/5) x (1 - 1.11e-17 ln 2 log2 a), a correction of up to 3e-16 over the
    // twenty decades an error norm spans — applied with the fp32 logarithm of the seed, which is far more than it needs.
    return __builtin_fma(y * -7.6954e-18, (double)L, y);
}
template <> __device__ __forceinline__ double strat_pow<double>(double g)
{
    const double y = inv_fifth_root<double>(g * g);
    return g < 0.0 ? __builtin_nan("") : y;         // a negative base to a fractional power is NaN in NumPy (-> dv/dt = 0, coupled_fast.py:150)
}
#endif

// cos(lat pi / 180) for a latitude on the sphere: the argument is within [-pi / 2, pi / 2], so no range reduction is needed and
// the Taylor polynomial to x ** 22 is below fp64 round-off in absolute terms (relative: 0.3 ulp on average, at most 1.2 ulp equatorward
// of 60 degrees and 4.4 ulp at 80 degrees, where cos = 0.17; the track stops there, bam_track.py:134).  ocml's cos spends ~150
// instructions, most of them on arguments this path never has.
template <typename R> __device__ __forceinline__ R cos_lat(R x) { return cos(x); }
#if TCR_FAST_COS
template <> __device__ __forceinline__ double cos_lat<double>(double x)
{
    const double z = x * x;
    double p = -1.0 / 1124000727777607680000.0;                   // -1 / 22!
    p = __builtin_fma(p, z, 1.0 / 2432902008176640000.0);         //  1 / 20!
    p = __builtin_fma(p, z, -1.0 / 6402373705728000.0);           // -1 / 18!
    p = __builtin_fma(p, z, 1.0 / 20922789888000.0);              //  1 / 16!
    p = __builtin_fma(p, z, -1.0 / 87178291200.0);                // -1 / 14!
    p = __builtin_fma(p, z, 1.0 / 479001600.0);                   //  1 / 12!
    p = __builtin_fma(p, z, -1.0 / 3628800.0);                    // -1 / 10!
    p = __builtin_fma(p, z, 1.0 / 40320.0);                       //  1 / 8!
    p = __builtin_fma(p, z, -1.0 / 720.0);                        // -1 / 6!
    p = __builtin_fma(p, z, 1.0 / 24.0);                          //  1 / 4!
    p = __builtin_fma(p, z, -0.5);
    return __builtin_fma(p, z, 1.0);
}
#endif

// fpbisp.f (clamp + interval search) and fpbspl.f (k = 1 weights).
// Affine axes (ERA5's 1 deg / 0.25 deg grids, CMIP regular grids) need no memory
// traffic at all: the host has verified that x0 + i*dx reproduces every knot
// bit for bit and that 1/(x[i+1]-x[i]) is the same number in every cell.
template <typename R, bool AFFINE>
__device__ __forceinline__ CellT<R> locate_t(const AxisT<R> &A, R arg)
{
    TCR_FP_EXACT
    const R ax0 = RD(A.x0), axn = RD(A.xn);
    arg = (arg < ax0) ? ax0 : arg;
    arg = (arg > axn) ? axn : arg;
    const int n = RD(A.n);
    int i = (int)((arg - ax0) * RD(A.inv_step));
    i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
    CellT<R> c;
    if (AFFINE) {
        const R dx = RD(A.dx), x0 = ax0, rdx = RD(A.rdx);
        const R xl = x0 + (R)i * dx, xr = x0 + (R)(i + 1) * dx;
        const bool up = (i < n - 2) && (arg >= xr);
        const bool dn = !up && (i > 0) && (arg < xl);
        i += up ? 1 : (dn ? -1 : 0);
        const R xl2 = up ? xr : (dn ? x0 + (R)i * dx : xl);
        const R xr2 = up ? x0 + (R)(i + 1) * dx : (dn ? xl : xr);
        c.i = i;
        c.w0 = R(0.0) + rdx * (xr2 - arg);
        c.w1 = rdx * (arg - xl2);
    } else {
        const R *__restrict__ x = RD(A.x);
        while (i < n - 2 && arg >= x[i + 1]) ++i;
        while (i > 0 && arg < x[i]) --i;
        const R f = RD(A.rx)[i];
        c.i = i;
        c.w0 = R(0.0) + f * (x[i + 1] - arg);
        c.w1 = f * (arg - x[i]);
    }
    return c;
}

__device__ __forceinline__ Cell locate(const DevAxis &A, double arg)
{
    return RD(A.affine) ? locate_t<double, true>(A, arg) : locate_t<double, false>(A, arg);
}

// The four corners of NF interleaved fields as vector gathers of L elements (fp64: L = 2, 16 bytes;
// fp32: L = 4 for the wind / thermo lines, 2 for land + bathymetry) ...
template <typename R, int NF, int L>
struct CornersT {
    typedef typename VecT<R, L>::type V;
    static constexpr int NV = (NF + L - 1) / L;
    V c00[NV], c01[NV], c10[NV], c11[NV];
};

template <typename R, int NF, int STRIDE, int L>
__device__ __forceinline__ void gather(const R *__restrict__ base, int nlon, const CellT<R> &cx,
                                       const CellT<R> &cy, CornersT<R, NF, L> &C)
{
    const R *p00 = base + ((size_t)cy.i * nlon + cx.i) * STRIDE;
    const R *p01 = p00 + (size_t)nlon * STRIDE;
#pragma unroll
    for (int k = 0; k < CornersT<R, NF, L>::NV; ++k) {
        C.c00[k] = ldg<R, L>(p00 + L * k); C.c01[k] = ldg<R, L>(p01 + L * k);
        C.c10[k] = ldg<R, L>(p00 + STRIDE + L * k); C.c11[k] = ldg<R, L>(p01 + STRIDE + L * k);
    }
}

// ... and their bilinear sums in fpbisp.f's order: (x0,y0), (x0,y1), (x1,y0), (x1,y1),
// each term (c*hx)*hy.  FUSED (wind means and covariances inside fun(t, y) only — fields no discrete decision reads): the
// same terms in the same order, each accumulated with one fma.
template <typename R, int NF, int L, bool FUSED = false>
__device__ __forceinline__ void blend(const CornersT<R, NF, L> &C, const CellT<R> &cx, const CellT<R> &cy, R (&out)[NF])
{
    if (FUSED && TCR_FUSE_RHS) {
        TCR_FP_FUSE
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            R sp = C.c00[f / L][f % L] * cx.w0 * cy.w0;
            sp = sp + C.c01[f / L][f % L] * cx.w0 * cy.w1;
            sp = sp + C.c10[f / L][f % L] * cx.w1 * cy.w0;
            sp = sp + C.c11[f / L][f % L] * cx.w1 * cy.w1;
            out[f] = sp;
        }
    } else {
        TCR_FP_EXACT
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const R a = C.c00[f / L][f % L];
            const R b = C.c01[f / L][f % L];
            const R c = C.c10[f / L][f % L];
            const R d = C.c11[f / L][f % L];
            R sp = R(0.0);
            sp = sp + a * cx.w0 * cy.w0;
            sp = sp + b * cx.w0 * cy.w1;
            sp = sp + c * cx.w1 * cy.w0;
            sp = sp + d * cx.w1 * cy.w1;
            out[f] = sp;
        }
    }
}

// fp64 one-shot lookup (seeding)
template <int NF, int STRIDE>
__device__ __forceinline__ void bilinear(const double *__restrict__ base, int nlon, const Cell &cx,
                                         const Cell &cy, double (&out)[NF])
{
    CornersT<double, NF, 2> C;
    gather<double, NF, STRIDE, 2>(base, nlon, cx, cy, C);
    blend<double, NF, 2>(C, cx, cy, out);
}

// gather widths (elements per load) of the wind / thermo / static / forcing-table reads
template <typename R> struct Widths;
template <> struct Widths<double> { static constexpr int W = 2, T = 2, H = 2, F = 2; };
template <> struct Widths<float> { static constexpr int W = 4, T = 4, H = 2, F = 4; };

__device__ __forceinline__ double ts_at(const tcr_params &P, int i)
{
    // np.linspace(0, total_time, n_steps)[i]
    return (i == P.n_steps - 1) ? P.total_time : (double)i * (P.total_time / (double)(P.n_steps - 1));
}

template <typename R>
__device__ __forceinline__ double ts_k(const EvalKT<R> &K, int i)
{
    return (i == RD(K.n_steps) - 1) ? RD(K.total_time) : (double)i * RD(K.tstep);
}

// interp1d(t_s, Fs, axis=1)(t) (scipy/interpolate/_interpolate.py:457-486), split into the
// bracket search (searchsorted side='left', clipped to [1, n-1]; branch-free: the guess from
// t * (1/step) is at most one off) and the blend, so the gather sits with the others.
struct FsBracket {
    int lo;
    double x_lo, dx;
};

template <typename R>
__device__ __forceinline__ FsBracket fs_bracket(const EvalKT<R> &K, double t)
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

// the two bracketing samples of the four series: fs[sample][4]
template <typename R>
struct FsPairT {
    static constexpr int L = Widths<R>::F;
    typedef typename VecT<R, L>::type V;
    V a[4 / L], b[4 / L];
};

template <typename R>
__device__ __forceinline__ void fs_gather(const R *__restrict__ fs, const FsBracket &b, FsPairT<R> &p)
{
    constexpr int L = FsPairT<R>::L;
    const R *q = fs + (size_t)b.lo * 4;
#pragma unroll
    for (int k = 0; k < 4 / L; ++k) { p.a[k] = ldg<R, L>(q + L * k); p.b[k] = ldg<R, L>(q + 4 + L * k); }
}

template <typename R>
__device__ __forceinline__ void fs_blend(const FsPairT<R> &p, const FsBracket &b, double t, R (&F)[4])
{
    TCR_FP_FUSE
    constexpr int L = FsPairT<R>::L;
    const R dt = (R)(t - b.x_lo), dx = (R)b.dx;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const R ya = p.a[s / L][s % L], yb = p.b[s / L][s % L];
        F[s] = qdiv_nz<R>(yb - ya, dx) * dt + ya;
    }
}

// track/bam_track.py:116-128 after the 14 lookups: mean + chol(cov) · F(t).  Cholesky as
// LAPACK dpotrf('L') unblocked: ajj = a_jj - dot; fail if ajj <= 0 (-> zero winds, the
// LinAlgError branch :124-126); sub-column scaled by 1/ajj.  NaN lon/t -> zeros (:117-118).
template <typename R>
__device__ __forceinline__ void winds_from_lookups(const R (&q)[14], const R (&F)[4], R lon, double t, R (&w)[4])
{
    TCR_FP_FUSE
    // packed lower triangle: q[4]=a00 q[5]=a10 q[6]=a11 q[7]=a20 q[8]=a21 q[9]=a22 q[10]=a30 q[11]=a31 q[12]=a32 q[13]=a33
    const R z = R(0.0), one = R(1.0);
    bool ok = !(lon != lon) && !(t != t);
    R l00, l10, l20, l30, l11, l21, l31, l22, l32, l33;
    {
        const R ajj = q[4] - z;
        ok = ok && (ajj > z);
        l00 = qsqrt_pos<R>(ajj);
        const R r = qdiv_nz<R>(one, l00);
        l10 = (q[5] - z) * r; l20 = (q[7] - z) * r; l30 = (q[10] - z) * r;
    }
    {
        const R ajj = q[6] - (z + l10 * l10);
        ok = ok && (ajj > z);
        l11 = qsqrt_pos<R>(ajj);
        const R r = qdiv_nz<R>(one, l11);
        l21 = (q[8] - (z + l20 * l10)) * r;
        l31 = (q[11] - (z + l30 * l10)) * r;
    }
    {
        const R ajj = q[9] - ((z + l20 * l20) + l21 * l21);
        ok = ok && (ajj > z);
        l22 = qsqrt_pos<R>(ajj);
        const R r = qdiv_nz<R>(one, l22);
        l32 = (q[12] - ((z + l30 * l20) + l31 * l21)) * r;
    }
    {
        const R ajj = q[13] - (((z + l30 * l30) + l31 * l31) + l32 * l32);
        ok = ok && (ajj > z);
        l33 = qsqrt_pos<R>(ajj);
    }
    const R w0 = q[0] + ((((z + l00 * F[0]) + z * F[1]) + z * F[2]) + z * F[3]);
    const R w1 = q[1] + ((((z + l10 * F[0]) + l11 * F[1]) + z * F[2]) + z * F[3]);
    const R w2 = q[2] + ((((z + l20 * F[0]) + l21 * F[1]) + l22 * F[2]) + z * F[3]);
    const R w3 = q[3] + ((((z + l30 * F[0]) + l31 * F[1]) + l32 * F[2]) + l33 * F[3]);
    w[0] = ok ? w0 : z; w[1] = ok ? w1 : z; w[2] = ok ? w2 : z; w[3] = ok ? w3 : z;
}

// _env_winds(lon, lat, t) on its own (output samples, probes)
template <typename R, bool AFFINE>
__device__ __forceinline__ void env_winds(const EvalKT<R> &K, const R *__restrict__ wind, const R *__restrict__ fs,
                                          R lon, R lat, double t, R (&w)[4])
{
    const CellT<R> cx = locate_t<R, AFFINE>(K.wx, lon);
    const CellT<R> cy = locate_t<R, AFFINE>(K.wy, lat);
    const FsBracket fb = fs_bracket(K, t);
    CornersT<R, 14, Widths<R>::W> CW;
    FsPairT<R> fp;
    gather<R, 14, kWindStride, Widths<R>::W>(wind, RD(K.wx.n), cx, cy, CW);
    fs_gather<R>(fs, fb, fp);
    R q[14], F[4];
    blend<R, 14, Widths<R>::W, true>(CW, cx, cy, q);
    fs_blend<R>(fp, fb, t, F);
    winds_from_lookups<R>(q, F, lon, t, w);
}

template <typename R> __device__ __forceinline__ R sign_of(R x) { return (R)((x > R(0)) - (x < R(0))); }
template <typename R> __device__ __forceinline__ R np_min(R a, R b) { return (a != a) ? a : (a < b ? a : b); }
template <typename R> __device__ __forceinline__ R np_max(R a, R b) { return (a != a) ? a : (a > b ? a : b); }

template <typename R>
struct RhsT {
    R d[4];     // d lon/dt, d lat/dt, dv/dt, dm/dt
    R w[4];     // raw env winds at the point (what _env_winds returns)
    R alpha;    // ocean feedback (probe only)
    R vpot, chi;   // what the ventilation gate needs besides the raw winds (coupled_fast.py:238-244)
    int dec;    // decision probe (tests only; dead code elsewhere): bit0 `land == 1`, bit1 PI != 0, bit2 |land - 1| <= 1e-12
};

// dydt after the lookups, in two parts so that the integrator can start the next stage point's gathers between them.
// Part 1 — what the *position* derivatives need: steering coefficients, beta-advection (bam_track.py:131-144).
template <typename R>
struct TrackMidT {
    R vb0, vb1;     // translation velocity (zero poleward of 80 deg)
};

template <typename R>
__device__ __forceinline__ void rhs_track(const EvalKT<R> &K, R lat, R v, RhsT<R> &r, TrackMidT<R> &mid)
{
    TCR_FP_FUSE
    const R z = R(0.0);
    // steering coefficients
    R c0, c1;
    {
        R a0 = (v * R(1.94384)) * RD(K.m_alpha[0]) + RD(K.y_alpha[0]);
        R a1 = (v * R(1.94384)) * RD(K.m_alpha[1]) + RD(K.y_alpha[1]);
        a0 = np_max(np_min(a0, RD(K.alpha_max[0])), RD(K.alpha_min[0]));
        a1 = np_max(np_min(a1, RD(K.alpha_max[1])), RD(K.alpha_min[1]));
        const bool bad = (a0 != a0) || (a1 != a1);
        a0 = bad ? RD(K.y_alpha[0]) : a0;
        a1 = bad ? RD(K.y_alpha[1]) : a1;
        c0 = RD(K.coupled_track) ? a0 : RD(K.steering_coefs[0]);
        c1 = RD(K.coupled_track) ? a1 : RD(K.steering_coefs[1]);
    }
    // beta-advection; |lat| >= 80 -> zero motion and zero winds (bam_track.py:134-135)
    const bool polar = fabs(lat) >= R(80);
    const R w0 = polar ? z : r.w[0], w1 = polar ? z : r.w[1];
    const R w2 = polar ? z : r.w[2], w3 = polar ? z : r.w[3];
    const R cl = cos_lat<R>(lat * R(kPi / 180.0));                     // np.deg2rad
    R vb0 = (w0 * c0 + w2 * c1) + RD(K.u_beta) * cl;
    R vb1 = (w1 * c0 + w3 * c1) + (sign_of(lat) * RD(K.v_beta)) * cl;
    vb0 = polar ? z : vb0;
    vb1 = polar ? z : vb1;
    // bam_track.py / coupled_fast.py:203-204: dlon/dt = v / R * 180 / pi / cos(lat * pi / 180), dlat/dt = v / R * 180 / pi
#if TCR_SHARE_COS
    // the factor 180 / (pi R) is one host-computed constant and cos(lat pi / 180) is the beta-drift's cos(deg2rad(lat)): the two
    // arguments differ by at most one rounding of lat * pi / 180 (2 ulp of the cosine at 80 deg)
    r.d[0] = polar ? z : qdiv_nz<R>(vb0 * RD(K.deg_per_m), cl);
    r.d[1] = vb1 * RD(K.deg_per_m);
#else
    r.d[0] = qdiv_nz<R>(qdiv_nz<R>(qdiv_nz<R>(vb0, RD(K.earth_R)) * R(180.), R(kPi)), cos(qdiv_nz<R>(lat * R(kPi), R(180.))));
    r.d[1] = qdiv_nz<R>(qdiv_nz<R>(vb1, RD(K.earth_R)) * R(180.), R(kPi));
#endif
    mid.vb0 = vb0; mid.vb1 = vb1;
}

// Part 2 — intensity and moisture: _dvdt (coupled_fast.py:141-150) with _get_current_vpot (:54-58),
// _calc_alpha / _calc_z (:65-94), _dmdt (:175-180).  ck_h = 0.5 * Ck / h_bl, the common prefix of both expressions
// (:148, :179): constant over a storm's life, so the integrator divides once per storm (ck_over_h), not per evaluation.
template <typename R>
__device__ __forceinline__ R ck_over_h(const EvalKT<R> &K, R h_bl)
{
    TCR_FP_EXACT
    return R(0.5) * RD(K.Ck) / h_bl;
}

template <typename R>
__device__ __forceinline__ void rhs_intensity(const EvalKT<R> &K, R ck_h, R lat, R v, R m, const R (&th)[4],
                                              const R (&lb)[2], const TrackMidT<R> &mid, RhsT<R> &r)
{
    TCR_FP_FUSE
    const R z = R(0.0);
    const bool polar = fabs(lat) >= R(80);
    const R w0 = polar ? z : r.w[0], w1 = polar ? z : r.w[1];
    const R w2 = polar ? z : r.w[2], w3 = polar ? z : r.w[3];
    const R vb0 = mid.vb0, vb1 = mid.vb1;
    const R vp = (lb[0] == R(1.0)) ? z : th[0];                        // coupled_fast.py:35-58
    r.dec = (lb[0] == R(1.0) ? 1 : 0) | (th[0] != z ? 2 : 0) | (fabs(lb[0] - R(1.0)) <= R(1e-12) ? 4 : 0);
    const R h_m = th[2], gam = th[3], bathy = lb[1];
    const bool no_mix = (bathy >= z) || (-h_m <= bathy) || (gam == z);
    const R uT = qsqrt<R>(vb0 * vb0 + vb1 * vb1);
    const R zz = R(0.01) * strat_pow<R>(gam) * h_m * uT * vp / v;
    const R zc = np_min(np_max(zz, z), R(100.0));
    const R al = no_mix ? R(1.0) : R(1) - R(0.87) * exp(-zc);
    r.alpha = al;
    const R beta = R(1) - RD(K.epsilon) - RD(K.kappa);
    const R gamma = RD(K.epsilon) + al * RD(K.kappa);
    const R m3 = m * m * m;
    const R dv = ck_h * (al * beta * (vp * vp) * m3 - (R(1) - gamma * m3) * (v * v));
    const R du = w0 - w2, dw = w1 - w3;
    const R venti = qsqrt<R>(du * du + dw * dw) * th[1];
    r.vpot = vp; r.chi = th[1];
    r.d[2] = (dv != dv) ? z : dv;
    r.d[3] = ck_h * ((R(1) - m) * v - venti * m);
}

template <typename R>
__device__ __forceinline__ void rhs_tail(const EvalKT<R> &K, R h_bl, R lat, R v, R m,
                                         const R (&th)[4], const R (&lb)[2], RhsT<R> &r)
{
    TrackMidT<R> mid;
    rhs_track<R>(K, lat, v, r, mid);
    rhs_intensity<R>(K, ck_over_h<R>(K, h_bl), lat, v, m, th, lb, mid, r);
}

// loads of 2 / 4 / 8 bytes at the natural alignment of the *element* (1 / 2 / 4 bytes), i.e. possibly straddling the
// load's own size: gfx950 under HSA runs with unaligned global access enabled, and the compiler (unaligned-access-mode)
// emits one global_load_ushort / _dword / _dwordx2 for them
__device__ __forceinline__ uint32_t ldg_u16x1_pair_u8(const uint8_t *p)
{
    typedef uint16_t u16_a1 __attribute__((aligned(1)));
    return *(const __attribute__((address_space(1))) u16_a1 *)(p);
}
__device__ __forceinline__ uint32_t ldg_u32_pair_u16(const uint16_t *p)
{
    typedef uint32_t u32_a2 __attribute__((aligned(2)));
    return *(const __attribute__((address_space(1))) u32_a2 *)(p);
}
__device__ __forceinline__ uint4 ldg_pair_u64(const uint64_t *p)
{
    typedef uint32_t u4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    const u4_a8 v = *(const __attribute__((address_space(1))) u4_a8 *)(p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float2 ldg_pair_f32(const float *p)
{
    typedef float f2_a4 __attribute__((ext_vector_type(2), aligned(4)));
    const f2_a4 v = *(const __attribute__((address_space(1))) f2_a4 *)(p);
    return make_float2(v[0], v[1]);
}

// bilinear sum of four corner values in fpbisp.f's order (the body of blend for one field)
template <typename R>
__device__ __forceinline__ R blend4(R a, R b, R c, R d, const CellT<R> &cx, const CellT<R> &cy)
{
    R sp = R(0.0);
    sp = sp + a * cx.w0 * cy.w0;
    sp = sp + b * cx.w0 * cy.w1;
    sp = sp + c * cx.w1 * cy.w0;
    sp = sp + d * cx.w1 * cy.w1;
    return sp;
}

// land and bathymetry at (lon, lat) in the storage of static mode SM (EvalKT): the gathers are issued by issue(), the values
// — widened to R, then blended in FITPACK's order — come out of finish()
template <typename R, bool AFFINE, int SM>
struct StaticLookup {
    CellT<R> hx, hy, bx, by;
    CornersT<R, 2, Widths<R>::H> CH;        // kStatF64: (land, bathy) pairs
    CornersT<R, 1, 1> CL, CB;               // kStatF64Split
    uint32_t p0, p1;                        // kStatPack16: rows y0 / y1, elements (x0, x1); kStatU8F32: the land bytes likewise
    float2 b0, b1;                          // kStatU8F32: bathymetry rows y0 / y1
    uint4 q0, q1;                           // kStatPack64: rows y0 / y1, (bathymetry bits, land) of x0 then of x1
    __device__ __forceinline__ void issue(const EvalKT<R> &K, R lon, R lat)
    {
        hx = locate_t<R, AFFINE>(K.hx, lon); hy = locate_t<R, AFFINE>(K.hy, lat);
        if (SM == kStatF64Split) {
            bx = locate_t<R, AFFINE>(K.bx, lon); by = locate_t<R, AFFINE>(K.by, lat);
            gather<R, 1, 1, 1>(reinterpret_cast<const R *>(RD(K.stat)), RD(K.hx.n), hx, hy, CL);
            gather<R, 1, 1, 1>(reinterpret_cast<const R *>(RD(K.bathy)), RD(K.bx.n), bx, by, CB);
        } else if (SM == kStatPack16) {
            const int nl = RD(K.hx.n);
            const uint16_t *q = reinterpret_cast<const uint16_t *>(RD(K.stat)) + ((size_t)hy.i * nl + hx.i);
            p0 = ldg_u32_pair_u16(q); p1 = ldg_u32_pair_u16(q + nl);
        } else if (SM == kStatPack64) {
            const int nl = RD(K.hx.n);
            const uint64_t *q = reinterpret_cast<const uint64_t *>(RD(K.stat)) + ((size_t)hy.i * nl + hx.i);
            q0 = ldg_pair_u64(q); q1 = ldg_pair_u64(q + nl);
        } else if (SM == kStatU8F32) {
            bx = hx; by = hy;
            if (!RD(K.hb_same)) { bx = locate_t<R, AFFINE>(K.bx, lon); by = locate_t<R, AFFINE>(K.by, lat); }      // wave-uniform
            const int nl = RD(K.hx.n), nb = RD(K.bx.n);
            const uint8_t *q = reinterpret_cast<const uint8_t *>(RD(K.stat)) + ((size_t)hy.i * nl + hx.i);
            const float *f = reinterpret_cast<const float *>(RD(K.bathy)) + ((size_t)by.i * nb + bx.i);
            p0 = ldg_u16x1_pair_u8(q); p1 = ldg_u16x1_pair_u8(q + nl);
            b0 = ldg_pair_f32(f); b1 = ldg_pair_f32(f + nb);
        } else {
            gather<R, 2, kStaticStride, Widths<R>::H>(reinterpret_cast<const R *>(RD(K.stat)), RD(K.hx.n), hx, hy, CH);
        }
    }
    __device__ __forceinline__ void finish(R (&lb)[2]) const
    {
        if (SM == kStatF64Split) {
            R a[1], b[1];
            blend<R, 1, 1>(CL, hx, hy, a);
            blend<R, 1, 1>(CB, bx, by, b);
            lb[0] = a[0]; lb[1] = b[0];
        } else if (SM == kStatPack16) {
            // corners (x0,y0) = low half of row y0, (x1,y0) = its high half; land = bit 0, bathymetry = the rest minus the bias
            const uint32_t c00 = p0 & 0xffffu, c10 = p0 >> 16, c01 = p1 & 0xffffu, c11 = p1 >> 16;
            lb[0] = blend4<R>((R)(int)(c00 & 1u), (R)(int)(c01 & 1u), (R)(int)(c10 & 1u), (R)(int)(c11 & 1u), hx, hy);
            lb[1] = blend4<R>((R)((int)(c00 >> 1) - kPack16Bias), (R)((int)(c01 >> 1) - kPack16Bias),
                              (R)((int)(c10 >> 1) - kPack16Bias), (R)((int)(c11 >> 1) - kPack16Bias), hx, hy);
        } else if (SM == kStatPack64) {
            lb[0] = blend4<R>((R)(int)(q0.y & 0xffu), (R)(int)(q1.y & 0xffu), (R)(int)(q0.w & 0xffu), (R)(int)(q1.w & 0xffu), hx, hy);
            lb[1] = blend4<R>((R)__uint_as_float(q0.x), (R)__uint_as_float(q1.x), (R)__uint_as_float(q0.z), (R)__uint_as_float(q1.z), hx, hy);
        } else if (SM == kStatU8F32) {
            lb[0] = blend4<R>((R)(int)(p0 & 0xffu), (R)(int)(p1 & 0xffu), (R)(int)((p0 >> 8) & 0xffu), (R)(int)((p1 >> 8) & 0xffu), hx, hy);
            lb[1] = blend4<R>((R)b0.x, (R)b1.x, (R)b0.y, (R)b1.y, bx, by);
        } else {
            blend<R, 2, Widths<R>::H>(CH, hx, hy, lb);
        }
    }
};

// fun(t, y) = Coupled_FAST.dydt (coupled_fast.py:196-207): _calc_steering_coefs (:183-192),
// _step_bam_track (bam_track.py:131-144) on _env_winds (:116-128), _dvdt (:141-150) with
// _get_current_vpot (:54-58), _calc_alpha/_calc_z (:65-94), _dmdt (:175-180).
template <typename R, bool AFFINE, int SM>
__device__ __forceinline__ RhsT<R> rhs_eval(const EvalKT<R> &K, const R *__restrict__ wind, const R *__restrict__ thermo,
                                            const R *__restrict__ fs, R h_bl, double t, R lon, R lat, R v, R m)
{
    typedef Widths<R> Wd;
    // ---- address generation: pure ALU on affine grids
    const CellT<R> wx = locate_t<R, AFFINE>(K.wx, lon), wy = locate_t<R, AFFINE>(K.wy, lat);
    CellT<R> tx = wx, ty = wy;
    if (!RD(K.tw_same)) { tx = locate_t<R, AFFINE>(K.tx, lon); ty = locate_t<R, AFFINE>(K.ty, lat); }      // wave-uniform
    const FsBracket fb = fs_bracket(K, t);
    // ---- one round of independent gathers
    CornersT<R, 14, Wd::W> CW;
    CornersT<R, 4, Wd::T> CT;
    StaticLookup<R, AFFINE, SM> SL;
    FsPairT<R> fp;
    gather<R, 14, kWindStride, Wd::W>(wind, RD(K.wx.n), wx, wy, CW);
    fs_gather<R>(fs, fb, fp);
    gather<R, 4, kThermoStride, Wd::T>(thermo, RD(K.tx.n), tx, ty, CT);
    SL.issue(K, lon, lat);
    // ---- straight-line math
    RhsT<R> r;
    R q[14], F[4], th[4], lb[2];
    blend<R, 14, Wd::W, true>(CW, wx, wy, q);
    fs_blend<R>(fp, fb, t, F);
    winds_from_lookups<R>(q, F, lon, t, r.w);
    blend<R, 4, Wd::T>(CT, tx, ty, th);
    SL.finish(lb);
    rhs_tail<R>(K, h_bl, lat, v, m, th, lb, r);
    return r;
}

// Per-lane corner cache for the sequential integrator.  Consecutive RK stage points of a storm
// mostly fall into the same grid cell (a stage moves a storm by a fraction of a degree), and the
// cost of a gather was measured to be per lane-request: keeping the last cell's corner data in
// registers and re-gathering (exec-masked) only when a lane's cell changes cuts the requests per
// evaluation from 44 to ~15 without changing a single value.  The cache registers are the load
// destinations the direct path needs anyway.
template <typename R>
struct CornerCacheT {
    int wi, wj;                      // cached wind cell (-1 = empty)
    CornersT<R, 14, Widths<R>::W> CW;
};

template <typename R> __device__ __forceinline__ void cache_reset(CornerCacheT<R> &C) { C.wi = C.wj = -1; }

// One evaluation of fun for the sequential integrator, cut at the two points where its data dependencies allow the
// next stage point's memory round trip to start early (k_integrate):
//   issue      cell searches and every gather of the point (lon, lat, t) — needs nothing but the point;
//   track      raw env winds and d lon/dt, d lat/dt — all the *next* stage point depends on; also reduces the thermo /
//              static corners to their bilinear values, so the load registers are free for the next issue;
//   intensity  dv/dt, dm/dt — runs in the shadow of the next point's gathers.
// The operations and their order per value are those of rhs_eval; only independent work is reordered.
template <typename R, bool AFFINE, int SM>
struct RhsPipeT {
    typedef Widths<R> Wd;
    CellT<R> wx, wy, tx, ty;
    FsBracket fb;
    FsPairT<R> fp;
    CornersT<R, 4, Wd::T> CT;
    StaticLookup<R, AFFINE, SM> SL;

    __device__ __forceinline__ void issue(CornerCacheT<R> &C, const EvalKT<R> &K, const R *__restrict__ wind,
                                          const R *__restrict__ thermo, const R *__restrict__ fs, double t, R lon, R lat)
    {
        wx = locate_t<R, AFFINE>(K.wx, lon); wy = locate_t<R, AFFINE>(K.wy, lat);
        tx = wx; ty = wy;
        if (!RD(K.tw_same)) { tx = locate_t<R, AFFINE>(K.tx, lon); ty = locate_t<R, AFFINE>(K.ty, lat); }      // wave-uniform
        fb = fs_bracket(K, t);
#ifdef TCR_ABLATE_UNIFORM_ADDR
        // timing experiment only (values are wrong): every lane gathers at lane 0's cells — one cache line per load
        // instruction instead of up to 64 — which separates the per-lane cost of a gather from its per-line cost
        wx.i = __builtin_amdgcn_readfirstlane(wx.i); wy.i = __builtin_amdgcn_readfirstlane(wy.i);
        tx.i = __builtin_amdgcn_readfirstlane(tx.i); ty.i = __builtin_amdgcn_readfirstlane(ty.i);
        fb.lo = __builtin_amdgcn_readfirstlane(fb.lo);
        {
            auto first64 = [](unsigned long long v) {
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
                return ((unsigned long long)hi << 32) | lo;
            };
            fs = reinterpret_cast<const R *>(first64((unsigned long long)fs));
            wind = reinterpret_cast<const R *>(first64((unsigned long long)wind));
            thermo = reinterpret_cast<const R *>(first64((unsigned long long)thermo));
            lon = (R)__longlong_as_double((long long)first64((unsigned long long)__double_as_longlong((double)lon)));
            lat = (R)__longlong_as_double((long long)first64((unsigned long long)__double_as_longlong((double)lat)));
        }
#endif
#if defined(TCR_ABLATE_FS_READ) || defined(TCR_ABLATE_ALL_READS)
        // timing experiment of DESIGN.md §9 only (values are wrong): what the integrator would gain if the forcing-table
        // read cost nothing.  The bracket arithmetic stays; the gather becomes a cheap per-lane constant.
        for (int k = 0; k < 4 / FsPairT<R>::L; ++k) for (int q = 0; q < FsPairT<R>::L; ++q) { fp.a[k][q] = (R)(1e-3 * fb.lo); fp.b[k][q] = (R)(1e-3 * fb.lo); }
#else
        fs_gather<R>(fs, fb, fp);
#endif
#ifdef TCR_ABLATE_ALL_READS
        // timing experiment only (values are not physical): every gather of the evaluation replaced by per-lane constants
        {
            const R eps = (R)(1e-6 * wx.i);
            for (int f = 0; f < 14; ++f) {
                const R c = ((f == 4 || f == 6 || f == 9 || f == 13) ? R(10) : (f < 4 ? R(3) : R(0.1))) + eps;
                C.CW.c00[f / Wd::W][f % Wd::W] = c; C.CW.c01[f / Wd::W][f % Wd::W] = c;
                C.CW.c10[f / Wd::W][f % Wd::W] = c; C.CW.c11[f / Wd::W][f % Wd::W] = c;
            }
            const R tv[4] = {R(60), R(0.5), R(50), R(0.05)};
            for (int f = 0; f < 4; ++f) {
                const R c = tv[f] + eps;
                CT.c00[f / Wd::T][f % Wd::T] = c; CT.c01[f / Wd::T][f % Wd::T] = c;
                CT.c10[f / Wd::T][f % Wd::T] = c; CT.c11[f / Wd::T][f % Wd::T] = c;
            }
            SL.hx = locate_t<R, AFFINE>(K.hx, lon); SL.hy = locate_t<R, AFFINE>(K.hy, lat);
            const R sv[2] = {R(0), R(-3000)};
            for (int f = 0; f < 2; ++f) {
                const R c = sv[f] + eps;
                SL.CH.c00[f / Wd::H][f % Wd::H] = c; SL.CH.c01[f / Wd::H][f % Wd::H] = c;
                SL.CH.c10[f / Wd::H][f % Wd::H] = c; SL.CH.c11[f / Wd::H][f % Wd::H] = c;
            }
        }
#else
        if (wx.i != C.wi || wy.i != C.wj) {
            gather<R, 14, kWindStride, Wd::W>(wind, RD(K.wx.n), wx, wy, C.CW);
            C.wi = wx.i; C.wj = wy.i;
        }
        gather<R, 4, kThermoStride, Wd::T>(thermo, RD(K.tx.n), tx, ty, CT);
        SL.issue(K, lon, lat);
#endif
    }

    __device__ __forceinline__ void track(const CornerCacheT<R> &C, const EvalKT<R> &K, double t, R lon, R lat, R v,
                                          RhsT<R> &r, TrackMidT<R> &mid, R (&th)[4], R (&lb)[2]) const
    {
        R q[14], F[4];
        blend<R, 14, Wd::W, true>(C.CW, wx, wy, q);
        fs_blend<R>(fp, fb, t, F);
        winds_from_lookups<R>(q, F, lon, t, r.w);
        blend<R, 4, Wd::T>(CT, tx, ty, th);
        SL.finish(lb);
        rhs_track<R>(K, lat, v, r, mid);
    }
};

// coupled_fast.py:246-256 with util/basins.py:32-37 (dx = 1); always >= 0
template <typename R>
__device__ __forceinline__ R event_fn(const tcr_params &P, R lon, R lat, R v)
{
    const bool inside = ((R)P.box[0] + R(1) < lon) && (lon < (R)P.box[2] - R(1)) &&
                        ((R)P.box[1] + R(1) < lat) && (lat < (R)P.box[3] - R(1));
    if (!inside) return R(0.0);
    if (fabs(lat) <= R(2)) return R(0.0);
    const R g = v - (R)P.v_dissipate;
    return g > R(0) ? g : R(0.0);
}

}  // namespace tcr
