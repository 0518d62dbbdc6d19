// HIP kernels of the storm hot path (gfx950, fp64).
//
// The hot path is split where its parallelism changes:
//   k_integrate  the inherently sequential part of a storm (adaptive RK45 steps up to the
//                terminal event): ~60 RHS evaluations per storm on average, one lane per
//                storm at a time, leaving one dense-output record per accepted step;
//   k_emit       everything that is independent per hourly output sample (2/3 of all field
//                evaluations): dense-output evaluation, env-wind recompute, vmax, accept
//                flags, NaN-padded planes — one thread per sample, full occupancy.
//
// k_integrate — persistent, one lane per storm *at a time*:
//   * every lane runs a small state machine whose only expensive state is "evaluate
//     fun(t, y)"; RK stages and the two evaluations of SciPy's initial-step heuristic
//     funnel through that one evaluation, so lanes of a wave stay converged on the
//     costly code whatever step / stage each storm is at;
//   * a lane whose storm ends (dissipation, basin exit, 15 days) pulls the next storm
//     from a device-wide queue (wave-aggregated atomic), which removes the 3x
//     lifetime imbalance between storms from the critical path;
//   * storm state lives in VGPRs, the 7 RK stage derivatives live in LDS,
//     lane-contiguous (conflict-free ds_read_b64); fields are gathered through L1/L2 from the interleaved HBM
//     layout of tcr_device.h with 16-B loads.
//
// The integrator is SciPy's RK45 restated per lane:
//   scipy/integrate/_ivp/rk.py:14-79 (rk_step), :111-176 (_step_impl), :293-420
//   (tableau), :552-574 (dense output); common.py:63-134 (norm, initial step);
//   ivp.py:654-725 (events at step end, t_eval emission) as called from
//   intensity/coupled_fast.py:264-266.
#include "tcr_device.h"

namespace tcr {

struct KArgs {
    tcr_params P;
    DevFields D;
    int64_t n;
    const double *lon0, *lat0, *v0, *m0, *h_bl;
    const int32_t *slot;
    const double *phases;    // [n][4][n_series]
    double *fs;              // [n][n_steps][4]
    double *srec;            // [n][max_rk_steps][kStepRec] accepted-step records
    int32_t *n_valid, *status, *nfev, *n_accept, *n_reject;
    unsigned long long *queue;   // next storm index to hand out (zeroed before the launch)
    int max_rk_steps;
};

// ---------------------------------------------------------------------------
// gen_f (track/bam_track.py:23-31), direct form: one thread per (storm, sample),
// 4 series x n_series sines, NumPy's evaluation order 2π·((n·t)/T + x).
// Used when the Fourier period is not a whole number of output intervals.
__global__ __launch_bounds__(256) void k_fourier_direct(tcr_params P, int64_t n,
                                                         const double *__restrict__ phases,
                                                         double *__restrict__ fs)
{
    const int ns = P.n_steps, N = P.n_series;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * ns) return;
    const int64_t storm = gid / ns;
    const int i = (int)(gid - storm * ns);
    const double t = ts_at(P, i);
    const double two_pi = 2. * kPi;
    const double *ph = phases + storm * 4 * N;
    double out[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        double acc = 0.0;
        for (int k = 0; k < N; ++k) {
            const double arg = two_pi * (((double)(k + 1) * t) / P.T_Fs + ph[s * N + k]);
            const double term = P.fs_wgt[k] * sin(arg);
            acc = (k == 0) ? term : acc + term;
        }
        out[s] = P.fs_amp * acc;
    }
    double2 *o = reinterpret_cast<double2 *>(fs + gid * 4);
    o[0] = make_double2(out[0], out[1]);
    o[1] = make_double2(out[2], out[3]);
}

// gen_f, periodic form.  With T_Fs = period * dt_out (the reference's defaults: 20 d
// and 1 h -> 480) the phase of harmonic n at output sample k is
//     2π·(n·k mod period)/period  +  2π·x_{s,n},
// so  sin(.) = S[j]·cos(2πx) + C[j]·sin(2πx)  with j = n·k mod period and (S, C) a
// host-computed (long-double accurate) table of one period.  Per storm that is 60
// sincospi instead of 21 660 sines.  It evaluates the same series the reference
// does; the values differ from NumPy's only by the rounding of NumPy's own argument
// 2π·(n t/T + x) (|Δ| ~ 1e-14, tests/test_gpu_parity.py states the bound).
// One workgroup per storm; thread = (sample, series) pairs, coalesced stores.
constexpr int kFsThreads = 128;

__global__ __launch_bounds__(kFsThreads) void k_fourier_periodic(tcr_params P, int64_t n, int period,
                                                                  const double2 *__restrict__ sc_table,
                                                                  const double *__restrict__ phases,
                                                                  double *__restrict__ fs)
{
    extern __shared__ double2 lds[];            // [period] table, then [4*N] (sin, cos)(2π x)
    const int N = P.n_series, ns = P.n_steps;
    double2 *tab = lds, *ph = lds + period;
    const int64_t storm = blockIdx.x;
    for (int j = threadIdx.x; j < period; j += kFsThreads) tab[j] = sc_table[j];
    if (threadIdx.x < 4 * N) {
        // amplitude-weighted phase factors: n^-1.5 * (sin, cos)(2π x)
        const double x = phases[storm * 4 * N + threadIdx.x];
        const double wgt = P.fs_wgt[threadIdx.x % N];
        ph[threadIdx.x] = make_double2(wgt * sinpi(2.0 * x), wgt * cospi(2.0 * x));
    }
    __syncthreads();
    double *out = fs + storm * ns * 4;
    // one thread = one output sample, all four series: the table entry of harmonic h is shared by
    // the four series (4 independent accumulators, 1 table read + 4 broadcast reads per harmonic)
    for (int k = threadIdx.x; k < ns; k += kFsThreads) {
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        int j = 0;                               // (n * k) mod period, built incrementally
        const int kk = k % period;
        for (int h = 0; h < N; ++h) {            // (fully unrolling this loop was measured 2x slower)
            j += kk; if (j >= period) j -= period;
            const double2 a = tab[j];
            const double2 b0 = ph[h], b1 = ph[N + h], b2 = ph[2 * N + h], b3 = ph[3 * N + h];
            // sin(A + B) = sinA cosB + cosA sinB, accumulated with explicit FMAs (2 per term)
            acc0 = fma(a.x, b0.y, fma(a.y, b0.x, acc0)); acc1 = fma(a.x, b1.y, fma(a.y, b1.x, acc1));
            acc2 = fma(a.x, b2.y, fma(a.y, b2.x, acc2)); acc3 = fma(a.x, b3.y, fma(a.y, b3.x, acc3));
        }
        double2 *o = reinterpret_cast<double2 *>(out + (size_t)k * 4);
        o[0] = make_double2(P.fs_amp * acc0, P.fs_amp * acc1);
        o[1] = make_double2(P.fs_amp * acc2, P.fs_amp * acc3);
    }
}

// ---------------------------------------------------------------------------
// Dormand–Prince tableau exactly as SciPy spells it (rk.py:384-408)
#define A10 (1. / 5)
#define A20 (3. / 40)
#define A21 (9. / 40)
#define A30 (44. / 45)
#define A31 (-56. / 15)
#define A32 (32. / 9)
#define A40 (19372. / 6561)
#define A41 (-25360. / 2187)
#define A42 (64448. / 6561)
#define A43 (-212. / 729)
#define A50 (9017. / 3168)
#define A51 (-355. / 33)
#define A52 (46732. / 5247)
#define A53 (49. / 176)
#define A54 (-5103. / 18656)
__device__ constexpr double RK_C[7] = {0, 1. / 5, 3. / 10, 4. / 5, 8. / 9, 1, 1};
__device__ constexpr double RK_B[6] = {35. / 384, 0, 500. / 1113, 125. / 192, -2187. / 6784, 11. / 84};
__device__ constexpr double RK_E[7] = {-71. / 57600, 0, 71. / 16695, -71. / 1920, 17253. / 339200, -22. / 525, 1. / 40};
__device__ constexpr double RK_P[7][4] = {
    {1, -8048581381. / 2820520608, 8663915743. / 2820520608, -12715105075. / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200. / 32700410799, -68118460800. / 10900136933, 87487479700. / 32700410799},
    {0, -1754552775. / 470086768, 14199869525. / 1410260304, -10690763975. / 1880347072},
    {0, 127303824393. / 49829197408, -318862633887. / 49829197408, 701980252875. / 199316789632},
    {0, -282668133. / 205662961, 2019193451. / 616988883, -1453857185. / 822651844},
    {0, 40617522. / 29380423, -110615467. / 29380423, 69997945. / 29380423}};

__device__ __forceinline__ double rms4(double a, double b, double c, double d)
{
    // np.linalg.norm(x) / x.size ** 0.5
    return sqrt(a * a + b * b + c * c + d * d) / 2.0;
}

// Number of t_eval samples <= t_emit: np.searchsorted(t_eval, t, side='right') (ivp.py:708)
__device__ __forceinline__ int samples_upto(const tcr_params &P, double t_emit)
{
    const int ns = P.n_steps;
    int k = (int)(t_emit / (P.total_time / (double)(ns - 1)));
    k = k < 0 ? 0 : (k > ns - 1 ? ns - 1 : k);
    while (k + 1 < ns && ts_at(P, k + 1) <= t_emit) ++k;
    while (k >= 0 && ts_at(P, k) > t_emit) --k;
    return k + 1;
}

constexpr int kWave = 64;
#ifndef TCR_INT_WPS
#define TCR_INT_WPS 1      // waves per SIMD the integrator is register-budgeted for
#endif
constexpr int kRunning = 99;

// k_integrate: the sequential part of a storm — RK45 steps until the terminal event.
//
// Wave-synchronous cycles: one cycle = one attempt of _step_impl for every lane = six
// evaluations of fun (stages 2..6 and f_new; stage 1 is FSAL).  All lanes evaluate the
// same stage in the same slot, so the per-stage bookkeeping is wave-uniform code instead
// of a divergent state machine.  A lane whose storm ends pulls the next one from the
// queue at the cycle boundary; a fresh storm spends slots 0 and 1 of its first cycle on
// the two evaluations of RungeKutta.__init__ (f0 and select_initial_step's f1) and idles
// for the other four.  Every accepted step leaves one kStepRec-double record
// (t_old, h, t_new, y_old, K[7][4]) from which k_emit evaluates the hourly samples.
template <bool AFFINE>
__global__ __launch_bounds__(kWave, TCR_INT_WPS) void k_integrate(KArgs a)
{
    // Kl[(stage*4 + component)*64 + lane]
    __shared__ double Kl[7 * 4 * kWave];
    __shared__ EvalK K;
    const tcr_params &P = a.P;
    const DevFields &D = a.D;
    const int lane = threadIdx.x;
    if (lane == 0) make_eval_k(P, D, K);
    __syncthreads();
    const int ns = P.n_steps;
    const double tb = P.total_time;
#define KS(j, i) Kl[((j) * 4 + (i)) * kWave + lane]

    // ---- per-lane storm state
    long long sid = -1;
    int status = kRunning, nfev = 0, nacc = 0, nrej = 0, next_out = 0;
    bool active = false, fresh = false, exhausted = false, rejected = false;
    DevSlot S{};
    const double *fs = nullptr;
    double *srec = nullptr;
    double h_bl = 0.0;
    double y[4] = {0, 0, 0, 0}, f[4] = {0, 0, 0, 0}, yn[4] = {0, 0, 0, 0};
    double e[5] = {0, 0, 0, 0, 0};          // evaluation point: t, lon, lat, v, m
    double t = 0, h = 0, ha = 0, h_abs = 0, t_new = 0, g = 0;
    CornerCache CC;
    cache_reset(CC);

    auto finalize = [&]() {
        a.n_valid[sid] = next_out;
        a.status[sid] = status;
        a.nfev[sid] = (status == TCR_STATUS_GATED) ? 0 : nfev;
        a.n_accept[sid] = nacc;
        a.n_reject[sid] = nrej;
        active = false;
    };
    // one attempt of _step_impl's while-loop: clip to t_bound, stage-2 input (rk.py:137-146, 62-66)
    auto attempt_setup = [&]() {
        const double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        if (ha < min_step) { status = TCR_STATUS_STEP_FAIL; finalize(); return; }
        h = ha;
        t_new = t + h;
        if (t_new - tb > 0) t_new = tb;
        h = t_new - t;
        ha = fabs(h);
        for (int i = 0; i < 4; ++i) {
            KS(0, i) = f[i];
            const double dy = 0.0 + f[i] * A10;
            e[1 + i] = y[i] + dy * h;
        }
        e[0] = t + RK_C[1] * h;
    };
    auto begin_step = [&]() {
        const double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        ha = h_abs;
        if (ha > P.max_step) ha = P.max_step;
        else if (ha < min_step) ha = min_step;
        rejected = false;
        attempt_setup();
    };

    for (;;) {
        // ---- cycle boundary: refill idle lanes from the storm queue (wave-aggregated atomic)
        const unsigned long long want = __ballot(!active && !exhausted);
        if (want) {
            const int leader = __ffsll((long long)want) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(a.queue, (unsigned long long)__popcll(want));
            base = __shfl(base, leader);
            if (!active && !exhausted) {
                sid = (long long)(base + (unsigned long long)__popcll(want & ((1ull << lane) - 1ull)));
                if (sid >= a.n) {
                    exhausted = true;
                } else {
                    S = D.slots[a.slot[sid]];
                    fs = a.fs + sid * ns * 4;
                    srec = a.srec + sid * (long long)a.max_rk_steps * kStepRec;
                    h_bl = a.h_bl[sid];
                    y[0] = a.lon0[sid]; y[1] = a.lat0[sid]; y[2] = a.v0[sid]; y[3] = a.m0[sid];
                    status = kRunning; nfev = 0; nacc = 0; nrej = 0; next_out = 0;
                    t = 0.0;
                    e[0] = 0.0; e[1] = y[0]; e[2] = y[1]; e[3] = y[2]; e[4] = y[3];
                    active = true; fresh = true;
                    cache_reset(CC);
                }
            }
        }
        if (!__ballot(active)) break;

        // ---- six evaluation slots
#pragma unroll 1
        for (int slot = 0; slot < 6; ++slot) {
            const bool live = active && !(fresh && slot >= 2);
            Rhs r{};
            if (live) r = rhs_eval_cached<AFFINE>(CC, K, S, fs, h_bl, e[0], e[1], e[2], e[3], e[4]);
            if (live && !fresh) {
                // rk_step (rk.py:62-70): K[s] = fun(...); next stage input dy = dot(K[:s].T, a[:s]) * h
                ++nfev;
                const int st = slot + 1;
                for (int i = 0; i < 4; ++i) KS(st, i) = r.d[i];
                if (slot < 5) {
                    for (int i = 0; i < 4; ++i) {
                        double dy = 0.0;
                        switch (slot) {
                        case 0: dy += KS(0, i) * A20; dy += KS(1, i) * A21; break;
                        case 1: dy += KS(0, i) * A30; dy += KS(1, i) * A31; dy += KS(2, i) * A32; break;
                        case 2: dy += KS(0, i) * A40; dy += KS(1, i) * A41; dy += KS(2, i) * A42; dy += KS(3, i) * A43; break;
                        case 3: dy += KS(0, i) * A50; dy += KS(1, i) * A51; dy += KS(2, i) * A52; dy += KS(3, i) * A53;
                                dy += KS(4, i) * A54; break;
                        default:
                            for (int j = 0; j < 6; ++j) dy += KS(j, i) * RK_B[j];
                            break;
                        }
                        if (slot < 4) e[1 + i] = y[i] + dy * h;
                        else { e[1 + i] = y[i] + h * dy; yn[i] = e[1 + i]; }     // y_new (rk.py:68)
                    }
                    e[0] = (slot < 4) ? t + RK_C[slot + 2] * h : t + h;
                } else {
                    // f_new is in K[6]: error estimate and step-size control (rk.py:147-165)
                    double er[4];
                    for (int i = 0; i < 4; ++i) {
                        const double sc = P.atol + fmax(fabs(y[i]), fabs(yn[i])) * P.rtol;
                        double acc = 0.0;
                        for (int j = 0; j < 7; ++j) acc += KS(j, i) * RK_E[j];
                        er[i] = (acc * h) / sc;
                    }
                    const double err = rms4(er[0], er[1], er[2], er[3]);
                    const double pw = 0.9 * pow(err, -0.2);
                    if (err < 1) {
                        double fac = (err == 0) ? 10.0 : fmin(10.0, pw);
                        if (rejected && fac > 1) fac = 1;
                        ha *= fac;
                        // step record for k_emit: t_old, h, t_new, -, y_old[4], K[7][4]
                        if (nacc < a.max_rk_steps) {
                            double2 *o = reinterpret_cast<double2 *>(srec + (size_t)nacc * kStepRec);
                            o[0] = make_double2(t, h);
                            o[1] = make_double2(t_new, 0.0);
                            o[2] = make_double2(y[0], y[1]);
                            o[3] = make_double2(y[2], y[3]);
                            for (int j = 0; j < 7; ++j) {
                                o[4 + 2 * j] = make_double2(KS(j, 0), KS(j, 1));
                                o[5 + 2 * j] = make_double2(KS(j, 2), KS(j, 3));
                            }
                        }
                        ++nacc;
                        const double t_old = t;
                        t = t_new;
                        for (int i = 0; i < 4; ++i) { y[i] = yn[i]; f[i] = r.d[i]; }
                        h_abs = ha;
                        if (t - tb >= 0) status = TCR_STATUS_FINISHED;
                        // terminal event at the step end (ivp.py:673-693); g >= 0 always, so a trigger
                        // is g_new == 0 (root = step end) or g0 == 0 on the first step (root = t0)
                        const double g_new = event_fn(P, y[0], y[1], y[2]);
                        double t_emit = t;
                        if (g == 0.0) { status = TCR_STATUS_EVENT; t_emit = t_old; }
                        else if (g_new == 0.0) status = TCR_STATUS_EVENT;
                        g = g_new;
                        next_out = samples_upto(P, t_emit);         // t_eval emission count (ivp.py:706-723)
                        if (status == kRunning && nacc >= a.max_rk_steps) status = TCR_STATUS_STEP_OVERFLOW;
                        if (status != kRunning) finalize();
                        else begin_step();
                    } else {
                        ha *= fmax(0.2, pw);
                        rejected = true;
                        ++nrej;
                        attempt_setup();
                    }
                }
            } else if (live && slot == 0) {
                // fresh storm, fun(t0, y0): ventilation gate (coupled_fast.py:238-244) on the same
                // lookups, then RungeKutta.__init__'s f0 and select_initial_step part 1 (common.py:112-126)
                if (r.vpot > 0 && r.shear * r.chi / r.vpot >= 1) {
                    status = TCR_STATUS_GATED;
                    finalize();
                    fresh = false;
                } else {
                    nfev = 1;
                    double sc[4];
                    for (int i = 0; i < 4; ++i) { f[i] = r.d[i]; sc[i] = P.atol + fabs(y[i]) * P.rtol; }
                    const double d0 = rms4(y[0] / sc[0], y[1] / sc[1], y[2] / sc[2], y[3] / sc[3]);
                    const double d1 = rms4(f[0] / sc[0], f[1] / sc[1], f[2] / sc[2], f[3] / sc[3]);
                    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
                    h0 = h0 < tb ? h0 : tb;
                    for (int i = 0; i < 4; ++i) e[1 + i] = y[i] + h0 * 1.0 * f[i];
                    e[0] = t + h0 * 1.0;
                    h = h0;
                }
            } else if (live && slot == 1) {
                // fresh storm, f1: select_initial_step part 2 (common.py:127-134); first attempt set up
                ++nfev;
                const double h0 = h;
                double sc[4];
                for (int i = 0; i < 4; ++i) sc[i] = P.atol + fabs(y[i]) * P.rtol;
                const double d1 = rms4(f[0] / sc[0], f[1] / sc[1], f[2] / sc[2], f[3] / sc[3]);
                const double d2 = rms4((r.d[0] - f[0]) / sc[0], (r.d[1] - f[1]) / sc[1],
                                       (r.d[2] - f[2]) / sc[2], (r.d[3] - f[3]) / sc[3]) / h0;
                double h1;
                if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
                else h1 = pow(0.01 / fmax(d1, d2), 0.2);
                h_abs = fmin(fmin(100 * h0, h1), fmin(tb, P.max_step));
                g = event_fn(P, y[0], y[1], y[2]);
                begin_step();
            }
        }
        fresh = false;
    }
#undef KS
}

// ---------------------------------------------------------------------------
// k_emit: everything of run_tracks that is independent per output sample, fused:
//   * t_eval emission from the dense output of the step that contains the sample
//     (ivp.py:706-723, rk.py:552-574),
//   * the env-wind recompute at every emitted sample (util/compute.py:201-202),
//   * axi_to_max_wind (wind/tc_wind.py:6-21 with util/sphere.py:15-30,58-83),
//   * accept tests 1 and 2 (compute.py:185-189, 205),
//   * the reference's per-variable [n_tracks][n_steps] planes with NaN padding
//     (compute.py:124-133), written coalesced.
// One workgroup per storm; a thread owns samples tid, tid+T, ...
struct EArgs {
    tcr_params P;
    DevFields D;
    int64_t n;
    int max_rk_steps;
    const double *srec;          // [n][max_rk_steps][kStepRec]
    const double *fs;            // [n][n_steps][4]
    const int32_t *slot;
    const int32_t *n_valid, *status, *n_accept;
    double *lon, *lat, *v, *m, *vmax, *envw;
    int32_t *flags;
};

__device__ __forceinline__ double haversine_km(const tcr_params &P, double lon1, double lat1,
                                               double lon2, double lat2)
{
    const double d = kPi / 180.0;
    lon1 *= d; lat1 *= d; lon2 *= d; lat2 *= d;
    const double sa = sin((lat2 - lat1) / 2), sb = sin((lon2 - lon1) / 2);
    const double aa = sa * sa + cos(lat1) * cos(lat2) * (sb * sb);
    return (P.earth_R / 1000.) * (2 * asin(sqrt(aa)));
}

// The two haversine calls of calc_translational_speed (sphere.py:71-76) have either equal latitudes
// or equal longitudes; the vanishing term is sin(0)^2 = 0 exactly, so dropping it is bit-identical.
__device__ __forceinline__ double haversine_same_lat_km(const tcr_params &P, double lon1, double lon2, double lat)
{
    const double d = kPi / 180.0;
    lon1 *= d; lon2 *= d; lat *= d;
    const double sb = sin((lon2 - lon1) / 2), c = cos(lat);
    const double aa = 0.0 + c * c * (sb * sb);
    return (P.earth_R / 1000.) * (2 * asin(sqrt(aa)));
}

__device__ __forceinline__ double haversine_same_lon_km(const tcr_params &P, double lat1, double lat2)
{
    const double d = kPi / 180.0;
    lat1 *= d; lat2 *= d;
    const double sa = sin((lat2 - lat1) / 2);
    const double aa = sa * sa + cos(lat1) * cos(lat2) * 0.0;
    return (P.earth_R / 1000.) * (2 * asin(sqrt(aa)));
}

constexpr int kEmitThreads = 128;
#ifndef TCR_EMIT_WPS
#define TCR_EMIT_WPS 2     // waves per SIMD k_emit is register-budgeted for (it is a throughput kernel)
#endif

template <bool AFFINE>
__global__ __launch_bounds__(kEmitThreads, TCR_EMIT_WPS) void k_emit(EArgs a)
{
    // LDS: per accepted step {t_new, t_old, h, y_old[4], Q[16]} = 23 doubles; per sample {lon, lat, v, us, vs}
    extern __shared__ double esh[];
    __shared__ EvalK K;
    __shared__ double s_best[kEmitThreads / 64];
    __shared__ int s_any[kEmitThreads / 64];
    __shared__ double s_v2d[2];
    const tcr_params &P = a.P;
    const int64_t sid = blockIdx.x;
    const int ns = P.n_steps;
    const int n = a.n_valid[sid];
    const int status = a.status[sid];
    int nst = a.n_accept[sid];
    nst = nst < a.max_rk_steps ? nst : a.max_rk_steps;
    double *s_step = esh;                                   // [max_rk_steps][23]
    double *s_lon = esh + (size_t)a.max_rk_steps * 23, *s_lat = s_lon + ns, *s_v = s_lat + ns;
    double *s_us = s_v + ns, *s_vs = s_us + ns;
    const double *srec = a.srec + sid * (int64_t)a.max_rk_steps * kStepRec;
    const double *fs = a.fs + sid * ns * 4;
    const DevSlot S = a.D.slots[a.slot[sid]];
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    // stage t_new, t_old, h, y_old and the dense-output matrix Q = K^T P (rk.py:179-181)
    for (int p = threadIdx.x; p < nst * 23; p += kEmitThreads) {
        const int j = p / 23, c = p - j * 23;
        const double *rj = srec + (size_t)j * kStepRec;
        double val;
        if (c == 0) val = rj[2];
        else if (c == 1) val = rj[0];
        else if (c == 2) val = rj[1];
        else if (c < 7) val = rj[4 + (c - 3)];
        else {
            const int i = (c - 7) >> 2, k = (c - 7) & 3;
            double acc = 0.0;
            for (int q = 0; q < 7; ++q) acc += rj[8 + q * 4 + i] * RK_P[q][k];
            val = acc;
        }
        s_step[p] = val;
    }
    if (threadIdx.x == 0) make_eval_k(P, a.D, K);
    __syncthreads();

    // ---- pass 1: dense output + env winds per valid sample; NaN padding beyond
    const double step_out = P.total_time / (double)(ns - 1);
    const double t2d = 2 * 86400.0;
    int any15 = 0;
    for (int i0 = 0; i0 < ns; i0 += kEmitThreads) {
        const int i = i0 + threadIdx.x;
        double lon = nan, lat = nan, v = nan, m = nan, w[4] = {nan, nan, nan, nan};
        const bool valid = i < n;
        double te = 0.0, ye[4] = {0.0, 0.0, 0.0, 0.0};
        if (valid) {
            te = ts_at(P, i);
            // the sample belongs to the first accepted step whose end is >= te
            int lo = 0, hi = nst - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_step[mid * 23] >= te) hi = mid; else lo = mid + 1; }
            const double *st = s_step + lo * 23;
            const double hh = st[2];
            const double x = (te - st[1]) / hh;
            const double p1 = x, p2 = p1 * x, p3 = p2 * x, p4 = p3 * x;
            for (int c = 0; c < 4; ++c) {
                const double *Q = st + 7 + c * 4;
                double acc = 0.0;
                acc += Q[0] * p1; acc += Q[1] * p2; acc += Q[2] * p3; acc += Q[3] * p4;
                ye[c] = hh * acc + st[3 + c];
            }
        }
        if (valid) {
            env_winds<AFFINE>(K, S, fs, ye[0], ye[1], te, w);
            lon = ye[0]; lat = ye[1]; v = ye[2]; m = ye[3];
            if (v >= P.v_thresh) any15 = 1;
            s_lon[i] = lon; s_lat[i] = lat; s_v[i] = v;
            s_us[i] = w[0] - w[2]; s_vs[i] = w[1] - w[3];
            // np.interp(2 d, res.t, v) needs v at the two samples bracketing 2 d (or the last one)
            const int j2 = (int)floor(t2d / step_out);
            if (t2d >= ts_at(P, n - 1)) { if (i == n - 1) s_v2d[0] = s_v2d[1] = v; }
            else { if (i == j2) s_v2d[0] = v; if (i == j2 + 1) s_v2d[1] = v; }
        }
        if (i < ns) {
            const size_t o = (size_t)sid * ns + i;
            a.lon[o] = lon; a.lat[o] = lat; a.v[o] = v; a.m[o] = m;
            double2 *eo = reinterpret_cast<double2 *>(a.envw + o * 4);
            eo[0] = make_double2(w[0], w[1]);
            eo[1] = make_double2(w[2], w[3]);
        }
    }
    __syncthreads();

    // ---- pass 2: translation speed by centred differences -> vmax (needs the neighbours)
    double best = -INFINITY;
    for (int i = threadIdx.x; i < ns; i += kEmitThreads) {
        double vm = nan;
        if (i < n && n > 1) {
            const double lon = s_lon[i], lat = s_lat[i], v = s_v[i];
            // linear extrapolation at both ends (sphere.py:66-69)
            const double lom = (i == 0) ? 2 * lon - s_lon[1] : s_lon[i - 1];
            const double lam = (i == 0) ? 2 * lat - s_lat[1] : s_lat[i - 1];
            const double lop = (i == n - 1) ? 2 * lon - s_lon[n - 2] : s_lon[i + 1];
            const double lap = (i == n - 1) ? 2 * lat - s_lat[n - 2] : s_lat[i + 1];
            const double dlon = 0.5 * (sign_of(lop - lom) * haversine_same_lat_km(P, lop, lom, lat));
            const double dlat = 0.5 * (sign_of(lap - lam) * haversine_same_lon_km(P, lap, lam));
            const double ut = dlon * 1000. / P.dt_out, vt = dlat * 1000. / P.dt_out;
            const double G = fmin(1., 0.8 + 0.35 * (1. + tanh((lat - 35.) / 10.)));
            const double Ui = G * ut + 0.1 * s_us[i] * v / 15.;
            const double Vi = G * vt + 0.1 * s_vs[i] * v / 15.;
            const double mag = sqrt(Ui * Ui + Vi * Vi);
            const double fac = np_min((v * 0.50) / mag, 1.0);
            const double th = atan2(-Ui, Vi);
            const double ug = v * -sin(th) + Ui * fac;
            const double vg = v * cos(th) + Vi * fac;
            vm = sqrt(ug * ug + vg * vg);
            if (vm > best) best = vm;
        }
        a.vmax[(size_t)sid * ns + i] = vm;
    }
    // ---- accept flags: wave reductions, then one thread
    for (int off = 32; off > 0; off >>= 1) {
        best = fmax(best, __shfl_down(best, off));
        any15 |= __shfl_down(any15, off);
    }
    if ((threadIdx.x & 63) == 0) { s_best[threadIdx.x >> 6] = best; s_any[threadIdx.x >> 6] = any15; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < kEmitThreads / 64; ++wv) { best = fmax(best, s_best[wv]); any15 |= s_any[wv]; }
        int fl = 0;
        if (n > 0 && status != TCR_STATUS_GATED) {
            double v2d;
            if (t2d >= ts_at(P, n - 1)) v2d = s_v2d[0];
            else {
                const int j = (int)floor(t2d / step_out);
                v2d = (s_v2d[1] - s_v2d[0]) / (ts_at(P, j + 1) - ts_at(P, j)) * (t2d - ts_at(P, j)) + s_v2d[0];
            }
            if (any15 && v2d >= P.v_2d_thresh) {
                fl |= TCR_FLAG_IS_TC;
                if (n > 1 && best >= P.vmax_thresh) fl |= TCR_FLAG_ACCEPTED;
            }
        }
        a.flags[sid] = fl;
    }
}

template <bool AFFINE>
__global__ __launch_bounds__(64) void k_probe_rhs(tcr_params P, DevFields D, int slot, double h_bl, const double *fs,
                            int64_t n, const double *t, const double *lon, const double *lat,
                            const double *v, const double *m, double *dydt, double *envw, double *alpha)
{
    __shared__ EvalK K;
    if (threadIdx.x == 0) make_eval_k(P, D, K);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevSlot S = D.slots[slot];
    const Rhs r = rhs_eval<AFFINE>(K, S, fs, h_bl, t[i], lon[i], lat[i], v[i], m[i]);
    for (int k = 0; k < 4; ++k) dydt[i * 4 + k] = r.d[k];
    alpha[i] = r.alpha;
    double w[4];
    env_winds<AFFINE>(K, S, fs, lon[i], lat[i], t[i], w);
    for (int k = 0; k < 4; ++k) envw[i * 4 + k] = w[k];
}

}  // namespace tcr
