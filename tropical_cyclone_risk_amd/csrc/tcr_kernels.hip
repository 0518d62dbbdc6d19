// HIP kernels of the storm hot path (gfx950).  One wave64 per workgroup, one
// lane per storm: the state of a storm (y, f, step size, event value, output
// cursor) lives in VGPRs, the seven Runge–Kutta stage derivatives live in LDS
// (lane-contiguous, conflict-free ds_read_b64), fields are read through L1/L2
// from the interleaved HBM layout described in tcr_device.h.
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
    double *rec;             // [n][n_steps][kRec]
    int32_t *n_valid, *status, *nfev, *n_accept, *n_reject;
};

// ---------------------------------------------------------------------------
// gen_f (track/bam_track.py:23-31): one thread per (storm, sample), 4 series.
// arg keeps NumPy's evaluation order 2π·((n·t)/T + x).
__global__ __launch_bounds__(256) void k_fourier_table(tcr_params P, int64_t n,
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

// ---------------------------------------------------------------------------
__constant__ double RK_C[6] = {0, 1. / 5, 3. / 10, 4. / 5, 8. / 9, 1};
__constant__ double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1. / 5, 0, 0, 0, 0},
    {3. / 40, 9. / 40, 0, 0, 0},
    {44. / 45, -56. / 15, 32. / 9, 0, 0},
    {19372. / 6561, -25360. / 2187, 64448. / 6561, -212. / 729, 0},
    {9017. / 3168, -355. / 33, 46732. / 5247, 49. / 176, -5103. / 18656}};
__constant__ double RK_B[6] = {35. / 384, 0, 500. / 1113, 125. / 192, -2187. / 6784, 11. / 84};
__constant__ double RK_E[7] = {-71. / 57600, 0, 71. / 16695, -71. / 1920, 17253. / 339200, -22. / 525, 1. / 40};
__constant__ double RK_P[7][4] = {
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

constexpr int kWave = 64;

__global__ __launch_bounds__(kWave) void k_integrate(KArgs a)
{
    // K[stage][component][lane]
    __shared__ double K[7][4][kWave];
    const tcr_params &P = a.P;
    const DevFields &D = a.D;
    const int lane = threadIdx.x;
    const int64_t sid = (int64_t)blockIdx.x * kWave + lane;
    if (sid >= a.n) return;

    const int ns = P.n_steps;
    const DevSlot S = D.slots[a.slot[sid]];
    const double *fs = a.fs + sid * ns * 4;
    const double h_bl = a.h_bl[sid];
    double y[4] = {a.lon0[sid], a.lat0[sid], a.v0[sid], a.m0[sid]};
    double *rec = a.rec + sid * ns * kRec;

    int status = 99, nfev = 0, nacc = 0, nrej = 0, next_out = 0;

    // ventilation gate (coupled_fast.py:238-244)
    {
        double w[4];
        env_winds(P, D, S, fs, y[0], y[1], 0.0, w);
        const double du = w[0] - w[2], dw = w[1] - w[3];
        const double Sh = sqrt(du * du + dw * dw);
        const Cell tx = locate(D.tg.lon, D.tg.rlon, D.tg.nlon, D.tg.lon_inv_step, y[0]);
        const Cell ty = locate(D.tg.lat, D.tg.rlat, D.tg.nlat, D.tg.lat_inv_step, y[1]);
        double th[4];
        bilinear<4, kThermoStride>(S.thermo, D.tg.nlon, tx, ty, th);
        const Cell hx = locate(D.hg.lon, D.hg.rlon, D.hg.nlon, D.hg.lon_inv_step, y[0]);
        const Cell hy = locate(D.hg.lat, D.hg.rlat, D.hg.nlat, D.hg.lat_inv_step, y[1]);
        double lb[2];
        bilinear<2, kStaticStride>(D.stat, D.hg.nlon, hx, hy, lb);
        const double vp = (lb[0] == 1.0) ? 0.0 : th[0];
        if (vp > 0 && Sh * th[1] / vp >= 1) status = TCR_STATUS_GATED;
    }

    if (status == 99) {
        const double tb = P.total_time;
        double t = 0.0, f[4], h_abs;
        // RungeKutta.__init__: f0 = fun(t0, y0); select_initial_step (common.py:68-134)
        {
            double sc[4], y1[4], h0 = 0.0, d1 = 0.0;
            double yy[4] = {y[0], y[1], y[2], y[3]};
            double tt = t;
            for (int q = 0; q < 2; ++q) {
                const Rhs r = rhs_eval(P, D, S, fs, h_bl, tt, yy[0], yy[1], yy[2], yy[3]);
                ++nfev;
                if (q == 0) {
                    for (int i = 0; i < 4; ++i) { f[i] = r.d[i]; sc[i] = P.atol + fabs(y[i]) * P.rtol; }
                    const double d0 = rms4(y[0] / sc[0], y[1] / sc[1], y[2] / sc[2], y[3] / sc[3]);
                    d1 = rms4(f[0] / sc[0], f[1] / sc[1], f[2] / sc[2], f[3] / sc[3]);
                    h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
                    h0 = h0 < tb ? h0 : tb;
                    for (int i = 0; i < 4; ++i) { y1[i] = y[i] + h0 * 1.0 * f[i]; yy[i] = y1[i]; }
                    tt = t + h0 * 1.0;
                } else {
                    const double d2 = rms4((r.d[0] - f[0]) / sc[0], (r.d[1] - f[1]) / sc[1],
                                           (r.d[2] - f[2]) / sc[2], (r.d[3] - f[3]) / sc[3]) / h0;
                    double h1;
                    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
                    else h1 = pow(0.01 / fmax(d1, d2), 0.2);
                    h_abs = fmin(fmin(100 * h0, h1), fmin(tb, P.max_step));
                }
            }
        }
        double g = event_fn(P, y[0], y[1], y[2]);

        while (status == 99) {
            const double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
            double ha = h_abs;
            if (ha > P.max_step) ha = P.max_step;
            else if (ha < min_step) ha = min_step;
            bool accepted = false, rejected = false, failed = false;
            double h = 0.0, t_new = 0.0, y_new[4], f_new[4];
            while (!accepted) {
                if (ha < min_step) { failed = true; break; }
                h = ha;
                t_new = t + h;
                if (t_new - tb > 0) t_new = tb;
                h = t_new - t;
                ha = fabs(h);
                // rk_step: K[s] = fun(t + c_s h, y + h * sum_j a_sj K[j])
                for (int i = 0; i < 4; ++i) K[0][i][lane] = f[i];
                for (int st = 1; st <= 6; ++st) {
                    double ys[4];
                    for (int i = 0; i < 4; ++i) {
                        double dy = 0.0;
                        if (st < 6) {
                            for (int j = 0; j < st; ++j) dy += K[j][i][lane] * RK_A[st][j];
                            ys[i] = y[i] + dy * h;
                        } else {
                            for (int j = 0; j < 6; ++j) dy += K[j][i][lane] * RK_B[j];
                            ys[i] = y[i] + h * dy;
                        }
                    }
                    const double ts = (st < 6) ? t + RK_C[st] * h : t + h;
                    const Rhs r = rhs_eval(P, D, S, fs, h_bl, ts, ys[0], ys[1], ys[2], ys[3]);
                    ++nfev;
                    for (int i = 0; i < 4; ++i) K[st][i][lane] = r.d[i];
                    if (st == 6) for (int i = 0; i < 4; ++i) { y_new[i] = ys[i]; f_new[i] = r.d[i]; }
                }
                double er[4];
                for (int i = 0; i < 4; ++i) {
                    const double sc = P.atol + fmax(fabs(y[i]), fabs(y_new[i])) * P.rtol;
                    double acc = 0.0;
                    for (int j = 0; j < 7; ++j) acc += K[j][i][lane] * RK_E[j];
                    er[i] = (acc * h) / sc;
                }
                const double err = rms4(er[0], er[1], er[2], er[3]);
                if (err < 1) {
                    double fac = (err == 0) ? 10.0 : fmin(10.0, 0.9 * pow(err, -0.2));
                    if (rejected && fac > 1) fac = 1;
                    ha *= fac;
                    accepted = true;
                } else {
                    ha *= fmax(0.2, 0.9 * pow(err, -0.2));
                    rejected = true;
                    ++nrej;
                }
            }
            if (failed) { status = TCR_STATUS_STEP_FAIL; break; }
            ++nacc;
            const double t_old = t;
            const double y_old[4] = {y[0], y[1], y[2], y[3]};
            t = t_new;
            for (int i = 0; i < 4; ++i) { y[i] = y_new[i]; f[i] = f_new[i]; }
            h_abs = ha;
            if (t - tb >= 0) status = TCR_STATUS_FINISHED;
            // dense output Q = K^T P (rk.py:179-181)
            double Q[4][4];
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 4; ++k) {
                    double acc = 0.0;
                    for (int j = 0; j < 7; ++j) acc += K[j][i][lane] * RK_P[j][k];
                    Q[i][k] = acc;
                }
            // terminal event at the step end (ivp.py:673-693); g >= 0 always, so a
            // trigger is g_new == 0 (root = step end) or g0 == 0 on the first step (root = t0)
            const double g_new = event_fn(P, y[0], y[1], y[2]);
            double t_emit = t;
            if (g == 0.0) { status = TCR_STATUS_EVENT; t_emit = t_old; }
            else if (g_new == 0.0) status = TCR_STATUS_EVENT;
            g = g_new;
            // t_eval emission (ivp.py:706-723) fused with the env-wind recompute of
            // util/compute.py:201-202 at the emitted sample
            while (next_out < ns) {
                const double te = ts_at(P, next_out);
                if (te > t_emit) break;
                const double x = (te - t_old) / h;
                const double p1 = x, p2 = p1 * x, p3 = p2 * x, p4 = p3 * x;
                double ye[4];
                for (int i = 0; i < 4; ++i) {
                    double acc = 0.0;
                    acc += Q[i][0] * p1; acc += Q[i][1] * p2; acc += Q[i][2] * p3; acc += Q[i][3] * p4;
                    ye[i] = h * acc + y_old[i];
                }
                double w[4];
                env_winds(P, D, S, fs, ye[0], ye[1], te, w);
                double2 *o = reinterpret_cast<double2 *>(rec + (size_t)next_out * kRec);
                o[0] = make_double2(ye[0], ye[1]);
                o[1] = make_double2(ye[2], ye[3]);
                o[2] = make_double2(w[0], w[1]);
                o[3] = make_double2(w[2], w[3]);
                ++next_out;
            }
        }
    }
    a.n_valid[sid] = next_out;
    a.status[sid] = status;
    a.nfev[sid] = (status == TCR_STATUS_GATED) ? 0 : nfev;
    a.n_accept[sid] = nacc;
    a.n_reject[sid] = nrej;
}

// ---------------------------------------------------------------------------
// Post-step: accept test 1 (compute.py:185-189), axi_to_max_wind (wind/tc_wind.py:6-21
// with util/sphere.py:15-30,58-83), accept test 2 (compute.py:205), and the unpack of
// the [storm][sample][8] records into the reference's per-variable planes with NaN
// padding (compute.py:124-133).  One workgroup per storm, threads stride over samples.
struct PArgs {
    tcr_params P;
    int64_t n;
    const double *rec;
    const int32_t *n_valid, *status;
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

constexpr int kPostThreads = 128;

__global__ __launch_bounds__(kPostThreads) void k_post_unpack(PArgs a)
{
    __shared__ double s_best[kPostThreads];
    __shared__ int s_any[kPostThreads];
    const tcr_params &P = a.P;
    const int64_t sid = blockIdx.x;
    const int ns = P.n_steps;
    const int n = a.n_valid[sid];
    const int status = a.status[sid];
    const double *rec = a.rec + sid * ns * kRec;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    double best = -INFINITY;
    int any15 = 0;
    for (int i = threadIdx.x; i < ns; i += kPostThreads) {
        double lon = nan, lat = nan, v = nan, m = nan, vm = nan, w0 = nan, w1 = nan, w2 = nan, w3 = nan;
        if (i < n) {
            const double2 *q = reinterpret_cast<const double2 *>(rec + (size_t)i * kRec);
            const double2 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3];
            lon = r0.x; lat = r0.y; v = r1.x; m = r1.y; w0 = r2.x; w1 = r2.y; w2 = r3.x; w3 = r3.y;
            if (v >= P.v_thresh) any15 = 1;
            if (n > 1) {
                // linear extrapolation at both ends, centred differences (sphere.py:66-77)
                double lom, lam, lop, lap;
                if (i == 0) { lom = 2 * lon - rec[kRec]; lam = 2 * lat - rec[kRec + 1]; }
                else { lom = rec[(size_t)(i - 1) * kRec]; lam = rec[(size_t)(i - 1) * kRec + 1]; }
                if (i == n - 1) { lop = 2 * lon - rec[(size_t)(n - 2) * kRec]; lap = 2 * lat - rec[(size_t)(n - 2) * kRec + 1]; }
                else { lop = rec[(size_t)(i + 1) * kRec]; lap = rec[(size_t)(i + 1) * kRec + 1]; }
                const double dlon = 0.5 * (sign_of(lop - lom) * haversine_km(P, lop, lat, lom, lat));
                const double dlat = 0.5 * (sign_of(lap - lam) * haversine_km(P, lon, lap, lon, lam));
                const double ut = dlon * 1000. / P.dt_out, vt = dlat * 1000. / P.dt_out;
                const double G = fmin(1., 0.8 + 0.35 * (1. + tanh((lat - 35.) / 10.)));
                const double Ui = G * ut + 0.1 * (w0 - w2) * v / 15.;
                const double Vi = G * vt + 0.1 * (w1 - w3) * v / 15.;
                const double mag = sqrt(Ui * Ui + Vi * Vi);
                const double fac = np_min((v * 0.50) / mag, 1.0);
                const double th = atan2(-Ui, Vi);
                const double ug = v * -sin(th) + Ui * fac;
                const double vg = v * cos(th) + Vi * fac;
                vm = sqrt(ug * ug + vg * vg);
                if (vm > best) best = vm;
            }
        }
        const size_t o = (size_t)sid * ns + i;
        a.lon[o] = lon; a.lat[o] = lat; a.v[o] = v; a.m[o] = m; a.vmax[o] = vm;
        double2 *e = reinterpret_cast<double2 *>(a.envw + o * 4);
        e[0] = make_double2(w0, w1);
        e[1] = make_double2(w2, w3);
    }
    s_best[threadIdx.x] = best;
    s_any[threadIdx.x] = any15;
    __syncthreads();
    for (int s = kPostThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            s_best[threadIdx.x] = fmax(s_best[threadIdx.x], s_best[threadIdx.x + s]);
            s_any[threadIdx.x] |= s_any[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int fl = 0;
        if (n > 0 && status != TCR_STATUS_GATED) {
            // np.interp(2 d, res.t, v): clamps to the last sample of a short track
            const double t2d = 2 * 86400.0;
            double v2d;
            if (t2d >= ts_at(P, n - 1)) v2d = rec[(size_t)(n - 1) * kRec + 2];
            else {
                int j = (int)floor(t2d / (P.total_time / (double)(ns - 1)));
                while (j > 0 && ts_at(P, j) > t2d) --j;
                while (j < n - 2 && ts_at(P, j + 1) <= t2d) ++j;
                const double va = rec[(size_t)j * kRec + 2], vb = rec[(size_t)(j + 1) * kRec + 2];
                v2d = (vb - va) / (ts_at(P, j + 1) - ts_at(P, j)) * (t2d - ts_at(P, j)) + va;
            }
            if (s_any[0] && v2d >= P.v_2d_thresh) {
                fl |= TCR_FLAG_IS_TC;
                if (n > 1 && s_best[0] >= P.vmax_thresh) fl |= TCR_FLAG_ACCEPTED;
            }
        }
        a.flags[sid] = fl;
    }
}

// ---------------------------------------------------------------------------
// Probe: dydt / _env_winds / _calc_alpha at arbitrary points of one slot with one
// forcing table (parity tests of the seam's leaf methods).
__global__ __launch_bounds__(64) void k_probe_rhs(tcr_params P, DevFields D, int slot, double h_bl, const double *fs,
                            int64_t n, const double *t, const double *lon, const double *lat,
                            const double *v, const double *m, double *dydt, double *envw, double *alpha)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevSlot S = D.slots[slot];
    const Rhs r = rhs_eval(P, D, S, fs, h_bl, t[i], lon[i], lat[i], v[i], m[i]);
    for (int k = 0; k < 4; ++k) dydt[i * 4 + k] = r.d[k];
    alpha[i] = r.alpha;
    double w[4];
    env_winds(P, D, S, fs, lon[i], lat[i], t[i], w);
    for (int k = 0; k < 4; ++k) envw[i * 4 + k] = w[k];
}

}  // namespace tcr
