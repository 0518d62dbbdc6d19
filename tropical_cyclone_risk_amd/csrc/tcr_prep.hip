// Monthly wind statistics: the step right before the hot path.
//
// calc_wnd_stat (track/env_wind.py:180-228): from the month's samples of (ua250, va250, ua850,
// va850) — first averaged per calendar day when the record is coarser than daily (`groupby("time.day")
// .mean`, :199-203) — the mean of each component (:218), its variance with ddof = 0 (`.var`, :221)
// and the covariances with ddof = 1 (`xr.cov`, :223): 14 statistics per grid point, in the order
// of `wnd_stats` (:226-229): 4 means, then the lower triangle row by row.
//
// The kernel reproduces what those xarray calls execute on NumPy-backed data *in the dtype of the
// file* (T = float for ERA5's float32 u / v, T = double otherwise), including `skipna`:
//   mean   np.nanmean: NaN -> 0, sum over time in T in time order, count of non-NaN, sum / count in
//          fp64 rounded back to T;
//   var    np.nanvar(ddof=0): x - mean (T), NaN -> 0, squares summed in T, / count via fp64, rounded to T;
//   cov    xr.cov(ddof=1): per pair, only the samples where BOTH components are finite; each
//          component's mean over those samples (as np.nanmean above); products of the deviations
//          summed in T; the sum divided by (count - 1) in fp64 — the one statistic that is fp64;
//   and every statistic is stored as fp64 (`wnd_stats[i] = stats[i]` into an np.zeros array).
// With no NaN in the month every pair uses the plain means, and with T = double this is the round-1
// kernel bit for bit.
//
// A pure streaming reduction: one thread per grid point, planes are [sample][point] so a wave reads
// 64 x sizeof(T) contiguous bytes per plane and sample; two passes over the month (sums, then centred
// products).  Bound: HBM reads, 2 x 4 planes x sizeof(T) per point and sample (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tcr {

template <typename T>
struct WindStatArgsT {
    const T *w[4];               // ua250, va250, ua850, va850: [n_samples][n_points]
    const int32_t *day_start;    // [n_days + 1] sample offsets of the calendar days (NULL: every sample is a day)
    int32_t n_days;
    int64_t n_points;
    double *out;                 // [14][n_points]
};

// one "day" of the four components: the sample itself, or np.nanmean over the day's samples
template <typename T>
__device__ __forceinline__ void day_means(const WindStatArgsT<T> &a, int d, int64_t p, T (&x)[4])
{
    if (!a.day_start) {
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = a.w[c][(size_t)d * a.n_points + p];
        return;
    }
    const int s0 = a.day_start[d], s1 = a.day_start[d + 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        T acc = T(0);
        int cnt = 0;
        for (int s = s0; s < s1; ++s) {
            const T v = a.w[c][(size_t)s * a.n_points + p];
            const bool ok = v == v;
            acc += ok ? v : T(0);
            cnt += ok ? 1 : 0;
        }
        x[c] = (T)((double)acc / (double)cnt);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_wind_stats(WindStatArgsT<T> a)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_points) return;
    const int D = a.n_days;
    // pass 1: sums and counts — per component, and per pair over the samples where both are finite
    T s1[4] = {0, 0, 0, 0};
    int n1[4] = {0, 0, 0, 0};
    T sa[6] = {0, 0, 0, 0, 0, 0}, sb[6] = {0, 0, 0, 0, 0, 0};     // pair k = (i, j), i > j: sums of x_i and x_j
    int n2[6] = {0, 0, 0, 0, 0, 0};
    for (int d = 0; d < D; ++d) {
        T x[4];
        day_means(a, d, p, x);
        bool ok[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { ok[c] = x[c] == x[c]; s1[c] += ok[c] ? x[c] : T(0); n1[c] += ok[c] ? 1 : 0; }
        int k = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const bool both = ok[i] && ok[j];
                sa[k] += both ? x[i] : T(0); sb[k] += both ? x[j] : T(0); n2[k] += both ? 1 : 0;
                ++k;
            }
    }
    T m1[4], ma[6], mb[6];
#pragma unroll
    for (int c = 0; c < 4; ++c) m1[c] = (T)((double)s1[c] / (double)n1[c]);
#pragma unroll
    for (int k = 0; k < 6; ++k) { ma[k] = (T)((double)sa[k] / (double)n2[k]); mb[k] = (T)((double)sb[k] / (double)n2[k]); }
    // pass 2: centred squares / products
    T q1[4] = {0, 0, 0, 0}, q2[6] = {0, 0, 0, 0, 0, 0};
    for (int d = 0; d < D; ++d) {
        T x[4];
        day_means(a, d, p, x);
        bool ok[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ok[c] = x[c] == x[c];
            const T dev = x[c] - m1[c];
            q1[c] += ok[c] ? dev * dev : T(0);
        }
        int k = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const T da = x[i] - ma[k], db = x[j] - mb[k];
                q2[k] += (ok[i] && ok[j]) ? da * db : T(0);
                ++k;
            }
    }
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
    for (int c = 0; c < 4; ++c) a.out[(size_t)c * a.n_points + p] = (double)m1[c];
    int o = 0, k = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double v;
            if (i == j) v = (double)(T)((double)q1[i] / (double)n1[i]);
            else { v = n2[k] >= 1 ? (double)q2[k] / (double)(n2[k] - 1) : nan; ++k; }
            a.out[(size_t)(4 + o) * a.n_points + p] = v;
            ++o;
        }
}

}  // namespace tcr
