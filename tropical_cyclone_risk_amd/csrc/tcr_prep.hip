// Monthly wind statistics: the step right before the hot path.
//
// calc_wnd_stat (track/env_wind.py:180-228): from the month's samples of (ua250, va250, ua850,
// va850) — first averaged per calendar day when the record is sub-daily (`groupby("time.day")
// .mean`, :199-203) — the mean of each component (:218), its variance with ddof = 0 (`.var`, :221)
// and the covariances with ddof = 1 (`xr.cov`, :223): 14 statistics per grid point, in the order
// of `wnd_stats` (:226-229): 4 means, then the lower triangle row by row.
//
// A pure streaming reduction: one thread per grid point, planes are [sample][point] so a wave reads
// 512 contiguous bytes per plane and sample.  Two passes over the month (mean, then centred
// products), the textbook form NumPy uses — the sums run over days in order, so with fp64 inputs the
// result equals `x.mean(0)`, `((x - m)**2).mean(0)` and `((x - mx)*(y - my)).sum(0) / (D - 1)`
// bit for bit (-ffp-contract=off).  Bound: HBM reads, 2 x 4 planes x 8 B per point and sample.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tcr {

struct WindStatArgs {
    const double *w[4];          // ua250, va250, ua850, va850: [n_samples][n_points]
    const int32_t *day_start;    // [n_days + 1] sample offsets of the calendar days (NULL: every sample is a day)
    int32_t n_days;
    int64_t n_points;
    double *out;                 // [14][n_points]
};

__device__ __forceinline__ void day_means(const WindStatArgs &a, int d, int64_t p, double (&x)[4])
{
    if (!a.day_start) {
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = a.w[c][(size_t)d * a.n_points + p];
        return;
    }
    const int s0 = a.day_start[d], s1 = a.day_start[d + 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double acc = 0.0;
        for (int s = s0; s < s1; ++s) acc += a.w[c][(size_t)s * a.n_points + p];
        x[c] = acc / (double)(s1 - s0);
    }
}

__global__ __launch_bounds__(256) void k_wind_stats(WindStatArgs a)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_points) return;
    const int D = a.n_days;
    double m[4] = {0, 0, 0, 0};
    for (int d = 0; d < D; ++d) {
        double x[4];
        day_means(a, d, p, x);
#pragma unroll
        for (int c = 0; c < 4; ++c) m[c] += x[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) m[c] = m[c] / (double)D;
    double cc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < D; ++d) {
        double x[4];
        day_means(a, d, p, x);
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = x[c] - m[c];
        int k = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) cc[k++] += x[i] * x[j];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) a.out[(size_t)c * a.n_points + p] = m[c];
    int k = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            a.out[(size_t)(4 + k) * a.n_points + p] = (i == j) ? cc[k] / (double)D : cc[k] / (double)(D - 1);
            ++k;
        }
}

}  // namespace tcr
