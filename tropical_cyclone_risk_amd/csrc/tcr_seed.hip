// Device-side genesis seeding: the rejection-sampling loop of
// util/compute.py:134-175 restated per *candidate* so that candidates are
// independent and can be drawn in any order on any GPU.
//
// The reference draws from NumPy's global MT19937, re-seeded from the wall clock
// for every storm (track/bam_track.py:37-42), so its streams are not reproducible;
// here every candidate owns a counter-based Philox4x32-10 stream keyed by
// (experiment_seed, year) with counter (candidate index, purpose, draw index),
// which makes results independent of batch size and GPU count.
//
// Draw order per candidate follows compute.py: U(lon), U(sin lat), [U(lon),
// U(lat)]* while the run-basin mask < 1e-2 (:146-148), month (:151), U for the
// low-latitude filter (:165), N(0,1) for v0 (:172), then the 60 Fourier phases
// gen_f consumes inside gen_track (bam_track.py:27).
#pragma once
#include "tcr_device.h"

namespace tcr {

// The key of a round of candidates.  Kernels take it by value; when a round is replayed from a captured graph
// (tcr_round_dev) the arguments of its nodes are frozen, so the kernels read it through `key` — a device copy that a
// one-thread launch in front of the graph refreshes — whenever that pointer is set.
struct RoundKey {
    uint64_t seed;
    int64_t cand0;
    int32_t year;
    int32_t pad_;
};

struct SeedArgs {
    tcr_params P;
    DevFields D;
    uint64_t seed;
    int32_t year;
    int64_t cand0;
    tcr_seeds out;
    const RoundKey *key;        // NULL: seed / year / cand0 above
};

__global__ void k_set_round_key(RoundKey *dst, uint64_t seed, int32_t year, int64_t cand0)
{
    dst->seed = seed; dst->cand0 = cand0; dst->year = year; dst->pad_ = 0;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&o)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// two uniforms in [0,1) with 53 random bits each (NumPy's legacy double recipe)
__device__ __forceinline__ void uniform2_raw(uint64_t seed, int32_t year, int64_t cand, uint32_t purpose,
                                             uint32_t idx, double &u0, double &u1)
{
    uint32_t o[4];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ ((uint32_t)year * 0x9E3779B9u);
    philox4x32_10((uint32_t)cand, (uint32_t)((uint64_t)cand >> 32), purpose, idx, k0, k1, o);
    u0 = ((double)(o[0] >> 5) * 67108864.0 + (double)(o[1] >> 6)) / 9007199254740992.0;
    u1 = ((double)(o[2] >> 5) * 67108864.0 + (double)(o[3] >> 6)) / 9007199254740992.0;
}

// the 4*N Fourier phases of one candidate (purpose 2), element k
__device__ __forceinline__ double phase_at(uint64_t seed, int32_t year, int64_t cand, int k)
{
    double p0, p1;
    uniform2_raw(seed, year, cand, 2u, (uint32_t)(k >> 1), p0, p1);
    return (k & 1) ? p1 : p0;
}

// The eight 0/1 mask planes (run basin + 7 basins) are staged as the bits of ONE byte per grid point, so a
// candidate's nine mask lookups read four bytes instead of thirty-six.
struct MaskCorners {
    unsigned c00, c01, c10, c11;        // the bytes at (x0,y0), (x0,y1), (x1,y0), (x1,y1)
};

__device__ __forceinline__ MaskCorners mask_corners(const DevGrid &g, const uint8_t *__restrict__ m, const Cell &cx, const Cell &cy)
{
    const uint8_t *r0 = m + (size_t)cy.i * g.nlon + cx.i, *r1 = r0 + g.nlon;
    MaskCorners c;
    c.c00 = r0[0]; c.c01 = r1[0]; c.c10 = r0[1]; c.c11 = r1[1];
    return c;
}

// bilinear lookup of mask plane `bit`, FITPACK order (mat.interp2_fx on a bool DataArray)
__device__ __forceinline__ double mask_at(const MaskCorners &c, int bit, const Cell &cx, const Cell &cy)
{
    double sp = 0.0;
    sp = sp + (double)((c.c00 >> bit) & 1u) * cx.w0 * cy.w0;
    sp = sp + (double)((c.c01 >> bit) & 1u) * cx.w0 * cy.w1;
    sp = sp + (double)((c.c10 >> bit) & 1u) * cx.w1 * cy.w0;
    sp = sp + (double)((c.c11 >> bit) & 1u) * cx.w1 * cy.w1;
    return sp;
}

constexpr int kMaxRedraw = 1 << 14;

// Register budget: at most 80 VGPRs (six waves per SIMD), so that a wave fits into the 88 registers an integrator wave of
// another batch leaves on its SIMD; the unbudgeted kernel took 92 (96 allocated).
__global__ __launch_bounds__(256, 6) void k_seed(SeedArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.out.n) return;
    const tcr_params &P = a.P;
    const DevFields &D = a.D;
    const uint64_t seed = a.key ? a.key->seed : a.seed;
    const int32_t year = a.key ? a.key->year : a.year;
    const int64_t cand = (a.key ? a.key->cand0 : a.cand0) + i;
    const double deg = kPi / 180;

    // compute.py:140-145.  np.sign(-0.0) >= 0 is True, so a '0S' upper bound still
    // selects the 45 N cap exactly as in the reference: test with !(x < 0).
    const double lat_min = !(P.box[1] < 0) ? 3 : -45;
    const double lat_max = !(P.box[3] < 0) ? 45 : -3;
    const double y_min = sin(deg * lat_min), y_max = sin(deg * lat_max);
    double u0, u1;
    uniform2_raw(seed, year, cand, 0u, 0u, u0, u1);
    double lon = P.box[0] + (P.box[2] - P.box[0]) * u0;
    double lat = asin(y_min + (y_max - y_min) * u1) * 180 / kPi;
    Cell cx = locate(D.mg.ax, lon);
    Cell cy = locate(D.mg.ay, lat);
    int redraw = 0;
    MaskCorners mc = mask_corners(D.mg, D.mask_bits, cx, cy);
    while (mask_at(mc, 7, cx, cy) < 1e-2 && redraw < kMaxRedraw) {
        ++redraw;                                           // compute.py:146-148: uniform in lat
        uniform2_raw(seed, year, cand, 0u, (uint32_t)redraw, u0, u1);
        lon = P.box[0] + (P.box[2] - P.box[0]) * u0;
        lat = P.box[1] + (P.box[3] - P.box[1]) * u1;
        cx = locate(D.mg.ax, lon);
        cy = locate(D.mg.ay, lat);
        mc = mask_corners(D.mg, D.mask_bits, cx, cy);
    }
    double um, ul;
    uniform2_raw(seed, year, cand, 1u, 0u, um, ul);
    int month = (int)(um * 12.0) + 1;                       // randint(1, 13)
    month = month > 12 ? 12 : month;

    // genesis basin = argmax of the 7 basin masks (compute.py:155-158), first max wins
    double best = mask_at(mc, 0, cx, cy);
    int bidx = 0;
    for (int b = 1; b < TCR_N_BASINS; ++b) {
        const double val = mask_at(mc, b, cx, cy);
        if (val > best) { best = val; bidx = b; }
    }
    // PI at genesis from the month's field set (compute.py:162)
    const DevSlot S = D.slots[month - 1];
    const Cell tx = locate(D.tg.ax, lon);
    const Cell ty = locate(D.tg.ay, lat);
    double th[4];
    bilinear<4, kThermoStride>(S.thermo, D.tg.nlon, tx, ty, th);
    const double pi_gen = th[0];
    const double base = np_min(np_max((fabs(lat) - P.lat_vort_fac) / 12.0, 0.0), 1.0);
    const double prob = pow(base, P.lat_vort_power[bidx]);  // compute.py:163-164
    int flags = 0;
    if (redraw < kMaxRedraw && best > 1e-3 && ul < prob) {
        flags |= 1;                                         // counts toward n_seeds (compute.py:165-167)
        if (pi_gen > P.pi_gate) flags |= 2;                 // seed passes (compute.py:168-169)
    }
    // initial state (compute.py:172-175)
    double n0, n1;
    uniform2_raw(seed, year, cand, 1u, 1u, n0, n1);
    const double z = sqrt(-2.0 * log(1.0 - n0)) * cos(2. * kPi * n1);   // Box–Muller N(0,1)
    // m_init_fx[month].ev(lon, lat) on the uncropped thermo grid (compute.py:111,173)
    const Cell rx = locate(D.rg.ax, lon);
    const Cell ry = locate(D.rg.ay, lat);
    const double *r0 = S.rh + (size_t)ry.i * D.rg.nlon + rx.i, *r1 = r0 + D.rg.nlon;
    double rh = 0.0;
    rh = rh + r0[0] * rx.w0 * ry.w0;
    rh = rh + r1[0] * rx.w0 * ry.w1;
    rh = rh + r0[1] * rx.w1 * ry.w0;
    rh = rh + r1[1] * rx.w1 * ry.w1;
    const double m_init = P.minit_a / (1 + exp(-(rh - P.minit_b) * P.minit_c)) + P.minit_d;

    a.out.lon0[i] = lon;
    a.out.lat0[i] = lat;
    a.out.v0[i] = P.seed_v_init + z;
    a.out.m0[i] = np_max(0.0, m_init);        // np.maximum(0, f_mInit(rh)): a NaN rh_mid stays NaN (compute.py:174)
    a.out.h_bl[i] = P.atm_bl_depth[bidx];
    a.out.slot[i] = month - 1;
    a.out.basin_idx[i] = bidx;
    a.out.seed_flags[i] = flags;
    const int N = P.n_series;
    if (!a.out.phases) return;               // drawn later, only for the candidates that pass
    double *ph = a.out.phases + (size_t)i * 4 * N;
    for (int k = 0; k < 4 * N; k += 2) {
        double p0, p1;
        uniform2_raw(seed, year, cand, 2u, (uint32_t)(k >> 1), p0, p1);
        ph[k] = p0;
        if (k + 1 < 4 * N) ph[k + 1] = p1;
    }
}

}  // namespace tcr
