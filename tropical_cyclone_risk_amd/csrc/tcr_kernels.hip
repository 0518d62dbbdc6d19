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

template <typename R>
struct KArgsT {
    tcr_params P;
    DevFields D;
    EvalKT<R> K;             // host-built evaluation constants, copied to LDS by every workgroup
    int64_t n;
    const int64_t *n_dev;    // optional device scalar (tcr_storms.n_dev): only the first min(n, *n_dev) storms exist
    const double *lon0, *lat0, *v0, *m0, *h_bl;     // storm inputs are fp64 in both instantiations (tcr_storms)
    const int32_t *slot;
    const double *phases;    // [n][4][n_series]
    R *fs;                   // [n][n_steps][4]
    double *srec;            // [n][max_rk_steps][step_rec_doubles<R>()] accepted-step records
    int32_t *n_valid, *status, *nfev, *n_accept, *n_reject;
    unsigned long long *queue;   // [kMaxPasses] work-queue heads, [kMaxPasses] parked-storm counts, then per pass
                                 // {wave cycles, live-lane cycles, wave wall-clock ticks, wave shader-clock ticks} (all zeroed before pass 0)
    int max_rk_steps;
    // multi-pass tail compaction (see k_integrate)
    int pass;                    // 0: storms come fresh from the batch; >0: from the list parked by pass-1
    int threshold;               // park the wave's storms and exit once fewer lanes than this are live (0: run to the end)
    int fill_pct;                // pass > 0: only ceil(items * fill_pct / 100 / 64) waves take part, so that a lane works through
                                 // 100 / fill_pct parked storms in turn instead of one (0 or 100: one lane per parked storm)
    const double *park_in;       // [.. ][kParkRec] list written by the previous pass
    double *park_out;            // list this pass writes
    double t_limit;              // a storm whose next attempt ends beyond this time is parked for the next pass (the forcing table
                                 // behind it is written only for the storms that get there); huge: no limit
    // decision probe (PROBE instantiation only, tcr_integrate_probe_host): probe[storm][probe_cap], one byte
    // per evaluation of fun in call order (Rhs::dec), so tests can find the first `land == 1` decision
    // that lands differently from the oracle's (oracle/parity.py)
    uint8_t *probe;
    int probe_cap;
    // what accept test 1 needs of an accepted step, packed: t_old, h, t_new, v_old, K[0..6][v], - (12 doubles, 96 B);
    // k_screen reads these instead of pulling the whole 288-byte records of every storm through the cache
    double *vrec;                // [n][max_rk_steps][kVRec]
    // TC-rows-only mode: accept test 1 is `any(v >= 15) and v(2 d) >= 6.5`, and when 2 d is an output sample (sample
    // prune_sample, -1: off) v(2 d) is that sample's v.  The step that emits it evaluates it — with dense_at's arithmetic — and a
    // storm that fails is marked in screen_skip[storm]: it cannot become a TC, so k_screen skips it and the integrator stops
    // writing its step records (nothing downstream reads them).
    int prune_sample;
    uint8_t *screen_skip;        // [n]: written for every storm when it ends (1: failed the 2-day test in flight)
    int32_t *und_list;           // storms accept test 1 is still open for when they end (neither gated nor failed in flight), any order
    unsigned long long *und_count;
};
using KArgs = KArgsT<double>;
constexpr int kVRec = 12;
constexpr int kMaxPasses = 16;
constexpr int kParkRec = 18;     // doubles per parked storm: t, h, t_new, ha, g, y[4], f[4], 6 x int32, 0.5 Ck / h_bl, field slot

// ---------------------------------------------------------------------------
// gen_f (track/bam_track.py:23-31), direct form: one thread per (storm, sample),
// 4 series x n_series sines, NumPy's evaluation order 2π·((n·t)/T + x).
// Used when the Fourier period is not a whole number of output intervals.
__device__ __forceinline__ int64_t n_eff(int64_t n, const int64_t *n_dev)
{
    if (!n_dev) return n;
    const int64_t m = *n_dev;
    return m < n ? m : n;
}

// one output sample of the four series: 32 B (fp64) or 16 B (fp32, rounded from the fp64 sums) in one go
template <typename R>
__device__ __forceinline__ void store_fs(R *__restrict__ dst, double a, double b, double c, double d)
{
    typedef typename VecT<R, 4>::type V4;
    V4 v; v[0] = (R)a; v[1] = (R)b; v[2] = (R)c; v[3] = (R)d;
    *reinterpret_cast<V4 *>(dst) = v;
}

template <typename R>
__global__ __launch_bounds__(256) void k_fourier_direct(tcr_params P, int64_t n, const int64_t *__restrict__ n_dev,
                                                         const double *__restrict__ phases,
                                                         R *__restrict__ fs)
{
    const int ns = P.n_steps, N = P.n_series;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_eff(n, n_dev) * ns) return;
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
    store_fs<R>(fs + gid * 4, out[0], out[1], out[2], out[3]);
}

// gen_f, periodic form.  With T_Fs = period * dt_out (the reference's defaults: 20 d
// and 1 h -> 480) the phase of harmonic n at output sample k is
//     2π·(n·k mod period)/period  +  2π·x_{s,n},
// so  sin(.) = S[j]·cos(2πx) + C[j]·sin(2πx)  with j = n·k mod period and (S, C) a
// host-computed (long-double accurate) table of one period.  Per storm that is 60
// sincospi instead of 21 660 sines.  It evaluates the same series the reference
// does; the values differ from NumPy's only by the rounding of NumPy's own argument
// 2π·(n t/T + x) (|Δ| ~ 1e-14, tests/test_gpu_parity.py states the bound).
//
// k_phase_factors: amplitude-weighted phase factors n^-1.5 * (sin, cos)(2π x), laid out
// [storm][harmonic][series] so that one harmonic of a storm is one 64-byte scalar load.
__global__ __launch_bounds__(256) void k_phase_factors(tcr_params P, int64_t n, const int64_t *__restrict__ n_dev,
                                                       const double *__restrict__ phases, double2 *__restrict__ pf)
{
    const int N = P.n_series;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_eff(n, n_dev) * 4 * N) return;
    const int64_t storm = gid / (4 * N);
    const int r = (int)(gid - storm * 4 * N), s = r / N, h = r - s * N;      // phases are [storm][series][harmonic]
    const double x = phases[gid];
    const double wgt = P.fs_wgt[h];
    pf[storm * 4 * N + h * 4 + s] = make_double2(wgt * sinpi(2.0 * x), wgt * cospi(2.0 * x));
}

// k_fourier_periodic: one workgroup per storm; a thread owns three output samples and all four
// series of them, harmonics in the outer loop.  The phase factors of a harmonic are the same for
// the whole workgroup, so they are scalar loads (SGPR operands of the FMAs) and the only LDS
// traffic is one table entry per (sample, harmonic) — the first version read the factors from
// LDS as well and was LDS-issue bound.
#ifndef TCR_FS_THREADS
#define TCR_FS_THREADS 128
#endif
constexpr int kFsThreads = TCR_FS_THREADS;
#ifndef TCR_FS_PER_THREAD
#define TCR_FS_PER_THREAD 3
#endif
constexpr int kFsPerThread = TCR_FS_PER_THREAD;

template <typename R>
__global__ __launch_bounds__(kFsThreads) void k_fourier_periodic(tcr_params P, int64_t n, const int64_t *__restrict__ n_dev,
                                                                  int period,
                                                                  const double2 *__restrict__ sc_table,
                                                                  const double2 *__restrict__ pf,
                                                                  R *__restrict__ fs)
{
    extern __shared__ double2 lds[];            // [period] one period of (sin, cos)
    if ((int64_t)blockIdx.x >= n_eff(n, n_dev)) return;
    const int N = P.n_series, ns = P.n_steps;
    double2 *tab = lds;
    const int64_t storm = blockIdx.x;
    for (int j = threadIdx.x; j < period; j += kFsThreads) tab[j] = sc_table[j];
    __syncthreads();
    // (persistent workgroups that stage the table once were measured slower: 0.51 vs 0.37 ms)
    const double2 *__restrict__ pfs = pf + storm * 4 * N;       // wave-uniform
    R *out = fs + storm * ns * 4;
    for (int base = 0; base < ns; base += kFsThreads * kFsPerThread) {
        // (n * k) mod period, built incrementally and kept as the byte offset of the table entry; the wrap is an
        // unsigned min (j < period ? j : j - period) — three integer instructions per (sample, harmonic) instead of five
        unsigned kk16[kFsPerThread], j16[kFsPerThread];
        const unsigned p16 = (unsigned)period * (unsigned)sizeof(double2);
        double acc[kFsPerThread][4];
#pragma unroll
        for (int u = 0; u < kFsPerThread; ++u) {
            kk16[u] = (unsigned)((base + u * kFsThreads + (int)threadIdx.x) % period) * (unsigned)sizeof(double2);
            j16[u] = 0;
            acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.0;
        }
        for (int h = 0; h < N; ++h) {
            const double2 b0 = pfs[h * 4 + 0], b1 = pfs[h * 4 + 1], b2 = pfs[h * 4 + 2], b3 = pfs[h * 4 + 3];
#pragma unroll
            for (int u = 0; u < kFsPerThread; ++u) {
                j16[u] += kk16[u];
                j16[u] = min(j16[u], j16[u] - p16);
                const double2 a = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(tab) + j16[u]);
                // sin(A + B) = sinA cosB + cosA sinB, accumulated with explicit FMAs (2 per term)
                acc[u][0] = fma(a.x, b0.y, fma(a.y, b0.x, acc[u][0])); acc[u][1] = fma(a.x, b1.y, fma(a.y, b1.x, acc[u][1]));
                acc[u][2] = fma(a.x, b2.y, fma(a.y, b2.x, acc[u][2])); acc[u][3] = fma(a.x, b3.y, fma(a.y, b3.x, acc[u][3]));
            }
        }
#pragma unroll
        for (int u = 0; u < kFsPerThread; ++u) {
            const int k = base + u * kFsThreads + (int)threadIdx.x;
            if (k < ns)
                store_fs<R>(out + (size_t)k * 4, P.fs_amp * acc[u][0], P.fs_amp * acc[u][1], P.fs_amp * acc[u][2], P.fs_amp * acc[u][3]);
        }
    }
}

// k_fourier_mfma: the same table as a matrix product on the fp64 matrix cores.
//
//   fs[storm][sample][series] = amp * sum_h ( S[(h+1) k mod period] * cb[storm][h][series] + C[...] * sb[storm][h][series] )
//
// is Out[(series, storm), sample] = A[(series, storm), 2N] x B[2N, sample] with A the storms' phase factors and B the
// one-period table laid out per (harmonic, sample) — the same B for every storm.  A workgroup of two waves owns a column
// group of 12 tiles = 192 samples (wave w: tiles 6w .. 6w+5 of the group; two groups = 384 samples; group0 selects the
// first group of a launch, so a launch writes the whole table or one of its two segments), keeps its B fragments in
// registers for its whole life (96 VGPRs) and walks row tiles of 4 storms x 4 series: 8 coalesced loads of the A fragment
// (written in fragment order by k_phase_factors_frag), 8 x v_mfma_f64_16x16x4_f64 per column tile on six independent
// accumulator chains, and — rows ordered series-major — a lane ends up with the four series of one (storm, sample) in
// its four accumulator registers: one 32-byte store per lane, 512 contiguous bytes per storm and instruction.
// No LDS, no index arithmetic in the loop; 43 k FMAs per storm run on the matrix pipe (the fp64 matrix peak of gfx950
// equals its vector peak, so the gain is the table reads and integer work of k_fourier_periodic, not a higher ceiling).
// Summation order differs from k_fourier_periodic (k ascending within an MFMA, fused) at the 1e-16 level; the bound
// against NumPy's own evaluation stays the one stated in tests/test_gpu_parity.py.
#ifndef TCR_FS_MFMA
#define TCR_FS_MFMA 1      // 0: k_fourier_periodic (vector FMAs on an LDS table) for every shape
#endif
#ifndef TCR_FS_MFMA_TILES
#define TCR_FS_MFMA_TILES 6
#endif
#ifndef TCR_FS_MFMA_WPS
#define TCR_FS_MFMA_WPS 2
#endif
constexpr int kFsMfmaColTiles = TCR_FS_MFMA_TILES;          // column tiles per wave (6 / 3 / 2: one / two / four workgroups along y share the samples)
#ifndef TCR_FS_MFMA_WAVES
#define TCR_FS_MFMA_WAVES 2
#endif
constexpr int kFsMfmaWaves = TCR_FS_MFMA_WAVES;                       // waves per workgroup (2: two column groups of 12 tiles = the table's two segments; 4: one group)
constexpr int kFsMfmaColGroups = 24 / (kFsMfmaWaves * kFsMfmaColTiles);   // workgroups along y that together cover the 24 column tiles
constexpr int kFsMfmaWgsPerCu = TCR_FS_MFMA_WPS;
static_assert(kFsMfmaColGroups * kFsMfmaWaves * kFsMfmaColTiles == 24, "column tiles per wave must be 6, 3, 2 or 1");
constexpr int kFsMfmaKSteps = 8;            // K = 32 >= 2 * n_series
constexpr int kFsMfmaMaxSamples = 24 * 16;

// A fragments: [row tile of 4 storms][k step][lane]; lane l supplies A[i = l & 15][k = (l >> 4) + 4 * kstep],
// i = series * 4 + storm_in_tile, k = 2 * harmonic + (0: weight * cos 2 pi x, 1: weight * sin 2 pi x); zero padding.
// k_fourier_mfma<LIST>: row r of the product is storm list[r * list_stride], r < *list_count (the second segment of the table, written
// only for the storms the first integration pass parked: `list` points at the storm-id word of k_integrate's park records, stride
// kParkRec); otherwise row r is storm r.
// What a batch's kernels accumulate into, zeroed by the first kernel of the batch that runs anyway (k_phase_factors_frag;
// k_batch_reset when the forcing table takes another path): k_integrate's queue heads / parked counts / occupancy counters
// and — when the 2-day test is decided in flight — flags[0 .. n_flags) and the count of storms accept test 1 is still open for.
struct BatchReset {
    unsigned long long *queue;
    int queue_words;
    unsigned long long *und_count;
    int32_t *flags;
    int64_t n_flags;
    unsigned long long *tc_count;       // TC-rows-only: the length of the list k_screen appends to
};
__device__ __forceinline__ void batch_reset(const BatchReset &z, int64_t i, int64_t stride)
{
    if (!z.queue) return;
    for (int64_t k = i; k < z.queue_words; k += stride) z.queue[k] = 0ull;
    if (i == 0 && z.und_count) *z.und_count = 0ull;
    if (i == 0 && z.tc_count) *z.tc_count = 0ull;
    for (int64_t k = i; k < z.n_flags; k += stride) z.flags[k] = 0;
}
__global__ __launch_bounds__(256) void k_batch_reset(BatchReset z)
{
    batch_reset(z, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

__global__ __launch_bounds__(256) void k_phase_factors_frag(tcr_params P, int64_t n, const int64_t *__restrict__ n_dev,
                                                            const double *__restrict__ phases, double *__restrict__ frag,
                                                            const int64_t *__restrict__ list, const unsigned long long *__restrict__ list_count,
                                                            BatchReset z)
{
    batch_reset(z, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
    const int N = P.n_series;
    const int64_t ne = list ? (int64_t)*list_count : n_eff(n, n_dev);
    const int64_t tiles = (ne + 3) / 4;
    const int r = threadIdx.x, h = r >> 4, i = r & 15;                          // harmonic 0..15, row 0..15 of the tile
    const int s = i >> 2;
    // a workgroup per row tile; the grid is bounded (the row count may live on the device only), so it walks
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = tile * 4 + (i & 3);
        double cb = 0.0, sb = 0.0;
        if (h < N && row < ne) {
            const int64_t storm = list ? list[row] : row;
            const double x = phases[storm * 4 * N + s * N + h];                 // phases are [storm][series][harmonic]
            const double wgt = P.fs_wgt[h];
            sb = wgt * sinpi(2.0 * x); cb = wgt * cospi(2.0 * x);
        }
        const int k = 2 * h;                                                    // k & 3 is 0 or 2: k and k + 1 share a k step
        double *o = frag + tile * (kFsMfmaKSteps * 64) + (k >> 2) * 64 + i;
        o[(k & 3) * 16] = cb;
        o[((k & 3) + 1) * 16] = sb;
    }
}

template <typename R, bool LIST>
__global__ __launch_bounds__(64 * kFsMfmaWaves, TCR_FS_MFMA_WPS) void k_fourier_mfma(tcr_params P, int64_t n, const int64_t *__restrict__ n_dev,
                                                         int period, const double2 *__restrict__ sc_table,
                                                         const double *__restrict__ frag, R *__restrict__ fs, int group0,
                                                         const int64_t *__restrict__ list, int list_stride,
                                                         const unsigned long long *__restrict__ list_count)
{
    typedef double D4 __attribute__((ext_vector_type(4)));
    __shared__ double stage[kFsMfmaWaves][4 * 16 * 4];     // per wave: one output tile, [storm][sample][series]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = P.n_series, ns = P.n_steps;
    const int64_t ne = LIST ? (int64_t)*list_count : n_eff(n, n_dev);      // rows of the product (LIST: row r is storm list[r])
    const int64_t tiles = (ne + 3) / 4;
    // B fragments of this wave's column tiles: lane l supplies B[k = (l >> 4) + 4 * kstep][sample = tile * 16 + (l & 15)]
    double B[kFsMfmaColTiles][kFsMfmaKSteps];
#pragma unroll
    for (int t = 0; t < kFsMfmaColTiles; ++t) {
        const int col = (((blockIdx.y + group0) * kFsMfmaWaves + wave) * kFsMfmaColTiles + t) * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < kFsMfmaKSteps; ++ks) {
            const int k = (lane >> 4) + 4 * ks, h = k >> 1;
            // branch-free so that the 48 gathers of a lane go out together; (h + 1) * col < 2^31
            const bool used = h < N && col < ns;
            const double *e = reinterpret_cast<const double *>(sc_table + (used ? (unsigned)((h + 1) * col) % (unsigned)period : 0u));
            const double v = e[k & 1];              // k even: pairs with the cos factor -> sin entry; odd: cos entry
            B[t][ks] = used ? v : 0.0;
        }
    }
    const double amp = P.fs_amp;
    const int q = lane >> 4, j = lane & 15;
    // Epilogue of one 16 x 16 output tile, cut into eight pieces that are issued *between* the eight MFMAs of the next
    // tile (one MFMA occupies the matrix pipe for 64 cycles; the wave's other instructions fit underneath).
    // D: column = lane & 15 (sample), row = (lane >> 4) + 4 * reg = series * 4 + storm  ->  reg = series, lane >> 4 =
    // storm: a lane holds the four series of one (storm, sample) = 32 contiguous bytes of the table (fp32: 16).
    // fp64: two 16-byte stores per lane straight from the accumulators would each write half of every 32-byte sector
    // (measured: the kernel then runs at the speed of its stores, 0.36 ms for 1.16 GB, MFMAs or not), so the tile goes
    // through a wave-private 2 KB of LDS and leaves as full lines: 16 bytes per lane, 512 contiguous bytes per storm,
    // two storms per instruction.  LDS executes a wave's instructions in order; the wavefront fences only keep the
    // compiler from reordering the staging writes and reads, which are to different addresses of the same lane.
    struct Epilogue {
        D4 v;
        double2 d0, d1;
        int64_t tile;
        int64_t s0, s1;          // storms behind the rows this lane stores (fp64: rows 2 i + (lane >> 5) of the tile; fp32: row lane >> 4)
        int k0;
        bool live;
    } ep;
    ep.live = false; ep.tile = 0; ep.k0 = 0; ep.s0 = ep.s1 = 0; ep.v = D4{0, 0, 0, 0}; ep.d0 = ep.d1 = make_double2(0, 0);
    double *const stg = stage[wave];
    auto epilogue_piece = [&](int piece) {
        if (!ep.live) return;                                   // wave-uniform
#if defined(TCR_FS_ABLATE) && (TCR_FS_ABLATE & 8)
        // timing experiment: the store pattern alone (no scaling, no LDS staging; table contents wrong)
        if (sizeof(R) == 8 && piece != 5 && piece != 6) return;
        if (sizeof(R) == 8) { ep.d0 = make_double2(ep.v[0], ep.v[1]); ep.d1 = make_double2(ep.v[2], ep.v[3]); }
#endif
        if (sizeof(R) == 8) {
            switch (piece) {
            case 0: ep.v[0] = amp * ep.v[0]; ep.v[1] = amp * ep.v[1]; ep.v[2] = amp * ep.v[2]; ep.v[3] = amp * ep.v[3]; break;
            case 1: *reinterpret_cast<D4 *>(&stg[(q * 16 + j) * 4]) = ep.v; break;
            case 2: __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); break;
            case 3: ep.d0 = *reinterpret_cast<const double2 *>(&stg[lane * 2]);
                    ep.d1 = *reinterpret_cast<const double2 *>(&stg[128 + lane * 2]); break;
            case 4: __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); break;
            case 5: case 6: {
                const int i = piece - 5;
                const int64_t row = ep.tile * 4 + 2 * i + (lane >> 5);
                const int64_t storm = i ? ep.s1 : ep.s0;
                const int k = ep.k0 + ((lane & 31) >> 1);
                const double2 d = i ? ep.d1 : ep.d0;
#if defined(TCR_FS_ABLATE) && (TCR_FS_ABLATE & 1)
                if (row < ne && k < ns && d.x == 1.2345e300)    // timing experiment: no stores
#else
                if (row < ne && k < ns)
#endif
                {
                    double *dst = reinterpret_cast<double *>(fs) + (storm * (int64_t)ns + ep.k0) * 4 + (lane & 31) * 2;
#if defined(TCR_FS_NT) && TCR_FS_NT
                    // experiment: streaming stores, so that the 0.95 GB table does not sweep the
                    // field data out of the Infinity Cache on its way to HBM
                    typedef double nt2 __attribute__((ext_vector_type(2)));
                    nt2 v; v[0] = d.x; v[1] = d.y;
                    __builtin_nontemporal_store(v, reinterpret_cast<nt2 *>(dst));
#else
                    *reinterpret_cast<double2 *>(dst) = d;
#endif
                }
                break;
            }
            default: break;
            }
        } else if (piece == 5) {
            const int64_t row = ep.tile * 4 + q;
            const int64_t storm = ep.s0;
            const int k = ep.k0 + j;
#if defined(TCR_FS_ABLATE) && (TCR_FS_ABLATE & 1)
            if (row < ne && k < ns && ep.v[0] == 1.2345e300)
#else
            if (row < ne && k < ns)
#endif
                store_fs<R>(fs + (storm * (int64_t)ns + k) * 4, amp * ep.v[0], amp * ep.v[1], amp * ep.v[2], amp * ep.v[3]);
        }
    };
    // A fragment of row tile `t`.  Plain: the tile's own 4 KB of `frag`, coalesced.  LIST: `frag` holds the fragments of the WHOLE
    // batch (written before the first segment; row = storm), and row r of this product is storm list[r * list_stride] — the lane
    // that supplies (series, storm-in-tile) of the list tile picks the same (series, k) element out of that storm's own tile:
    // no second pass over the phases (k_phase_factors_frag ran for the list again until round 6), the same numbers
    auto load_frag = [&](int64_t t, double (&dst)[kFsMfmaKSteps]) {
        if (LIST) {
            const int i = lane & 15;
            const int64_t r = t * 4 + (i & 3);
            const bool have = t < tiles && r < ne;
            const int64_t storm = have ? list[r * (int64_t)list_stride] : 0;
            const double *src = frag + (storm >> 2) * (kFsMfmaKSteps * 64) + (lane >> 4) * 16 + ((i & ~3) | (int)(storm & 3));
#pragma unroll
            for (int ks = 0; ks < kFsMfmaKSteps; ++ks) { const double v = src[ks * 64]; dst[ks] = have ? v : 0.0; }
        } else {
#pragma unroll
            for (int ks = 0; ks < kFsMfmaKSteps; ++ks) dst[ks] = (t < tiles) ? frag[t * (kFsMfmaKSteps * 64) + ks * 64 + lane] : 0.0;
        }
    };
    double A[kFsMfmaKSteps];
    int64_t tile = blockIdx.x;
    load_frag(tile, A);
    for (; tile < tiles; tile += gridDim.x) {
        // the storms behind this row tile's rows (LIST: loaded here, with the fragment loads, long before the stores need
        // them — a load next to the stores would put a vmcnt(0), i.e. the stores' whole write latency, in front of each)
        const int64_t r0 = tile * 4 + (sizeof(R) == 8 ? (lane >> 5) : q), r1 = r0 + 2;
        int64_t cs0 = r0, cs1 = r1;
        if (LIST) { cs0 = list[(r0 < ne ? r0 : ne - 1) * (int64_t)list_stride]; cs1 = list[(r1 < ne ? r1 : ne - 1) * (int64_t)list_stride]; }
        // next row tile's fragment while this one multiplies
        double An[kFsMfmaKSteps];
        load_frag(tile + gridDim.x, An);
#pragma unroll
        for (int t = 0; t < kFsMfmaColTiles; ++t) {
            const int k0 = (((blockIdx.y + group0) * kFsMfmaWaves + wave) * kFsMfmaColTiles + t) * 16;
            if (k0 >= ns) break;                                // wave-uniform: this wave's last column tiles lie beyond the track
            D4 acc = D4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < kFsMfmaKSteps; ++ks) {
#if defined(TCR_FS_ABLATE) && (TCR_FS_ABLATE & 2)
                acc[ks & 3] += A[ks] * B[t][ks];              // timing experiment: no matrix instructions (values wrong)
#else
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ks], B[t][ks], acc, 0, 0, 0);
#endif
                epilogue_piece(ks);                              // of the previous tile
                __builtin_amdgcn_sched_barrier(0);               // keep this interleaving
            }
            ep.v = acc; ep.tile = tile; ep.k0 = k0; ep.s0 = cs0; ep.s1 = cs1; ep.live = true;
        }
#pragma unroll
        for (int ks = 0; ks < kFsMfmaKSteps; ++ks) A[ks] = An[ks];
    }
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) epilogue_piece(piece);
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
    TCR_FP_FUSE_RK
    // np.linalg.norm(x) / x.size ** 0.5
    return qsqrt<double>(a * a + b * b + c * c + d * d) / 2.0;
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

// np.interp(2 d, res.t, v) between two samples (compute.py:186-188): the fp64 expression on values widened from R
template <typename R>
__device__ __forceinline__ double interp_v2d(R v_lo, R v_hi, double t_lo, double t_hi, double t2d)
{
    return ((double)v_hi - (double)v_lo) / (t_hi - t_lo) * (t2d - t_lo) + (double)v_lo;
}

constexpr int kWave = 64;
#ifndef TCR_INT_WPS
#define TCR_INT_WPS 1      // waves per SIMD the fp64 integrator is register-budgeted for
#endif
#ifndef TCR_INT_WPS_F32
#define TCR_INT_WPS_F32 1  // ... and the fp32 instantiation: budgeted for 2 it spills 248 B per lane and is slower (1.38 vs 1.29 ms per step)
#endif
#ifndef TCR_INT_PIPELINE
#define TCR_INT_PIPELINE 0   // 1: the next stage point's gathers are issued ahead of the intensity half (measured: +4 % time)
#endif
constexpr int kRunning = 99;
constexpr int kIntSlotCache = 32;     // field-slot pointers kept in LDS by k_integrate
template <typename R> constexpr int int_wps() { return sizeof(R) == 8 ? TCR_INT_WPS : TCR_INT_WPS_F32; }
// Register cap of the integrator (build knob for the experiment in DESIGN.md §9): amdgpu_num_vgpr(N) makes the kernel
// allocate 256 + (2N - 256) registers of the SIMD's 512, so that the small kernels of the batches other streams have in
// flight (Fourier table, post-processing: 64-96 registers) could be resident on the same SIMD and issue in the
// integrator's stalls.  Measured: the spills cost more than the overlap gains (N = 224 / 208 / 192: +4 % / 0 / +20 % per step).
#ifdef TCR_INT_NUM_VGPR
#define TCR_INT_CAP __attribute__((amdgpu_num_vgpr(TCR_INT_NUM_VGPR)))
#else
#define TCR_INT_CAP
#endif

// k_integrate: the sequential part of a storm — RK45 steps until the terminal event.
//
// Wave-synchronous cycles: one cycle = one attempt of _step_impl for every lane = six
// evaluations of fun (stages 2..6 and f_new; stage 1 is FSAL).  All lanes evaluate the
// same stage in the same slot, so the per-stage bookkeeping is wave-uniform code instead
// of a divergent state machine.  A lane whose storm ends pulls the next one from the
// queue at the cycle boundary; a fresh storm spends slots 0 and 1 of its first cycle on
// the two evaluations of RungeKutta.__init__ (f0 and select_initial_step's f1) and idles
// for the other four.  Every accepted step leaves one record (doubles t_old, h, t_new, -;
// then R y_old[4], K[7][4]) from which k_emit evaluates the hourly samples.
//
// R = float (BASELINE config 5): state, stage derivatives and the RHS are fp32; what stays fp64 is
// *time* (t, h, t_new, the stage times, the output grid) and the step-size controller: the error norm
// is accumulated in fp64 from the fp32 stage derivatives, and err < 1, the factor 0.9 err^-0.2 and the
// min-step test are the fp64 expressions of the fp64 build (every (double) cast below is the identity there).
template <typename R, bool AFFINE, bool PROBE, int SM>
__global__ __launch_bounds__(kWave, (int_wps<R>())) TCR_INT_CAP void k_integrate(KArgsT<R> a)
{
    // TCR_FUSE_RK (tcr_device.h): stage inputs, y_new, the error norm and the initial-step norms below contract to fmas; the
    // in-flight 2-day sample does not (k_screen must reproduce it bit for bit), nor does anything inlined from another function
    TCR_FP_FUSE_RK
    // Kl[(stage*4 + component)*64 + lane]
    __shared__ R Kl[7 * 4 * kWave];
    __shared__ EvalKT<R> K;
    __shared__ const R *s_wind[kIntSlotCache], *s_thermo[kIntSlotCache];     // month slot -> field planes: a refill reads LDS, not HBM
    const tcr_params &P = a.P;
    const DevFields &D = a.D;
    const int lane = threadIdx.x;
    const long long n_items = a.pass == 0 ? (long long)n_eff(a.n, a.n_dev) : (long long)a.queue[kMaxPasses + a.pass - 1];
    if (a.pass > 0) {                                                       // the list fits the first waves
        const long long lanes = (a.fill_pct > 0 && a.fill_pct < 100) ? (n_items * a.fill_pct + 99) / 100 : n_items;
        if ((long long)blockIdx.x * kWave >= lanes) return;
    }
    unsigned long long *const q_head = a.queue + a.pass;
    for (unsigned w = lane; w < sizeof(EvalKT<R>) / 8; w += kWave)
        reinterpret_cast<uint64_t *>(&K)[w] = reinterpret_cast<const uint64_t *>(&a.K)[w];
    if (lane < kIntSlotCache && lane < D.n_slots) { const DevSlot S = D.slots[lane]; s_wind[lane] = slot_wind<R>(S); s_thermo[lane] = slot_thermo<R>(S); }
    __syncthreads();
    const int ns = P.n_steps;
    const double tb = P.total_time;
    constexpr int REC = step_rec_doubles<R>();
    typedef typename VecT<R, 4>::type V4;
#define KS(j, i) Kl[((j) * 4 + (i)) * kWave + lane]

    // ---- per-lane storm state
    long long sid = -1;
    int status = kRunning, nfev = 0, nacc = 0, nrej = 0, next_out = 0;
    bool active = false, fresh = false, exhausted = false, rejected = false;
    const R *wind = nullptr, *thermo = nullptr;
    const R *fs = nullptr;
    double *srec = nullptr;
    R ck_h = R(0.0);                        // 0.5 Ck / h_bl of the lane's storm (ck_over_h; travels in the park record)
    int cur_slot = 0;                       // the storm's field slot (travels in the park record)
    bool doomed = false;                    // failed the 2-day test already (prune_sample): no more records
    R y[4] = {0, 0, 0, 0}, f[4] = {0, 0, 0, 0}, yn[4] = {0, 0, 0, 0};
    R e[4] = {0, 0, 0, 0};                  // evaluation point: lon, lat, v, m
    double et = 0;                          // ... and its time
    double t = 0, h = 0, ha = 0, h_abs = 0, t_new = 0;
    R g = R(0.0);
    CornerCacheT<R> CC;
    cache_reset(CC);
    RhsPipeT<R, AFFINE, SM> PIPE;
    bool pre = false;                       // the gathers of the point (e, et) are already in flight
    double etn = 0;

    auto finalize = [&]() {
        a.n_valid[sid] = next_out;
        a.status[sid] = status;
        a.nfev[sid] = (status == TCR_STATUS_GATED) ? 0 : nfev;
        a.n_accept[sid] = nacc;
        a.n_reject[sid] = nrej;
        if (a.screen_skip) a.screen_skip[sid] = doomed ? 1 : 0;
        if (a.und_list && !doomed && status != TCR_STATUS_GATED && next_out > 0)
            a.und_list[atomicAdd(a.und_count, 1ull)] = (int32_t)sid;            // k_screen works through this list only
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
            const R dy = R(0.0) + f[i] * R(A10);
            e[i] = y[i] + dy * (R)h;
        }
        et = t + RK_C[1] * h;
    };
    auto begin_step = [&]() {
        const double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        ha = h_abs;
        if (ha > P.max_step) ha = P.max_step;
        else if (ha < min_step) ha = min_step;
        rejected = false;
        attempt_setup();
    };

    // park the storms of the lanes in `mask` (each between two attempts of _step_impl) for the next pass
    auto park = [&](unsigned long long mask) {
        const int leader = __ffsll((long long)mask) - 1;
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(a.queue + kMaxPasses + a.pass, (unsigned long long)__popcll(mask));
        base = __shfl(base, leader);
        if ((mask >> lane) & 1ull) {
            const size_t item = (size_t)(base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull)));
            double2 *o = reinterpret_cast<double2 *>(a.park_out + item * kParkRec);
            const long long c0 = (long long)(unsigned)nfev | ((long long)nacc << 32);
            const long long c1 = (long long)(unsigned)nrej | ((long long)(doomed ? 1 : 0) << 30) | ((long long)(rejected ? 1 : 0) << 31) | ((long long)next_out << 32);
            o[0] = make_double2(t, h); o[1] = make_double2(t_new, ha); o[2] = make_double2((double)g, (double)y[0]);
            o[3] = make_double2((double)y[1], (double)y[2]); o[4] = make_double2((double)y[3], (double)f[0]);
            o[5] = make_double2((double)f[1], (double)f[2]);
            o[6] = make_double2((double)f[3], __longlong_as_double(sid));
            o[7] = make_double2(__longlong_as_double(c0), __longlong_as_double(c1));
            o[8] = make_double2((double)ck_h, __longlong_as_double((long long)cur_slot));
        }
    };
    // occupancy accounting lives in LDS (lane 0 only): the kernel has no register to spare
    __shared__ unsigned long long occ[4];
    if (lane == 0) { occ[0] = 0; occ[1] = 0; occ[2] = wall_clock64(); occ[3] = clock64(); }
#ifdef TCR_INT_PHASE_CLOCKS
    // experiment of DESIGN.md §9: shader clocks per phase of an evaluation slot, lane 0 of block 0 (perturbs the schedule)
    __shared__ unsigned long long ph[8];
    if (lane == 0) for (int i = 0; i < 8; ++i) ph[i] = 0;
    unsigned long long ph_last = clock64();
#define TCR_PHASE_CLK(i) do { if (TCR_INT_PHASE_CLOCKS == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = clock64(); \
        if (lane == 0) { ph[i] += now_ - ph_last; if ((i) == 0) ph[5] += 1; } ph_last = now_; } while (0)
#else
#define TCR_PHASE_CLK(i) do { } while (0)
#endif
    for (;;) {
        // ---- cycle boundary.  First launch of a chain with a segmented forcing table: a storm whose next attempt would read
        // the table beyond the part written so far leaves for the next pass (its lane takes a new storm right away)
        {
            const unsigned long long late = __ballot(active && t_new > a.t_limit);
            if (late) { park(late); if ((late >> lane) & 1ull) active = false; }
        }
        // ---- refill idle lanes from the storm queue (wave-aggregated atomic)
        const unsigned long long want = __ballot(!active && !exhausted);
        if (want) {
            const int leader = __ffsll((long long)want) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(q_head, (unsigned long long)__popcll(want));
            base = __shfl(base, leader);
            if (!active && !exhausted) {
                const long long item = (long long)(base + (unsigned long long)__popcll(want & ((1ull << lane) - 1ull)));
                if (item >= n_items) {
                    exhausted = true;
                } else {
                    // Everything a lane needs to take a storm arrives in ONE round trip behind the queue's: the six
                    // per-storm inputs (or the park record, which carries h_bl and the field slot as well), then the
                    // slot's planes from LDS.  (As five dependent trips — inputs, slot, h_bl, slot table — the refill
                    // cost every cycle of the first pass, which always has a lane to refill, 3-4 of its 42 us.)
                    int slot_id;
                    if (a.pass == 0) {
                        sid = item;
                        const double lo0 = a.lon0[sid], la0 = a.lat0[sid], vv0 = a.v0[sid], mm0 = a.m0[sid], hb0 = a.h_bl[sid];
                        slot_id = a.slot[sid];
                        y[0] = (R)lo0; y[1] = (R)la0; y[2] = (R)vv0; y[3] = (R)mm0; ck_h = ck_over_h<R>(K, (R)hb0);
                        status = kRunning; nfev = 0; nacc = 0; nrej = 0; next_out = 0; doomed = false;
                        t = 0.0;
                        et = 0.0; e[0] = y[0]; e[1] = y[1]; e[2] = y[2]; e[3] = y[3];
                        fresh = true;
                    } else {
                        // restore a parked storm: the state between two attempts of _step_impl
                        const double2 *r = reinterpret_cast<const double2 *>(a.park_in + (size_t)item * kParkRec);
                        const double2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6], r7 = r[7], r8 = r[8];
                        t = r0.x; h = r0.y; t_new = r1.x; ha = r1.y; g = (R)r2.x;
                        y[0] = (R)r2.y; y[1] = (R)r3.x; y[2] = (R)r3.y; y[3] = (R)r4.x;
                        f[0] = (R)r4.y; f[1] = (R)r5.x; f[2] = (R)r5.y; f[3] = (R)r6.x;
                        sid = __double_as_longlong(r6.y);
                        const long long c0 = __double_as_longlong(r7.x), c1 = __double_as_longlong(r7.y);
                        nfev = (int)(c0 & 0xffffffffll); nacc = (int)(c0 >> 32);
                        nrej = (int)(c1 & 0x3fffffffll); doomed = (c1 >> 30) & 1; rejected = (c1 >> 31) & 1; next_out = (int)(c1 >> 32);
                        ck_h = (R)r8.x; slot_id = (int)__double_as_longlong(r8.y);
                        status = kRunning;
                        // stage-2 input exactly as attempt_setup left it
                        for (int i = 0; i < 4; ++i) {
                            KS(0, i) = f[i];
                            const R dy = R(0.0) + f[i] * R(A10);
                            e[i] = y[i] + dy * (R)h;
                        }
                        et = t + RK_C[1] * h;
                        fresh = false;
                    }
                    if (slot_id < kIntSlotCache) { wind = s_wind[slot_id]; thermo = s_thermo[slot_id]; }
                    else { const DevSlot S = D.slots[slot_id]; wind = slot_wind<R>(S); thermo = slot_thermo<R>(S); }
                    cur_slot = slot_id;
                    fs = a.fs + sid * ns * 4;
                    srec = a.srec + sid * (long long)a.max_rk_steps * REC;
                    active = true;
                    cache_reset(CC);
                    // the refill's loads complete here (vmcnt(0); expcnt / lgkmcnt untouched): otherwise their first use
                    // — h_bl, in the intensity half — makes the compiler wait for the *prefetched* gathers issued since
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
            }
        }
        const unsigned long long live_mask = __ballot(active);
        if (a.threshold > 0 && live_mask && !__ballot(fresh) && __popcll(live_mask) < a.threshold) {
            // queue empty (every idle lane tried it) and the wave is mostly idle: park and exit
            park(live_mask);
            break;
        }
        if (!live_mask) break;
        if (lane == 0) { occ[0] += 1; occ[1] += (unsigned long long)__popcll(live_mask); }

        // ---- six evaluation slots
#pragma unroll 1
        for (int slot = 0; slot < 6; ++slot) {
            const bool live = active && !(fresh && slot >= 2);
            RhsT<R> r{};
#if defined(TCR_K_KERNARG)
            const EvalKT<R> &Kq = a.K;           // experiment: constants through scalar loads from the kernel arguments
#elif defined(TCR_OPAQUE_K)
            int koff = 0;
            asm volatile("" : "+s"(koff));      // opaque per iteration: the ~100 EvalK constants stay in LDS instead of being hoisted into registers
            const EvalKT<R> &Kq = *reinterpret_cast<const EvalKT<R> *>(reinterpret_cast<const char *>(&K) + koff);
#else
            const EvalKT<R> &Kq = K;
#endif
            // Software pipeline over the stages of an attempt: the next stage *point* needs only this evaluation's
            // d lon/dt and d lat/dt, so its gathers are issued as soon as those exist and the intensity half of this
            // evaluation runs in their shadow.  Exposed round trips remain at the first slot of a cycle (the point
            // depends on the step-size decision) and for fresh storms.
            TCR_PHASE_CLK(0);
            if (live && !pre) PIPE.issue(CC, Kq, wind, thermo, fs, et, e[0], e[1]);
            TCR_PHASE_CLK(1);
            R th[4] = {0, 0, 0, 0}, lb[2] = {0, 0};
            TrackMidT<R> mid{};
            if (live) PIPE.track(CC, Kq, et, e[0], e[1], e[2], r, mid, th, lb);
            TCR_PHASE_CLK(2);
            // rk_step (rk.py:62-70): K[s] = fun(...); next stage input dy = dot(K[:s].T, a[:s]) * h, one component at a time
            R en[4] = {e[0], e[1], e[2], e[3]};
            auto stage_input = [&](int i) {
                R dy = R(0.0);
                switch (slot) {
                case 0: dy += KS(0, i) * R(A20); dy += KS(1, i) * R(A21); break;
                case 1: dy += KS(0, i) * R(A30); dy += KS(1, i) * R(A31); dy += KS(2, i) * R(A32); break;
                case 2: dy += KS(0, i) * R(A40); dy += KS(1, i) * R(A41); dy += KS(2, i) * R(A42); dy += KS(3, i) * R(A43); break;
                case 3: dy += KS(0, i) * R(A50); dy += KS(1, i) * R(A51); dy += KS(2, i) * R(A52); dy += KS(3, i) * R(A53);
                        dy += KS(4, i) * R(A54); break;
                default:
                    for (int j = 0; j < 6; ++j) dy += KS(j, i) * R(RK_B[j]);
                    break;
                }
                if (slot < 4) en[i] = y[i] + dy * (R)h;
                else { en[i] = y[i] + (R)h * dy; yn[i] = en[i]; }     // y_new (rk.py:68)
            };
            const bool stepping = live && !fresh;
            const R v_e = e[2], m_e = e[3], lat_e = e[1];
            pre = false;
            if (stepping) { KS(slot + 1, 0) = r.d[0]; KS(slot + 1, 1) = r.d[1]; }
            if (stepping && slot < 5) {
                stage_input(0); stage_input(1);
                etn = (slot < 4) ? t + RK_C[slot + 2] * h : t + h;
                if (TCR_INT_PIPELINE) { PIPE.issue(CC, Kq, wind, thermo, fs, etn, en[0], en[1]); pre = true; }
            }
            TCR_PHASE_CLK(3);
            if (live) rhs_intensity<R>(Kq, ck_h, lat_e, v_e, m_e, th, lb, mid, r);
            TCR_PHASE_CLK(4);
            if (PROBE && live) {
                const int ev = fresh ? slot : nfev;          // index of this evaluation in the storm's call order
                if (ev < a.probe_cap) a.probe[(size_t)sid * a.probe_cap + ev] = (uint8_t)r.dec;
            }
            if (stepping) {
                ++nfev;
                const int st = slot + 1;
                KS(st, 2) = r.d[2]; KS(st, 3) = r.d[3];
                if (slot < 5) {
                    stage_input(2); stage_input(3);
                    for (int i = 0; i < 4; ++i) e[i] = en[i];
                    et = etn;
                } else {
                    // f_new is in K[6]: error estimate and step-size control (rk.py:147-165), in fp64
                    double er[4];
                    for (int i = 0; i < 4; ++i) {
                        const double sc = P.atol + fmax(fabs((double)y[i]), fabs((double)yn[i])) * P.rtol;
                        double acc = 0.0;
                        for (int j = 0; j < 7; ++j) acc += (double)KS(j, i) * RK_E[j];
                        er[i] = qdiv_nz<double>(acc * h, sc);
                    }
                    const double err = rms4(er[0], er[1], er[2], er[3]);
                    const double pw = 0.9 * inv_fifth_root<double>(err);
                    if (err < 1) {
                        double fac = (err == 0) ? 10.0 : fmin(10.0, pw);
                        if (rejected && fac > 1) fac = 1;
                        ha *= fac;
                        // step record for k_emit: doubles t_old, h, t_new, -; then R y_old[4], K[7][4]
#ifdef TCR_ABLATE_STEP_STORES
                        if (nacc < 0) {         // timing experiment only: no step records (post-processing reads garbage)
#else
                        if (nacc < a.max_rk_steps && !doomed) {
#endif
                            double *rec = srec + (size_t)nacc * REC;
                            double2 *o = reinterpret_cast<double2 *>(rec);
                            o[0] = make_double2(t, h);
                            o[1] = make_double2(t_new, 0.0);
                            V4 *body = reinterpret_cast<V4 *>(rec + kStepHdr);
                            V4 v; v[0] = y[0]; v[1] = y[1]; v[2] = y[2]; v[3] = y[3];
                            body[0] = v;
                            for (int j = 0; j < 7; ++j) {
                                V4 kk; kk[0] = KS(j, 0); kk[1] = KS(j, 1); kk[2] = KS(j, 2); kk[3] = KS(j, 3);
                                body[1 + j] = kk;
                            }
                            double2 *vr = reinterpret_cast<double2 *>(a.vrec + (sid * (long long)a.max_rk_steps + nacc) * kVRec);
                            vr[0] = make_double2(t, h);
                            vr[1] = make_double2(t_new, (double)y[2]);
                            vr[2] = make_double2((double)KS(0, 2), (double)KS(1, 2));
                            vr[3] = make_double2((double)KS(2, 2), (double)KS(3, 2));
                            vr[4] = make_double2((double)KS(4, 2), (double)KS(5, 2));
                            vr[5] = make_double2((double)KS(6, 2), 0.0);
                        }
                        ++nacc;
                        const double t_old = t;
                        t = t_new;
                        const R v_old = y[2];
                        for (int i = 0; i < 4; ++i) { y[i] = yn[i]; f[i] = r.d[i]; }
                        h_abs = ha;
                        if (t - tb >= 0) status = TCR_STATUS_FINISHED;
                        // terminal event at the step end (ivp.py:673-693); g >= 0 always, so a trigger
                        // is g_new == 0 (root = step end) or g0 == 0 on the first step (root = t0)
                        const R g_new = event_fn<R>(P, y[0], y[1], y[2]);
                        double t_emit = t;
                        if (g == R(0.0)) { status = TCR_STATUS_EVENT; t_emit = t_old; }
                        else if (g_new == R(0.0)) status = TCR_STATUS_EVENT;
                        g = g_new;
                        const int first_out = next_out;
                        next_out = samples_upto(P, t_emit);         // t_eval emission count (ivp.py:706-723)
                        // the sample np.interp(2 d, res.t, v) reads: the 2-day sample itself, or — the track ends before
                        // it — the last one emitted (np.interp clamps), if this final step is the one that emits it
                        int s2d = -1;
                        if (a.prune_sample >= 0 && !doomed) {
                            if (first_out <= a.prune_sample && a.prune_sample < next_out) s2d = a.prune_sample;
                            else if (status != kRunning && next_out <= a.prune_sample && first_out < next_out) s2d = next_out - 1;
                        }
                        if (s2d >= 0) {
                            TCR_FP_EXACT
                            // its v exactly as dense_at / k_screen form it (row 2 of Q = K^T P, x = (t_i - t_old) / h), and
                            // accept test 1's second half on it
                            R Qv[4];
                            for (int k = 0; k < 4; ++k) {
                                R acc = R(0.0);
                                for (int q = 0; q < 7; ++q) acc += KS(q, 2) * R(RK_P[q][k]);
                                Qv[k] = acc;
                            }
                            const R x = (R)((ts_at(P, s2d) - t_old) / h);
                            const R p1 = x, p2 = p1 * x, p3 = p2 * x, p4 = p3 * x;
                            R acc = R(0.0);
                            acc += Qv[0] * p1; acc += Qv[1] * p2; acc += Qv[2] * p3; acc += Qv[3] * p4;
                            const R v2d = (R)h * acc + v_old;
                            doomed = !((double)v2d >= P.v_2d_thresh);
                        }
                        // (a doomed storm writes no records and nobody reads any: it cannot overflow its record — ADVICE r2)
                        if (status == kRunning && !doomed && nacc >= a.max_rk_steps) status = TCR_STATUS_STEP_OVERFLOW;
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
                // S = |shear| of the raw env winds (coupled_fast.py:115-122): only this branch needs it
                const R sdu = r.w[0] - r.w[2], sdw = r.w[1] - r.w[3];
                const R shear = sqrt(sdu * sdu + sdw * sdw);
                if (r.vpot > R(0) && shear * r.chi / r.vpot >= R(1)) {
                    status = TCR_STATUS_GATED;
                    finalize();
                    fresh = false;
                } else {
                    nfev = 1;
                    double sc[4];
                    for (int i = 0; i < 4; ++i) { f[i] = r.d[i]; sc[i] = P.atol + fabs((double)y[i]) * P.rtol; }
                    const double d0 = rms4((double)y[0] / sc[0], (double)y[1] / sc[1], (double)y[2] / sc[2], (double)y[3] / sc[3]);
                    const double d1 = rms4((double)f[0] / sc[0], (double)f[1] / sc[1], (double)f[2] / sc[2], (double)f[3] / sc[3]);
                    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
                    h0 = h0 < tb ? h0 : tb;
                    for (int i = 0; i < 4; ++i) e[i] = y[i] + (R)h0 * R(1.0) * f[i];
                    et = t + h0 * 1.0;
                    h = h0;
                }
            } else if (live && slot == 1) {
                // fresh storm, f1: select_initial_step part 2 (common.py:127-134); first attempt set up
                ++nfev;
                const double h0 = h;
                double sc[4];
                for (int i = 0; i < 4; ++i) sc[i] = P.atol + fabs((double)y[i]) * P.rtol;
                const double d1 = rms4((double)f[0] / sc[0], (double)f[1] / sc[1], (double)f[2] / sc[2], (double)f[3] / sc[3]);
                const double d2 = rms4((double)(r.d[0] - f[0]) / sc[0], (double)(r.d[1] - f[1]) / sc[1],
                                       (double)(r.d[2] - f[2]) / sc[2], (double)(r.d[3] - f[3]) / sc[3]) / h0;
                double h1;
                if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
                else h1 = pow(0.01 / fmax(d1, d2), 0.2);
                h_abs = fmin(fmin(100 * h0, h1), fmin(tb, P.max_step));
                g = event_fn<R>(P, y[0], y[1], y[2]);
                begin_step();
            }
        }
        fresh = false;
    }
#ifdef TCR_INT_PHASE_CLOCKS
    if (lane == 0 && blockIdx.x == 0 && a.pass == 0)
        printf("phase clocks (block 0, pass 0): issue %llu track %llu stage+prefetch %llu intensity %llu bookkeeping %llu; slots %llu\n",
               ph[1], ph[2], ph[3], ph[4], ph[0], ph[5]);
#endif
    if (lane == 0) {       // occupancy accounting of this pass (tcr_integrate_pass_stats)
        unsigned long long *st = a.queue + 2 * kMaxPasses + 4 * a.pass;
        atomicAdd(st + 0, occ[0]);
        atomicAdd(st + 1, occ[1]);
        atomicAdd(st + 2, wall_clock64() - occ[2]);
        atomicAdd(st + 3, (unsigned long long)clock64() - occ[3]);
    }
#undef KS
}

// ---------------------------------------------------------------------------
// Post-processing: everything of run_tracks that is independent per output sample, as three
// loop-free kernels split by register footprint (a fused one-workgroup-per-storm kernel was
// measured first: it ran everything at the occupancy of the gather-heavy part, idled at its
// barriers, and let the compiler hoist ~100 libm constants into registers around its loops):
//   k_dense  wave per storm, lane per (accepted step, component): dense-output matrix Q = K^T P
//            (rk.py:179-181) and the sample -> step map of t_eval emission (ivp.py:706-723);
//   k_emit   thread per sample slot: dense output (rk.py:552-574), the env-wind recompute at
//            every emitted sample (util/compute.py:201-202), axi_to_max_wind (wind/tc_wind.py:6-21
//            with util/sphere.py:15-30,58-83), NaN padding of the reference's [n_tracks][n_steps]
//            planes (compute.py:124-133), "any v >= 15" / "any vmax >= threshold" bits;
//   k_flags  thread per storm: accept tests 1 and 2 (compute.py:185-189, 205).
template <typename R>
struct EArgsT {
    tcr_params P;
    DevFields D;
    int64_t n;
    const int64_t *n_dev;        // optional device scalar: only the first min(n, *n_dev) storms exist; flags of the rest are set to 0
    int max_rk_steps;
    const double *srec;          // [n][max_rk_steps][step_rec_doubles<R>()]
    const double *vrec;          // [n][max_rk_steps][kVRec]: the v part of every step record (k_screen)
    const R *fs;                 // [n][n_steps][4]
    const int32_t *slot;
    const int32_t *n_valid, *status, *n_accept;
    R *lon, *lat, *v, *m, *vmax, *envw;
    int32_t *flags;
    const int32_t *pad_state;    // rows are already NaN from this sample on (NULL / <0: unknown), see tcrisk_hip.h
    // TC-rows-only mode (tcr_tracks.tc_rows_only): the kernels run over list[0 .. *count) — the storms k_screen
    // found to pass accept test 1 — instead of over every storm of the batch (list == NULL)
    const int32_t *list;
    const int64_t *count;
    int64_t item_base;           // k_emit's overflow launch: first list entry it handles
    const uint8_t *screen_skip;  // k_screen: storms the integrator already found to fail the 2-day test (KArgsT::screen_skip; NULL: none)
    const int32_t *und_list;     // k_screen: if set, only these storms are looked at (KArgsT::und_list); flags[] was zeroed beforehand
    const unsigned long long *und_count;
    int32_t *tc_list;            // k_screen (TC rows only): appends the storms that pass accept test 1 ...
    unsigned long long *tc_count;    // ... counted here (zeroed by the batch reset)
    EvalKT<R> K;                 // built on the host; k_emit's small workgroups copy it to LDS
};
using EArgs = EArgsT<double>;
static_assert(sizeof(EvalKT<double>) % 8 == 0 && sizeof(EvalKT<float>) % 8 == 0, "EvalK is copied to LDS in eight-byte words");
constexpr int kBitAny15 = 1 << 8, kBitVmax = 1 << 9;     // scratch bits in flags[] between the kernels
constexpr int kBitListed = 1 << 10;                       // ... and, TC-rows-only: the storm is on the list k_dense / k_emit walked (k_flags finalises exactly these)
#ifndef TCR_POST_THREADS
#define TCR_POST_THREADS 128
#endif
constexpr int kPostThreads = TCR_POST_THREADS;
#ifndef TCR_EMIT_GRID_CAP
#define TCR_EMIT_GRID_CAP 8192
#endif
constexpr int64_t kEmitGridCap = TCR_EMIT_GRID_CAP;      // workgroup rows of k_emit / k_dense over the TC list
constexpr int kEmitSlotCache = 32;      // field-slot wind pointers kept in LDS by k_emit
#ifndef TCR_EMIT_WPS
#define TCR_EMIT_WPS 3     // waves per SIMD k_emit is register-budgeted for (<= 168 VGPRs)
#endif
// Register budget of the post-processing kernels.  An integrator wave owns 256 VGPRs + 164 AGPRs = 424 of its SIMD's 512
// registers (allocated in eights), so a wave of another kernel can be resident NEXT to it — and issue in the 40 % of the
// issue slots the integrator's dependent fp64 chains leave empty — only if it needs at most 88.  k_emit took 90 and
// k_screen 92 (96 allocated: 424 + 96 > 512), i.e. every batch's post-processing had to wait for SIMDs without an
// integrator wave of the three other batches in flight.  Budgeted for 6 waves per SIMD they get at most 80.
// DESIGN.md §9 (round 3) has the measurement.
#ifndef TCR_SHADOW_WPS
#define TCR_SHADOW_WPS 6
#endif

// body of an accepted-step record: R y_old[4], then K[7][4] (k_integrate) / Q[4][4] in its place (k_dense)
template <typename R>
__device__ __forceinline__ const R *rec_body(const double *rec) { return reinterpret_cast<const R *>(rec + kStepHdr); }

// k_dense: one wave per storm, lane = (accepted step j, state component c).  Replaces the seven
// stage derivatives K[q][c] of the step record by row c of the dense-output matrix
// Q = K^T P (rk.py:179-181) — in place: every lane has loaded its K column before any lane
// stores (one wave, lock step) — and tells every hourly sample which step it belongs to.
template <typename R>
__global__ __launch_bounds__(kWave) void k_dense(EArgsT<R> a, uint16_t *__restrict__ sidx)
{
    const tcr_params &P = a.P;
    // work items: the listed storms (TC-rows-only: a bounded grid walks the list) or every row of the batch
    const int64_t n_items = a.list ? *a.count : a.n;
    const int64_t n_exist = a.list ? n_items : n_eff(a.n, a.n_dev);
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        if (item >= n_exist) {
            if (threadIdx.x == 0) a.flags[item] = 0;
            continue;
        }
        const int64_t sid = a.list ? (int64_t)a.list[item] : item;
        const int ns = P.n_steps;
        const int n = a.n_valid[sid];
        int nst = a.n_accept[sid];
        nst = nst < a.max_rk_steps ? nst : a.max_rk_steps;
        if (threadIdx.x == 0) a.flags[sid] = a.list ? kBitListed : 0;
        constexpr int REC = step_rec_doubles<R>();
        typedef typename VecT<R, 4>::type V4;
        const int c = threadIdx.x & 3;
        for (int j0 = 0; j0 < nst; j0 += kWave / 4) {
            const int j = j0 + (threadIdx.x >> 2);
            const bool on = j < nst;
            double *rj = const_cast<double *>(a.srec) + (sid * (int64_t)a.max_rk_steps + (on ? j : 0)) * REC;
            R *body = reinterpret_cast<R *>(rj + kStepHdr);
            R kq[7];
            for (int q = 0; q < 7; ++q) kq[q] = body[4 + q * 4 + c];
            const double t_old = rj[0], t_new = rj[2];
            R Q[4];
            for (int k = 0; k < 4; ++k) {
                R acc = R(0.0);
                for (int q = 0; q < 7; ++q) acc += kq[q] * R(RK_P[q][k]);
                Q[k] = acc;
            }
            // (all loads of the wave are complete here: Q depends on them)
            if (on) {
                V4 qv; qv[0] = Q[0]; qv[1] = Q[1]; qv[2] = Q[2]; qv[3] = Q[3];
                *reinterpret_cast<V4 *>(body + 4 + c * 4) = qv;
                if (c == 0) {
                    // samples of this step: t_old < ts <= t_new (the first step also owns ts = 0)
                    const int i_lo = (j == 0) ? 0 : samples_upto(P, t_old);
                    int i_hi = samples_upto(P, t_new);
                    i_hi = i_hi < n ? i_hi : n;
                    for (int i = i_lo; i < i_hi; ++i) sidx[(size_t)sid * ns + i] = (uint16_t)j;
                }
            }
        }
    }
}

// The two haversine calls of calc_translational_speed (sphere.py:71-76, haversine :15-30) have
// either equal latitudes or equal longitudes.  The vanishing term is sin(0)^2 = 0 (same
// latitude) or cos(lat1)*cos(lat2)*sin(0)^2 = +-0 (same longitude, latitudes finite), and
// x + (+-0) = x for x >= 0, so dropping it — and with it two cosines — is bit-identical.
template <typename R>
__device__ __forceinline__ R haversine_same_lat_km(const tcr_params &P, R lon1, R lon2, R lat)
{
    const R d = R(kPi / 180.0);
    lon1 *= d; lon2 *= d; lat *= d;
    const R sb = sin((lon2 - lon1) / R(2)), c = cos(lat);
    const R aa = R(0.0) + c * c * (sb * sb);
    return R(P.earth_R / 1000.) * (R(2) * asin(sqrt(aa)));
}

template <typename R>
__device__ __forceinline__ R haversine_same_lon_km(const tcr_params &P, R lat1, R lat2)
{
    const R d = R(kPi / 180.0);
    lat1 *= d; lat2 *= d;
    const R sa = sin((lat2 - lat1) / R(2));
    const R aa = sa * sa;
    return R(P.earth_R / 1000.) * (R(2) * asin(sqrt(aa)));
}

// axi_to_max_wind (wind/tc_wind.py:6-21) for one sample, given its neighbours along the track.
// tc_wind.py:17-20 turns (Ui, Vi) into an angle and back: th = arctan2(-Ui, Vi),
// ug = v*(-sin th) + Ui*fac, vg = v*cos th + Vi*fac.  -sin(th) = Ui/|U| and cos(th) = Vi/|U| with
// |U| = sqrt(Ui^2 + Vi^2), which the function has just computed, so the three transcendental calls
// reduce to two divisions (|U| = 0: th = -0, i.e. (0, 1)); differs from libm's round trip by ~1 ulp.
template <typename R>
__device__ __forceinline__ R vmax_at(const tcr_params &P, R lon, R lat, R v, R us, R vs, R lom, R lam, R lop, R lap)
{
    const R dlon = R(0.5) * (sign_of(lop - lom) * haversine_same_lat_km<R>(P, lop, lom, lat));
    const R dlat = R(0.5) * (sign_of(lap - lam) * haversine_same_lon_km<R>(P, lap, lam));
    const R ut = dlon * R(1000.) / (R)P.dt_out, vt = dlat * R(1000.) / (R)P.dt_out;
    const R G = fmin(R(1.), R(0.8) + R(0.35) * (R(1.) + tanh((lat - R(35.)) / R(10.))));
    const R Ui = G * ut + R(0.1) * us * v / R(15.);
    const R Vi = G * vt + R(0.1) * vs * v / R(15.);
    const R mag = sqrt(Ui * Ui + Vi * Vi);
    const R fac = np_min((v * R(0.50)) / mag, R(1.0));
    const R msin = (mag == R(0.0)) ? R(0.0) : Ui / mag;      // -sin(arctan2(-Ui, Vi))
    const R mcos = (mag == R(0.0)) ? R(1.0) : Vi / mag;      //  cos(arctan2(-Ui, Vi))
    const R ug = v * msin + Ui * fac;
    const R vg = v * mcos + Vi * fac;
    return sqrt(ug * ug + vg * vg);
}

// Dense output of a sample at time te inside accepted step `step` of a storm (rk.py:552-574):
// y = y_old + h * Q . (x, x^2, x^3, x^4).  x is formed in fp64 (times are fp64) and rounded to R.
template <typename R, int NC>
__device__ __forceinline__ void dense_at(const double *__restrict__ srec_storm, int step, double te, R (&ye)[NC])
{
    typedef typename VecT<R, 4>::type V4;
    const double *rec = srec_storm + (size_t)step * step_rec_doubles<R>();
    const double2 h0 = *reinterpret_cast<const double2 *>(rec);
    const V4 *body = reinterpret_cast<const V4 *>(rec + kStepHdr);
    const R hh = (R)h0.y;
    const R x = (R)((te - h0.x) / h0.y);
    const R p1 = x, p2 = p1 * x, p3 = p2 * x, p4 = p3 * x;
    const V4 y0 = body[0];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const V4 q = body[1 + c];
        R acc = R(0.0);
        acc += q[0] * p1; acc += q[1] * p2; acc += q[2] * p3; acc += q[3] * p4;
        ye[c] = hh * acc + y0[c];
    }
}

// k_emit: thread per sample slot.  Valid sample: dense output of its step, env winds there,
// vmax from the neighbouring samples (wave shuffles; the two edge lanes of a wave evaluate their
// outside neighbour themselves), planes written; the rest of the row is NaN padding.  The gathers
// make this kernel wait on memory ~60 % of the time, so the transcendental-heavy vmax math of
// the same sample runs in its shadow (as a kernel of its own it cost 0.34 ms per 100k storms).
template <typename R, bool AFFINE, bool LIST>
__global__ __launch_bounds__(kPostThreads, TCR_SHADOW_WPS) void k_emit(EArgsT<R> a, const uint16_t *__restrict__ sidx)
{
    __shared__ EvalKT<R> K_lds;
    __shared__ const R *s_wind[kEmitSlotCache];
    const tcr_params &P = a.P;
    // Work items along x: every existing row, or (TC rows only) the device-side list walked by a bounded grid —
    // a grid sized for the whole batch spent 40 us per 100 000 storms on workgroups that only read the count and left.
    const int64_t n_items = a.list ? *a.count : n_eff(a.n, a.n_dev);
    if ((int64_t)blockIdx.x + (LIST ? a.item_base : 0) >= n_items) return;       // uniform per workgroup
    const int ns = P.n_steps;
    const int i = blockIdx.y * kPostThreads + threadIdx.x;
    const R nan = (R)__longlong_as_double(0x7ff8000000000000LL);
    for (unsigned w = threadIdx.x; w < sizeof(EvalKT<R>) / 8; w += kPostThreads)       // any workgroup size
        reinterpret_cast<uint64_t *>(&K_lds)[w] = reinterpret_cast<const uint64_t *>(&a.K)[w];
    if (threadIdx.x < kEmitSlotCache && (int)threadIdx.x < a.D.n_slots) s_wind[threadIdx.x] = slot_wind<R>(a.D.slots[threadIdx.x]);
    __syncthreads();
    // LIST: this launch walks list entries item_base + blockIdx.x, + gridDim.x, ... (the overflow launch behind a bounded
    // grid); otherwise one entry per workgroup and straight-line code (90 instead of 130 VGPRs: 5 instead of 3 waves per SIMD)
    int64_t item = blockIdx.x + (LIST ? a.item_base : 0);
#pragma unroll 1
    do {
        // per-iteration opaque view of the constants: they stay in LDS instead of being hoisted into registers around the loop
        int koff = 0;
        asm volatile("" : "+s"(koff));
        const EvalKT<R> &K = *reinterpret_cast<const EvalKT<R> *>(reinterpret_cast<const char *>(&K_lds) + koff);
        const int64_t sid = a.list ? (int64_t)a.list[item] : item;
        // the three per-storm / per-sample indices are independent loads: one round trip, not three
        const int n = a.n_valid[sid];
        const int slot_id = a.slot[sid];
        const uint16_t *sidx_storm = sidx + (size_t)sid * ns;
        const int my_step = (i < ns) ? sidx_storm[i] : 0;          // meaningful only if i < n
        const size_t o = (size_t)sid * ns + i;
        typedef typename VecT<R, 4>::type V4;
        const bool valid = i < n;
        const double *srec_storm = a.srec + sid * (int64_t)a.max_rk_steps * step_rec_doubles<R>();
        R ye[4] = {0, 0, 0, 0}, w[4] = {0, 0, 0, 0};
        if (valid) {
            const double te = ts_at(P, i);
    #if defined(TCR_EMIT_ABLATE) && (TCR_EMIT_ABLATE & 1)
            ye[0] = R(-60) + R(0.05) * (R)i; ye[1] = R(15) + R(0.03) * (R)i; ye[2] = R(20); ye[3] = R(0.5) + R(1e-3) * (R)my_step;     // timing experiment: no record gather
    #else
            dense_at<R, 4>(srec_storm, my_step, te, ye);
    #endif
            const R *wind = (slot_id < kEmitSlotCache) ? s_wind[slot_id] : slot_wind<R>(a.D.slots[slot_id]);
    #if defined(TCR_EMIT_ABLATE) && (TCR_EMIT_ABLATE & 2)
            w[0] = ye[0] * R(0.1); w[1] = ye[1] * R(0.1); w[2] = R(1); w[3] = (R)(wind != nullptr);            // timing experiment: no wind / forcing gathers
    #else
            env_winds<R, AFFINE>(K, wind, a.fs + sid * ns * 4, ye[0], ye[1], te, w);
    #endif
            a.lon[o] = ye[0]; a.lat[o] = ye[1]; a.v[o] = ye[2]; a.m[o] = ye[3];
            V4 wv; wv[0] = w[0]; wv[1] = w[1]; wv[2] = w[2]; wv[3] = w[3];
            *reinterpret_cast<V4 *>(a.envw + o * 4) = wv;
        }
        // neighbours along the track: lanes +-1, except across the wave's edges
        const int lane = threadIdx.x & 63;
        R lom = __shfl_up(ye[0], 1), lam = __shfl_up(ye[1], 1);
        R lop = __shfl_down(ye[0], 1), lap = __shfl_down(ye[1], 1);
        if (valid && n > 1) {
            if (lane == 0 && i > 0) {
                R q[2];
                dense_at<R, 2>(srec_storm, sidx_storm[i - 1], ts_at(P, i - 1), q);
                lom = q[0]; lam = q[1];
            }
            if (lane == 63 && i < n - 1) {
                R q[2];
                dense_at<R, 2>(srec_storm, sidx_storm[i + 1], ts_at(P, i + 1), q);
                lop = q[0]; lap = q[1];
            }
            // linear extrapolation at both ends (sphere.py:66-69): the neighbour on the other side is
            // sample 1 / n-2, i.e. lop / lom of this very lane
            const R lop_in = lop, lap_in = lap, lom_in = lom, lam_in = lam;
            if (i == 0) { lom = R(2) * ye[0] - lop_in; lam = R(2) * ye[1] - lap_in; }
            if (i == n - 1) { lop = R(2) * ye[0] - lom_in; lap = R(2) * ye[1] - lam_in; }
            const R vm = vmax_at<R>(P, ye[0], ye[1], ye[2], w[0] - w[2], w[1] - w[3], lom, lam, lop, lap);
            a.vmax[o] = vm;
            const bool hit_v = ye[2] >= (R)P.v_thresh, hit_vm = vm >= (R)P.vmax_thresh;
            const int bits = (__ballot(hit_v) ? kBitAny15 : 0) | (__ballot(hit_vm) ? kBitVmax : 0);
            if (bits && lane == (__ffsll((long long)__ballot(true)) - 1)) atomicOr(a.flags + sid, bits);
        } else if (valid) {                                   // n == 1: no translation speed, vmax stays NaN
            a.vmax[o] = nan;
            if (ye[2] >= (R)P.v_thresh) atomicOr(a.flags + sid, kBitAny15);
        } else {
            // NaN padding, only where the row is not known to be padded already
            int pad_to = a.pad_state ? a.pad_state[sid] : ns;
            pad_to = (pad_to < 0 || pad_to > ns) ? ns : pad_to;
            if (i < pad_to) {
                a.lon[o] = nan; a.lat[o] = nan; a.v[o] = nan; a.m[o] = nan; a.vmax[o] = nan;
                V4 nv; nv[0] = nan; nv[1] = nan; nv[2] = nan; nv[3] = nan;
                *reinterpret_cast<V4 *>(a.envw + o * 4) = nv;
            }
        }
    } while (LIST && (item += gridDim.x) < n_items);
}

// k_screen: accept test 1 (util/compute.py:185-189) without producing a single row.  The reference
// recomputes env winds, computes vmax and writes rows only for candidates that pass this test
// (compute.py:190-204) — ~6 % of the integrated storms — and it needs only the hourly v series:
// `any(v >= 15)` and `np.interp(2 d, res.t, v) >= 6.5`.  So: 16 lanes per storm, lane = accepted step;
// a lane forms row 2 (v) of its step's dense-output matrix Q = K^T P (rk.py:179-181) and walks the
// hourly samples of its step (ivp.py:706-723).  The arithmetic is k_dense's and dense_at's, operation
// for operation, so the v values — and therefore the decision — are bit-identical to what k_emit
// writes for the same storm and what k_flags would decide from those rows.  Writes flags[] only.
constexpr int kScreenThreads = 256;
constexpr int kScreenGroup = 16;                                   // lanes per storm
constexpr int kScreenStorms = kScreenThreads / kScreenGroup;       // storms per workgroup

template <typename R>
__global__ __launch_bounds__(kScreenThreads, TCR_SHADOW_WPS) void k_screen(EArgsT<R> a)
{
    __shared__ R cap[kScreenStorms][3];      // v at sample j2d, j2d + 1, n - 1
    const tcr_params &P = a.P;
    const int g = threadIdx.x / kScreenGroup, l = threadIdx.x % kScreenGroup;
    const int64_t item = (int64_t)blockIdx.x * kScreenStorms + g;
    if (a.und_list && (int64_t)blockIdx.x * kScreenStorms >= (int64_t)*a.und_count) return;      // uniform per workgroup
    const bool listed = a.und_list && item < (int64_t)*a.und_count;
    const int64_t sid = a.und_list ? (listed ? (int64_t)a.und_list[item] : 0) : item;
    const bool on = a.und_list ? listed : sid < a.n;                       // writes a flag
    const bool exists = a.und_list ? listed : sid < n_eff(a.n, a.n_dev);   // has a storm behind it
    const int ns = P.n_steps;
    int n = 0, nst = 0, st = TCR_STATUS_GATED;
    if (exists) {
        n = a.n_valid[sid]; nst = a.n_accept[sid]; st = a.status[sid];
        nst = nst < a.max_rk_steps ? nst : a.max_rk_steps;
        if (a.screen_skip && a.screen_skip[sid]) n = 0;        // not a TC whatever else happened; its records stop at day 2
    }
    // np.interp(2 d, res.t, v): which samples it reads (compute.py:186-188; same as k_flags)
    const double step_out = P.total_time / (double)(ns - 1);
    const double t2d = 2 * 86400.0;
    const bool clamp2d = n > 0 && t2d >= ts_at(P, n - 1);
    const int j2d = clamp2d ? n - 1 : (int)floor(t2d / step_out);
    bool any15 = false;
    const double *vrec_storm = a.vrec + sid * (int64_t)a.max_rk_steps * kVRec;
    // One accepted step of the storm: the v row of Q = K^T P and the samples the step emits
    struct StepV {
        double t_old, h64;
        R hh, y0, Q[4];
        int i_lo, i_hi;
    };
    auto load_step = [&](int j) {
        // (values were stored widened to fp64; narrowing back gives the R values k_dense reads from the full record)
        const double2 *rj = reinterpret_cast<const double2 *>(vrec_storm + (size_t)j * kVRec);
        const double2 r0 = rj[0], r1 = rj[1], r2 = rj[2], r3 = rj[3], r4 = rj[4], r5 = rj[5];
        StepV s;
        s.t_old = r0.x; s.h64 = r0.y;
        const double t_new = r1.x;
        s.hh = (R)s.h64; s.y0 = (R)r1.y;
        const R kq[7] = {(R)r2.x, (R)r2.y, (R)r3.x, (R)r3.y, (R)r4.x, (R)r4.y, (R)r5.x};
        for (int k = 0; k < 4; ++k) {
            R acc = R(0.0);
            for (int q = 0; q < 7; ++q) acc += kq[q] * R(RK_P[q][k]);
            s.Q[k] = acc;
        }
        s.i_lo = (j == 0) ? 0 : samples_upto(P, s.t_old);
        s.i_hi = samples_upto(P, t_new);
        s.i_hi = s.i_hi < n ? s.i_hi : n;
        return s;
    };
    // dense output of v at sample i, exactly as dense_at forms it: x = (t_i - t_old) / h
    auto v_at = [&](const StepV &s, int i) -> R {
        const R x = (R)((ts_at(P, i) - s.t_old) / s.h64);
        const R p1 = x, p2 = p1 * x, p3 = p2 * x, p4 = p3 * x;
        R acc = R(0.0);
        acc += s.Q[0] * p1; acc += s.Q[1] * p2; acc += s.Q[2] * p3; acc += s.Q[3] * p4;
        return s.hh * acc + s.y0;
    };
    // ---- phase 1: the 2-day test (three exact samples at most) — most storms fail it, and those need no more
    auto caps = [&](const StepV &s) {
        if (s.i_lo <= j2d && j2d < s.i_hi) cap[g][0] = v_at(s, j2d);
        if (s.i_lo <= j2d + 1 && j2d + 1 < s.i_hi) cap[g][1] = v_at(s, j2d + 1);
        if (s.i_lo <= n - 1 && n - 1 < s.i_hi) cap[g][2] = v_at(s, n - 1);
    };
#if defined(TCR_SCREEN_ABLATE) && TCR_SCREEN_ABLATE == 2
    n = 0;
#endif
    const bool has0 = l < nst && n > 0;
    StepV s0{};                                          // the lane's first step stays in registers for phase 2
    if (has0) { s0 = load_step(l); caps(s0); }
    for (int j = l + kScreenGroup; j < nst && n > 0; j += kScreenGroup) caps(load_step(j));
    __syncthreads();
    bool pass2d = false;
    if (n > 0 && st != TCR_STATUS_GATED) {
        double v2d;
        if (clamp2d) v2d = (double)cap[g][2];
        else v2d = interp_v2d<R>(cap[g][0], cap[g][1], ts_at(P, j2d), ts_at(P, j2d + 1), t2d);
        pass2d = v2d >= P.v_2d_thresh;
    }
#if defined(TCR_SCREEN_ABLATE)
    pass2d = false;
#endif
    // ---- phase 2: `any(v >= v_thresh)` over the hourly samples, only for the storms that passed.
    // A step whose dense output cannot reach the threshold anywhere is skipped without looking at its samples:
    // on x in [0, 1] every term Q_k x^(k+1) is at most max(Q_k, 0), so v <= y0 + |h| sum max(Q_k, 0) (h > 0 here); the
    // margin (1e-3 m/s; fp32: 1e-2) is orders of magnitude above the rounding of either side.  A step that can
    // reach it is walked until its first hit.  Each sample is first judged with x = (t_i - t_old) * (1 / h) and
    // Horner's form — within 1e-13 (fp32: 1e-5) of the exact value — and only a sample that lands within `band` of the
    // threshold is evaluated exactly as dense_at does (a wave-uniform branch the compiler cannot turn into
    // straight-line code): the decision is the exact one in every case.
    const R thr = (R)P.v_thresh, band = sizeof(R) == 8 ? R(1e-9) : R(2e-3), margin = sizeof(R) == 8 ? R(1e-3) : R(1e-2);
    auto scan = [&](const StepV &s) {
        const R z = R(0.0);
        const R up = (s.Q[0] > z ? s.Q[0] : z) + (s.Q[1] > z ? s.Q[1] : z) + (s.Q[2] > z ? s.Q[2] : z) + (s.Q[3] > z ? s.Q[3] : z);
        const R ub = s.y0 + fabs(s.hh) * up;
        if (ub < thr - margin) return;                  // NaNs fall through to the walk
        const double rh = 1.0 / s.h64;
        for (int i = s.i_lo; i < s.i_hi && !any15; ++i) {
            const R x = (R)((ts_at(P, i) - s.t_old) * rh);
            const R v_f = s.hh * (x * (s.Q[0] + x * (s.Q[1] + x * (s.Q[2] + x * s.Q[3])))) + s.y0;
            bool hit = v_f >= thr + band;
            const bool amb = !hit && v_f >= thr - band;
            if (__ballot(amb)) {
                asm volatile("" ::: "memory");
                if (amb) hit = v_at(s, i) >= thr;
            }
            any15 = any15 || hit;
        }
    };
    if (pass2d) {
        if (has0) scan(s0);
        for (int j = l + kScreenGroup; j < nst; j += kScreenGroup) scan(load_step(j));
    }
    int hit = any15 ? 1 : 0;
    for (int off = kScreenGroup / 2; off > 0; off >>= 1) hit |= __shfl_xor(hit, off);       // every lane takes part
    any15 = hit != 0;
    if (on && l == 0) {
        const bool tc = any15 && pass2d;
        a.flags[sid] = tc ? TCR_FLAG_IS_TC : 0;
        // the list k_dense / k_emit walk is appended to here (a compaction launch of its own until round 4; the list's order does
        // not matter: every consumer works storm by storm)
        if (tc && a.tc_list) a.tc_list[atomicAdd(a.tc_count, 1ull)] = (int32_t)sid;
    }
}

// One thread per storm of the batch.  TC-rows-only mode (tc_list): the storms k_screen passed (exactly the
// entries of the list k_dense / k_emit walked, marked kBitListed by k_dense) are finalised, the others keep their 0.  stats != NULL: the sums of
// tcr_stats_dev over the batch are accumulated by the same launch (a round needs no k_stats dispatch; counters as there).
template <typename R>
__global__ __launch_bounds__(256) void k_flags(tcr_params P, int64_t n_storms, const int32_t *__restrict__ n_valid,
                                               const int32_t *__restrict__ status, const R *__restrict__ pv,
                                               int32_t *__restrict__ flags, int32_t *__restrict__ pad_state,
                                               int tc_list, const int64_t *__restrict__ n_dev,
                                               const int32_t *__restrict__ nfev, unsigned long long *__restrict__ stats,
                                               const int64_t *__restrict__ stats_n_dev)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ne = n_eff(n_storms, n_dev);
    int fl = 0, n = 0;
    const bool exists = gid < ne;
    if (gid < n_storms) {
        const int bits = exists ? flags[gid] : 0;
        const bool work = exists && (!tc_list || (bits & kBitListed));
        const int64_t sid = gid;
        n = exists ? n_valid[sid] : 0;
        if (work) {
            const int ns = P.n_steps;
            if (pad_state) pad_state[sid] = n;         // every k_emit block of the row has read the old value
            if (n > 0 && status[sid] != TCR_STATUS_GATED) {
                // np.interp(2 d, res.t, v) (compute.py:186-188)
                const double step_out = P.total_time / (double)(ns - 1);
                const double t2d = 2 * 86400.0;
                const R *v = pv + (size_t)sid * ns;
                double v2d;
                if (t2d >= ts_at(P, n - 1)) v2d = (double)v[n - 1];
                else {
                    const int j = (int)floor(t2d / step_out);
                    v2d = interp_v2d<R>(v[j], v[j + 1], ts_at(P, j), ts_at(P, j + 1), t2d);
                }
                if ((bits & kBitAny15) && v2d >= P.v_2d_thresh) {
                    fl |= TCR_FLAG_IS_TC;
                    if (n > 1 && (bits & kBitVmax)) fl |= TCR_FLAG_ACCEPTED;
                }
            }
        }
        if (work || !tc_list || !exists) flags[gid] = fl;          // (TC-rows-only: storms off the list already hold 0)
    }
    if (!stats) return;
    // ---- tcr_stats_dev's sums (k_stats), same counters, over the first min(n, *stats_n_dev) storms
    const int64_t nc = n_eff(n_storms, stats_n_dev);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (stats_n_dev && *stats_n_dev < n_storms) atomicAdd(stats + 6, 1ull);
        atomicAdd(stats + 7, (unsigned long long)(nc > 0 ? nc : 0));
        if (stats_n_dev && *stats_n_dev > n_storms) atomicAdd(stats + 9, (unsigned long long)(*stats_n_dev - n_storms));
    }
    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, bad = 0;
    if (gid < nc) {
        if (!exists) n = n_valid[gid];
        a0 = n > 1 ? n - 1 : 0; a1 = nfev[gid]; a2 = n;
        a3 = (fl & TCR_FLAG_ACCEPTED) ? 1 : 0; a4 = (fl & TCR_FLAG_IS_TC) ? 1 : 0; a5 = (fl & TCR_FLAG_IS_TC) ? n : 0;
        bad = status[gid] == TCR_STATUS_STEP_OVERFLOW ? 1 : 0;
    }
    __shared__ unsigned long long sh[4][7];
    for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off); a2 += __shfl_down(a2, off); a3 += __shfl_down(a3, off);
        a4 += __shfl_down(a4, off); a5 += __shfl_down(a5, off); bad += __shfl_down(bad, off);
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        sh[w][0] = a0; sh[w][1] = a1; sh[w][2] = a2; sh[w][3] = a3; sh[w][4] = a4; sh[w][5] = a5; sh[w][6] = bad;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const unsigned long long v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        if (v) atomicAdd(stats + (threadIdx.x < 6 ? threadIdx.x : 8), v);
    }
}

// Coupled_FAST._init_m(y, dvdt) (intensity/coupled_fast.py:153-173): the inner-core moisture gen_track(m=None) starts
// from (:258-261, dvdt = 0) — the m that makes dv/dt equal `dvdt` at t = 0, with PI taken as the maximum over the
// point and the four points 0.25 degrees diagonally off it.  One thread per storm; storms whose m0 is a number keep it.
template <bool AFFINE, int SM>
__global__ __launch_bounds__(64) void k_init_m(tcr_params P, DevFields D, EvalK K_host, int64_t n, const int64_t *__restrict__ n_dev,
                                               const double *__restrict__ lon0, const double *__restrict__ lat0,
                                               const double *__restrict__ v0, const double *__restrict__ m0,
                                               const double *__restrict__ h_bl, const int32_t *__restrict__ slot,
                                               const double *__restrict__ phases, double dvdt, double *__restrict__ m_out)
{
    __shared__ EvalK K;
    if (threadIdx.x == 0) K = K_host;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_eff(n, n_dev)) return;
    if (m0 && m0[i] == m0[i]) { m_out[i] = m0[i]; return; }
    const double lon = lon0[i], lat = lat0[i], v = v0[i];
    const DevSlot S = D.slots[slot[i]];
    // self.Fs_i(0): the forcing series at t = 0 (bam_track.py:23-31 with t = 0; interp1d at its first knot)
    double F[4];
    {
        const int N = P.n_series;
        const double two_pi = 2. * kPi;
        const double *ph = phases + i * 4 * N;
        for (int sidx = 0; sidx < 4; ++sidx) {
            double acc = 0.0;
            for (int k = 0; k < N; ++k) {
                const double arg = two_pi * (((double)(k + 1) * 0.0) / P.T_Fs + ph[sidx * N + k]);
                const double term = P.fs_wgt[k] * sin(arg);
                acc = (k == 0) ? term : acc + term;
            }
            F[sidx] = P.fs_amp * acc;
        }
    }
    // _calc_steering_coefs(v), _step_bam_track(lon, lat, 0, coefs) -> v_bam (coupled_fast.py:155-156)
    RhsT<double> r{};
    TrackMidT<double> mid{};
    {
        const Cell cx = locate_t<double, AFFINE>(K.wx, lon), cy = locate_t<double, AFFINE>(K.wy, lat);
        CornersT<double, 14, 2> CW;
        gather<double, 14, kWindStride, 2>(S.wind, K.wx.n, cx, cy, CW);
        double q[14];
        blend<double, 14, 2>(CW, cx, cy, q);
        winds_from_lookups<double>(q, F, lon, 0.0, r.w);
        rhs_track<double>(K, lat, v, r, mid);
    }
    // _get_current_vpot at a point (coupled_fast.py:35-58)
    auto vpot_at = [&](double x, double y) {
        StaticLookup<double, AFFINE, SM> SL;
        SL.issue(K, x, y);
        const Cell tx = locate_t<double, AFFINE>(K.tx, x), ty = locate_t<double, AFFINE>(K.ty, y);
        CornersT<double, 4, 2> CT;
        gather<double, 4, kThermoStride, 2>(S.thermo, K.tx.n, tx, ty, CT);
        double th[4], lb[2];
        blend<double, 4, 2>(CT, tx, ty, th);
        SL.finish(lb);
        return (lb[0] == 1.0) ? 0.0 : th[0];
    };
    // np.max over the five points: NaN propagates (coupled_fast.py:157-161)
    double vp = vpot_at(lon, lat);
    const double dx[4] = {-0.25, -0.25, 0.25, 0.25}, dy[4] = {-0.25, 0.25, -0.25, 0.25};
    for (int k = 0; k < 4; ++k) {
        const double c = vpot_at(lon + dx[k], lat + dy[k]);
        vp = (vp != vp) ? vp : ((c != c) ? c : (c > vp ? c : vp));
    }
    // alpha = _calc_alpha(lon, lat, v_bam, v) (coupled_fast.py:65-94): the intensity half of the RHS has it
    double th[4], lb[2];
    {
        StaticLookup<double, AFFINE, SM> SL;
        SL.issue(K, lon, lat);
        const Cell tx = locate_t<double, AFFINE>(K.tx, lon), ty = locate_t<double, AFFINE>(K.ty, lat);
        CornersT<double, 4, 2> CT;
        gather<double, 4, kThermoStride, 2>(S.thermo, K.tx.n, tx, ty, CT);
        blend<double, 4, 2>(CT, tx, ty, th);
        SL.finish(lb);
    }
    rhs_intensity<double>(K, ck_over_h<double>(K, h_bl[i]), lat, v, 0.5, th, lb, mid, r);
    const double al = r.alpha;
    const double gamma = K.epsilon + al * K.kappa;
    const double beta = 1 - K.epsilon - K.kappa;
    const double numer = 2 * h_bl[i] / K.Ck * dvdt + (v * v);
    const double denom = al * beta * (vp * vp) + gamma * (v * v);
    const double c = cbrt(numer / denom);
    m_out[i] = np_max(np_min(c, 1.0), 0.0);
}

template <bool AFFINE, int SM>
__global__ __launch_bounds__(64) void k_probe_rhs(tcr_params P, DevFields D, EvalK K_host, int slot, double h_bl, const double *fs,
                            int64_t n, const double *t, const double *lon, const double *lat,
                            const double *v, const double *m, double *dydt, double *envw, double *alpha)
{
    __shared__ EvalK K;
    if (threadIdx.x == 0) K = K_host;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevSlot S = D.slots[slot];
    const RhsT<double> r = rhs_eval<double, AFFINE, SM>(K, S.wind, S.thermo, fs, h_bl, t[i], lon[i], lat[i], v[i], m[i]);
    for (int k = 0; k < 4; ++k) dydt[i * 4 + k] = r.d[k];
    alpha[i] = r.alpha;
    double w[4];
    env_winds<double, AFFINE>(K, S.wind, fs, lon[i], lat[i], t[i], w);
    for (int k = 0; k < 4; ++k) envw[i * 4 + k] = w[k];
}

// The arithmetic helpers of tcr_device.h's policy on their own (tcr_probe_math_host: tests pin their accuracy against NumPy)
__global__ __launch_bounds__(256) void k_probe_math(int fn, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                    double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b ? b[i] : 0.0;
    double r = 0.0;
    switch (fn) {
    case 0: r = qdiv_nz<double>(x, y); break;
    case 1: r = qsqrt<double>(x); break;
    case 2: r = qsqrt_pos<double>(x); break;
    case 3: r = inv_fifth_root<double>(x); break;
    case 4: r = strat_pow<double>(x); break;
    case 5: r = cos_lat<double>(x); break;
    default: break;
    }
    out[i] = r;
}

}  // namespace tcr
