/*
 * tcrisk_hip.h — C ABI of libtcrisk_hip.so, the MI355X (gfx950) implementation of
 * the per-storm hot path of linjonathan/tropical_cyclone_risk.
 *
 * The reference has no FFI: the seam it exposes is the Python method boundary
 *     Coupled_FAST.init_fields / .gen_track / ._env_winds   (intensity/coupled_fast.py:217-267,
 *                                                            track/bam_track.py:116-128)
 * driven once per candidate seed by util/compute.py:run_tracks (compute.py:134-209).
 * This ABI is the *batched* form of that seam: fields are staged once per
 * (month) slot, then whole batches of storms are integrated per call.  Every
 * entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer it passes;
 *   - functions return 0 on success, <0 on error (tcr_last_error() has the text);
 *   - "host" pointers are ordinary memory, "dev" pointers are HIP device memory
 *     (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void*
 *     (NULL = the context's own stream);
 *   - all floating point is IEEE fp64 (the reference computes in fp64 throughout);
 *   - 2-D planes are [lat][lon] row-major with lat ascending, already cropped to
 *     the basin box exactly as TC_Basin.transform_global_field does
 *     (util/basins.py:57-75): lookups clamp to the cropped grid's edge like
 *     RectBivariateSpline(kx=1,ky=1).ev.
 */
#ifndef TCRISK_HIP_H
#define TCRISK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCR_ABI_VERSION 7
#define TCR_NW 4            /* ua250, va250, ua850, va850  (track/env_wind.py:22-26) */
#define TCR_NCOV 10         /* packed lower triangle (0,0),(1,0),(1,1),(2,0)..(3,3) (env_wind.py:31-42) */
#define TCR_MAX_SERIES 32
#define TCR_N_BASINS 7      /* sorted ids AU,EP,NA,NI,SI,SP,WP (util/compute.py:87) */

/* per-storm status (gen_track's return, intensity/coupled_fast.py:229-267) */
#define TCR_STATUS_GATED (-1)      /* ventilation gate: gen_track returned None (coupled_fast.py:238-244) */
#define TCR_STATUS_FINISHED 0      /* solve_ivp status 0: reached total_time */
#define TCR_STATUS_EVENT 1         /* solve_ivp status 1: tc_dissipates fired (coupled_fast.py:246-256) */
#define TCR_STATUS_STEP_FAIL (-2)  /* solve_ivp status -1: step size underflow */
#define TCR_STATUS_STEP_OVERFLOW (-3) /* more accepted RK steps than tcr_params.max_rk_steps: raise it */

/* flags bit field written by the post-step (util/compute.py:185-209) */
#define TCR_FLAG_IS_TC 1           /* any(v >= 15) and v(2 d) >= 6.5        (compute.py:185-189) */
#define TCR_FLAG_ACCEPTED 2        /* ... and nanmax(vmax) >= 18            (compute.py:205)     */

typedef struct tcr_ctx tcr_ctx;

/* 1-D coordinate pair of a rectilinear grid (strictly increasing, may be non-uniform) */
typedef struct {
    int32_t nlon, nlat;
    const double *lon;      /* host, [nlon] */
    const double *lat;      /* host, [nlat] */
} tcr_grid;

/* namelist scalars the path reads (namelist.py:56-94) + solve_ivp options
 * (coupled_fast.py:264-266) + Coupled_FAST constants (coupled_fast.py:23-27). */
typedef struct {
    double Ck, epsilon, kappa;              /* namelist.py:57; coupled_fast.py:25-26 */
    double u_beta, v_beta;                  /* namelist.py:77-78 */
    double T_Fs;                            /* namelist.T_days * 86400 (bam_track.py:56) */
    double y_alpha[2], m_alpha[2], alpha_max[2], alpha_min[2];   /* namelist.py:73-76 */
    double steering_coefs[2];               /* namelist.py:71 (used when !coupled_track) */
    double dt_out, total_time;              /* namelist.py:48-49 */
    double rtol, atol, max_step;            /* solve_ivp defaults + coupled_fast.py:266 */
    double v_thresh, v_2d_thresh, vmax_thresh;   /* namelist.py:81-83 */
    double v_dissipate;                     /* 4 m/s (coupled_fast.py:255) */
    double earth_R;                         /* util/constants.py:7 */
    double box[4];                          /* basin lon_min, lat_min, lon_max, lat_max (basins.py:42-50) */
    double fs_amp;                          /* sqrt(2 / sum(n^-3)), computed by the host with NumPy */
    double fs_wgt[TCR_MAX_SERIES];          /* n^-1.5 */
    int32_t n_series;                       /* 15 (bam_track.py:112) */
    int32_t n_steps;                        /* int(total_time/dt_out)+1 (bam_track.py:54) */
    int32_t coupled_track;                  /* namelist.py:72 */
    int32_t max_rk_steps;                   /* capacity of the per-storm accepted-step record (0 = 64) */
    /* seeding (util/compute.py:134-175) */
    double seed_v_init;                     /* namelist.py:80 */
    double pi_gate;                         /* 35 m/s (compute.py:168) */
    double lat_vort_fac;                    /* namelist.py:88 */
    double lat_vort_power[TCR_N_BASINS];    /* namelist.py:89-92, sorted-id order */
    double atm_bl_depth[TCR_N_BASINS];      /* namelist.py:85-86, sorted-id order */
    double minit_a, minit_b, minit_c, minit_d;   /* f_mInit = a/(1+exp(-(rh-b)*c))+d (namelist.py:94) */
} tcr_params;

/* A batch of storms; every pointer is [n] unless noted.  Used with host or
 * device memory depending on the entry point. */
typedef struct {
    int64_t n;
    const double *lon0, *lat0, *v0, *m0;    /* gen_track(clon, clat, v, m)  (coupled_fast.py:229) */
    const double *h_bl;                     /* fast.h_bl = atm_bl_depth[basin] (compute.py:175) */
    const int32_t *slot;                    /* field slot = genesis month - 1 (compute.py:151-152) */
    const double *phases;                   /* [n][4][n_series] uniforms of gen_f (bam_track.py:27) */
    /* Optional device scalar (tcr_integrate_dev only; NULL otherwise): only the first min(n, *n_dev) storms
     * exist — the count tcr_compact_dev left on the device — so a round of the accept loop needs no host
     * synchronisation between seeding and integration.  Grids are sized for n; flags of the rows beyond are 0. */
    const int64_t *n_dev;
} tcr_storms;

/* Outputs: the per-storm part of run_tracks' 9-tuple (compute.py:124-133, 210) */
typedef struct {
    double *lon, *lat, *v, *m, *vmax;       /* [n][n_steps], NaN after the track end */
    double *envw;                           /* [n][n_steps][4] (tc_env_wnds) */
    int32_t *n_valid;                       /* emitted samples (len(res.t)) */
    int32_t *status;                        /* TCR_STATUS_* */
    int32_t *flags;                         /* TCR_FLAG_* */
    int32_t *nfev;                          /* RHS evaluations (res.nfev) */
    int32_t *n_accept, *n_reject;           /* accepted steps / rejected attempts */
    /* Optional (may be NULL), [n]: padding state of the plane rows.  The NaN padding is 2/3 of
     * the plane bytes; a caller that reuses the same plane buffers batch after batch passes this
     * array with them, initialised to -1 ("unknown").  On entry pad_state[r] = k >= 0 promises
     * that row r of every plane is already NaN from sample k on, so only [n_valid, k) needs
     * re-padding; on return pad_state[r] = n_valid[r].  The planes come out bit-identical to a
     * full padding.  NULL or -1: the whole tail is written. */
    int32_t *pad_state;
    /* 0: rows are produced for every storm of the batch (what the parity tests compare).
     * 1: rows are produced only where the reference produces them — for candidates that pass accept test 1
     *    (`if is_tc:`, compute.py:190-204): env winds, vmax and the row writes are skipped for the others
     *    (~90 % of a batch); their rows and pad_state entries are left untouched (unspecified contents).
     *    n_valid, status, flags, nfev, n_accept, n_reject are complete in both modes, and the rows of
     *    is_tc storms are bit-identical in both.  tcr_integrate_host always produces every row. */
    int32_t tc_rows_only;
} tcr_tracks;

/* The same outputs in fp32 (BASELINE config 5, "fp32 intensity ODE"): float planes, same layout and meaning. */
typedef struct {
    float *lon, *lat, *v, *m, *vmax;        /* [n][n_steps], NaN after the track end */
    float *envw;                            /* [n][n_steps][4] */
    int32_t *n_valid, *status, *flags, *nfev, *n_accept, *n_reject;
    int32_t *pad_state;
    int32_t tc_rows_only;
} tcr_tracks_f32;

/* ---- lifecycle ----------------------------------------------------------- */
int tcr_abi_version(void);
/* replaces: constructing the 12 Coupled_FAST objects of a year (compute.py:119) */
int tcr_ctx_create(int device, tcr_ctx **out);
int tcr_ctx_destroy(tcr_ctx *ctx);
/* last error text of a context (ctx == NULL: of the last failed tcr_ctx_create) */
const char *tcr_last_error(const tcr_ctx *ctx);

/* replaces: `import namelist` reads scattered through the path */
int tcr_params_set(tcr_ctx *ctx, const tcr_params *p);

/* ---- field staging (once per experiment / year) --------------------------- */
/* replaces: geo.read_bathy / geo.read_land (intensity/geo.py:9-34, coupled_fast.py:30-31) */
int tcr_static_upload(tcr_ctx *ctx, const tcr_grid *hg, const double *land, const double *bathy);
/* the same with the land mask and the bathymetry each on its own grid, as the two independent interpolators
 * f_land / f_bath of the reference allow (intensity/geo.py:9-34); equal grids take the one-grid path */
int tcr_static_upload2(tcr_ctx *ctx, const tcr_grid *land_grid, const double *land,
                       const tcr_grid *bathy_grid, const double *bathy);
/* How the two planes are kept in HBM.  The reference hands float64 arrays to RectBivariateSpline whatever the files hold
 * (intensity/geo.py:15-19, 29-33) — its own land.nc is int8 0 / 1 on a 0.125-degree grid (1440 x 2880).  With pref 0 (auto,
 * the default) an upload inspects the VALUES and stores them narrow when that is exact: land 0 / 1 and whole-metre
 * bathymetry in [-16384, 16383] on one grid -> one uint16 per grid point (mode 2); land in 0 .. 255 and a bathymetry that
 * round-trips through float32 -> eight bytes per grid point on one grid (mode 4), a uint8 and a float plane on two (mode 3);
 * anything else the fp64 planes of pref 1 (mode 0, or 1
 * on two grids).  The kernels widen to the same doubles before any arithmetic, so results do not depend on the mode.
 * Call before tcr_static_upload*.  tcr_static_info reports the mode in use and the bytes the planes occupy.
 * A context may be re-staged with other planes (the mode is chosen again) — not while launches that read them are in flight. */
int tcr_static_store(tcr_ctx *ctx, int32_t pref);
int tcr_static_info(tcr_ctx *ctx, int32_t *mode, int64_t *bytes);
/* replaces: BetaAdvectionTrack._load_wnd_stat (bam_track.py:76-91) +
 *           Coupled_FAST.init_fields (coupled_fast.py:217-225) for one month slot.
 */
int tcr_fields_upload(tcr_ctx *ctx, int slot,
                      const tcr_grid *wg, const double *const mean[TCR_NW], const double *const cov[TCR_NCOV],
                      const tcr_grid *tg, const double *vpot, const double *chi,
                      const double *mld, const double *strat);
/* replaces: m_init_fx[month] = mat.interp2_fx(lon, lat, rh_mid_month) (compute.py:111): mid-level
 * RH of one month slot on the *uncropped* thermo grid, as the reference builds it; only the
 * device-side seeding reads it (initial m, compute.py:173-174). */
int tcr_rh_upload(tcr_ctx *ctx, int slot, const tcr_grid *rg, const double *rh_mid);
/* tcr_fields_upload + tcr_rh_upload of one month slot in ONE transfer (rg / rh_mid may both be NULL).  All three calls copy
 * the planes into pinned memory (tcr_tune.copy_threads host threads), enqueue one asynchronous transfer and the device kernel
 * that interleaves the planes into the slot's layouts, and return: the caller's arrays are free again, and the library waits
 * for the staged fields when they are first used.  Do not stage a slot while launches that read it are still in flight. */
int tcr_slot_upload(tcr_ctx *ctx, int slot,
                    const tcr_grid *wg, const double *const mean[TCR_NW], const double *const cov[TCR_NCOV],
                    const tcr_grid *tg, const double *vpot, const double *chi, const double *mld, const double *strat,
                    const tcr_grid *rg, const double *rh_mid);
/* measurement: host milliseconds the slot uploads of this context have spent since the last reset — [0] waiting for the pinned
 * half a transfer two slots back was still reading, [1] copying planes into pinned memory, [2] enqueuing the transfer,
 * [3] enqueuing the interleave kernel — and [4] the number of uploads */
int tcr_stage_timing(tcr_ctx *ctx, double ms[5], int32_t reset);
/* replaces: the land/<B>.nc interpolators f_b and f_basins (compute.py:87-97).
 * masks are uint8 0/1 planes on one global grid; run_mask is the run basin's. */
int tcr_masks_upload(tcr_ctx *ctx, const tcr_grid *mg, const uint8_t *run_mask,
                     const uint8_t *const basin_masks[TCR_N_BASINS]);

/* ---- the hot path ---------------------------------------------------------- */
/* replaces: the per-candidate body of run_tracks — gen_track (compute.py:176),
 * accept test 1 (:185-189), env-wind recompute (:201-202), axi_to_max_wind
 * (:203-204), accept test 2 (:205) — for a whole batch.  Host buffers. */
int tcr_integrate_host(tcr_ctx *ctx, const tcr_storms *in, const tcr_tracks *out);
/* same with device buffers, asynchronous on `stream`.  Results do not depend on how the library schedules a batch
 * (tcr_tune below; tests/test_gpu_parity.py::test_full_size_ensemble_properties). */
int tcr_integrate_dev(tcr_ctx *ctx, const tcr_storms *in_dev, const tcr_tracks *out_dev, void *stream);

/* Launch shape of batches that do not fill the chip (fewer than 64 storms per SIMD).  The reference has no counterpart: it
 * integrates one storm at a time.  storms_per_lane = 1 (default): one integrator lane per storm, the shortest time to a
 * single batch's results (its longest storm's ~300 sequential evaluations).  k > 1: n / (64 k) persistent waves whose lanes
 * take ~k storms in turn from the batch's queue — about a third of the SIMD time per batch at k = 4 for a ~1.4x longer
 * chain, which is what a caller with many batches in flight on other streams / contexts wants (bench.py, pipelined years).
 * Results do not depend on it. */
int tcr_schedule_set(tcr_ctx *ctx, int32_t storms_per_lane);

/* Launch-shape knobs of a context (the reference has no counterpart).  Results never depend on them.  A context starts from
 * the environment — TCR_WAVES, TCR_PARK, TCR_PARK_FINAL, TCR_TABLE_SEGMENTS, TCR_PRUNE, TCR_EMIT_GRID_CAP, TCR_COPY_THREADS,
 * read ONCE, in tcr_ctx_create; no entry point reads the environment afterwards — and tcr_tune_set replaces all of them.
 * A negative field = the library's own choice. */
typedef struct {
    int32_t waves;            /* persistent integrator waves per launch (default: from the batch size and tcr_schedule_set) */
    int32_t park;             /* tail-compaction threshold of k_integrate's chain of passes; 0 = one launch (default: 12 when a
                                 launch fills the chip, else 0) */
    int32_t park_final;       /* a pass of at most this many waves runs to the end (default 8) */
    int32_t table_segments;   /* 0: the forcing table in one piece instead of a second segment written only for the storms the
                                 first integration pass parks (default 1) */
    int32_t prune;            /* 0: tc_rows_only without the in-flight 2-day test (default 1) */
    int32_t emit_grid_cap;    /* workgroup rows walking the list of storms that pass accept test 1 (default 8192) */
    int32_t copy_threads;     /* host threads that copy a month slot's planes into the pinned staging buffer (default 4) */
    int32_t reserved;
} tcr_tune;
int tcr_tune_set(tcr_ctx *ctx, const tcr_tune *t);
int tcr_tune_get(tcr_ctx *ctx, tcr_tune *t);

/* The fp32 variant of the same path (BASELINE config 5; the reference itself is fp64 throughout, so this is a
 * documented departure with a stated tolerance, see DESIGN.md and profiles/r02_fp32_study.json): fields are
 * converted to fp32 on the device on first use, state / stage derivatives / right-hand side / dense output are
 * fp32, rows come out as fp32; time, the output grid and the step-size controller (error norm, accept test,
 * step factor) stay fp64.  Storm inputs are the same fp64 tcr_storms.  Device buffers, asynchronous on `stream`. */
int tcr_integrate_f32_dev(tcr_ctx *ctx, const tcr_storms *in_dev, const tcr_tracks_f32 *out_dev, void *stream);
int tcr_integrate_f32_host(tcr_ctx *ctx, const tcr_storms *in, const tcr_tracks_f32 *out);

/* replaces: the rejection-sampling seed loop (compute.py:134-175) for candidates
 * [cand0, cand0+n) of `year`, drawn from Philox4x32-10 keyed by
 * (experiment_seed, year, candidate index).  Device buffers (all [n]):
 * lon0, lat0, v0, m0, h_bl, slot, phases as tcr_storms; basin_idx (sorted-id
 * index), seed_flags bit0 = counts toward n_seeds (compute.py:165-167),
 * bit1 = passed the PI gate (compute.py:168).  phases may be NULL (see tcr_gather_seeds_dev). */
typedef struct {
    int64_t n;
    double *lon0, *lat0, *v0, *m0, *h_bl;
    int32_t *slot;
    double *phases;
    int32_t *basin_idx;
    int32_t *seed_flags;
} tcr_seeds;
int tcr_seed_dev(tcr_ctx *ctx, uint64_t experiment_seed, int32_t year, int64_t cand0,
                 const tcr_seeds *out_dev, void *stream);
int tcr_seed_host(tcr_ctx *ctx, uint64_t experiment_seed, int32_t year, int64_t cand0,
                  const tcr_seeds *out_host);

/* ---- order-preserving filters (the batched form of the sequential accept loop) -- */
/* replaces: `while not seed_passed` / `if is_tc` / `nt += 1` control flow of run_tracks
 * (compute.py:135-209): indices i in [0,n) with (flags[i] & mask) != 0, in increasing
 * order, first max_out of them -> idx; *count = how many matched (device scalar). */
int tcr_compact_dev(tcr_ctx *ctx, int64_t n, const int32_t *flags_dev, int32_t mask, int64_t max_out,
                    int32_t *idx_dev, int64_t *count_dev, void *stream);
/* Locality order of a selection (no reference counterpart: the reference integrates one storm at a time): reorders
 * idx_dev[0 .. min(n, *count_dev)) — candidate indices as tcr_compact_dev leaves them — by the cell_deg-degree cell of
 * the candidates' genesis points (latitude row major, stable: candidate order within a cell), so that the storms
 * tcr_gather_seeds_dev then places next to each other, and the integrator runs in one wave, read the same parts of the
 * fields.  Per-storm results do not depend on the order of a batch; a caller that needs candidate order (the accept loop
 * of run_tracks) sorts the few accepted rows back by their candidate index.  Cells that hold many entries (large cells over
 * a small basin; NaN positions, which all sort into cell 0) are sorted as segments, so the worst case stays O(n log^2 n).
 * Like every entry point it works in scratch owned by the context: one stream per context at a time.  (The scan kernel
 * holds 66 KB of LDS: gfx950-class parts.) */
int tcr_cell_order_dev(tcr_ctx *ctx, const tcr_seeds *cand_dev, int32_t *idx_dev, int64_t n, const int64_t *count_dev,
                       double cell_deg, void *stream);
/* dense storm batch dst[r] = src[idx[r]], r < min(n_out, *count_dev) (count_dev, the device scalar
 * written by tcr_compact_dev, may be NULL = all n_out rows are valid).  If src_dev->phases is NULL (tcr_seed_dev was
 * asked not to write them: most candidates never pass) the 4*n_series Fourier phases of the selected
 * candidates are drawn here from the same Philox stream (candidate = cand0 + idx[r]). */
int tcr_gather_seeds_dev(tcr_ctx *ctx, const tcr_seeds *src_dev, const int32_t *idx_dev, int64_t n_out,
                         const int64_t *count_dev, const tcr_seeds *dst_dev, uint64_t experiment_seed,
                         int32_t year, int64_t cand0, void *stream);
/* sums over a finished batch, for throughput accounting and round control: out_dev[0] = storm-steps
 * (sum of max(n_valid-1, 0)), [1] = RHS evaluations, [2] = output samples, [3] = accepted tracks,
 * [4] = storms that passed accept test 1 (is_tc), [5] = output samples of those storms (the rows
 * tc_rows_only produces), [6] = 1 if *n_dev < n (the batch was short of storms: a round control signal that
 * needs no host round trip), [7] = storms counted (min(n, *n_dev)), [8] = storms whose step record overflowed
 * (status TCR_STATUS_STEP_OVERFLOW), [9] = storms the batch had no room for (max(*n_dev - n, 0): passing seeds that
 * tcr_compact_dev clipped).  The counters are uint64 and are ADDED to (zero them first).  n_out is the capacity of
 * out_dev in words — 6, 8 or 10 (TCR_N_STATS): only counters below it are written (ABI v5; v4 wrote eight through an
 * unsized pointer). */
#define TCR_N_STATS 10
int tcr_stats_dev(tcr_ctx *ctx, int64_t n, const int64_t *n_dev, const tcr_tracks *tracks_dev, uint64_t *out_dev,
                  int32_t n_out, void *stream);
/* survivor records for the all-gather of final tracks (compute.py:233-242 concatenation):
 * packed[r] = { lon[ns], lat[ns], v[ns], m[ns], vmax[ns], envw[ns][4] } of track idx[r],
 * r < min(*count_dev, cap); rows are row_stride doubles apart (0 = 9 * ns; larger leaves room for the
 * caller's own columns, e.g. candidate index / month / basin). */
int tcr_pack_tracks_dev(tcr_ctx *ctx, const tcr_tracks *src_dev, const int32_t *idx_dev,
                        const int64_t *count_dev, int64_t cap, double *packed_dev, int64_t row_stride, void *stream);
/* the same from fp32 rows; the packed records are fp64 (the 9-tuple is float64 like the reference's) */
int tcr_pack_tracks_f32_dev(tcr_ctx *ctx, const tcr_tracks_f32 *src_dev, const int32_t *idx_dev,
                            const int64_t *count_dev, int64_t cap, double *packed_dev, int64_t row_stride, void *stream);
/* tcr_pack_tracks_dev with the three columns run_tracks keeps next to a track's rows (compute.py:206-208) appended to
 * every record — packed[r][9*ns + 0..2] = global candidate index (cand0 + cand_idx[j], cand_idx NULL: cand0 + j), month
 * (slot[j] + 1), genesis-basin index (basin_idx[j]) of dense-batch row j = idx[r]; row_stride >= 9*ns + 3.  `f32` != 0:
 * src_dev is a tcr_tracks_f32. */
int tcr_pack_tracks_meta_dev(tcr_ctx *ctx, const tcr_tracks *src_dev, int32_t f32, const int32_t *idx_dev,
                             const int64_t *count_dev, int64_t cap, double *packed_dev, int64_t row_stride,
                             const int32_t *cand_idx_dev, const int32_t *slot_dev, const int32_t *basin_idx_dev,
                             int64_t cand0, void *stream);
/* replaces: `n_seeds[basin_idx, month - 1] += 1` (compute.py:165-167) for a finished round of candidates: out_dev[7][12]
 * (int64, SET) = candidates [cand0, cand0 + cand_dev->n) that count toward n_seeds (seed_flags bit 0) per (genesis basin,
 * month); with cutoff_dev (device scalar, a candidate index held as a double as the survivor records hold it) only the
 * candidates with global index <= *cutoff_dev — the candidate that completed the quota.
 * One stream per context: the reduction goes through a scratch block (partial counts + a ticket word) the CONTEXT owns, like
 * every workspace of the integrator, so two calls of the same context — this one or a tcr_round_dev with seed_hist — must be
 * enqueued on the same stream (or ordered by events); calls of different contexts are independent. */
int tcr_seed_hist_dev(tcr_ctx *ctx, const tcr_seeds *cand_dev, int64_t cand0, const double *cutoff_dev,
                      int64_t *out_dev, void *stream);

/* ---- one round of the accept loop in one call ------------------------------------------------------------------ */
/* replaces: one pass of the body of run_tracks' `while nt < n_tracks` loop (compute.py:134-209) over a block of
 * candidates — and, one level up, the fan-out of run_downscaling (compute.py:223-242), which hands run_tracks calls to
 * dask workers: here a round is ONE library call that enqueues
 *     tcr_seed_dev -> tcr_compact_dev (seeds that passed) -> tcr_cell_order_dev -> tcr_gather_seeds_dev ->
 *     tcr_integrate[_f32]_dev -> tcr_stats_dev -> tcr_compact_dev (accepted tracks) -> tcr_pack_tracks_meta_dev ->
 *     tcr_seed_hist_dev
 * on `stream`, with nothing returning to the host in between; every stage is optional through a NULL buffer.  The results
 * are exactly those of the separate calls, bit for bit (the same arithmetic; the round fuses a few of the small launches:
 * the locality order's cell keys come out of the compaction, the statistics out of the last post-processing kernel).  All
 * pointers are device memory and the caller owns them; they must stay valid until the stream has run the round. */
typedef struct {
    int64_t n_cand;             /* candidates of the round: [cand0, cand0 + n_cand) */
    int64_t n_storms;           /* capacity of the dense batch: the first n_storms passing seeds are integrated */
    tcr_seeds cand;             /* [n_cand] candidate arrays (cand.n is ignored; phases may be NULL, see tcr_seed_dev) */
    tcr_seeds storms;           /* [n_storms] the dense batch (storms.n is ignored) */
    int32_t *cand_idx;          /* [n_storms] position in the candidate block of every dense-batch storm */
    int64_t *n_passed;          /* [1] seeds that passed (may exceed n_storms) */
    double cell_deg;            /* > 0: dense batch in locality order (tcr_cell_order_dev); 0: candidate order */
    int32_t exact_count;        /* 1: integrate min(n_storms, *n_passed) storms (tcr_storms.n_dev); 0: the caller sized the
                                   round so that n_storms seeds pass (short rounds show in stats[6]) */
    int32_t f32;                /* 1: `tracks` is a tcr_tracks_f32 (tcr_integrate_f32_dev) */
    tcr_tracks tracks;          /* [n_storms] outputs */
    uint64_t *stats;            /* [TCR_N_STATS] added to (tcr_stats_dev), or NULL */
    int32_t *acc_idx;           /* [n_storms] dense-batch rows of the accepted tracks, or NULL (then no packing) */
    int64_t *n_accepted;        /* [1] */
    double *packed;             /* [pack_cap][pack_stride] survivor records (tcr_pack_tracks_dev; with the meta columns of
                                   tcr_pack_tracks_meta_dev when pack_stride >= 9 * n_steps + 3), or NULL */
    int64_t pack_cap, pack_stride;
    int64_t *seed_hist;         /* [7][12] the round's n_seeds contribution (tcr_seed_hist_dev, no cutoff), or NULL */
    int64_t n_expected;         /* 0, or how many seeds the caller expects to pass (<= n_storms): the integrator's launch is
                                   shaped for that many storms instead of the batch's capacity; never changes a result */
} tcr_round;
/* use_graph != 0: the round is captured into a hipGraph the first time a (ctx, descriptor) pair is seen and replayed from
 * then on — one graph launch instead of ~20 kernel launches: half the host time of a round (it is not what bounds small
 * rounds: DESIGN.md section 9, round 4).  The graph
 * is keyed by the descriptor's bytes; it is dropped whenever the context allocates or its parameters change.  seed / year /
 * cand0 reach the replayed kernels through a device-side key that a one-thread launch refreshes in front of the graph.
 * The TCR_* scheduling knobs are read when the graph is captured.  Timing events (tcr_timing_enable) are not recorded by
 * replayed rounds.  A context is used from one stream at a time, as for every other entry point.
 * Capture is a process-wide affair in HIP: while a round is being captured (its first use), a legacy-default-stream
 * operation on ANY thread of the process — a synchronous hipMemcpy, a torch op on the default stream — fails there ("would
 * make the legacy stream depend on a capturing blocking stream") and invalidates the capture here; the descriptor then keeps
 * the direct form and no error is returned for it.  The library's own synchronous copies (field staging, host entry
 * points) go through the context's stream for that reason (tools/capture_race_probe.py: 3 700 captures next to threads
 * that stage fields, allocate and grow workspaces, no error); a caller that captures keeps its other threads off the
 * default stream.  The replay saves ~40 us of host time per round and no GPU time. */
int tcr_round_dev(tcr_ctx *ctx, const tcr_round *round, uint64_t experiment_seed, int32_t year, int64_t cand0,
                  int32_t use_graph, void *stream);
/* Stage trace of directly enqueued rounds (measurement aid; replayed rounds record nothing): with it enabled tcr_round_dev
 * records a HIP event behind every stage, and tcr_stage_trace_sum returns, per stage, the summed time from the previous
 * event of the stream to the stage's own — i.e. how long the stream took to get through that stage, waiting included —
 * over the *n_rounds rounds since the trace was enabled (which also resets it). */
enum { TCR_STAGE_START = 0, TCR_STAGE_SEED, TCR_STAGE_SELECT, TCR_STAGE_ORDER, TCR_STAGE_GATHER, TCR_STAGE_FOURIER,
       TCR_STAGE_INTEGRATE, TCR_STAGE_SCREEN, TCR_STAGE_SELECT_TC, TCR_STAGE_DENSE, TCR_STAGE_EMIT, TCR_STAGE_FLAGS,
       TCR_STAGE_STATS, TCR_STAGE_PACK, TCR_N_STAGES };
int tcr_stage_trace_enable(tcr_ctx *ctx, int on);
int tcr_stage_trace_sum(tcr_ctx *ctx, double ms[TCR_N_STAGES], int64_t *n_rounds);
/* graphs this context holds / replays since it was created (test and measurement aid) */
int tcr_round_graph_stats(tcr_ctx *ctx, int64_t *n_graphs, int64_t *n_replays);

/* ---- preprocessing next to the path ----------------------- */
/* replaces: calc_wnd_stat (track/env_wind.py:180-228) for one month: wnd[c] = ua250, va250,
 * ua850, va850 as [n_samples][n_points] planes (points = the flattened lat x lon grid); day_start
 * (NULL: samples are days) holds n_days + 1 sample offsets of the calendar days of a sub-daily
 * record (the reference's groupby("time.day").mean).  out = [14][n_points]: 4 means, then the lower
 * triangle of the covariance row by row, variances with ddof 0 and covariances with ddof 1 exactly as
 * the reference's .var / xr.cov give them.  _dev: device pointers, asynchronous on `stream`. */
int tcr_wind_stats_dev(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const double *const wnd[4],
                       const int32_t *day_start, int32_t n_days, double *out, void *stream);
int tcr_wind_stats_host(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const double *const wnd[4],
                        const int32_t *day_start, int32_t n_days, double *out);
/* The same for float32 planes — the dtype ERA5 u / v files hold and xarray keeps through .mean / .var / xr.cov:
 * sums over time in float32 in time order, divisions through fp64 rounded back to float32, covariances
 * (float32 sum / int64 count) in fp64, every statistic stored as fp64.  Both variants skip NaN samples as
 * xarray's skipna does (per pair for the covariances). */
int tcr_wind_stats_f32_dev(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const float *const wnd[4],
                           const int32_t *day_start, int32_t n_days, double *out, void *stream);
int tcr_wind_stats_f32_host(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const float *const wnd[4],
                            const int32_t *day_start, int32_t n_days, double *out);

/* ---- thermodynamic preprocessing ---------------------------- */
/* replaces: np.load(thermo/entropy_table.npz) in CAPE_PI_vectorized (thermo/thermo.py:272-277):
 * T[np][ns] over ascending pressure (Pa) and entropy (J/kg/K) axes. */
int tcr_entropy_table_upload(tcr_ctx *ctx, int32_t np, int32_t ns, const double *p, const double *s, const double *T);
/* replaces: thermo.CAPE_PI_vectorized(sst, p_surf, p_env, T_env, r_env) (thermo/thermo.py:266-412;
 * select_thermo = 1, select_interp = 2) for n_points columns: p_env[n_lev] in Pa from the lowest level up,
 * T_env / r_env as [n_lev][n_points] planes, ck_over_cd = namelist.Ck / namelist.Cd; pi[n_points] in m/s. */
int tcr_potential_intensity_host(tcr_ctx *ctx, int64_t n_points, int32_t n_lev, const double *p_env,
                                 const double *sst, const double *psl, const double *T_env, const double *r_env,
                                 double ck_over_cd, double *pi);
int tcr_potential_intensity_dev(tcr_ctx *ctx, int64_t n_points, int32_t n_lev, const double *p_env,
                                const double *sst, const double *psl, const double *T_env, const double *r_env,
                                double ck_over_cd, double *pi, void *stream);
/* replaces: thermo.sat_deficit (thermo/thermo.py:92-104, unclipped) and thermo.conv_q_to_rh (:41-46) at the
 * mid level p_mid (Pa), as thermo/calc_thermo.py:66-74 calls them. */
int tcr_chi_rh_host(tcr_ctx *ctx, int64_t n_points, const double *sst, const double *psl, const double *T_mid,
                    const double *q_mid, double p_mid, double *chi, double *rh_mid);

/* ---- gen_track(clon, clat, v, m=None) ---------------------------------------- */
/* replaces: Coupled_FAST._init_m(y, dvdt) (intensity/coupled_fast.py:153-173), which gen_track calls with dvdt = 0
 * when it is given no m (coupled_fast.py:258-261): the inner-core moisture that makes dv/dt equal `dvdt` at t = 0,
 * with the potential intensity taken as the maximum over the point and the four points 0.25 degrees diagonally off
 * it.  m_out[i] = storms->m0[i] where that is a number (or storms->m0 is not NULL and not NaN), else _init_m's value
 * (storms->m0 may be NULL: every storm is initialised).  run_tracks always passes m (compute.py:173-176), so the
 * integrate entry points require m0; a caller that wants the reference's m=None behaviour calls this first and hands
 * m_out to tcr_integrate_* as m0.  Uses lon0, lat0, v0, h_bl, slot and phases (the env winds at t = 0). */
int tcr_init_m_dev(tcr_ctx *ctx, const tcr_storms *storms_dev, double dvdt, double *m_out_dev, void *stream);
int tcr_init_m_host(tcr_ctx *ctx, const tcr_storms *storms_host, double dvdt, double *m_out_host);

/* ---- single-point probes (parity tests of the seam's leaf methods) -------- */
/* (tests) the arithmetic helpers of the integrator on their own (csrc/tcr_device.h, "Arithmetic policy"): out[i] = fn(a[i][, b[i]]),
 * fn 0: a / b without range scaling (qdiv_nz), 1: sqrt (qsqrt), 2: sqrt of a positive argument (qsqrt_pos), 3: a ** (-1/5)
 * (inv_fifth_root: err ** -0.2 of scipy/integrate/_ivp/rk.py:160), 4: a ** -0.4 (strat_pow: t_strat ** -0.4,
 * intensity/coupled_fast.py:91), 5: cos(a) for a in [-pi / 2, pi / 2] (cos_lat: np.cos(np.deg2rad(lat)), track/bam_track.py:139).
 * Host arrays; b may be NULL except for fn 0. */
int tcr_probe_math_host(tcr_ctx *ctx, int32_t fn, int64_t n, const double *a, const double *b, double *out);
/* replaces: Coupled_FAST.dydt (coupled_fast.py:196-207), ._env_winds
 * (bam_track.py:116-128) and ._calc_alpha (coupled_fast.py:65-94) at n points
 * of one slot with one forcing table Fs[4][n_steps] (host buffers). */
int tcr_probe_rhs_host(tcr_ctx *ctx, int slot, double h_bl, const double *Fs, int64_t n,
                       const double *t, const double *lon, const double *lat,
                       const double *v, const double *m,
                       double *dydt /*[n][4]*/, double *envw /*[n][4]*/, double *alpha /*[n]*/);
/* Test instrument of Coupled_FAST._get_over_land (coupled_fast.py:35-38): tcr_integrate_host plus, per storm,
 * one byte per evaluation of dydt in call order — bit0 = `f_land.ev(lon, lat) == 1`, bit1 = the interpolated
 * PI is non-zero, bit2 = the land value is within 1e-12 of 1; 0xff = not evaluated.  dec is [n][cap] host
 * memory.  That decision is taken by rounding in the interior of land, so parity tests compare tracks up
 * to the first evaluation where it lands differently (oracle/parity.py).  Results equal tcr_integrate_host's. */
int tcr_integrate_probe_host(tcr_ctx *ctx, const tcr_storms *in, const tcr_tracks *out, uint8_t *dec, int32_t cap);
/* replaces: gen_f (bam_track.py:23-31): Fs[n][4][n_steps] from phases[n][4][n_series] */
int tcr_fourier_table_host(tcr_ctx *ctx, int64_t n, const double *phases, double *Fs);

/* ---- measurement ----------------------------------------------------------- */
/* HIP-event durations (ms) of the kernels of the last tcr_integrate_dev/_host
 * call on this context: [0] fourier table, [1] integrate, [2] post/unpack.
 * Enabled by tcr_timing_enable(ctx, 1) (which also resets the record); reading
 * waits for the recorded events.  Events are recorded on the launch stream. */
int tcr_timing_enable(tcr_ctx *ctx, int on);
int tcr_timing_last(tcr_ctx *ctx, double ms[3]);
/* sums over every timed call since tcr_timing_enable(ctx, 1); *n_calls = how many */
int tcr_timing_sum(tcr_ctx *ctx, double ms[3], int64_t *n_calls);
/* Occupancy accounting of the last tcr_integrate_* call (synchronises the device first is the
 * caller's job): k_integrate runs as a chain of passes (tail compaction); for each pass p
 * out[6p..6p+5] = { queue requests, storms parked for pass p+1, wave cycles, live-lane cycles,
 * wave wall-clock ticks at 100 MHz, wave shader-clock ticks }.  One cycle = one RK45 step attempt (six evaluations of
 * the reference's ode_rhs, intensity/coupled_fast.py:150-209) of up to 64 storms; live-lane
 * cycles / (64 x wave cycles) is the lane utilisation.  Returns the number of passes. */
int tcr_integrate_pass_stats(tcr_ctx *ctx, int64_t *out, int max_passes);
int tcr_sync(tcr_ctx *ctx, void *stream);

/* ---- multi-GPU: the exchange of final tracks, RCCL over xGMI ---------------------------------------------------- */
/* replaces: the fan-out / collection of run_downscaling (util/compute.py:223-242) — the reference hands run_tracks calls to
 * dask worker processes and gets their 9-tuples pickled back.  Here one process owns one GPU and one context; every rank
 * integrates its share (a block of each round's candidate indices, or whole years) and the survivor records are exchanged with
 * ONE fixed-shape ncclAllGather per round / per run, device to device.  RCCL is loaded at run time (dlopen of librccl.so.1 —
 * the copy the process already holds, e.g. PyTorch's, when there is one): a single-GPU user never needs it.
 *   rank 0: tcr_comm_unique_id(id) -> the 128 bytes travel to the other ranks by any host channel (a file, MPI, a TCP store)
 *   all   : tcr_comm_create(ctx, id, rank, world, &comm)       collective: every rank of the job calls it
 * A communicator borrows its context (error text, device, default stream): destroy it before the context.
 * Calls on one communicator are collective and must be issued in the same order on every rank; they are asynchronous on
 * `stream` (NULL: the context's).  Ragged contributions travel padded to a common row count (`cap`), each rank's real count
 * next to them (tcr_allgather_counts_dev); tcr_concat_rows_dev packs the received blocks in rank order — which is candidate
 * order when ranks own contiguous candidate blocks in rank order (the order the reference's sequential loop meets them). */
#define TCR_COMM_ID_BYTES 128
typedef struct tcr_comm tcr_comm;
int tcr_comm_unique_id(uint8_t id[TCR_COMM_ID_BYTES]);     /* errors: tcr_last_error(NULL) */
int tcr_comm_create(tcr_ctx *ctx, const uint8_t id[TCR_COMM_ID_BYTES], int32_t rank, int32_t world, tcr_comm **out);
int tcr_comm_destroy(tcr_comm *comm);
int tcr_comm_rank(const tcr_comm *comm);
int tcr_comm_world(const tcr_comm *comm);
/* recv_dev[world][bytes] <- every rank's send_dev[bytes] (the same size on every rank) */
int tcr_allgather_dev(tcr_comm *comm, const void *send_dev, void *recv_dev, int64_t bytes, void *stream);
/* gathered_dev[world][n_rows][row_stride] <- every rank's rows_dev[n_rows][row_stride] (survivor records: row_stride =
 * 9 * n_steps (+ 3 meta columns), tcr_pack_tracks[_meta]_dev) */
int tcr_allgather_rows_dev(tcr_comm *comm, const double *rows_dev, int64_t n_rows, int64_t row_stride, double *gathered_dev, void *stream);
/* counts_dev[world] <- every rank's *count_dev */
int tcr_allgather_counts_dev(tcr_comm *comm, const int64_t *count_dev, int64_t *counts_dev, void *stream);
/* buf_dev[n] <- sum over ranks, in place (n_seeds[7][12], compute.py:167; round control) */
int tcr_allreduce_sum_i64_dev(tcr_comm *comm, int64_t *buf_dev, int64_t n, void *stream);
/* out_dev[sum_r min(counts[r], cap)][row_stride] <- the first min(counts[r], cap) rows of every rank's block of
 * gathered_dev[n_blocks][cap][row_stride] (n_blocks = world), in rank order; at most out_cap rows are written;
 * *n_out_dev (optional) = rows written.  Local to the calling rank (no communication). */
int tcr_concat_rows_dev(tcr_ctx *ctx, int32_t n_blocks, const double *gathered_dev, const int64_t *counts_dev, int64_t cap, int64_t row_stride,
                        double *out_dev, int64_t out_cap, int64_t *n_out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TCRISK_HIP_H */
