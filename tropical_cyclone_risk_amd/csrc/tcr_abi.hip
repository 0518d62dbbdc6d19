// C-ABI host side of libtcrisk_hip.so (see include/tcrisk_hip.h for the contract
// and the reference interface each entry point replaces).  Owns: the HIP stream,
// the staged field sets in HBM (interleaved layouts of tcr_device.h), grow-only
// workspaces (forcing tables, track records), and kernel launches.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tcr_kernels.hip"
#include "tcr_seed.hip"
#include "tcr_compact.hip"
#include "tcr_prep.hip"
#include "tcr_thermo.hip"
#ifdef TCR_EXPERIMENTS
#include "tcr_experiments.h"       // scheduling probes and per-call environment knobs: tools/build_variant.py NAME -DTCR_EXPERIMENTS
#endif

using namespace tcr;

namespace {

thread_local std::string g_create_error;

struct GridStore {
    std::vector<double> lon, lat;
    double *d_lon = nullptr, *d_lat = nullptr, *d_rlon = nullptr, *d_rlat = nullptr;
    bool affine_lon = false, affine_lat = false;
    bool set = false;
    // fp32 view of the same grid (built on first use of an fp32 entry point): knots rounded to float,
    // reciprocal widths and the affine test redone in float arithmetic
    std::vector<float> lon32, lat32;
    float *d_lon32 = nullptr, *d_lat32 = nullptr, *d_rlon32 = nullptr, *d_rlat32 = nullptr;
    bool affine_lon32 = false, affine_lat32 = false;
    bool set32 = false;
};

// An axis is "affine" when x0 + i*dx reproduces every knot bit for bit and every cell's
// reciprocal width 1.0/(x[i+1]-x[i]) is the same double; the kernels then never load knots.
template <typename R>
bool axis_is_affine(const std::vector<R> &x)
{
    const int n = (int)x.size();
    const R x0 = x[0], dx = x[1] - x[0];
    const R rdx = R(1.0) / dx;
    for (int i = 0; i < n; ++i) {
        volatile R prod = (R)i * dx;      // no FMA contraction: mirror the device expression
        volatile R xi = x0 + prod;
        if (xi != x[i]) return false;
    }
    for (int i = 0; i + 1 < n; ++i)
        if (R(1.0) / (x[i + 1] - x[i]) != rdx) return false;
    return true;
}

struct SlotStore {
    double *wind = nullptr, *thermo = nullptr, *rh = nullptr;
    float *wind32 = nullptr, *thermo32 = nullptr;       // fp32 copies (tcr_integrate_f32_*), converted on the device
    bool f32_stale = true;
};

// Host threads that copy a month slot's planes from the caller's (pageable) arrays into the pinned staging buffer: one
// thread moves ~10 GB/s, a slot of the global basin is 9.9 MB and crosses PCIe in 0.2 ms — the copy into pinned memory, not
// the link, is what a year's staging waits for.  Fork / join per call; the caller takes part.
struct CopyPool {
    struct Seg { char *dst; const char *src; size_t bytes; };
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const Seg *segs = nullptr;
    size_t n_segs = 0;
    std::atomic<size_t> next{0};
    int active = 0;
    uint64_t gen = 0;
    bool stop = false;

    explicit CopyPool(int helpers)
    {
        for (int i = 0; i < helpers; ++i) th.emplace_back([this] { loop(); });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
    void work()
    {
        for (size_t i; (i = next.fetch_add(1)) < n_segs;) memcpy(segs[i].dst, segs[i].src, segs[i].bytes);
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            work();
            std::lock_guard<std::mutex> lk(mu);
            if (--active == 0) cv_done.notify_one();
        }
    }
    void run(const std::vector<Seg> &v)
    {
        if (v.empty()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            segs = v.data(); n_segs = v.size(); next = 0; active = (int)th.size(); ++gen;
        }
        cv_go.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return active == 0; });
    }
};

}  // namespace

struct tcr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    tcr_params prm;
    bool have_prm = false;
    GridStore wg, tg, hg, bg, mg, rg;           // wind, thermo, land (+ bathymetry when shared), bathymetry (when on its own grid), masks, rh
    std::vector<SlotStore> slots;
    DevSlot *d_slots = nullptr;
    size_t d_slots_cap = 0;
    bool slots_dirty = true;
    double *d_stat = nullptr;                   // [lat][lon][2] land, bathymetry interleaved (one shared grid)
    double *d_land = nullptr, *d_bathy = nullptr;   // separate planes when the two grids differ (split_static)
    float *d_land32 = nullptr, *d_bathy32 = nullptr;
    bool split_static = false;                  // land and bathymetry on two grids (hg, bg)
    // exact narrow storage of the two (tcr_device.h, StaticMode): kStatPack16 -> d_nstat = uint16 [lat][lon];
    // kStatU8F32 -> d_nstat = the uint8 land plane, d_nbathy = the float bathymetry plane
    int static_mode = kStatF64;
    int static_pref = 0;                        // tcr_static_store: 0 auto, 1 fp64 planes
    void *d_nstat = nullptr;
    float *d_nbathy = nullptr;
    size_t static_bytes = 0;                    // footprint of the staged land + bathymetry
    uint8_t *d_mask_bits = nullptr;             // the eight mask planes as the bits of one byte per grid point
    // workspaces
    double *d_fs = nullptr, *d_srec = nullptr;    // forcing tables, accepted-step records
    size_t fs_cap = 0, srec_cap = 0;
    double *d_vrec = nullptr;                     // the v part of the step records, packed (k_integrate -> k_screen)
    size_t vrec_cap = 0;
    unsigned long long *d_tiles = nullptr;      // scratch of k_compact: ticket, generation, one word per tile (zeroed when allocated)
    size_t tiles_cap = 0;
    unsigned long long *d_queue = nullptr;      // work-queue heads and parked-storm counts of k_integrate's passes
    uint16_t *d_sidx = nullptr;                 // sample -> accepted-step map (k_dense -> k_emit)
    size_t sidx_cap = 0;
    double *d_park[2] = {nullptr, nullptr};     // ping-pong lists of parked storms
    size_t park_cap[2] = {0, 0};
    double2 *d_sc_table = nullptr;              // one period of (sin, cos)(2π j / period)
    double *d_tab = nullptr;                    // entropy table: p[np], s[ns], T[np][ns]
    int tab_np = 0, tab_ns = 0;
    double2 *d_pf = nullptr;                    // weighted phase factors [n][n_series][4] of the periodic Fourier kernel
    size_t pf_cap = 0;
    int fs_period = 0;                          // 0: direct Fourier kernel
    int cu_count = 256;
    uint8_t *d_screen_skip = nullptr; // storms the integrator found to fail the 2-day test (TC rows only)
    size_t screen_skip_cap = 0;
    int32_t *d_und_list = nullptr;    // ... and the storms accept test 1 is still open for (k_screen's work list)
    size_t und_list_cap = 0;
    unsigned long long *d_und_count = nullptr;
    float *d_stat32 = nullptr;                  // fp32 copy of the land / bathymetry planes
    bool stat32_stale = true;
    int32_t *d_tc_idx = nullptr;                // storms that passed accept test 1 (k_screen -> compaction), tc_rows_only
    size_t tc_idx_cap = 0;
    int64_t *d_tc_count = nullptr;
    double *d_cell = nullptr; size_t cell_cap = 0; int cell_bins = 0;     // scratch of tcr_cell_order_dev
    bool cell_counts_open = false;              // a compaction that counts cells has been enqueued, the rank kernel that zeroes them again not yet
    unsigned int *d_hist_partial = nullptr;     // k_seed_hist: per-workgroup counts + ticket (zero between launches)
    uint8_t *d_probe = nullptr;                 // decision probe of the next tcr_integrate_dev (tcr_integrate_probe_host)
    int probe_cap = 0;
    // rounds replayed from captured graphs (tcr_round_dev): the device copy of the round key the replayed kernels read, the
    // graphs keyed by the bytes of their descriptor, and an epoch that every allocation / parameter change bumps (a graph
    // holds the workspaces' addresses and the parameters by value)
    int storms_per_lane = 1;                    // tcr_schedule_set
    tcr_tune tune;                              // launch-shape knobs: the environment at tcr_ctx_create, then tcr_tune_set
    // field staging: two pinned halves (the host fills one while the interleave kernel reads the other over the link); nothing
    // waits on the host until the fields are used (fields_pending)
    double *h_stage[2] = {nullptr, nullptr};
    size_t h_stage_cap = 0;                     // doubles per half
    hipEvent_t h_stage_ev[2] = {nullptr, nullptr};
    int h_stage_next = 0;
    bool fields_pending = false;
    double stage_ms[5] = {0, 0, 0, 0, 0};       // host time of the slot uploads: wait for the pinned half, copy, enqueue transfer, enqueue kernel; [4] calls
    std::unique_ptr<CopyPool> pool;
    RoundKey *d_round_key = nullptr;
    uint64_t epoch = 0;
    bool capturing = false;
    struct RoundGraph { std::vector<uint8_t> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; bool failed = false; };
    std::vector<RoundGraph> graphs;
    int64_t n_replays = 0;
    // timing: four events per timed tcr_integrate_dev call since tcr_timing_enable(ctx, 1)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    // stage trace (tcr_stage_trace_enable): one event behind every stage of a directly enqueued round
    bool stage_trace = false;
    std::vector<hipEvent_t> st_pool;
    std::vector<int> st_id;
    size_t st_used = 0;
};

namespace {

int fail(tcr_ctx *ctx, const char *fmt, const char *a = "", const char *b = "")
{
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return -1;
}

#define HIPCHK(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) return fail(ctx, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// Synchronous copies go through the context's own stream, never the legacy default stream: a legacy-stream operation
// fails ("would make the legacy stream depend on a capturing blocking stream") whenever ANY thread of the process is
// capturing a round (tcr_round_dev use_graph), and invalidates that capture.
inline hipError_t copy_sync(hipStream_t st, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (!bytes) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

template <typename T>
int dev_alloc(tcr_ctx *ctx, T **p, size_t count)
{
    if (ctx && ctx->capturing) return fail(ctx, "internal: allocation while a round is being captured");
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T) + 256));
    if (ctx) ++ctx->epoch;          // captured rounds hold addresses: they are re-captured (drop_graphs)
    return 0;
}

bool same(const std::vector<double> &a, const double *b, int n)
{
    return (int)a.size() == n && memcmp(a.data(), b, sizeof(double) * n) == 0;
}

// Upload a grid (or check that it equals the one already staged): all month slots
// of one experiment share one wind grid and one thermo grid, as in the reference
// where every Coupled_FAST of a year reads the same env_wnd/thermo datasets.
int stage_grid(tcr_ctx *ctx, GridStore &g, const tcr_grid *in, const char *what)
{
    if (!in || in->nlon < 2 || in->nlat < 2 || !in->lon || !in->lat)
        return fail(ctx, "%s grid: need >= 2 points per axis", what);
    for (int i = 1; i < in->nlon; ++i)
        if (!(in->lon[i] > in->lon[i - 1])) return fail(ctx, "%s grid: lon not strictly increasing", what);
    for (int i = 1; i < in->nlat; ++i)
        if (!(in->lat[i] > in->lat[i - 1])) return fail(ctx, "%s grid: lat not strictly increasing", what);
    if (g.set) {
        if (same(g.lon, in->lon, in->nlon) && same(g.lat, in->lat, in->nlat)) return 0;
        return fail(ctx, "%s grid differs from the grid already staged in this context", what);
    }
    g.lon.assign(in->lon, in->lon + in->nlon);
    g.lat.assign(in->lat, in->lat + in->nlat);
    std::vector<double> rlon(in->nlon - 1), rlat(in->nlat - 1);
    for (int i = 0; i + 1 < in->nlon; ++i) rlon[i] = 1.0 / (in->lon[i + 1] - in->lon[i]);   // fpbspl.f
    for (int i = 0; i + 1 < in->nlat; ++i) rlat[i] = 1.0 / (in->lat[i + 1] - in->lat[i]);
    if (dev_alloc(ctx, &g.d_lon, in->nlon) || dev_alloc(ctx, &g.d_lat, in->nlat) ||
        dev_alloc(ctx, &g.d_rlon, in->nlon) || dev_alloc(ctx, &g.d_rlat, in->nlat)) return -1;
    HIPCHK(ctx, copy_sync(ctx->stream, g.d_lon, g.lon.data(), sizeof(double) * in->nlon, hipMemcpyHostToDevice));
    HIPCHK(ctx, copy_sync(ctx->stream, g.d_lat, g.lat.data(), sizeof(double) * in->nlat, hipMemcpyHostToDevice));
    HIPCHK(ctx, copy_sync(ctx->stream, g.d_rlon, rlon.data(), sizeof(double) * (in->nlon - 1), hipMemcpyHostToDevice));
    HIPCHK(ctx, copy_sync(ctx->stream, g.d_rlat, rlat.data(), sizeof(double) * (in->nlat - 1), hipMemcpyHostToDevice));
    g.affine_lon = axis_is_affine(g.lon);
    g.affine_lat = axis_is_affine(g.lat);
    g.set = true;
    return 0;
}

DevAxis dev_axis(const std::vector<double> &x, const double *d_x, const double *d_rx, bool affine)
{
    DevAxis a{};
    a.n = (int)x.size();
    a.affine = affine ? 1 : 0;
    a.x = d_x; a.rx = d_rx;
    a.x0 = x.front(); a.xn = x.back();
    a.dx = x[1] - x[0];
    a.rdx = 1.0 / a.dx;
    a.inv_step = (a.n - 1) / (x.back() - x.front());
    return a;
}

DevGrid dev_grid(const GridStore &g)
{
    DevGrid d{};
    if (!g.set) return d;
    d.nlon = (int)g.lon.size(); d.nlat = (int)g.lat.size();
    d.ax = dev_axis(g.lon, g.d_lon, g.d_rlon, g.affine_lon);
    d.ay = dev_axis(g.lat, g.d_lat, g.d_rlat, g.affine_lat);
    return d;
}

int sync_slots(tcr_ctx *ctx)
{
    if (!ctx->slots_dirty) return 0;
    const size_t n = ctx->slots.size();
    if (n > ctx->d_slots_cap) {
        if (ctx->d_slots) HIPCHK(ctx, hipFree(ctx->d_slots));
        if (dev_alloc(ctx, &ctx->d_slots, n)) return -1;
        ctx->d_slots_cap = n;
    }
    std::vector<DevSlot> h(n);
    for (size_t i = 0; i < n; ++i) {
        h[i].wind = ctx->slots[i].wind; h[i].thermo = ctx->slots[i].thermo; h[i].rh = ctx->slots[i].rh;
        h[i].wind32 = ctx->slots[i].wind32; h[i].thermo32 = ctx->slots[i].thermo32;
    }
    if (n) HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_slots, h.data(), sizeof(DevSlot) * n, hipMemcpyHostToDevice));
    ctx->slots_dirty = false;
    return 0;
}

DevFields dev_fields(const tcr_ctx *ctx)
{
    DevFields D{};
    D.wg = dev_grid(ctx->wg); D.tg = dev_grid(ctx->tg); D.hg = dev_grid(ctx->hg); D.mg = dev_grid(ctx->mg);
    D.rg = dev_grid(ctx->rg);
    D.slots = ctx->d_slots; D.n_slots = (int)ctx->slots.size();
    D.mask_bits = ctx->d_mask_bits;
    D.all_affine = (ctx->wg.affine_lon && ctx->wg.affine_lat && ctx->tg.affine_lon && ctx->tg.affine_lat &&
                    ctx->hg.affine_lon && ctx->hg.affine_lat &&
                    (!ctx->split_static || (ctx->bg.affine_lon && ctx->bg.affine_lat))) ? 1 : 0;
    return D;
}

int ready(tcr_ctx *ctx, bool need_masks)
{
    if (!ctx) return -1;
    if (!ctx->have_prm) return fail(ctx, "tcr_params_set has not been called");
    if (!ctx->wg.set || !ctx->tg.set || ctx->slots.empty()) return fail(ctx, "no field slot staged (tcr_fields_upload)");
    if (!ctx->hg.set) return fail(ctx, "static fields not staged (tcr_static_upload)");
    if (need_masks && !ctx->mg.set) return fail(ctx, "basin masks not staged (tcr_masks_upload)");
    if (ctx->fields_pending) {
        // the slots staged since the last use: their copies and interleave kernels run on the context's stream, the caller's
        // launches may be on any stream
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->fields_pending = false;
    }
    return sync_slots(ctx);
}

int grow(tcr_ctx *ctx, double **p, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    if (*p) HIPCHK(ctx, hipFree(*p));
    *p = nullptr; *cap = 0;
    if (dev_alloc(ctx, p, need)) return -1;
    *cap = need;
    return 0;
}

}  // namespace

namespace {

// Number of persistent waves of k_integrate.  Lanes pull storms from a queue, so fewer
// lanes than storms is fine (and balances storm lifetimes); up to one wave per SIMD while
// the batch is small, two per SIMD once every lane would still get several storms.
unsigned integrate_waves(const tcr_ctx *ctx, int64_t n, int wps)
{
    // one wave per SIMD per launch also when the register budget allows wps > 1 resident waves (fp32): the second
    // slot is for the waves of the batches other streams have in flight, and a launch of more waves than storms / 64
    // would leave lanes without work from the start
    (void)wps;
    const int64_t simds = (int64_t)ctx->cu_count * 4;
    int64_t waves = (n + kWave - 1) / kWave;
    if (ctx->tune.waves > 0) return (unsigned)(ctx->tune.waves < waves ? ctx->tune.waves : waves);
    if (waves > simds) waves = (n >= simds * kWave * 4) ? 2 * simds : simds;
    else if (ctx->storms_per_lane > 1) {
        // a batch that does not fill the chip, on a context set up for throughput (tcr_schedule_set): fewer waves whose lanes
        // take several storms in turn — the queue balances the lifetimes, so a batch costs a third of the SIMD time (lane
        // utilisation 0.44 -> 0.8) and leaves SIMDs to the batches other streams have in flight, for a ~1.4x longer chain
        const int64_t w = (n + (int64_t)kWave * ctx->storms_per_lane - 1) / ((int64_t)kWave * ctx->storms_per_lane);
        waves = std::max<int64_t>(std::min<int64_t>(waves, 16), w);
    }
    return (unsigned)(waves < 1 ? 1 : waves);
}

// Tail compaction of k_integrate: a wave parks its storms once fewer than this many lanes are
// live and the queue is empty.  Only worth it when the launch fills the chip (one wave per SIMD):
// it trades latency of the chain (pass barriers) for SIMD time, and SIMD time is only scarce then.
// tcr_tune.park = 0 disables the chain (one launch runs every storm to its end), k > 0 forces k.
int park_threshold(const tcr_ctx *ctx, unsigned waves, int wps)
{
    if (ctx->tune.park >= 0) return ctx->tune.park > 63 ? 63 : ctx->tune.park;
    (void)wps;
    // 12: measured on 100 000-storm steps (4 streams, segmented forcing table) — threshold 8 / 12 / 16 / 24 give 1.41 / 1.39 / 1.39 /
    // 1.38 ms per step and chains of 2.20 / 2.14-2.20 / 2.20-2.24 / 2.31-2.34 ms (5 / 5 / 6 / 7 passes): a higher threshold buys lane
    // utilisation (0.69 ... 0.83) with pass barriers; 12 has the shortest chain at (within noise) the best step time
    return waves >= (unsigned)ctx->cu_count * 4u ? 12 : 0;
}

// TCR_PASS_FILL=p (experiment builds): passes after the first launch only p % as many lanes as they have parked storms
// (k_integrate, fill_pct)
int pass_fill_pct()
{
#ifdef TCR_EXPERIMENTS
    return tcr_exp::pass_fill_pct();
#else
    return 100;
#endif
}
// np.linspace(0, total_time, n_steps)[i] as the kernels form it (ts_at)
double ts_host(const tcr_params &P, int i)
{
    return (i == P.n_steps - 1) ? P.total_time : (double)i * (P.total_time / (double)(P.n_steps - 1));
}

constexpr size_t kQueueWords = 6 * kMaxPasses;     // heads, parked counts, 4 occupancy counters per pass
// A pass this small runs to the end.  Default: one wave per CU (256 on MI355X) — for a chip-filling batch that is three passes
// (1 024, 1 024 behind the table's second segment, 176 waves) instead of the five of rounds 2-5 (… 30, 5 waves; the default was
// 8): the two tiny passes held 0.2 % of the SIMD time and each cost a dispatch in the chain.  Round 6, same box: exclusive
// chain -4 %, one-stream step -5 %, 12-stream step -1…1.5 %, SIMD time +1 % (profiles/r06_dispatch_and_knobs.txt).
unsigned park_final_waves(const tcr_ctx *ctx)
{
    return ctx->tune.park_final > 0 ? (unsigned)ctx->tune.park_final : (unsigned)std::max(8, ctx->cu_count);
}

// The launch-shape knobs a context starts from: the environment, read ONCE (tcr_ctx_create); tcr_tune_set replaces them.
long env_long(const char *name, long dflt)
{
    const char *e = getenv(name);
    return e ? atol(e) : dflt;
}
tcr_tune tune_from_env()
{
    tcr_tune t;
    t.waves = (int32_t)env_long("TCR_WAVES", -1);
    t.park = (int32_t)std::min<long>(env_long("TCR_PARK", -1), 63);     // (tcr_tune_set's own bound: a get / modify / set round trip must not fail on it)
    t.park_final = (int32_t)env_long("TCR_PARK_FINAL", -1);
    t.table_segments = (int32_t)env_long("TCR_TABLE_SEGMENTS", -1);
    t.prune = (int32_t)env_long("TCR_PRUNE", -1);
    t.emit_grid_cap = (int32_t)env_long("TCR_EMIT_GRID_CAP", -1);
    t.copy_threads = (int32_t)env_long("TCR_COPY_THREADS", -1);
    t.reserved = 0;
    return t;
}

struct DevBuf {
    std::vector<void *> ptrs;
    ~DevBuf() { for (void *p : ptrs) (void)hipFree(p); }
    template <typename T>
    T *get(size_t count)
    {
        void *p = nullptr;
        if (hipMalloc(&p, count * sizeof(T) + 256) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return static_cast<T *>(p);
    }
    template <typename T>
    T *put(const T *host, size_t count)
    {
        T *d = get<T>(count);
        if (d && hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    }
};

// stage trace: an event labelled with the stage that has just been enqueued (not while a round is being captured)
int stage_mark(tcr_ctx *ctx, hipStream_t st, int id)
{
    if (!ctx->stage_trace || ctx->capturing) return 0;
    if (ctx->st_used == ctx->st_pool.size()) {
        const size_t old = ctx->st_pool.size();
        ctx->st_pool.resize(old + 256, nullptr);
        ctx->st_id.resize(old + 256, 0);
        for (size_t i = old; i < ctx->st_pool.size(); ++i) HIPCHK(ctx, hipEventCreate(&ctx->st_pool[i]));
    }
    ctx->st_id[ctx->st_used] = id;
    HIPCHK(ctx, hipEventRecord(ctx->st_pool[ctx->st_used], st));
    ++ctx->st_used;
    return 0;
}
#define STAGE(id) do { if (stage_mark(ctx, (hipStream_t)st, id)) return -1; } while (0)

// per timed call: [0] start, [1] table (first segment) written, [2] integration chain done, [3] post-processing done,
// [4], [5] around the table's second segment inside the chain (recorded back to back when there is none)
constexpr size_t kEvPerCall = 6;
int timing_events(tcr_ctx *ctx, hipEvent_t **quad)
{
    if (ctx->ev_used + kEvPerCall > ctx->ev_pool.size()) {
        const size_t old = ctx->ev_pool.size();
        ctx->ev_pool.resize(old + 64, nullptr);
        for (size_t i = old; i < ctx->ev_pool.size(); ++i) HIPCHK(ctx, hipEventCreate(&ctx->ev_pool[i]));
    }
    *quad = &ctx->ev_pool[ctx->ev_used];
    ctx->ev_used += kEvPerCall;
    return 0;
}


// k_integrate<R, AFFINE, PROBE, SM>: pick the instantiation.  The narrow static modes exist for affine grids only (a context
// whose grids are not all affine has had its static planes widened to fp64 by settle_static).
template <typename R, bool PROBE>
void launch_integrate_rp(const KArgsT<R> &a, bool affine, int sm, unsigned waves, hipStream_t st)
{
#define TCR_LI(A, S) hipLaunchKernelGGL((k_integrate<R, A, PROBE, S>), dim3(waves), dim3(kWave), 0, st, a)
    if (affine) {
        switch (sm) {
        case kStatF64Split: TCR_LI(true, kStatF64Split); break;
        case kStatPack16: TCR_LI(true, kStatPack16); break;
        case kStatU8F32: TCR_LI(true, kStatU8F32); break;
        case kStatPack64: TCR_LI(true, kStatPack64); break;
        default: TCR_LI(true, kStatF64); break;
        }
    } else if (sm == kStatF64Split) TCR_LI(false, kStatF64Split);
    else TCR_LI(false, kStatF64);
#undef TCR_LI
}
void launch_integrate(const KArgsT<double> &a, bool affine, bool probe, int sm, unsigned waves, hipStream_t st)
{
    if (probe) launch_integrate_rp<double, true>(a, affine, sm, waves, st);
    else launch_integrate_rp<double, false>(a, affine, sm, waves, st);
}
void launch_integrate(const KArgsT<float> &a, bool affine, bool, int sm, unsigned waves, hipStream_t st)
{
    launch_integrate_rp<float, false>(a, affine, sm, waves, st);      // the decision probe is an fp64 instrument
}

// kStatPack16 / kStatU8F32 -> the fp64 planes the general (non-affine) kernels read: the same values, widened
__global__ __launch_bounds__(256) void k_static_widen(int mode, const void *__restrict__ nstat, const float *__restrict__ nbathy,
                                                      size_t n_land, size_t n_bathy, double *__restrict__ stat,
                                                      double *__restrict__ land, double *__restrict__ bathy)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == kStatPack16) {
        if (i >= n_land) return;
        const unsigned v = reinterpret_cast<const uint16_t *>(nstat)[i];
        stat[2 * i] = (double)(int)(v & 1u);
        stat[2 * i + 1] = (double)((int)(v >> 1) - kPack16Bias);
    } else if (mode == kStatPack64) {
        if (i >= n_land) return;
        const uint64_t v = reinterpret_cast<const uint64_t *>(nstat)[i];
        stat[2 * i] = (double)(int)((v >> 32) & 0xffu);
        stat[2 * i + 1] = (double)__uint_as_float((unsigned)v);
    } else {
        // (stat != NULL: one shared grid, interleaved; else two planes)
        if (i < n_land) { const double v = (double)reinterpret_cast<const uint8_t *>(nstat)[i]; if (stat) stat[2 * i] = v; else land[i] = v; }
        if (i < n_bathy) { const double v = (double)nbathy[i]; if (stat) stat[2 * i + 1] = v; else bathy[i] = v; }
    }
}

// The planes of one month slot — src = 14 wind planes [nw], 4 thermo planes [nt], rh [nr], in PINNED HOST memory: the
// coalesced plane reads are the transfer — into the slot layouts of tcr_device.h: wind [point][16] (NaN -> 0: _interp_basin_field, bam_track.py:72-74; two pads),
// thermo [point][4], rh as it is.  One thread per grid point.
struct StageSlotArgs {
    const double *src;
    size_t nw, nt, nr;
    double *wind, *thermo, *rh;
};
__global__ __launch_bounds__(256) void k_stage_slot(StageSlotArgs a)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.nw) {
        double v[16];
#pragma unroll
        for (int f = 0; f < 14; ++f) { const double x = a.src[(size_t)f * a.nw + i]; v[f] = (x != x) ? 0.0 : x; }
        v[14] = 0.0; v[15] = 0.0;
        double2 *o = reinterpret_cast<double2 *>(a.wind + i * kWindStride);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = make_double2(v[2 * k], v[2 * k + 1]);
    }
    if (i < a.nt) {
        const double *t = a.src + 14 * a.nw;
        double2 *o = reinterpret_cast<double2 *>(a.thermo + i * kThermoStride);
        o[0] = make_double2(t[i], t[a.nt + i]);
        o[1] = make_double2(t[2 * a.nt + i], t[3 * a.nt + i]);
    }
    if (i < a.nr) a.rh[i] = a.src[14 * a.nw + 4 * a.nt + i];
}

// ---- fp32 staging: float knots (+ reciprocal widths and the affine test in float arithmetic) and float
// copies of the interleaved field layouts, converted on the device from the fp64 arrays already staged
__global__ __launch_bounds__(256) void k_to_f32(const double *__restrict__ src, float *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

int grid_f32(tcr_ctx *ctx, GridStore &g, const char *what)
{
    if (g.set32) return 0;
    auto one = [&](const std::vector<double> &x, std::vector<float> &x32, float **d_x, float **d_rx, bool *affine) -> int {
        const int n = (int)x.size();
        x32.resize(n);
        for (int i = 0; i < n; ++i) x32[i] = (float)x[i];
        for (int i = 1; i < n; ++i)
            if (!(x32[i] > x32[i - 1])) return fail(ctx, "%s grid: knots are not strictly increasing once rounded to fp32", what);
        std::vector<float> r(n - 1);
        for (int i = 0; i + 1 < n; ++i) r[i] = 1.0f / (x32[i + 1] - x32[i]);
        if (dev_alloc(ctx, d_x, (size_t)n) || dev_alloc(ctx, d_rx, (size_t)n)) return -1;
        HIPCHK(ctx, copy_sync(ctx->stream, *d_x, x32.data(), sizeof(float) * n, hipMemcpyHostToDevice));
        HIPCHK(ctx, copy_sync(ctx->stream, *d_rx, r.data(), sizeof(float) * (n - 1), hipMemcpyHostToDevice));
        *affine = axis_is_affine(x32);
        return 0;
    };
    if (one(g.lon, g.lon32, &g.d_lon32, &g.d_rlon32, &g.affine_lon32)) return -1;
    if (one(g.lat, g.lat32, &g.d_lat32, &g.d_rlat32, &g.affine_lat32)) return -1;
    g.set32 = true;
    return 0;
}

int ensure_f32(tcr_ctx *ctx, hipStream_t st)
{
    if (grid_f32(ctx, ctx->wg, "wind") || grid_f32(ctx, ctx->tg, "thermo") || grid_f32(ctx, ctx->hg, "static")) return -1;
    if (ctx->split_static && grid_f32(ctx, ctx->bg, "bathymetry")) return -1;
    const size_t nw = ctx->wg.lon.size() * ctx->wg.lat.size() * kWindStride;
    const size_t nt = ctx->tg.lon.size() * ctx->tg.lat.size() * kThermoStride;
    const size_t nh = ctx->hg.lon.size() * ctx->hg.lat.size() * kStaticStride;
    auto conv = [&](const double *src, float **dst, size_t n) -> int {
        if (!*dst && dev_alloc(ctx, dst, n)) return -1;
        hipLaunchKernelGGL(k_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, *dst, n);
        return 0;
    };
    for (auto &s : ctx->slots) {
        if (!s.wind || !s.f32_stale) continue;
        if (conv(s.wind, &s.wind32, nw) || conv(s.thermo, &s.thermo32, nt)) return -1;
        s.f32_stale = false;
        ctx->slots_dirty = true;
    }
    if (ctx->stat32_stale && (ctx->static_mode == kStatF64 || ctx->static_mode == kStatF64Split)) {
        if (ctx->split_static) {
            if (conv(ctx->d_land, &ctx->d_land32, nh / kStaticStride)) return -1;
            if (conv(ctx->d_bathy, &ctx->d_bathy32, ctx->bg.lon.size() * ctx->bg.lat.size())) return -1;
        } else if (conv(ctx->d_stat, &ctx->d_stat32, nh)) return -1;
        ctx->stat32_stale = false;
    }
    HIPCHK(ctx, hipGetLastError());
    return sync_slots(ctx);
}

template <typename R> AxisT<R> axis_of(const GridStore &g, bool lon);
template <> AxisT<double> axis_of<double>(const GridStore &g, bool lon)
{
    return lon ? dev_axis(g.lon, g.d_lon, g.d_rlon, g.affine_lon) : dev_axis(g.lat, g.d_lat, g.d_rlat, g.affine_lat);
}
template <> AxisT<float> axis_of<float>(const GridStore &g, bool lon)
{
    const std::vector<float> &x = lon ? g.lon32 : g.lat32;
    AxisT<float> a{};
    a.n = (int)x.size();
    a.affine = (lon ? g.affine_lon32 : g.affine_lat32) ? 1 : 0;
    a.x = lon ? g.d_lon32 : g.d_lat32; a.rx = lon ? g.d_rlon32 : g.d_rlat32;
    a.x0 = x.front(); a.xn = x.back();
    a.dx = x[1] - x[0];
    a.rdx = 1.0f / a.dx;
    a.inv_step = (float)(a.n - 1) / (x.back() - x.front());
    return a;
}

// evaluation constants of one precision (copied to LDS by the kernels)
template <typename R>
void host_eval_k(const tcr_ctx *ctx, EvalKT<R> &K, bool *all_affine)
{
    K.wx = axis_of<R>(ctx->wg, true); K.wy = axis_of<R>(ctx->wg, false);
    K.tx = axis_of<R>(ctx->tg, true); K.ty = axis_of<R>(ctx->tg, false);
    K.hx = axis_of<R>(ctx->hg, true); K.hy = axis_of<R>(ctx->hg, false);
    const bool f64 = std::is_same<R, double>::value;
    if (ctx->split_static) { K.bx = axis_of<R>(ctx->bg, true); K.by = axis_of<R>(ctx->bg, false); }
    else { K.bx = K.hx; K.by = K.hy; }
    switch (ctx->static_mode) {
    case kStatF64Split:
        K.stat = f64 ? static_cast<const void *>(ctx->d_land) : static_cast<const void *>(ctx->d_land32);
        K.bathy = f64 ? static_cast<const void *>(ctx->d_bathy) : static_cast<const void *>(ctx->d_bathy32);
        break;
    case kStatPack16: case kStatPack64: K.stat = ctx->d_nstat; K.bathy = nullptr; break;
    case kStatU8F32: K.stat = ctx->d_nstat; K.bathy = ctx->d_nbathy; break;
    default:
        K.stat = f64 ? static_cast<const void *>(ctx->d_stat) : static_cast<const void *>(ctx->d_stat32);
        K.bathy = nullptr;
        break;
    }
    eval_k_scalars<R>(ctx->prm, K);
    K.tw_same = (ctx->wg.lon == ctx->tg.lon && ctx->wg.lat == ctx->tg.lat) ? 1 : 0;       // same knots: the same cells and weights
    K.hb_same = ctx->split_static ? 0 : 1;
    *all_affine = K.wx.affine && K.wy.affine && K.tx.affine && K.ty.affine && K.hx.affine && K.hy.affine &&
                  K.bx.affine && K.by.affine;
}

// The narrow static modes are instantiated for affine grids only.  A context whose grids are not all affine in the precision
// of the call (a Gaussian latitude axis, say) gets its land / bathymetry widened to the fp64 planes once — the same values —
// and keeps them.
int widen_static(tcr_ctx *ctx, hipStream_t st)
{
    const int mode = ctx->static_mode;
    if (mode != kStatPack16 && mode != kStatU8F32 && mode != kStatPack64) return 0;
    // (synchronises and frees: never legal inside a stream capture.  tcr_round_dev runs every new descriptor directly before it
    // captures it, and a static re-upload changes the epoch, so the widening always happens in that direct run — guarded anyway)
    if (ctx->capturing) return fail(ctx, "internal: static fields would be widened inside a stream capture");
    const size_t np = ctx->hg.lon.size() * ctx->hg.lat.size();
    const size_t nb = ctx->split_static ? ctx->bg.lon.size() * ctx->bg.lat.size() : np;
    if (ctx->split_static) { if (dev_alloc(ctx, &ctx->d_land, np) || dev_alloc(ctx, &ctx->d_bathy, nb)) return -1; }
    else if (dev_alloc(ctx, &ctx->d_stat, np * kStaticStride)) return -1;
    hipLaunchKernelGGL(k_static_widen, dim3((unsigned)((std::max(np, nb) + 255) / 256)), dim3(256), 0, st, mode, ctx->d_nstat, ctx->d_nbathy,
                       np, nb, ctx->d_stat, ctx->d_land, ctx->d_bathy);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));
    HIPCHK(ctx, hipFree(ctx->d_nstat)); ctx->d_nstat = nullptr;
    if (ctx->d_nbathy) { HIPCHK(ctx, hipFree(ctx->d_nbathy)); ctx->d_nbathy = nullptr; }
    ctx->static_mode = ctx->split_static ? kStatF64Split : kStatF64;
    ctx->static_bytes = sizeof(double) * (np + nb);
    ctx->stat32_stale = true;
    ++ctx->epoch;
    return 0;
}

template <typename R>
int eval_k_ready(tcr_ctx *ctx, EvalKT<R> &K, bool *affine, hipStream_t st)
{
    host_eval_k<R>(ctx, K, affine);
    if (!*affine && (ctx->static_mode == kStatPack16 || ctx->static_mode == kStatU8F32 || ctx->static_mode == kStatPack64)) {
        if (widen_static(ctx, st)) return -1;
        if (!std::is_same<R, double>::value && ensure_f32(ctx, st)) return -1;
        host_eval_k<R>(ctx, K, affine);
    }
    return 0;
}

// Outputs of one precision: tcr_tracks (double planes) or tcr_tracks_f32 (float planes), same layout
template <typename R>
struct TracksT {
    R *lon, *lat, *v, *m, *vmax, *envw;
    int32_t *n_valid, *status, *flags, *nfev, *n_accept, *n_reject, *pad_state;
    int32_t tc_rows_only;
};
template <typename R, typename T>
TracksT<R> tracks_of(const T *o)
{
    TracksT<R> t{};
    t.lon = o->lon; t.lat = o->lat; t.v = o->v; t.m = o->m; t.vmax = o->vmax; t.envw = o->envw;
    t.n_valid = o->n_valid; t.status = o->status; t.flags = o->flags; t.nfev = o->nfev;
    t.n_accept = o->n_accept; t.n_reject = o->n_reject; t.pad_state = o->pad_state; t.tc_rows_only = o->tc_rows_only;
    return t;
}

// The forcing table can be written in two segments of 12 column tiles (192 samples) each: the first for every storm before
// the chain starts, the second only for the storms the first integration pass parks.
constexpr int kFsSegSamples = kFsMfmaWaves * kFsMfmaColTiles * 16;
enum FsPart { kFsAll = -1, kFsFirst = 0, kFsRest = 1 };
bool fourier_on_matrix_cores(const tcr_ctx *ctx)
{
    const tcr_params &P = ctx->prm;
    return TCR_FS_MFMA && ctx->fs_period > 0 && 2 * P.n_series <= 4 * kFsMfmaKSteps && P.n_steps <= kFsMfmaMaxSamples;
}

// part = kFsRest: rows are the storms of the park list `park` (count on the device, at most n)
template <typename R>
int launch_fourier(tcr_ctx *ctx, int64_t n, const int64_t *n_dev, const double *phases, R *fs, hipStream_t st,
                   FsPart part = kFsAll, const double *park = nullptr, const unsigned long long *park_count = nullptr,
                   const BatchReset *reset = nullptr)
{
    // the batch's counters are zeroed by the phase-factor kernel when there is one, else by a launch of their own
    BatchReset z{};
    if (reset) z = *reset;
    if (reset && !(ctx->fs_period > 0 && fourier_on_matrix_cores(ctx)))
        hipLaunchKernelGGL(k_batch_reset, dim3((unsigned)((std::max<int64_t>(z.n_flags, z.queue_words) + 255) / 256)), dim3(256), 0, st, z);
    const tcr_params &P = ctx->prm;
    if (ctx->fs_period > 0) {
        const size_t lds = sizeof(double2) * (size_t)ctx->fs_period;
        const int64_t nf = n * 4 * (int64_t)P.n_series;
        if (part != kFsRest) {          // (kFsRest reads what the first segment's call left in d_pf: no reallocation in between)
            double *p = reinterpret_cast<double *>(ctx->d_pf);
            if (grow(ctx, &p, &ctx->pf_cap, (size_t)nf * 2)) { ctx->d_pf = nullptr; return -1; }
            ctx->d_pf = reinterpret_cast<double2 *>(p);
        } else if (!ctx->d_pf) return fail(ctx, "internal: second table segment without the first");
        if (fourier_on_matrix_cores(ctx)) {
            // matrix-core form: phase factors in MFMA fragment order (4 KB per 4 storms <= the 3.84 KB of d_pf's layout + padding)
            const int64_t tiles = (n + 3) / 4;
            // kFsRest: the second segment for the parked storms is ONE launch (round 6; three until then): the table kernel reads the
            // storm ids straight out of the park records (word 13 of each, k_integrate's park()) and picks every storm's phase
            // factors out of the fragments the first segment's k_phase_factors_frag wrote for the whole batch — d_pf is not written
            // between the two segments (only this function writes it, and a batch's two calls are on one stream)
            const int64_t *list = nullptr;
            int list_stride = 1;
            if (part == kFsRest) {
                list = reinterpret_cast<const int64_t *>(park) + 13;
                list_stride = kParkRec;
            } else {
                double *p0 = reinterpret_cast<double *>(ctx->d_pf);
                if (grow(ctx, &p0, &ctx->pf_cap, (size_t)tiles * kFsMfmaKSteps * 64)) { ctx->d_pf = nullptr; return -1; }
                ctx->d_pf = reinterpret_cast<double2 *>(p0);
                hipLaunchKernelGGL(k_phase_factors_frag, dim3((unsigned)std::min<int64_t>(tiles, 8192)), dim3(256), 0, st, P, n, n_dev, phases, p0, nullptr, nullptr, z);
            }
            double *p = reinterpret_cast<double *>(ctx->d_pf);
            const int groups = part == kFsAll ? kFsMfmaColGroups : 1;
            // workgroups per launch: TCR_FS_WGS=<total> overrides (scheduling experiment: a workgroup's two 191-register waves
            // keep integrator waves of other batches off their SIMDs)
            int64_t want = (int64_t)ctx->cu_count * kFsMfmaWgsPerCu * (4 / kFsMfmaWaves) / groups;
#ifdef TCR_EXPERIMENTS
            want = tcr_exp::fs_workgroups(want, groups);
#endif
            const unsigned wgs = (unsigned)std::min<int64_t>(tiles, std::max<int64_t>(1, want));
            if (part == kFsRest)
                hipLaunchKernelGGL((k_fourier_mfma<R, true>), dim3(wgs, groups), dim3(64 * kFsMfmaWaves), 0, st, P, n, n_dev, ctx->fs_period,
                                   ctx->d_sc_table, p, fs, 1, list, list_stride, park_count);
            else
                hipLaunchKernelGGL((k_fourier_mfma<R, false>), dim3(wgs, groups), dim3(64 * kFsMfmaWaves), 0, st, P, n, n_dev, ctx->fs_period,
                                   ctx->d_sc_table, p, fs, 0, list, list_stride, park_count);
        } else {
            hipLaunchKernelGGL(k_phase_factors, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, P, n, n_dev, phases, ctx->d_pf);
            hipLaunchKernelGGL(k_fourier_periodic<R>, dim3((unsigned)n), dim3(kFsThreads), lds, st, P, n, n_dev,
                               ctx->fs_period, ctx->d_sc_table, ctx->d_pf, fs);
        }
    } else {
        const int64_t total = n * (int64_t)P.n_steps;
        hipLaunchKernelGGL(k_fourier_direct<R>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, n, n_dev, phases, fs);
    }
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// The per-candidate body of run_tracks for a batch (see tcr_integrate_dev in the header), in precision R.
// stats: optional device counter block (TCR_N_STATS words) the batch's sums are added to by k_flags (tcr_round_dev)
template <typename R>
int integrate_impl(tcr_ctx *ctx, const tcr_storms *in, const TracksT<R> out, void *stream_, uint64_t *stats = nullptr,
                   const int64_t *stats_n_dev = nullptr, int64_t n_expected = 0)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t n = in->n;
    if (n < 0) return fail(ctx, "tcr_integrate: negative n");
    if (n == 0) return 0;
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    const tcr_params &P = ctx->prm;
    const size_t ns = (size_t)P.n_steps;
    constexpr size_t REC = (size_t)step_rec_doubles<R>();
    if (grow(ctx, &ctx->d_fs, &ctx->fs_cap, ((size_t)n * ns * 4 * sizeof(R) + 7) / 8)) return -1;
    const int max_rk = P.max_rk_steps > 0 ? P.max_rk_steps : 64;
    if (grow(ctx, &ctx->d_srec, &ctx->srec_cap, (size_t)n * max_rk * REC)) return -1;
    if (grow(ctx, &ctx->d_vrec, &ctx->vrec_cap, (size_t)n * max_rk * kVRec)) return -1;
    if (max_rk > 65535) return fail(ctx, "tcr_params.max_rk_steps must be <= 65535");
    {
        double *p = reinterpret_cast<double *>(ctx->d_sidx);        // [n][n_steps] uint16: step of each sample
        if (grow(ctx, &p, &ctx->sidx_cap, ((size_t)n * ns * sizeof(uint16_t) + 7) / 8)) { ctx->d_sidx = nullptr; return -1; }
        ctx->d_sidx = reinterpret_cast<uint16_t *>(p);
    }
    R *fs = reinterpret_cast<R *>(ctx->d_fs);
    EvalKT<R> EK{};
    bool affine = false;
    if (eval_k_ready<R>(ctx, EK, &affine, st)) return -1;

    hipEvent_t *ev = nullptr;
    if (ctx->timing && !ctx->capturing && timing_events(ctx, &ev)) return -1;
    if (ev) HIPCHK(ctx, hipEventRecord(ev[0], st));
    // Chain of launches with tail compaction (k_integrate): a pass parks the storms of waves that
    // fall under `thr` live lanes, the next pass needs at most waves*(thr-1)/64 waves for them.
    const int wps = std::is_same<R, double>::value ? TCR_INT_WPS : TCR_INT_WPS_F32;
    // (n_expected: how many of the n rows the caller expects to hold storms — a round's dense batch has capacity to spare)
    unsigned waves = integrate_waves(ctx, (n_expected > 0 && n_expected < n) ? n_expected : n, wps);
    const int thr = park_threshold(ctx, waves, wps);
    // With a chain, the forcing table is written in two segments: samples [0, 192) for every storm now, the rest after the
    // first pass and only for the storms that pass parks — the pass parks a storm (its lane takes the next one) as soon as
    // its next attempt could read beyond the first segment.  58 % of the storms never get there.
    // (only when the first pass is not also the last one: a last pass never parks, so a table cut at sample 191 would strand
    // every storm that lives beyond it — ADVICE r2: TCR_PARK forced on a batch of <= 8 waves)
    const unsigned final_waves = park_final_waves(ctx);
    const bool segmented = kFsMfmaColGroups == 2 && thr > 0 && waves > final_waves && kMaxPasses > 1 && fourier_on_matrix_cores(ctx) &&
                           (int)P.n_steps > kFsSegSamples + 16 && ctx->tune.table_segments != 0;
    // TC rows only: the 2-day half of accept test 1 is decided in flight when 2 d is an output sample (KArgsT::prune_sample)
    int prune_sample = -1;
    if (out.tc_rows_only && ctx->tune.prune != 0) {
        const double t2d = 2 * 86400.0, step_out = P.total_time / (double)(P.n_steps - 1);
        const int j = (int)floor(t2d / step_out);
        if (j >= 1 && j < P.n_steps - 1 && ts_host(P, j) == t2d) {
            prune_sample = j;
            double *q = reinterpret_cast<double *>(ctx->d_screen_skip);
            if (grow(ctx, &q, &ctx->screen_skip_cap, ((size_t)n + 7) / 8)) { ctx->d_screen_skip = nullptr; return -1; }
            ctx->d_screen_skip = reinterpret_cast<uint8_t *>(q);
            double *u = reinterpret_cast<double *>(ctx->d_und_list);
            if (grow(ctx, &u, &ctx->und_list_cap, ((size_t)n + 1) / 2 + 1)) { ctx->d_und_list = nullptr; return -1; }
            ctx->d_und_list = reinterpret_cast<int32_t *>(u);
            if (!ctx->d_und_count && dev_alloc(ctx, &ctx->d_und_count, (size_t)1)) return -1;
        }
    }
    {
        BatchReset z{};
        z.queue = ctx->d_queue; z.queue_words = (int)kQueueWords;
        if (prune_sample >= 0) { z.und_count = ctx->d_und_count; z.flags = out.flags; z.n_flags = n; }
        if (out.tc_rows_only) z.tc_count = reinterpret_cast<unsigned long long *>(ctx->d_tc_count);
        if (launch_fourier<R>(ctx, n, in->n_dev, in->phases, fs, st, segmented ? kFsFirst : kFsAll, nullptr, nullptr, &z)) return -1;
    }
    STAGE(TCR_STAGE_FOURIER);
    if (ev) HIPCHK(ctx, hipEventRecord(ev[1], st));
    {
        KArgsT<R> a{};
        a.P = P; a.D = dev_fields(ctx); a.K = EK; a.n = n; a.n_dev = in->n_dev;
        a.lon0 = in->lon0; a.lat0 = in->lat0; a.v0 = in->v0; a.m0 = in->m0; a.h_bl = in->h_bl;
        a.slot = in->slot; a.phases = in->phases; a.fs = fs; a.srec = ctx->d_srec; a.vrec = ctx->d_vrec; a.max_rk_steps = max_rk;
        a.n_valid = out.n_valid; a.status = out.status; a.nfev = out.nfev;
        a.n_accept = out.n_accept; a.n_reject = out.n_reject;
        a.queue = ctx->d_queue;
        a.prune_sample = prune_sample;
        a.screen_skip = prune_sample >= 0 ? ctx->d_screen_skip : nullptr;
        if (prune_sample >= 0) {
            // storms still open for accept test 1 when they end: the list k_screen works through (flags of the others stay 0)
            a.und_list = ctx->d_und_list; a.und_count = ctx->d_und_count;
        }
        // (the work queue / pass counters and, with the in-flight 2-day test, flags[] and the open-storm count were zeroed by
        // the batch's first kernel — BatchReset; three hipMemsetAsync until round 4, which as captured memset nodes of a
        // replayed round did not reliably clear flags[]: tests/test_round.py)
        // (a segmented first pass can park any number of its storms)
        const size_t park_items = segmented ? (size_t)n : (size_t)waves * kWave;
        if (thr > 0 && grow(ctx, &ctx->d_park[0], &ctx->park_cap[0], park_items * kParkRec)) return -1;
        if (thr > 0 && grow(ctx, &ctx->d_park[1], &ctx->park_cap[1], park_items * kParkRec)) return -1;
        const bool probe = std::is_same<R, double>::value && ctx->d_probe;
        if (probe) { a.probe = ctx->d_probe; a.probe_cap = ctx->probe_cap; }
        bool seg_events = false;
        for (int pass = 0; pass < kMaxPasses; ++pass) {
            const bool last = thr <= 0 || waves <= final_waves || pass == kMaxPasses - 1;
            a.pass = pass;
            a.fill_pct = pass_fill_pct();
            a.threshold = last ? 0 : thr;
            a.park_in = ctx->d_park[(pass + 1) & 1];
            a.park_out = ctx->d_park[pass & 1];
            // the stage times of an attempt lie in [t, t_new] and the table lookup at t reads samples up to ceil(t / dt) + 1
            a.t_limit = (segmented && pass == 0) ? ts_host(P, kFsSegSamples - 2) : 1e300;
            launch_integrate(a, affine, probe, ctx->static_mode, waves, st);
            if (last) break;
            if (segmented && pass == 0) {
                // the rest of the table, for the storms that are still alive (park list of pass 0, count on the device);
                // pass 1 may have as many storms as pass 0 had, so it keeps pass 0's waves
                if (ev) HIPCHK(ctx, hipEventRecord(ev[4], st));
                if (launch_fourier<R>(ctx, n, in->n_dev, in->phases, fs, st, kFsRest, ctx->d_park[0], ctx->d_queue + kMaxPasses)) return -1;
                if (ev) HIPCHK(ctx, hipEventRecord(ev[5], st));
                seg_events = true;
                continue;
            }
            waves = (unsigned)(((size_t)waves * (size_t)(thr - 1) + kWave - 1) / kWave);
        }
        if (ev && !seg_events) { HIPCHK(ctx, hipEventRecord(ev[4], st)); HIPCHK(ctx, hipEventRecord(ev[5], st)); }
    }
    STAGE(TCR_STAGE_INTEGRATE);
    if (ev) HIPCHK(ctx, hipEventRecord(ev[2], st));
    {
        EArgsT<R> a{};
        a.P = P; a.D = dev_fields(ctx); a.n = n; a.n_dev = in->n_dev; a.max_rk_steps = max_rk; a.srec = ctx->d_srec; a.vrec = ctx->d_vrec; a.fs = fs;
        a.slot = in->slot; a.n_valid = out.n_valid; a.status = out.status; a.n_accept = out.n_accept;
        a.lon = out.lon; a.lat = out.lat; a.v = out.v; a.m = out.m; a.vmax = out.vmax;
        a.envw = out.envw; a.flags = out.flags; a.pad_state = out.pad_state;
        a.K = EK;
        a.screen_skip = prune_sample >= 0 ? ctx->d_screen_skip : nullptr;
        if (prune_sample >= 0) { a.und_list = ctx->d_und_list; a.und_count = ctx->d_und_count; }
        const unsigned chunks = (unsigned)((ns + kPostThreads - 1) / kPostThreads);
        if (out.tc_rows_only) {
            // Only what the reference does (compute.py:185-204): accept test 1 from the v series alone, then env
            // winds, vmax and rows for the storms that passed.  The list stays on the device; the grids are
            // sized for the whole batch and workgroups beyond *count leave at once.
            if ((size_t)n > ctx->tc_idx_cap) {
                if (ctx->d_tc_idx) HIPCHK(ctx, hipFree(ctx->d_tc_idx));
                ctx->d_tc_idx = nullptr; ctx->tc_idx_cap = 0;
                if (dev_alloc(ctx, &ctx->d_tc_idx, (size_t)n)) return -1;
                ctx->tc_idx_cap = (size_t)n;
            }
            a.tc_list = ctx->d_tc_idx; a.tc_count = reinterpret_cast<unsigned long long *>(ctx->d_tc_count);
            hipLaunchKernelGGL(k_screen<R>, dim3((unsigned)((n + kScreenStorms - 1) / kScreenStorms)), dim3(kScreenThreads), 0, st, a);
            STAGE(TCR_STAGE_SCREEN);
            STAGE(TCR_STAGE_SELECT_TC);          // (the list is k_screen's own since round 4: no compaction launch)
            a.list = ctx->d_tc_idx; a.count = ctx->d_tc_count;
        }
        // k_dense, TC rows only: a bounded grid of waves walks the device-side list (one wave per storm otherwise)
        int64_t cap = kEmitGridCap;
        if (ctx->tune.emit_grid_cap > 0) cap = ctx->tune.emit_grid_cap;     // tests: force several list entries per workgroup
        // (a batch of at most two grid caps gets a row of workgroups per storm: no overflow launch — a dispatch costs a small
        // batch more than 12 000 workgroups that find nothing to do)
        const unsigned gx = (out.tc_rows_only && n > 2 * cap) ? (unsigned)cap : (unsigned)n;
        hipLaunchKernelGGL(k_dense<R>, dim3(gx), dim3(kWave), 0, st, a, ctx->d_sidx);
        STAGE(TCR_STAGE_DENSE);
        // TC rows only: a bounded grid walks the device-side list (all rows: one row of workgroups per storm)
        const unsigned ex = gx;
        if (out.tc_rows_only) {
            // list entries [0, ex): one per workgroup row, straight-line kernel; entries beyond the bounded grid (none
            // unless more than `cap` storms pass accept test 1): a small grid that loops
            if (affine) hipLaunchKernelGGL((k_emit<R, true, false>), dim3(ex, chunks), dim3(kPostThreads), 0, st, a, ctx->d_sidx);
            else hipLaunchKernelGGL((k_emit<R, false, false>), dim3(ex, chunks), dim3(kPostThreads), 0, st, a, ctx->d_sidx);
            if ((int64_t)ex < n) {
                a.item_base = ex;
                const unsigned ov = (unsigned)std::min<int64_t>(n - ex, 256);
                if (affine) hipLaunchKernelGGL((k_emit<R, true, true>), dim3(ov, chunks), dim3(kPostThreads), 0, st, a, ctx->d_sidx);
                else hipLaunchKernelGGL((k_emit<R, false, true>), dim3(ov, chunks), dim3(kPostThreads), 0, st, a, ctx->d_sidx);
                a.item_base = 0;
            }
        } else {
            if (affine) hipLaunchKernelGGL((k_emit<R, true, false>), dim3(ex, chunks), dim3(kPostThreads), 0, st, a, ctx->d_sidx);
            else hipLaunchKernelGGL((k_emit<R, false, false>), dim3(ex, chunks), dim3(kPostThreads), 0, st, a, ctx->d_sidx);
        }
        STAGE(TCR_STAGE_EMIT);
        hipLaunchKernelGGL(k_flags<R>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, n, out.n_valid,
                           out.status, out.v, out.flags, out.pad_state, out.tc_rows_only ? 1 : 0, a.n_dev, out.nfev,
                           reinterpret_cast<unsigned long long *>(stats), stats_n_dev);
        STAGE(TCR_STAGE_FLAGS);
    }
    if (ev) HIPCHK(ctx, hipEventRecord(ev[3], st));
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int tcr_abi_version(void) { return TCR_ABI_VERSION; }

const char *tcr_last_error(const tcr_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int tcr_ctx_create(int device, tcr_ctx **out)
{
    if (!out) return fail(nullptr, "tcr_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= count) return fail(nullptr, "tcr_ctx_create: device index out of range");
    tcr_ctx *ctx = new tcr_ctx();
    ctx->device = device;
    ctx->tune = tune_from_env();
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return fail(nullptr, "tcr_ctx_create: hipSetDevice/hipStreamCreate failed");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        ctx->cu_count = prop.multiProcessorCount;
        if (prop.sharedMemPerBlock < kCellScanLds) {
            (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return fail(nullptr, "tcr_ctx_create: this library is built for gfx950 (160 KB of LDS per CU); the device offers less than k_cell_scan's 67 KB per workgroup");
        }
    }
    if (hipMalloc(reinterpret_cast<void **>(&ctx->d_queue), kQueueWords * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&ctx->d_tc_count), 64) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&ctx->d_round_key), 64) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(nullptr, "tcr_ctx_create: hipMalloc failed");
    }
    *out = ctx;
    return 0;
}

int tcr_ctx_destroy(tcr_ctx *ctx)
{
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (GridStore *g : {&ctx->wg, &ctx->tg, &ctx->hg, &ctx->bg, &ctx->mg, &ctx->rg}) {
        (void)hipFree(g->d_lon); (void)hipFree(g->d_lat); (void)hipFree(g->d_rlon); (void)hipFree(g->d_rlat);
        (void)hipFree(g->d_lon32); (void)hipFree(g->d_lat32); (void)hipFree(g->d_rlon32); (void)hipFree(g->d_rlat32);
    }
    for (auto &s : ctx->slots) { (void)hipFree(s.wind); (void)hipFree(s.thermo); (void)hipFree(s.rh); (void)hipFree(s.wind32); (void)hipFree(s.thermo32); }
    (void)hipFree(ctx->d_stat32); (void)hipFree(ctx->d_land); (void)hipFree(ctx->d_bathy); (void)hipFree(ctx->d_land32); (void)hipFree(ctx->d_bathy32);
    (void)hipFree(ctx->d_slots); (void)hipFree(ctx->d_stat); (void)hipFree(ctx->d_mask_bits); (void)hipFree(ctx->d_nstat); (void)hipFree(ctx->d_nbathy);
    (void)hipFree(ctx->d_fs); (void)hipFree(ctx->d_srec); (void)hipFree(ctx->d_vrec);
    for (auto &ev : ctx->ev_pool) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : ctx->st_pool) if (ev) (void)hipEventDestroy(ev);
    for (auto &g : ctx->graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
    (void)hipFree(ctx->d_round_key);
    for (int i = 0; i < 2; ++i) { if (ctx->h_stage[i]) (void)hipHostFree(ctx->h_stage[i]); if (ctx->h_stage_ev[i]) (void)hipEventDestroy(ctx->h_stage_ev[i]); }
    (void)hipFree(ctx->d_hist_partial); (void)hipFree(ctx->d_cell); (void)hipFree(ctx->d_tiles); (void)hipFree(ctx->d_tc_idx); (void)hipFree(ctx->d_tc_count); (void)hipFree(ctx->d_queue); (void)hipFree(ctx->d_sidx); (void)hipFree(ctx->d_park[0]); (void)hipFree(ctx->d_park[1]); (void)hipFree(ctx->d_sc_table); (void)hipFree(ctx->d_pf); (void)hipFree(ctx->d_screen_skip); (void)hipFree(ctx->d_und_list); (void)hipFree(ctx->d_und_count); (void)hipFree(ctx->d_tab);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int tcr_params_set(tcr_ctx *ctx, const tcr_params *p)
{
    if (!ctx || !p) return -1;
    if (p->n_series < 1 || p->n_series > TCR_MAX_SERIES) return fail(ctx, "n_series out of range");
    if (p->n_steps < 2 || p->n_steps > 4096) return fail(ctx, "n_steps must be in [2, 4096]");
    if (p->max_rk_steps < 0 || p->max_rk_steps > 4096) return fail(ctx, "max_rk_steps out of range");
    if (!(p->total_time > 0) || !(p->dt_out > 0)) return fail(ctx, "total_time and dt_out must be positive");
    ctx->prm = *p;
    ctx->have_prm = true;
    ++ctx->epoch;
    // Periodic Fourier kernel when the series period is a whole number of output intervals
    // and the output times are exactly k*dt_out (np.linspace with an exact step).
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->fs_period = 0;
    const double per = p->T_Fs / p->dt_out;
    const int iper = (int)per;
    const bool regular = p->dt_out * (double)(p->n_steps - 1) == p->total_time;
    if (regular && iper >= 2 && iper <= 4096 && (double)iper == per && (double)iper * p->dt_out == p->T_Fs) {
        std::vector<double2> tab(iper);
        const long double two_pi = 6.283185307179586476925286766559005768L;
        for (int j = 0; j < iper; ++j) {
            const long double ang = two_pi * (long double)j / (long double)iper;
            tab[j] = make_double2((double)sinl(ang), (double)cosl(ang));
        }
        if (ctx->d_sc_table) HIPCHK(ctx, hipFree(ctx->d_sc_table));
        ctx->d_sc_table = nullptr;
        if (dev_alloc(ctx, &ctx->d_sc_table, (size_t)iper)) return -1;
        HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_sc_table, tab.data(), sizeof(double2) * iper, hipMemcpyHostToDevice));
        ctx->fs_period = iper;
    }
    return 0;
}

int tcr_static_store(tcr_ctx *ctx, int32_t pref)
{
    if (!ctx) return -1;
    if (pref != 0 && pref != 1) return fail(ctx, "tcr_static_store: 0 (auto) or 1 (fp64 planes)");
    ctx->static_pref = pref;
    return 0;
}

int tcr_static_info(tcr_ctx *ctx, int32_t *mode, int64_t *bytes)
{
    if (!ctx) return -1;
    if (!ctx->hg.set) return fail(ctx, "static fields not staged (tcr_static_upload)");
    if (mode) *mode = ctx->static_mode;
    if (bytes) *bytes = (int64_t)ctx->static_bytes;
    return 0;
}

int tcr_static_upload2(tcr_ctx *ctx, const tcr_grid *lg, const double *land, const tcr_grid *bg, const double *bathy)
{
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!land || !bathy || !lg || !bg) return fail(ctx, "tcr_static_upload: NULL argument");
    const bool shared = lg->nlon == bg->nlon && lg->nlat == bg->nlat && lg->lon && bg->lon && lg->lat && bg->lat &&
                        memcmp(lg->lon, bg->lon, sizeof(double) * lg->nlon) == 0 &&
                        memcmp(lg->lat, bg->lat, sizeof(double) * lg->nlat) == 0;
    if (ctx->hg.set && shared == ctx->split_static)
        return fail(ctx, "static fields were staged %s before; a context keeps one arrangement", ctx->split_static ? "on two grids" : "on one grid");
    if (stage_grid(ctx, ctx->hg, lg, shared ? "static" : "land")) return -1;
    if (!shared && stage_grid(ctx, ctx->bg, bg, "bathymetry")) return -1;
    const size_t np = (size_t)lg->nlon * lg->nlat, nb = (size_t)bg->nlon * bg->nlat;
    // Exact narrow storage where the VALUES allow it (the reference's land.nc is int8 0 / 1 on a 0.125-degree grid,
    // intensity/geo.py:23-34; bathymetry products are whole metres or float32): the kernels widen to the same doubles, so
    // nothing downstream — the `land == 1` decision included — can tell the difference.
    bool land01 = true, land_u8 = true, bathy_i15 = true, bathy_f32 = true;
    for (size_t i = 0; i < np && land_u8; ++i) {
        const double v = land[i];
        if (!(v >= 0.0 && v <= 255.0 && v == (double)(int)v)) land_u8 = false;
        if (v != 0.0 && v != 1.0) land01 = false;
    }
    for (size_t i = 0; i < nb && bathy_f32; ++i) {
        const double v = bathy[i];
        if (!((double)(float)v == v)) bathy_f32 = false;                  // (a NaN keeps the fp64 planes)
        if (!(v >= -(double)kPack16Bias && v < (double)kPack16Bias && v == (double)(int)v)) bathy_i15 = false;
    }
    int mode = shared ? kStatF64 : kStatF64Split;
    if (ctx->static_pref == 0) {
        if (shared && land_u8 && land01 && bathy_f32 && bathy_i15) mode = kStatPack16;
        else if (shared && land_u8 && bathy_f32) mode = kStatPack64;
        else if (land_u8 && bathy_f32) mode = kStatU8F32;
    }
    // a context may be re-staged with other planes (another mode, even): start from nothing
    for (void **q : {reinterpret_cast<void **>(&ctx->d_stat), reinterpret_cast<void **>(&ctx->d_land), reinterpret_cast<void **>(&ctx->d_bathy),
                     reinterpret_cast<void **>(&ctx->d_stat32), reinterpret_cast<void **>(&ctx->d_land32), reinterpret_cast<void **>(&ctx->d_bathy32),
                     &ctx->d_nstat, reinterpret_cast<void **>(&ctx->d_nbathy)}) {
        if (*q) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(*q)); *q = nullptr; ++ctx->epoch; }
    }
    if (mode == kStatPack16) {
        std::vector<uint16_t> h(np + 8, 0);
        for (size_t i = 0; i < np; ++i) h[i] = (uint16_t)((((int)bathy[i] + kPack16Bias) << 1) | (int)land[i]);
        uint16_t *d = nullptr;
        if (dev_alloc(ctx, &d, h.size())) return -1;
        ctx->d_nstat = d;
        HIPCHK(ctx, copy_sync(ctx->stream, d, h.data(), sizeof(uint16_t) * h.size(), hipMemcpyHostToDevice));
        ctx->static_bytes = sizeof(uint16_t) * np;
    } else if (mode == kStatPack64) {
        std::vector<uint64_t> h(np + 2, 0);
        for (size_t i = 0; i < np; ++i) {
            const float b = (float)bathy[i];
            uint32_t bits;
            memcpy(&bits, &b, 4);
            h[i] = (uint64_t)bits | ((uint64_t)(uint8_t)(int)land[i] << 32);
        }
        uint64_t *d = nullptr;
        if (dev_alloc(ctx, &d, h.size())) return -1;
        ctx->d_nstat = d;
        HIPCHK(ctx, copy_sync(ctx->stream, d, h.data(), sizeof(uint64_t) * h.size(), hipMemcpyHostToDevice));
        ctx->static_bytes = sizeof(uint64_t) * np;
    } else if (mode == kStatU8F32) {
        std::vector<uint8_t> hl(np + 8, 0);
        std::vector<float> hb(nb + 4, 0.f);
        for (size_t i = 0; i < np; ++i) hl[i] = (uint8_t)(int)land[i];
        for (size_t i = 0; i < nb; ++i) hb[i] = (float)bathy[i];
        uint8_t *d = nullptr;
        if (dev_alloc(ctx, &d, hl.size()) || dev_alloc(ctx, &ctx->d_nbathy, hb.size())) return -1;
        ctx->d_nstat = d;
        HIPCHK(ctx, copy_sync(ctx->stream, d, hl.data(), hl.size(), hipMemcpyHostToDevice));
        HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_nbathy, hb.data(), sizeof(float) * hb.size(), hipMemcpyHostToDevice));
        ctx->static_bytes = np + sizeof(float) * nb;
    } else if (shared) {
        std::vector<double> h(np * kStaticStride);
        for (size_t i = 0; i < np; ++i) { h[i * 2] = land[i]; h[i * 2 + 1] = bathy[i]; }
        if (dev_alloc(ctx, &ctx->d_stat, h.size())) return -1;
        HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_stat, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
        ctx->static_bytes = sizeof(double) * h.size();
    } else {
        if (dev_alloc(ctx, &ctx->d_land, np) || dev_alloc(ctx, &ctx->d_bathy, nb)) return -1;
        HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_land, land, sizeof(double) * np, hipMemcpyHostToDevice));
        HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_bathy, bathy, sizeof(double) * nb, hipMemcpyHostToDevice));
        ctx->static_bytes = sizeof(double) * (np + nb);
    }
    ctx->split_static = !shared;
    ctx->static_mode = mode;
    ctx->stat32_stale = true;
    return 0;
}

int tcr_static_upload(tcr_ctx *ctx, const tcr_grid *hg, const double *land, const double *bathy)
{
    return tcr_static_upload2(ctx, hg, land, hg, bathy);
}

extern "C++" {
namespace {
// One month slot: the planes go to the device as they are — copied by the pool into a pinned half, one asynchronous
// transfer — and k_stage_slot interleaves them there (wind NaN -> 0 as _interp_basin_field does, bam_track.py:72-74).
// Nothing here waits for the GPU except for the pinned half it is about to overwrite (two slots back).
// wind / thermo == false: that part is not given (tcr_rh_upload alone / tcr_fields_upload without rh).
int slot_stage(tcr_ctx *ctx, int slot, const tcr_grid *wg, const double *const mean[TCR_NW], const double *const cov[TCR_NCOV],
               const tcr_grid *tg, const double *vpot, const double *chi, const double *mld, const double *strat,
               const tcr_grid *rg, const double *rh_mid)
{
    const bool fields = wg != nullptr, rh = rg != nullptr;
    if (fields && (stage_grid(ctx, ctx->wg, wg, "wind") || stage_grid(ctx, ctx->tg, tg, "thermo"))) return -1;
    if (rh && stage_grid(ctx, ctx->rg, rg, "rh")) return -1;
    if ((size_t)slot >= ctx->slots.size()) ctx->slots.resize(slot + 1);
    SlotStore &s = ctx->slots[slot];
    const size_t nw = fields ? (size_t)wg->nlon * wg->nlat : 0, nt = fields ? (size_t)tg->nlon * tg->nlat : 0;
    const size_t nr = rh ? (size_t)rg->nlon * rg->nlat : 0;
    const size_t need = nw * 14 + nt * 4 + nr;
    if (need > ctx->h_stage_cap) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->h_stage_cap = 0;             // a failure below leaves "no staging buffer", never a half-built pair at the old capacity
        for (int i = 0; i < 2; ++i) {
            if (ctx->h_stage[i]) HIPCHK(ctx, hipHostFree(ctx->h_stage[i]));
            ctx->h_stage[i] = nullptr;
            HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_stage[i]), need * sizeof(double), hipHostMallocDefault));
            if (!ctx->h_stage_ev[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->h_stage_ev[i], hipEventDisableTiming));
        }
        ctx->h_stage_cap = need;
    }
    if (fields && !s.wind && dev_alloc(ctx, &s.wind, nw * kWindStride)) return -1;
    if (fields && !s.thermo && dev_alloc(ctx, &s.thermo, nt * kThermoStride)) return -1;
    if (rh && !s.rh && dev_alloc(ctx, &s.rh, nr)) return -1;
    if (!ctx->pool) ctx->pool.reset(new CopyPool((ctx->tune.copy_threads > 0 ? std::min(ctx->tune.copy_threads, 16) : 4) - 1));
    const int half = ctx->h_stage_next;
    ctx->h_stage_next ^= 1;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    HIPCHK(ctx, hipEventSynchronize(ctx->h_stage_ev[half]));           // the kernel that last read this half, two uploads ago (long done)
    const double t1 = now();
    double *h = ctx->h_stage[half];
    std::vector<CopyPool::Seg> segs;
    auto add = [&](double *dst, const double *src, size_t n) {
        const size_t chunk = 16384;                                    // 128 KB pieces: a slot is ~75 of them
        for (size_t o = 0; o < n; o += chunk)
            segs.push_back({reinterpret_cast<char *>(dst + o), reinterpret_cast<const char *>(src + o), sizeof(double) * std::min(chunk, n - o)});
    };
    if (fields) {
        for (int f = 0; f < 14; ++f) add(h + (size_t)f * nw, f < 4 ? mean[f] : cov[f - 4], nw);
        const double *th[4] = {vpot, chi, mld, strat};
        for (int k = 0; k < 4; ++k) add(h + 14 * nw + (size_t)k * nt, th[k], nt);
    }
    if (rh) add(h + 14 * nw + 4 * nt, rh_mid, nr);
    ctx->pool->run(segs);
    const double t2 = now();
    // The interleave kernel reads the pinned half ITSELF, over the link (hipHostMalloc memory is mapped into the device's
    // address space): no copy engine, no landing buffer, one stream.  A transfer followed by the kernel that consumes it makes
    // the next transfer wait for that kernel — a copy-engine job depending on a compute-queue signal, which the runtime
    // resolves on a host thread: inside run_downscaling that hop took 1-2 ms per slot (tools/stage_in_run_probe.py) although
    // the 9.9 MB transfer itself takes 0.26 ms.
    const double t3 = now();
    StageSlotArgs a{};
    a.src = h; a.nw = nw; a.nt = nt; a.nr = nr; a.wind = s.wind; a.thermo = s.thermo; a.rh = s.rh;
    const size_t nmax = std::max(nw, std::max(nt, nr));
    hipLaunchKernelGGL(k_stage_slot, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->h_stage_ev[half], ctx->stream));
    const double t4 = now();
    ctx->stage_ms[0] += t1 - t0; ctx->stage_ms[1] += t2 - t1; ctx->stage_ms[2] += t3 - t2; ctx->stage_ms[3] += t4 - t3; ctx->stage_ms[4] += 1.0;
    if (fields) s.f32_stale = true;
    ctx->slots_dirty = true;
    ctx->fields_pending = true;
    return 0;
}
}  // namespace
}  // extern "C++"

int tcr_slot_upload(tcr_ctx *ctx, int slot, const tcr_grid *wg, const double *const mean[TCR_NW],
                    const double *const cov[TCR_NCOV], const tcr_grid *tg, const double *vpot,
                    const double *chi, const double *mld, const double *strat, const tcr_grid *rg, const double *rh_mid)
{
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (slot < 0 || slot >= 4096) return fail(ctx, "tcr_slot_upload: slot out of range");
    if (!wg || !tg || !mean || !cov || !vpot || !chi || !mld || !strat) return fail(ctx, "tcr_slot_upload: NULL plane");
    for (int f = 0; f < 14; ++f)
        if (!(f < 4 ? mean[f] : cov[f - 4])) return fail(ctx, "tcr_slot_upload: NULL wind plane");
    if ((rg == nullptr) != (rh_mid == nullptr)) return fail(ctx, "tcr_slot_upload: rh grid and plane go together");
    return slot_stage(ctx, slot, wg, mean, cov, tg, vpot, chi, mld, strat, rg, rh_mid);
}

int tcr_fields_upload(tcr_ctx *ctx, int slot, const tcr_grid *wg, const double *const mean[TCR_NW],
                      const double *const cov[TCR_NCOV], const tcr_grid *tg, const double *vpot,
                      const double *chi, const double *mld, const double *strat)
{
    return tcr_slot_upload(ctx, slot, wg, mean, cov, tg, vpot, chi, mld, strat, nullptr, nullptr);
}

int tcr_rh_upload(tcr_ctx *ctx, int slot, const tcr_grid *rg, const double *rh_mid)
{
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (slot < 0 || slot >= 4096) return fail(ctx, "tcr_rh_upload: slot out of range");
    if (!rg || !rh_mid) return fail(ctx, "tcr_rh_upload: NULL plane");
    return slot_stage(ctx, slot, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rg, rh_mid);
}

int tcr_stage_timing(tcr_ctx *ctx, double ms[5], int32_t reset)
{
    if (!ctx || !ms) return -1;
    for (int i = 0; i < 5; ++i) { ms[i] = ctx->stage_ms[i]; if (reset) ctx->stage_ms[i] = 0.0; }
    return 0;
}

int tcr_masks_upload(tcr_ctx *ctx, const tcr_grid *mg, const uint8_t *run_mask,
                     const uint8_t *const basin_masks[TCR_N_BASINS])
{
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!run_mask || !basin_masks) return fail(ctx, "tcr_masks_upload: NULL mask");
    if (stage_grid(ctx, ctx->mg, mg, "mask")) return -1;
    const size_t np = (size_t)mg->nlon * mg->nlat;
    static_assert(TCR_N_BASINS == 7, "seven basin masks + the run basin's fill one byte");
    std::vector<uint8_t> bits(np, 0);
    for (int b = 0; b < TCR_N_BASINS; ++b) {
        if (!basin_masks[b]) return fail(ctx, "tcr_masks_upload: NULL basin mask");
        for (size_t i = 0; i < np; ++i) bits[i] |= (uint8_t)((basin_masks[b][i] ? 1u : 0u) << b);
    }
    for (size_t i = 0; i < np; ++i) bits[i] |= (uint8_t)((run_mask[i] ? 1u : 0u) << 7);
    if (!ctx->d_mask_bits && dev_alloc(ctx, &ctx->d_mask_bits, np)) return -1;
    HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_mask_bits, bits.data(), np, hipMemcpyHostToDevice));
    return 0;
}

int tcr_timing_enable(tcr_ctx *ctx, int on)
{
    if (!ctx) return -1;
    ctx->timing = on != 0;
    ctx->ev_used = 0;
    return 0;
}

// ms[0] forcing table (both segments), ms[1] integration chain without the table's second segment, ms[2] post-processing
static int timing_of_call(tcr_ctx *ctx, hipEvent_t *q, double ms[3])
{
    float f[3] = {0.f, 0.f, 0.f}, seg = 0.f;
    for (int i = 0; i < 3; ++i) HIPCHK(ctx, hipEventElapsedTime(&f[i], q[i], q[i + 1]));
    HIPCHK(ctx, hipEventElapsedTime(&seg, q[4], q[5]));
    ms[0] = f[0] + seg; ms[1] = f[1] - seg; ms[2] = f[2];
    return 0;
}

int tcr_timing_sum(tcr_ctx *ctx, double ms[3], int64_t *n_calls)
{
    if (!ctx || !ms) return -1;
    ms[0] = ms[1] = ms[2] = 0.0;
    const size_t calls = ctx->ev_used / kEvPerCall;
    if (n_calls) *n_calls = (int64_t)calls;
    if (!calls) return fail(ctx, "no timed launch recorded (tcr_timing_enable + tcr_integrate_*)");
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_pool[ctx->ev_used - 1]));
    for (size_t c = 0; c < calls; ++c) {
        double one[3];
        if (timing_of_call(ctx, &ctx->ev_pool[c * kEvPerCall], one)) return -1;
        for (int i = 0; i < 3; ++i) ms[i] += one[i];
    }
    return 0;
}

int tcr_timing_last(tcr_ctx *ctx, double ms[3])
{
    if (!ctx || !ms) return -1;
    if (ctx->ev_used < kEvPerCall) return fail(ctx, "no timed launch recorded (tcr_timing_enable + tcr_integrate_*)");
    hipEvent_t *q = &ctx->ev_pool[ctx->ev_used - kEvPerCall];
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_pool[ctx->ev_used - 1]));
    return timing_of_call(ctx, q, ms);
}

int tcr_integrate_pass_stats(tcr_ctx *ctx, int64_t *out, int max_passes)
{
    if (!ctx || !out) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    unsigned long long h[kQueueWords];
    HIPCHK(ctx, copy_sync(ctx->stream, h, ctx->d_queue, sizeof(h), hipMemcpyDeviceToHost));
    int np = 0;
    for (int p = 0; p < kMaxPasses && p < max_passes; ++p) {
        if (h[2 * kMaxPasses + 4 * p] == 0 && h[p] == 0) break;
        out[6 * p + 0] = (int64_t)h[p];                               // items requested from the queue (>= items)
        out[6 * p + 1] = (int64_t)h[kMaxPasses + p];                  // storms parked for the next pass
        out[6 * p + 2] = (int64_t)h[2 * kMaxPasses + 4 * p + 0];      // wave cycles
        out[6 * p + 3] = (int64_t)h[2 * kMaxPasses + 4 * p + 1];      // live-lane cycles
        out[6 * p + 4] = (int64_t)h[2 * kMaxPasses + 4 * p + 2];      // wave wall-clock ticks (100 MHz)
        out[6 * p + 5] = (int64_t)h[2 * kMaxPasses + 4 * p + 3];      // wave shader-clock ticks
        ++np;
    }
    return np;
}

int tcr_sync(tcr_ctx *ctx, void *stream)
{
    if (!ctx) return -1;
    HIPCHK(ctx, hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    return 0;
}

int tcr_integrate_dev(tcr_ctx *ctx, const tcr_storms *in, const tcr_tracks *out, void *stream_)
{
    if (ready(ctx, false)) return -1;
    if (!in || !out) return fail(ctx, "tcr_integrate_dev: NULL argument");
    return integrate_impl<double>(ctx, in, tracks_of<double>(out), stream_);
}

int tcr_integrate_f32_dev(tcr_ctx *ctx, const tcr_storms *in, const tcr_tracks_f32 *out, void *stream_)
{
    if (ready(ctx, false)) return -1;
    if (!in || !out) return fail(ctx, "tcr_integrate_f32_dev: NULL argument");
    if (ctx->d_probe) return fail(ctx, "the decision probe is an fp64 instrument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ensure_f32(ctx, stream_ ? (hipStream_t)stream_ : ctx->stream)) return -1;
    return integrate_impl<float>(ctx, in, tracks_of<float>(out), stream_);
}


extern "C++" {
namespace {
template <typename R, typename T>
int integrate_host_impl(tcr_ctx *ctx, const tcr_storms *in, const T *out)
{
    if (ready(ctx, false)) return -1;
    if (!in || !out) return fail(ctx, "tcr_integrate_host: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)in->n, ns = (size_t)ctx->prm.n_steps, N = (size_t)ctx->prm.n_series;
    if (n == 0) return 0;
    for (size_t i = 0; i < n; ++i)
        if (in->slot[i] < 0 || (size_t)in->slot[i] >= ctx->slots.size() || !ctx->slots[in->slot[i]].wind)
            return fail(ctx, "tcr_integrate_host: storm references a slot that is not staged");
    DevBuf B;
    tcr_storms di{};
    di.n = in->n;
    di.lon0 = B.put(in->lon0, n); di.lat0 = B.put(in->lat0, n); di.v0 = B.put(in->v0, n);
    di.m0 = B.put(in->m0, n); di.h_bl = B.put(in->h_bl, n); di.slot = B.put(in->slot, n);
    di.phases = B.put(in->phases, n * 4 * N);
    di.n_dev = nullptr;
    TracksT<R> dout{};
    dout.lon = B.get<R>(n * ns); dout.lat = B.get<R>(n * ns); dout.v = B.get<R>(n * ns);
    dout.m = B.get<R>(n * ns); dout.vmax = B.get<R>(n * ns); dout.envw = B.get<R>(n * ns * 4);
    dout.n_valid = B.get<int32_t>(n); dout.status = B.get<int32_t>(n); dout.flags = B.get<int32_t>(n);
    dout.nfev = B.get<int32_t>(n); dout.n_accept = B.get<int32_t>(n); dout.n_reject = B.get<int32_t>(n);
    if (!di.lon0 || !di.lat0 || !di.v0 || !di.m0 || !di.h_bl || !di.slot || !di.phases || !dout.lon ||
        !dout.lat || !dout.v || !dout.m || !dout.vmax || !dout.envw || !dout.n_valid || !dout.status ||
        !dout.flags || !dout.nfev || !dout.n_accept || !dout.n_reject)
        return fail(ctx, "tcr_integrate_host: device allocation / upload failed");
    dout.tc_rows_only = 0;             // host buffers come back whole: every row is produced
    if (!std::is_same<R, double>::value && ensure_f32(ctx, ctx->stream)) return -1;
    if (integrate_impl<R>(ctx, &di, dout, ctx->stream)) return -1;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
#define D2H(field, count, TT) \
    if (out->field) HIPCHK(ctx, copy_sync(ctx->stream, out->field, dout.field, (count) * sizeof(TT), hipMemcpyDeviceToHost))
    D2H(lon, n * ns, R); D2H(lat, n * ns, R); D2H(v, n * ns, R); D2H(m, n * ns, R);
    D2H(vmax, n * ns, R); D2H(envw, n * ns * 4, R);
    D2H(n_valid, n, int32_t); D2H(status, n, int32_t); D2H(flags, n, int32_t); D2H(nfev, n, int32_t);
    D2H(n_accept, n, int32_t); D2H(n_reject, n, int32_t);
#undef D2H
    return 0;
}
}  // namespace
}  // extern "C++"

int tcr_integrate_host(tcr_ctx *ctx, const tcr_storms *in, const tcr_tracks *out)
{
    return integrate_host_impl<double>(ctx, in, out);
}

int tcr_integrate_f32_host(tcr_ctx *ctx, const tcr_storms *in, const tcr_tracks_f32 *out)
{
    if (ctx && ctx->d_probe) return fail(ctx, "the decision probe is an fp64 instrument");
    return integrate_host_impl<float>(ctx, in, out);
}

int tcr_integrate_probe_host(tcr_ctx *ctx, const tcr_storms *in, const tcr_tracks *out, uint8_t *dec, int32_t cap)
{
    if (!ctx) return -1;
    if (!in || !dec || cap <= 0) return fail(ctx, "tcr_integrate_probe_host: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (in->n <= 0) return 0;
    DevBuf B;
    const size_t bytes = (size_t)in->n * (size_t)cap;
    uint8_t *d = B.get<uint8_t>(bytes);
    if (!d) return fail(ctx, "tcr_integrate_probe_host: device allocation failed");
    HIPCHK(ctx, hipMemsetAsync(d, 0xff, bytes, ctx->stream)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->d_probe = d; ctx->probe_cap = cap;
    const int rc = tcr_integrate_host(ctx, in, out);
    ctx->d_probe = nullptr; ctx->probe_cap = 0;
    if (rc) return rc;
    HIPCHK(ctx, copy_sync(ctx->stream, dec, d, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int tcr_fourier_table_host(tcr_ctx *ctx, int64_t n, const double *phases, double *Fs)
{
    if (!ctx) return -1;
    if (!ctx->have_prm) return fail(ctx, "tcr_params_set has not been called");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n <= 0) return 0;
    const size_t ns = (size_t)ctx->prm.n_steps, N = (size_t)ctx->prm.n_series;
    DevBuf B;
    const double *d_ph = B.put(phases, (size_t)n * 4 * N);
    double *d_fs = B.get<double>((size_t)n * ns * 4);
    if (!d_ph || !d_fs) return fail(ctx, "tcr_fourier_table_host: device allocation failed");
    if (launch_fourier(ctx, n, nullptr, d_ph, d_fs, ctx->stream)) return -1;
    std::vector<double> h((size_t)n * ns * 4);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), d_fs, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // device layout [n][n_steps][4] -> reference layout [n][4][n_steps]
    for (int64_t s = 0; s < n; ++s)
        for (size_t i = 0; i < ns; ++i)
            for (int k = 0; k < 4; ++k) Fs[((size_t)s * 4 + k) * ns + i] = h[((size_t)s * ns + i) * 4 + k];
    return 0;
}

int tcr_entropy_table_upload(tcr_ctx *ctx, int32_t np, int32_t ns, const double *p, const double *s, const double *T)
{
    if (!ctx) return -1;
    if (np < 2 || ns < 2 || !p || !s || !T) return fail(ctx, "tcr_entropy_table_upload: bad argument");
    for (int i = 1; i < np; ++i) if (!(p[i] > p[i - 1])) return fail(ctx, "tcr_entropy_table_upload: pressure axis must ascend");
    for (int i = 1; i < ns; ++i) if (!(s[i] > s[i - 1])) return fail(ctx, "tcr_entropy_table_upload: entropy axis must ascend");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->d_tab) HIPCHK(ctx, hipFree(ctx->d_tab));
    ctx->d_tab = nullptr;
    const size_t n = (size_t)np + ns + (size_t)np * ns;
    if (dev_alloc(ctx, &ctx->d_tab, n)) return -1;
    HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_tab, p, sizeof(double) * np, hipMemcpyHostToDevice));
    HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_tab + np, s, sizeof(double) * ns, hipMemcpyHostToDevice));
    HIPCHK(ctx, copy_sync(ctx->stream, ctx->d_tab + np + ns, T, sizeof(double) * (size_t)np * ns, hipMemcpyHostToDevice));
    ctx->tab_np = np; ctx->tab_ns = ns;
    return 0;
}

int tcr_potential_intensity_dev(tcr_ctx *ctx, int64_t n_points, int32_t n_lev, const double *p_env, const double *sst,
                                const double *psl, const double *T_env, const double *r_env, double ck_over_cd,
                                double *pi, void *stream_)
{
    if (!ctx) return -1;
    if (!ctx->d_tab) return fail(ctx, "tcr_potential_intensity: no entropy table (tcr_entropy_table_upload)");
    if (n_points <= 0 || n_lev < 2 || !p_env || !sst || !psl || !T_env || !r_env || !pi)
        return fail(ctx, "tcr_potential_intensity: bad argument (at least two levels)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    th::PiArgs a{};
    a.tab.np = ctx->tab_np; a.tab.ns = ctx->tab_ns;
    a.tab.p = ctx->d_tab; a.tab.s = ctx->d_tab + ctx->tab_np; a.tab.T = ctx->d_tab + ctx->tab_np + ctx->tab_ns;
    a.n_points = n_points; a.n_lev = n_lev; a.p_env = p_env; a.sst = sst; a.psl = psl; a.T_env = T_env; a.r_env = r_env;
    a.cecd = ck_over_cd; a.pi = pi;
    hipLaunchKernelGGL(th::k_potential_intensity, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int tcr_potential_intensity_host(tcr_ctx *ctx, int64_t n_points, int32_t n_lev, const double *p_env, const double *sst,
                                 const double *psl, const double *T_env, const double *r_env, double ck_over_cd, double *pi)
{
    if (!ctx) return -1;
    if (n_points <= 0 || n_lev < 2) return fail(ctx, "tcr_potential_intensity: bad argument (at least two levels)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int k = 1; k < n_lev; ++k)
        if (!(p_env[k] < p_env[k - 1])) return fail(ctx, "tcr_potential_intensity: levels must run from the lowest (highest pressure) up");
    DevBuf B;
    const double *d_p = B.put(p_env, (size_t)n_lev), *d_sst = B.put(sst, (size_t)n_points), *d_psl = B.put(psl, (size_t)n_points);
    const double *d_T = B.put(T_env, (size_t)n_lev * n_points), *d_r = B.put(r_env, (size_t)n_lev * n_points);
    double *d_pi = B.get<double>((size_t)n_points);
    if (!d_p || !d_sst || !d_psl || !d_T || !d_r || !d_pi) return fail(ctx, "tcr_potential_intensity_host: device allocation failed");
    if (tcr_potential_intensity_dev(ctx, n_points, n_lev, d_p, d_sst, d_psl, d_T, d_r, ck_over_cd, d_pi, ctx->stream)) return -1;
    HIPCHK(ctx, hipMemcpyAsync(pi, d_pi, sizeof(double) * (size_t)n_points, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int tcr_chi_rh_host(tcr_ctx *ctx, int64_t n_points, const double *sst, const double *psl, const double *T_mid,
                    const double *q_mid, double p_mid, double *chi, double *rh_mid)
{
    if (!ctx) return -1;
    if (n_points <= 0 || !sst || !psl || !T_mid || !q_mid || !chi || !rh_mid) return fail(ctx, "tcr_chi_rh_host: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf B;
    const size_t n = (size_t)n_points;
    const double *d_sst = B.put(sst, n), *d_psl = B.put(psl, n), *d_T = B.put(T_mid, n), *d_q = B.put(q_mid, n);
    double *d_chi = B.get<double>(n), *d_rh = B.get<double>(n);
    if (!d_sst || !d_psl || !d_T || !d_q || !d_chi || !d_rh) return fail(ctx, "tcr_chi_rh_host: device allocation failed");
    hipLaunchKernelGGL(th::k_chi_rh, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, ctx->stream, n_points,
                       d_sst, d_psl, d_T, d_q, p_mid, d_chi, d_rh);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(chi, d_chi, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(rh_mid, d_rh, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C++" {
namespace {
template <typename T>
int wind_stats_dev_impl(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const T *const wnd[4],
                        const int32_t *day_start, int32_t n_days, double *out, void *stream_)
{
    if (!ctx) return -1;
    if (!wnd || !out || n_samples <= 0 || n_points <= 0) return fail(ctx, "tcr_wind_stats_dev: bad argument");
    if (!day_start) n_days = (int32_t)n_samples;
    if (n_days < 2) return fail(ctx, "tcr_wind_stats_dev: a covariance needs at least two days");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    WindStatArgsT<T> a{};
    for (int c = 0; c < 4; ++c) a.w[c] = wnd[c];
    a.day_start = day_start; a.n_days = n_days; a.n_points = n_points; a.out = out;
    hipEvent_t *ev = nullptr;
    if (ctx->timing && timing_events(ctx, &ev)) return -1;
    if (ev) { HIPCHK(ctx, hipEventRecord(ev[0], st)); HIPCHK(ctx, hipEventRecord(ev[1], st)); }
    hipLaunchKernelGGL(k_wind_stats<T>, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, a);
    if (ev) { HIPCHK(ctx, hipEventRecord(ev[2], st)); HIPCHK(ctx, hipEventRecord(ev[3], st)); HIPCHK(ctx, hipEventRecord(ev[4], st)); HIPCHK(ctx, hipEventRecord(ev[5], st)); }
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

template <typename T>
int wind_stats_host_impl(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const T *const wnd[4],
                         const int32_t *day_start, int32_t n_days, double *out)
{
    if (!ctx) return -1;
    if (!wnd || !out || n_samples <= 0 || n_points <= 0) return fail(ctx, "tcr_wind_stats_host: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf B;
    const T *d_w[4];
    for (int c = 0; c < 4; ++c)
        if (!(d_w[c] = B.put(wnd[c], (size_t)n_samples * n_points))) return fail(ctx, "tcr_wind_stats_host: device allocation failed");
    const int32_t *d_ds = nullptr;
    if (day_start && !(d_ds = B.put(day_start, (size_t)n_days + 1))) return fail(ctx, "tcr_wind_stats_host: device allocation failed");
    double *d_out = B.get<double>((size_t)14 * n_points);
    if (!d_out) return fail(ctx, "tcr_wind_stats_host: device allocation failed");
    if (wind_stats_dev_impl<T>(ctx, n_samples, n_points, d_w, d_ds, n_days, d_out, ctx->stream)) return -1;
    HIPCHK(ctx, hipMemcpyAsync(out, d_out, sizeof(double) * 14 * (size_t)n_points, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
}  // namespace
}  // extern "C++"

int tcr_wind_stats_dev(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const double *const wnd[4],
                       const int32_t *day_start, int32_t n_days, double *out, void *stream_)
{
    return wind_stats_dev_impl<double>(ctx, n_samples, n_points, wnd, day_start, n_days, out, stream_);
}

int tcr_wind_stats_f32_dev(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const float *const wnd[4],
                           const int32_t *day_start, int32_t n_days, double *out, void *stream_)
{
    return wind_stats_dev_impl<float>(ctx, n_samples, n_points, wnd, day_start, n_days, out, stream_);
}

int tcr_wind_stats_host(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const double *const wnd[4],
                        const int32_t *day_start, int32_t n_days, double *out)
{
    return wind_stats_host_impl<double>(ctx, n_samples, n_points, wnd, day_start, n_days, out);
}

int tcr_wind_stats_f32_host(tcr_ctx *ctx, int64_t n_samples, int64_t n_points, const float *const wnd[4],
                            const int32_t *day_start, int32_t n_days, double *out)
{
    return wind_stats_host_impl<float>(ctx, n_samples, n_points, wnd, day_start, n_days, out);
}

int tcr_probe_math_host(tcr_ctx *ctx, int32_t fn, int64_t n, const double *a, const double *b, double *out)
{
    if (!ctx) return -1;
    if (fn < 0 || fn > 5 || n < 0 || (n > 0 && (!a || !out)) || (fn == 0 && n > 0 && !b)) return fail(ctx, "tcr_probe_math_host: bad argument");
    if (n == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf B;
    const double *d_a = B.put(a, n), *d_b = b ? B.put(b, n) : nullptr;
    double *d_o = B.get<double>(n);
    if (!d_a || (b && !d_b) || !d_o) return fail(ctx, "tcr_probe_math_host: device allocation failed");
    hipLaunchKernelGGL(k_probe_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)fn, n, d_a, d_b, d_o);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, copy_sync(ctx->stream, out, d_o, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}

int tcr_probe_rhs_host(tcr_ctx *ctx, int slot, double h_bl, const double *Fs, int64_t n, const double *t,
                       const double *lon, const double *lat, const double *v, const double *m,
                       double *dydt, double *envw, double *alpha)
{
    if (ready(ctx, false)) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (slot < 0 || (size_t)slot >= ctx->slots.size() || !ctx->slots[slot].wind)
        return fail(ctx, "tcr_probe_rhs_host: slot not staged");
    if (n <= 0) return 0;
    const size_t ns = (size_t)ctx->prm.n_steps;
    std::vector<double> fs(ns * 4);
    for (size_t i = 0; i < ns; ++i)
        for (int k = 0; k < 4; ++k) fs[i * 4 + k] = Fs[(size_t)k * ns + i];
    DevBuf B;
    const double *d_fs = B.put(fs.data(), fs.size());
    const double *d_t = B.put(t, n), *d_lon = B.put(lon, n), *d_lat = B.put(lat, n), *d_v = B.put(v, n), *d_m = B.put(m, n);
    double *d_dy = B.get<double>(n * 4), *d_w = B.get<double>(n * 4), *d_al = B.get<double>(n);
    if (!d_fs || !d_t || !d_lon || !d_lat || !d_v || !d_m || !d_dy || !d_w || !d_al)
        return fail(ctx, "tcr_probe_rhs_host: device allocation failed");
    const DevFields DF = dev_fields(ctx);
    EvalK EK{};
    bool affine = false;
    if (eval_k_ready<double>(ctx, EK, &affine, ctx->stream)) return -1;
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
#define PROBE_RHS(A, S) hipLaunchKernelGGL((k_probe_rhs<A, S>), grid, block, 0, ctx->stream, ctx->prm, DF, EK, slot, h_bl, \
                                           d_fs, n, d_t, d_lon, d_lat, d_v, d_m, d_dy, d_w, d_al)
    const int sm = ctx->static_mode;
    if (affine && sm == kStatPack16) PROBE_RHS(true, kStatPack16);
    else if (affine && sm == kStatU8F32) PROBE_RHS(true, kStatU8F32);
    else if (affine && sm == kStatPack64) PROBE_RHS(true, kStatPack64);
    else if (affine && sm == kStatF64) PROBE_RHS(true, kStatF64);
    else if (affine) PROBE_RHS(true, kStatF64Split);
    else if (sm == kStatF64) PROBE_RHS(false, kStatF64);
    else PROBE_RHS(false, kStatF64Split);
#undef PROBE_RHS
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, copy_sync(ctx->stream, dydt, d_dy, sizeof(double) * n * 4, hipMemcpyDeviceToHost));
    HIPCHK(ctx, copy_sync(ctx->stream, envw, d_w, sizeof(double) * n * 4, hipMemcpyDeviceToHost));
    HIPCHK(ctx, copy_sync(ctx->stream, alpha, d_al, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C++" {
namespace {
int init_m_launch(tcr_ctx *ctx, const tcr_storms *in, double dvdt, double *m_out, hipStream_t st)
{
    const DevFields DF = dev_fields(ctx);
    EvalK EK{};
    bool affine = false;
    if (eval_k_ready<double>(ctx, EK, &affine, st)) return -1;
    const dim3 grid((unsigned)((in->n + 63) / 64)), block(64);
#define INIT_M(A, S) hipLaunchKernelGGL((k_init_m<A, S>), grid, block, 0, st, ctx->prm, DF, EK, in->n, in->n_dev, in->lon0, in->lat0, \
                                        in->v0, in->m0, in->h_bl, in->slot, in->phases, dvdt, m_out)
    const int sm = ctx->static_mode;
    if (affine && sm == kStatPack16) INIT_M(true, kStatPack16);
    else if (affine && sm == kStatU8F32) INIT_M(true, kStatU8F32);
    else if (affine && sm == kStatPack64) INIT_M(true, kStatPack64);
    else if (affine && sm == kStatF64) INIT_M(true, kStatF64);
    else if (affine) INIT_M(true, kStatF64Split);
    else if (sm == kStatF64) INIT_M(false, kStatF64);
    else INIT_M(false, kStatF64Split);
#undef INIT_M
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
}  // namespace
}  // extern "C++"

int tcr_init_m_dev(tcr_ctx *ctx, const tcr_storms *in, double dvdt, double *m_out_dev, void *stream_)
{
    if (ready(ctx, false)) return -1;
    if (!in || !m_out_dev || !in->lon0 || !in->lat0 || !in->v0 || !in->h_bl || !in->slot || !in->phases)
        return fail(ctx, "tcr_init_m_dev: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (in->n <= 0) return 0;
    return init_m_launch(ctx, in, dvdt, m_out_dev, stream_ ? (hipStream_t)stream_ : ctx->stream);
}

int tcr_init_m_host(tcr_ctx *ctx, const tcr_storms *in, double dvdt, double *m_out)
{
    if (ready(ctx, false)) return -1;
    if (!in || !m_out || !in->lon0 || !in->lat0 || !in->v0 || !in->h_bl || !in->slot || !in->phases)
        return fail(ctx, "tcr_init_m_host: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t n = in->n;
    if (n <= 0) return 0;
    for (int64_t i = 0; i < n; ++i)
        if (in->slot[i] < 0 || (size_t)in->slot[i] >= ctx->slots.size() || !ctx->slots[in->slot[i]].wind)
            return fail(ctx, "tcr_init_m_host: a storm uses a slot that is not staged");
    DevBuf B;
    tcr_storms d = *in;
    d.n_dev = nullptr;
    d.lon0 = B.put(in->lon0, n); d.lat0 = B.put(in->lat0, n); d.v0 = B.put(in->v0, n);
    d.m0 = in->m0 ? B.put(in->m0, n) : nullptr;
    d.h_bl = B.put(in->h_bl, n); d.slot = B.put(in->slot, n);
    d.phases = B.put(in->phases, n * 4 * (int64_t)ctx->prm.n_series);
    double *d_out = B.get<double>(n);
    if (!d.lon0 || !d.lat0 || !d.v0 || (in->m0 && !d.m0) || !d.h_bl || !d.slot || !d.phases || !d_out)
        return fail(ctx, "tcr_init_m_host: device allocation failed");
    if (init_m_launch(ctx, &d, dvdt, d_out, ctx->stream)) return -1;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, copy_sync(ctx->stream, m_out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C++" {
namespace {
int seed_impl(tcr_ctx *ctx, uint64_t experiment_seed, int32_t year, int64_t cand0, const RoundKey *key,
              const tcr_seeds *out, void *stream_)
{
    if (ready(ctx, true)) return -1;
    if (!out) return fail(ctx, "tcr_seed_dev: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (out->n <= 0) return 0;
    for (auto &s : ctx->slots) if (!s.rh) return fail(ctx, "tcr_seed_dev: a slot has no rh_mid (tcr_rh_upload)");
    if (!ctx->rg.set) return fail(ctx, "tcr_seed_dev: rh_mid not staged (tcr_rh_upload)");
    if (ctx->slots.size() < 12) return fail(ctx, "tcr_seed_dev: needs the 12 month slots staged");
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    SeedArgs a{};
    a.P = ctx->prm; a.D = dev_fields(ctx); a.seed = experiment_seed; a.year = year; a.cand0 = cand0; a.out = *out; a.key = key;
    hipLaunchKernelGGL(k_seed, dim3((unsigned)((out->n + 255) / 256)), dim3(256), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
}  // namespace
}  // extern "C++"

int tcr_seed_dev(tcr_ctx *ctx, uint64_t experiment_seed, int32_t year, int64_t cand0,
                 const tcr_seeds *out, void *stream_)
{
    return seed_impl(ctx, experiment_seed, year, cand0, nullptr, out, stream_);
}

int tcr_seed_host(tcr_ctx *ctx, uint64_t experiment_seed, int32_t year, int64_t cand0, const tcr_seeds *out)
{
    if (ready(ctx, true)) return -1;
    if (!out) return fail(ctx, "tcr_seed_host: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)out->n, N = (size_t)ctx->prm.n_series;
    if (n == 0) return 0;
    DevBuf B;
    tcr_seeds d{};
    d.n = out->n;
    d.lon0 = B.get<double>(n); d.lat0 = B.get<double>(n); d.v0 = B.get<double>(n); d.m0 = B.get<double>(n);
    d.h_bl = B.get<double>(n); d.slot = B.get<int32_t>(n); d.phases = B.get<double>(n * 4 * N);
    d.basin_idx = B.get<int32_t>(n); d.seed_flags = B.get<int32_t>(n);
    if (!d.lon0 || !d.lat0 || !d.v0 || !d.m0 || !d.h_bl || !d.slot || !d.phases || !d.basin_idx || !d.seed_flags)
        return fail(ctx, "tcr_seed_host: device allocation failed");
    if (tcr_seed_dev(ctx, experiment_seed, year, cand0, &d, ctx->stream)) return -1;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
#define D2H(field, count, T) \
    if (out->field) HIPCHK(ctx, copy_sync(ctx->stream, out->field, d.field, (count) * sizeof(T), hipMemcpyDeviceToHost))
    D2H(lon0, n, double); D2H(lat0, n, double); D2H(v0, n, double); D2H(m0, n, double); D2H(h_bl, n, double);
    D2H(slot, n, int32_t); D2H(phases, n * 4 * N, double); D2H(basin_idx, n, int32_t); D2H(seed_flags, n, int32_t);
#undef D2H
    return 0;
}

extern "C++" {
namespace {
// cell != NULL: the selection's cell keys and counts are produced on the way (k_compact<true>), for cell_order_impl(key_done)
int compact_impl(tcr_ctx *ctx, int64_t n, const int32_t *flags, int32_t mask, int64_t max_out,
                 int32_t *idx, int64_t *count, void *stream_, const CellOrderArgs *cell)
{
    if (!ctx) return -1;
    if (!flags || !idx || !count || n < 0 || max_out < 0) return fail(ctx, "tcr_compact_dev: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    const int64_t tiles = std::max<int64_t>(1, (n + kScanTile - 1) / kScanTile);       // (n == 0: one tile writes *count = 0)
    if ((size_t)tiles + 2 > ctx->tiles_cap) {
        if (ctx->d_tiles) HIPCHK(ctx, hipFree(ctx->d_tiles));
        ctx->d_tiles = nullptr; ctx->tiles_cap = 0;
        if (dev_alloc(ctx, &ctx->d_tiles, (size_t)tiles + 1024)) return -1;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_tiles, 0, sizeof(unsigned long long) * ((size_t)tiles + 1024), st));
        ctx->tiles_cap = (size_t)tiles + 1024;
    }
    if (cell) hipLaunchKernelGGL(k_compact<true>, dim3((unsigned)tiles), dim3(kScanThreads), 0, st, n, flags, mask, max_out, idx, count, ctx->d_tiles, (int)tiles, *cell);
    else hipLaunchKernelGGL(k_compact<false>, dim3((unsigned)tiles), dim3(kScanThreads), 0, st, n, flags, mask, max_out, idx, count, ctx->d_tiles, (int)tiles, CellOrderArgs{});
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
}  // namespace
}  // extern "C++"

int tcr_compact_dev(tcr_ctx *ctx, int64_t n, const int32_t *flags, int32_t mask, int64_t max_out,
                    int32_t *idx, int64_t *count, void *stream_)
{
    return compact_impl(ctx, n, flags, mask, max_out, idx, count, stream_, nullptr);
}

extern "C++" {
namespace {
// scratch and arguments of the locality order of idx[0 .. min(n, *count))
int cell_order_args(tcr_ctx *ctx, const tcr_seeds *cand, int32_t *idx, int64_t n, const int64_t *count, double cell_deg,
                    hipStream_t st, CellOrderArgs &a)
{
    if (!cand || !cand->lon0 || !cand->lat0 || !idx || n < 0) return fail(ctx, "tcr_cell_order_dev: bad argument");
    if (!(cell_deg >= 0.25 && cell_deg <= 90.0)) return fail(ctx, "tcr_cell_order_dev: cell_deg must be in [0.25, 90]");
    a = CellOrderArgs{};
    a.ncol = (int)ceil(360.0 / cell_deg);
    const int nrow = (int)ceil(180.0 / cell_deg) + 1;
    a.nbins = a.ncol * nrow;
    if (a.nbins > (1 << 20)) return fail(ctx, "tcr_cell_order_dev: too many cells");
    // scratch (int32): cell counters [nbins + 1] (zero between calls: k_cell_rank leaves them so), offsets [nbins + 1], the list
    // of long cells [nbins + 2], then idx copy [n], key [n], tmp [n], tmp_key [n]
    const size_t head = 3 * ((size_t)a.nbins + 1) + 8;
    const size_t words = head + 4 * (size_t)n;
    if ((words + 1) / 2 > ctx->cell_cap || ctx->cell_bins != a.nbins) {
        if (grow(ctx, &ctx->d_cell, &ctx->cell_cap, (words + 1) / 2 + 4096)) return -1;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_cell, 0, sizeof(int32_t) * head, st));
        ctx->cell_bins = a.nbins;
        ctx->cell_counts_open = false;
    } else if (ctx->cell_counts_open && !ctx->capturing) {
        // an earlier ordering was abandoned between the kernel that counts the cells and the one that leaves the counters zero
        // (an enqueue error in between): the counters are not relied on, they are cleared
        HIPCHK(ctx, hipMemsetAsync(ctx->d_cell, 0, sizeof(int32_t) * head, st));
        ctx->cell_counts_open = false;
    }
    int32_t *w = reinterpret_cast<int32_t *>(ctx->d_cell);
    a.hist = w; a.start = w + a.nbins + 1; a.long_cells = w + 2 * (a.nbins + 1);
    int32_t *body = w + head;
    a.idx_in = body; a.key = body + n; a.tmp = body + 2 * n; a.tmp_key = body + 3 * n;
    a.lon0 = cand->lon0; a.lat0 = cand->lat0; a.idx_out = idx; a.count = count; a.n = n;
    a.inv_cell = 1.0 / cell_deg;
    return 0;
}

// key_done: the keys and cell counts came with the compaction (k_compact<true>)
int cell_order_launch(tcr_ctx *ctx, const CellOrderArgs &a, bool key_done, hipStream_t st)
{
    const int64_t n = a.n;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    ctx->cell_counts_open = true;
    if (!key_done) hipLaunchKernelGGL(k_cell_key, dim3(blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_cell_scan, dim3(1), dim3(kCellScanThreads), 0, st, a);
    hipLaunchKernelGGL(k_cell_scatter, dim3(blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_cell_rank, dim3((unsigned)((std::max<int64_t>(n, a.nbins + 1) + 255) / 256)), dim3(256), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    ctx->cell_counts_open = false;              // k_cell_rank leaves the counters zero
    return 0;
}
}  // namespace
}  // extern "C++"

int tcr_cell_order_dev(tcr_ctx *ctx, const tcr_seeds *cand, int32_t *idx, int64_t n, const int64_t *count,
                       double cell_deg, void *stream_)
{
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    CellOrderArgs a{};
    if (cell_order_args(ctx, cand, idx, n, count, cell_deg, st, a)) return -1;
    if (n == 0) return 0;
    return cell_order_launch(ctx, a, false, st);
}

extern "C++" {
namespace {
int gather_impl(tcr_ctx *ctx, const tcr_seeds *src, const int32_t *idx, int64_t n_out, const int64_t *count,
                const tcr_seeds *dst, uint64_t experiment_seed, int32_t year, int64_t cand0, const RoundKey *key, void *stream_)
{
    if (!ctx) return -1;
    if (!ctx->have_prm) return fail(ctx, "tcr_params_set has not been called");
    if (!src || !dst || !idx) return fail(ctx, "tcr_gather_seeds_dev: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n_out <= 0) return 0;
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    GatherSeedArgs a{};
    a.src = *src; a.dst = *dst; a.idx = idx; a.n_out = n_out; a.phases_per_storm = 4 * ctx->prm.n_series;
    a.seed = experiment_seed; a.year = year; a.cand0 = cand0; a.count = count; a.key = key;
    const int64_t threads = n_out * kGatherLanes;
    hipLaunchKernelGGL(k_gather_seeds, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
}  // namespace
}  // extern "C++"

int tcr_gather_seeds_dev(tcr_ctx *ctx, const tcr_seeds *src, const int32_t *idx, int64_t n_out,
                         const int64_t *count, const tcr_seeds *dst, uint64_t experiment_seed, int32_t year,
                         int64_t cand0, void *stream_)
{
    return gather_impl(ctx, src, idx, n_out, count, dst, experiment_seed, year, cand0, nullptr, stream_);
}

int tcr_stats_dev(tcr_ctx *ctx, int64_t n, const int64_t *n_dev, const tcr_tracks *t, uint64_t *out, int32_t n_out, void *stream_)
{
    if (!ctx) return -1;
    if (!t || !out) return fail(ctx, "tcr_stats_dev: NULL argument");
    if (n_out != 6 && n_out != 8 && n_out != TCR_N_STATS) return fail(ctx, "tcr_stats_dev: n_out must be 6, 8 or 10 (the capacity of out_dev in words)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n <= 0) return 0;
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_stats, dim3((unsigned)blocks), dim3(256), 0, st, n, n_dev, t->n_valid, t->nfev, t->flags, t->status,
                       reinterpret_cast<unsigned long long *>(out), (int)n_out);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

extern "C++" {
namespace {
template <typename R, typename T>
int pack_impl(tcr_ctx *ctx, const T *src, const int32_t *idx, const int64_t *count, int64_t cap, double *packed,
              int64_t row_stride, void *stream_, const int32_t *cand_idx = nullptr, const int32_t *slot = nullptr,
              const int32_t *basin_idx = nullptr, int64_t cand0 = 0, const RoundKey *key = nullptr)
{
    if (!ctx) return -1;
    if (!ctx->have_prm) return fail(ctx, "tcr_params_set has not been called");
    if (!src || !idx || !count || !packed) return fail(ctx, "tcr_pack_tracks: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (cap <= 0) return 0;
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    PackArgsT<R> a{};
    a.lon = src->lon; a.lat = src->lat; a.v = src->v; a.m = src->m; a.vmax = src->vmax; a.envw = src->envw;
    a.idx = idx; a.count = count; a.cap = cap; a.ns = ctx->prm.n_steps; a.packed = packed;
    a.row_stride = row_stride > 0 ? row_stride : 9 * (int64_t)ctx->prm.n_steps;
    if (a.row_stride < 9 * (int64_t)ctx->prm.n_steps) return fail(ctx, "tcr_pack_tracks: row_stride < 9 * n_steps");
    if (slot) {
        if (!basin_idx) return fail(ctx, "tcr_pack_tracks_meta_dev: slot and basin_idx go together");
        if (a.row_stride < 9 * (int64_t)ctx->prm.n_steps + 3) return fail(ctx, "tcr_pack_tracks_meta_dev: row_stride < 9 * n_steps + 3");
        a.cand_idx = cand_idx; a.slot = slot; a.basin_idx = basin_idx; a.cand0 = cand0; a.key = key;
    }
    hipLaunchKernelGGL(k_pack_tracks<R>, dim3((unsigned)cap), dim3(256), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
}  // namespace
}  // extern "C++"

int tcr_pack_tracks_dev(tcr_ctx *ctx, const tcr_tracks *src, const int32_t *idx, const int64_t *count,
                        int64_t cap, double *packed, int64_t row_stride, void *stream_)
{
    return pack_impl<double>(ctx, src, idx, count, cap, packed, row_stride, stream_);
}

int tcr_pack_tracks_f32_dev(tcr_ctx *ctx, const tcr_tracks_f32 *src, const int32_t *idx, const int64_t *count,
                            int64_t cap, double *packed, int64_t row_stride, void *stream_)
{
    return pack_impl<float>(ctx, src, idx, count, cap, packed, row_stride, stream_);
}

int tcr_pack_tracks_meta_dev(tcr_ctx *ctx, const tcr_tracks *src, int32_t f32, const int32_t *idx, const int64_t *count,
                             int64_t cap, double *packed, int64_t row_stride, const int32_t *cand_idx, const int32_t *slot,
                             const int32_t *basin_idx, int64_t cand0, void *stream_)
{
    if (ctx && (!slot || !basin_idx)) return fail(ctx, "tcr_pack_tracks_meta_dev: NULL argument");
    if (f32) return pack_impl<float>(ctx, reinterpret_cast<const tcr_tracks_f32 *>(src), idx, count, cap, packed, row_stride, stream_,
                                     cand_idx, slot, basin_idx, cand0);
    return pack_impl<double>(ctx, src, idx, count, cap, packed, row_stride, stream_, cand_idx, slot, basin_idx, cand0);
}

extern "C++" {
namespace {
int seed_hist_impl(tcr_ctx *ctx, const tcr_seeds *cand, int64_t n, int64_t cand0, const RoundKey *key, const double *cutoff,
                   int64_t *out, void *stream_)
{
    if (!ctx) return -1;
    if (!cand || !cand->seed_flags || !cand->basin_idx || !cand->slot || !out) return fail(ctx, "tcr_seed_hist_dev: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    SeedHistArgs a{};
    a.seed_flags = cand->seed_flags; a.basin_idx = cand->basin_idx; a.slot = cand->slot; a.n = n > 0 ? n : 0; a.cand0 = cand0;
    a.key = key; a.cutoff = cutoff; a.out = reinterpret_cast<unsigned long long *>(out);
    if (!ctx->d_hist_partial) {
        const size_t words = (size_t)kSeedHistBlocks * TCR_N_BASINS * 12 + 16;
        if (dev_alloc(ctx, &ctx->d_hist_partial, words)) return -1;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_hist_partial, 0, sizeof(unsigned int) * words, st));
    }
    a.partial = ctx->d_hist_partial;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(kSeedHistBlocks, (a.n + 1023) / 1024));
    hipLaunchKernelGGL(k_seed_hist, dim3((unsigned)blocks), dim3(256), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// Everything a round enqueues (see tcr_round_dev in the header).  key != NULL: the replayable form — seed / year / cand0
// are read on the device.
int enqueue_round(tcr_ctx *ctx, const tcr_round *r, uint64_t seed, int32_t year, int64_t cand0, const RoundKey *key, void *st)
{
    tcr_seeds cand = r->cand, storms = r->storms;
    cand.n = r->n_cand; storms.n = r->n_storms;
    STAGE(TCR_STAGE_START);
    if (seed_impl(ctx, seed, year, cand0, key, &cand, st)) return -1;
    STAGE(TCR_STAGE_SEED);
#ifdef TCR_EXPERIMENTS
    tcr_exp::dummy_launches((hipStream_t)st, reinterpret_cast<unsigned long long *>(ctx->d_cell));
#endif
    if (r->cell_deg > 0) {
        // locality order: with many cells the keys and cell counts are produced by the compaction itself (one launch less)
        CellOrderArgs ca{};
        if (cell_order_args(ctx, &cand, r->cand_idx, r->n_storms, r->n_passed, r->cell_deg, (hipStream_t)st, ca)) return -1;
        const bool fused = ca.nbins > kCellLdsBins;
        if (fused) ctx->cell_counts_open = true;
        if (compact_impl(ctx, r->n_cand, cand.seed_flags, 2, r->n_storms, r->cand_idx, r->n_passed, st, fused ? &ca : nullptr)) return -1;
        STAGE(TCR_STAGE_SELECT);
        if (cell_order_launch(ctx, ca, fused, (hipStream_t)st)) return -1;
    } else {
        if (compact_impl(ctx, r->n_cand, cand.seed_flags, 2, r->n_storms, r->cand_idx, r->n_passed, st, nullptr)) return -1;
        STAGE(TCR_STAGE_SELECT);
    }
    STAGE(TCR_STAGE_ORDER);
    if (gather_impl(ctx, &cand, r->cand_idx, r->n_storms, r->n_passed, &storms, seed, year, cand0, key, st)) return -1;
    STAGE(TCR_STAGE_GATHER);
    tcr_storms in{};
    in.n = r->n_storms; in.lon0 = storms.lon0; in.lat0 = storms.lat0; in.v0 = storms.v0; in.m0 = storms.m0; in.h_bl = storms.h_bl;
    in.slot = storms.slot; in.phases = storms.phases; in.n_dev = r->exact_count ? r->n_passed : nullptr;
    if (r->f32) {
        if (ensure_f32(ctx, (hipStream_t)st)) return -1;
        if (integrate_impl<float>(ctx, &in, tracks_of<float>(reinterpret_cast<const tcr_tracks_f32 *>(&r->tracks)), st, r->stats, r->n_passed, r->n_expected)) return -1;
    } else if (integrate_impl<double>(ctx, &in, tracks_of<double>(&r->tracks), st, r->stats, r->n_passed, r->n_expected)) return -1;
    // (the stats of tcr_stats_dev are accumulated by the batch's last kernel, k_flags)
    STAGE(TCR_STAGE_STATS);
    if (r->acc_idx) {
        if (!r->n_accepted) return fail(ctx, "tcr_round_dev: acc_idx without n_accepted");
        if (tcr_compact_dev(ctx, r->n_storms, r->tracks.flags, TCR_FLAG_ACCEPTED, r->n_storms, r->acc_idx, r->n_accepted, st)) return -1;
        if (r->packed && r->pack_cap > 0) {
            // the meta columns go with a record that has room for them (pack_stride >= 9 * n_steps + 3)
            const bool meta = r->pack_stride >= 9 * (int64_t)ctx->prm.n_steps + 3;
            const int32_t *ci = meta ? r->cand_idx : nullptr, *sl = meta ? storms.slot : nullptr, *bi = meta ? storms.basin_idx : nullptr;
            if (r->f32 ? pack_impl<float>(ctx, reinterpret_cast<const tcr_tracks_f32 *>(&r->tracks), r->acc_idx, r->n_accepted, r->pack_cap,
                                          r->packed, r->pack_stride, st, ci, sl, bi, cand0, key)
                       : pack_impl<double>(ctx, &r->tracks, r->acc_idx, r->n_accepted, r->pack_cap, r->packed, r->pack_stride, st,
                                           ci, sl, bi, cand0, key)) return -1;
        }
    }
    if (r->seed_hist && seed_hist_impl(ctx, &cand, r->n_cand, cand0, key, nullptr, r->seed_hist, st)) return -1;
    STAGE(TCR_STAGE_PACK);
    return 0;
}

void drop_graphs(tcr_ctx *ctx)
{
    for (auto &g : ctx->graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
    ctx->graphs.clear();
}
}  // namespace
}  // extern "C++"

int tcr_seed_hist_dev(tcr_ctx *ctx, const tcr_seeds *cand, int64_t cand0, const double *cutoff, int64_t *out, void *stream_)
{
    return seed_hist_impl(ctx, cand, cand ? cand->n : 0, cand0, nullptr, cutoff, out, stream_);
}

int tcr_round_dev(tcr_ctx *ctx, const tcr_round *r, uint64_t seed, int32_t year, int64_t cand0, int32_t use_graph, void *stream_)
{
    if (ready(ctx, true)) return -1;
    if (!r) return fail(ctx, "tcr_round_dev: NULL argument");
    if (r->n_cand <= 0 || r->n_storms <= 0) return fail(ctx, "tcr_round_dev: n_cand and n_storms must be positive");
    if (r->n_storms > r->n_cand) return fail(ctx, "tcr_round_dev: n_storms > n_cand");
    if (!r->cand_idx || !r->n_passed || !r->cand.seed_flags || !r->cand.lon0 || !r->storms.lon0 || !r->storms.phases ||
        !r->storms.basin_idx || !r->tracks.flags || !r->tracks.n_valid)
        return fail(ctx, "tcr_round_dev: NULL buffer");
    if (r->cell_deg != 0 && !(r->cell_deg >= 0.25 && r->cell_deg <= 90.0)) return fail(ctx, "tcr_round_dev: cell_deg must be 0 or in [0.25, 90]");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    if (!use_graph || ctx->d_probe) return enqueue_round(ctx, r, seed, year, cand0, nullptr, st);

    // ---- replayed form
    std::vector<uint8_t> key(sizeof(tcr_round) + sizeof(uint64_t) + sizeof(void *));
    memcpy(key.data(), r, sizeof(tcr_round));
    {   // the one run of padding bytes in the descriptor (behind tcr_tracks.tc_rows_only) must not take part in the comparison
        constexpr size_t pad0 = offsetof(tcr_round, tracks) + offsetof(tcr_tracks, tc_rows_only) + sizeof(int32_t);
        constexpr size_t pad1 = offsetof(tcr_round, tracks) + sizeof(tcr_tracks);
        static_assert(pad1 >= pad0 && sizeof(tcr_seeds) == 80 && sizeof(tcr_round) == 384, "tcr_round layout changed: revisit the key");
        memset(key.data() + pad0, 0, pad1 - pad0);
    }
    memcpy(key.data() + sizeof(tcr_round), &ctx->epoch, sizeof(uint64_t));
    memcpy(key.data() + sizeof(tcr_round) + sizeof(uint64_t), &st, sizeof(void *));
    tcr_ctx::RoundGraph *g = nullptr;
    for (auto &c : ctx->graphs) if (c.key == key) { g = &c; break; }
    if (!g) {
        // First sight of this descriptor (or the context has allocated / changed parameters since): run the round directly
        // — that is this call's result and it sizes every workspace — then capture the same enqueue for the calls to come.
        if (!ctx->graphs.empty() && memcmp(ctx->graphs.front().key.data() + sizeof(tcr_round), &ctx->epoch, sizeof(uint64_t)) != 0)
            drop_graphs(ctx);                    // graphs of an older epoch hold stale addresses
        if (ctx->graphs.size() >= 64) drop_graphs(ctx);       // a caller that keeps changing its buffers: start over rather than grow
        if (enqueue_round(ctx, r, seed, year, cand0, nullptr, st)) return -1;
        memcpy(key.data() + sizeof(tcr_round), &ctx->epoch, sizeof(uint64_t));     // the direct run may have grown workspaces
        ctx->graphs.emplace_back();
        g = &ctx->graphs.back();
        g->key = key;
        // Capture.  HIP's capture state is fragile against the rest of the process: a legacy-stream operation on ANY thread
        // while this stream captures (a synchronous hipMemcpy of another context's field upload, say) fails there with
        // "would make the legacy stream depend on a capturing blocking stream" and invalidates the capture here.  Whatever
        // happens, the stream must leave capture mode and the sticky error must be cleared; the descriptor then keeps the
        // direct form (this call's round has already run).
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { g->failed = true; (void)hipGetLastError(); return 0; }
        ctx->capturing = true;
        const int rc = enqueue_round(ctx, r, 0, 0, 0, ctx->d_round_key, st);
        ctx->capturing = false;
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(st, &graph);
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            hipGraph_t junk = nullptr;                      // still (invalidly) capturing: end it once more
            (void)hipStreamEndCapture(st, &junk);
            if (junk) (void)hipGraphDestroy(junk);
            e = hipErrorStreamCaptureInvalidated;
        }
        for (int k = 0; k < 4 && hipGetLastError() != hipSuccess; ++k) {}
        if (rc || e != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            ctx->err.clear();
            g->failed = true;                    // this descriptor keeps the direct form; the round above has run
            return 0;
        }
#ifdef TCR_EXPERIMENTS
        tcr_exp::graph_dot(graph);
#endif
        hipGraphExec_t exec = nullptr;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || !exec) {
            (void)hipGraphDestroy(graph); (void)hipGetLastError();
            g->failed = true;
            return 0;
        }
        g->graph = graph; g->exec = exec;
        return 0;
    }
    if (g->failed) return enqueue_round(ctx, r, seed, year, cand0, nullptr, st);
    // (the fp32 copies of re-staged fields are refreshed here: the captured enqueue found them up to date and holds no conversion)
    if (r->f32 && ensure_f32(ctx, st)) return -1;
    hipLaunchKernelGGL(k_set_round_key, dim3(1), dim3(1), 0, st, ctx->d_round_key, seed, year, cand0);
    HIPCHK(ctx, hipGraphLaunch(g->exec, st));
    ++ctx->n_replays;
    return 0;
}

int tcr_schedule_set(tcr_ctx *ctx, int32_t storms_per_lane)
{
    if (!ctx) return -1;
    if (storms_per_lane < 1 || storms_per_lane > 64) return fail(ctx, "tcr_schedule_set: storms_per_lane must be in [1, 64]");
    if (storms_per_lane != ctx->storms_per_lane) ++ctx->epoch;        // captured rounds hold the launch shape
    ctx->storms_per_lane = storms_per_lane;
    return 0;
}

int tcr_tune_set(tcr_ctx *ctx, const tcr_tune *t)
{
    if (!ctx || !t) return -1;
    if (t->park > 63) return fail(ctx, "tcr_tune_set: park must be < 64");
    if (t->copy_threads != ctx->tune.copy_threads) ctx->pool.reset();       // (the next slot upload starts the pool it asks for)
    ctx->tune = *t;
    ++ctx->epoch;                  // captured rounds hold the launch shape
    return 0;
}

int tcr_tune_get(tcr_ctx *ctx, tcr_tune *t)
{
    if (!ctx || !t) return -1;
    *t = ctx->tune;
    return 0;
}

int tcr_stage_trace_enable(tcr_ctx *ctx, int on)
{
    if (!ctx) return -1;
    ctx->stage_trace = on != 0;
    ctx->st_used = 0;
    return 0;
}

int tcr_stage_trace_sum(tcr_ctx *ctx, double ms[TCR_N_STAGES], int64_t *n_rounds)
{
    if (!ctx || !ms) return -1;
    for (int i = 0; i < TCR_N_STAGES; ++i) ms[i] = 0.0;
    int64_t rounds = 0;
    if (ctx->st_used) HIPCHK(ctx, hipEventSynchronize(ctx->st_pool[ctx->st_used - 1]));
    for (size_t i = 0; i < ctx->st_used; ++i) {
        const int id = ctx->st_id[i];
        if (id == TCR_STAGE_START) { ++rounds; continue; }
        if (i == 0 || id < 0 || id >= TCR_N_STAGES) continue;
        float f = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&f, ctx->st_pool[i - 1], ctx->st_pool[i]));
        ms[id] += f;
    }
    if (n_rounds) *n_rounds = rounds;
    return 0;
}

int tcr_round_graph_stats(tcr_ctx *ctx, int64_t *n_graphs, int64_t *n_replays)
{
    if (!ctx) return -1;
    int64_t n = 0;
    for (auto &g : ctx->graphs) n += g.exec ? 1 : 0;
    if (n_graphs) *n_graphs = n;
    if (n_replays) *n_replays = ctx->n_replays;
    return 0;
}

}  // extern "C"

#include "tcr_comm.hip"                  // multi-GPU exchange (RCCL, loaded at run time)
