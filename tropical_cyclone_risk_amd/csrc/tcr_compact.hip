// Order-preserving stream compaction and row gathers.
//
// The reference's accept loop is sequential: "keep drawing candidates until
// n_tracks have survived" (util/compute.py:134-209).  Batched, that becomes
// "seed a round of candidates, keep the ones that pass *in candidate order*,
// integrate, keep the accepted ones in candidate order".  Order preservation is
// what makes the result independent of batch size and GPU count, so compaction
// is a scan, not an atomic append.
#include "tcr_device.h"
#include "tcr_seed.hip"

namespace tcr {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;                       // items per thread
constexpr int kScanTile = kScanThreads * kScanItems;

// ---------------------------------------------------------------------------------------------------------------
// Locality order of the dense batch (round 3).  The step is bound by L2 misses on the 142 MB field set:
// storms are drawn at random positions, so the 64 lanes of an integrator wave gather from 64 unrelated places.  Ordering
// the selected candidates by the 2-degree cell of their genesis point (latitude row major) — a stable counting sort of the
// index list, before the seeds are gathered, so no storm data moves — puts neighbours into the same wave: the
// integrator's L2 misses drop by ~20 %, the 100 000-storm step by 5-7 %.  Per-storm results do not depend on the order.
// key = row * ncol + col; bins <= 65536.
struct CellOrderArgs {
    const double *lon0, *lat0;       // candidate arrays (indexed by idx[])
    int32_t *idx_in;                 // [n] scratch: the selection (ascending candidate indices), copied aside by k_cell_key
    int32_t *idx_out;
    const int64_t *count;            // device scalar: how many entries of idx_in are valid (may be NULL: n)
    int64_t n;
    double inv_cell;
    int ncol, nbins;
    int32_t *key;                    // [n] scratch
    int32_t *hist;                   // [nbins + 1]: counts -> exclusive offsets (k_cell_scan) -> running cursors (k_cell_scatter)
    int32_t *start;                  // [nbins + 1]: a copy of the exclusive offsets for k_cell_rank
    int32_t *tmp, *tmp_key;          // [n] each: the scatter's output (cell-grouped, unordered inside a cell) and its keys
    int32_t *long_cells;             // [nbins + 1]: [0] how many cells hold more than kCellLong entries, then those cells (k_cell_scan)
};

// A cell with more entries than this is not ranked entry by entry (one thread scanning the whole segment per entry: quadratic
// in the cell's population) but sorted as a segment by a workgroup (k_cell_rank's second half).  Typical cells — 2-degree
// cells, 100 000 storms over a basin — hold 5-30 entries; a small basin with large cells, or NaN genesis points (all in
// cell 0), put thousands into one.
constexpr int kCellLong = 96;
constexpr int kCellSortChunk = 4096;

// the cell of a genesis point (k_cell_key; also inside k_compact when a round asks for the locality order)
__device__ __forceinline__ int cell_of(const CellOrderArgs &a, double lo, double la)
{
    lo = lo - 360.0 * floor(lo / 360.0);
    int col = (int)(lo * a.inv_cell), row = (int)((la + 90.0) * a.inv_cell);
    const int nrow = a.nbins / a.ncol;
    col = col < 0 ? 0 : (col >= a.ncol ? a.ncol - 1 : col);
    row = row < 0 ? 0 : (row >= nrow ? nrow - 1 : row);
    return (lo == lo && la == la) ? row * a.ncol + col : 0;          // (a NaN position: cell 0)
}

// One launch (three until round 4: count, a 1024-thread scan workgroup, write).  A tile's workgroup counts its selected
// items, publishes the count, adds up the counts of the tiles in front of it — a look-back over their published words, one
// word per thread, no chain between tiles — and writes its indices at their ranks.  Small workgroups matter as much as the
// saved launches: next to an integrator wave a SIMD has 88 registers left, so a 1024-thread workgroup (four waves per SIMD)
// had to wait for a CU without integrator waves.
//   state[0]      ticket counter: a workgroup's tile is its ticket, so every tile in front of it is already running
//   state[1]      generation of the last finished launch
//   state[2 + t]  (generation << 32) | count of tile t — valid for this launch when the generation matches
// The last tile writes *total and leaves ticket = 0, generation + 1 behind for the next launch on this scratch (launches
// that share a scratch are ordered by their stream).
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// KEY: the selection is about to be put into locality order (tcr_round_dev): what k_cell_key would do in a launch of its own
// — copy the entry aside, compute its cell, count the cell — happens where the entry is written.
template <bool KEY>
__global__ __launch_bounds__(kScanThreads) void k_compact(int64_t n, const int32_t *__restrict__ flags, int32_t mask,
                                                          int64_t max_out, int32_t *__restrict__ idx, int64_t *__restrict__ total,
                                                          unsigned long long *__restrict__ state, int n_tiles, CellOrderArgs cell)
{
    __shared__ int s[kScanThreads];
    __shared__ int s_tile;
    __shared__ unsigned s_gen;
    __shared__ long long s_prefix[kScanThreads / 64];
    if (threadIdx.x == 0) {
        s_tile = (int)__hip_atomic_fetch_add(state, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_gen = (unsigned)ld_agent(state + 1) + 1u;
    }
    __syncthreads();
    const int tile = s_tile;
    const unsigned gen = s_gen;
    const int64_t base = (int64_t)tile * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int c = 0;
    bool sel[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        sel[k] = i < n && (flags[i] & mask);
        c += sel[k];
    }
    s[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < kScanThreads; off <<= 1) {
        const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    const int tile_total = s[kScanThreads - 1];
    if (threadIdx.x == 0) st_agent(state + 2 + tile, ((unsigned long long)gen << 32) | (unsigned)tile_total);
    // look-back: the counts of tiles [0, tile), a word per thread
    long long before = 0;
    for (int p = threadIdx.x; p < tile; p += kScanThreads) {
        unsigned long long w = ld_agent(state + 2 + p);
        while ((unsigned)(w >> 32) != gen) { __builtin_amdgcn_s_sleep(1); w = ld_agent(state + 2 + p); }
        before += (long long)(unsigned)w;
    }
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
    if ((threadIdx.x & 63) == 0) s_prefix[threadIdx.x >> 6] = before;
    __syncthreads();
    long long prefix = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) prefix += s_prefix[w];
    if (tile == n_tiles - 1 && threadIdx.x == 0) {
        *total = prefix + tile_total;
        st_agent(state, 0ull);                          // every ticket of this launch has been taken
        st_agent(state + 1, (unsigned long long)gen);   // ... and every tile in front has read the generation (it published with it)
    }
    int64_t rank = prefix + s[threadIdx.x] - c;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (sel[k]) {
            if (rank < max_out) {
                const int32_t c = (int32_t)(base + k);
                idx[rank] = c;
                if (KEY) {
                    cell.idx_in[rank] = c;
                    const int kk = cell_of(cell, cell.lon0[c], cell.lat0[c]);
                    cell.key[rank] = kk;
                    atomicAdd(cell.hist + kk, 1);
                }
            }
            ++rank;
        }
}

// Gather candidate rows idx[0..n_out) into a dense storm batch.
struct GatherSeedArgs {
    tcr_seeds src, dst;
    const int32_t *idx;
    int64_t n_out;
    int phases_per_storm;
    uint64_t seed;
    int32_t year;
    int64_t cand0;
    const int64_t *count;       // device scalar from tcr_compact_dev: rows >= *count are not valid (NULL: all are)
    const RoundKey *key;        // NULL: seed / year / cand0 above (tcr_seed.hip)
};

// 16 lanes per output row (four rows per wave: the kernel is bound by its two dependent loads — idx[row], then
// src[idx] — so rows per wave is what counts): lane 15 copies the scalars, the others draw / copy the phases.
constexpr int kGatherLanes = 16;

__global__ __launch_bounds__(256) void k_gather_seeds(GatherSeedArgs a)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = gid / kGatherLanes;
    const int lane = (int)(gid % kGatherLanes);
    if (row >= a.n_out || (a.count && row >= *a.count)) return;
    const int32_t j = a.idx[row];
    if (lane == kGatherLanes - 1) {
        a.dst.lon0[row] = a.src.lon0[j]; a.dst.lat0[row] = a.src.lat0[j];
        a.dst.v0[row] = a.src.v0[j]; a.dst.m0[row] = a.src.m0[j]; a.dst.h_bl[row] = a.src.h_bl[j];
        a.dst.slot[row] = a.src.slot[j];
        if (a.dst.basin_idx) a.dst.basin_idx[row] = a.src.basin_idx[j];
        if (a.dst.seed_flags) a.dst.seed_flags[row] = a.src.seed_flags[j];
    }
    double *dp = a.dst.phases + (size_t)row * a.phases_per_storm;
    if (a.src.phases) {
        const double *sp = a.src.phases + (size_t)j * a.phases_per_storm;
        for (int k = lane; k < a.phases_per_storm; k += kGatherLanes) dp[k] = sp[k];
    } else {
        const uint64_t seed = a.key ? a.key->seed : a.seed;
        const int32_t year = a.key ? a.key->year : a.year;
        const int64_t cand0 = a.key ? a.key->cand0 : a.cand0;
        // one Philox block yields the pair (2p, 2p + 1): a lane per pair, not per phase
        for (int p = lane; 2 * p < a.phases_per_storm; p += kGatherLanes) {
            double p0, p1;
            uniform2_raw(seed, year, cand0 + j, 2u, (uint32_t)p, p0, p1);
            dp[2 * p] = p0;
            if (2 * p + 1 < a.phases_per_storm) dp[2 * p + 1] = p1;
        }
    }
}

__device__ __forceinline__ int64_t cell_n(const CellOrderArgs &a) { return (a.count && *a.count < a.n) ? *a.count : a.n; }

// With few cells (large cells / a small basin) thousands of entries hit the same counter: the workgroup counts in LDS first
// and adds each non-empty cell to the global counter once (50 000 storms in 6 cells: 2.4 -> 0.1 ms for the whole order).
constexpr int kCellLdsBins = 2048;

__global__ __launch_bounds__(256) void k_cell_key(CellOrderArgs a)
{
    __shared__ int h[kCellLdsBins];
    const bool local = a.nbins <= kCellLdsBins;
    if (local) {
        for (int b = threadIdx.x; b < a.nbins; b += 256) h[b] = 0;
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < cell_n(a)) {
        const int32_t c = a.idx_out[i];                   // the selection as tcr_compact_dev left it (in place: idx_out == the list)
        a.idx_in[i] = c;                                  // ... copied aside: the ranked result overwrites the list
        const int k = cell_of(a, a.lon0[c], a.lat0[c]);
        a.key[i] = k;
        if (local) atomicAdd(&h[k], 1); else atomicAdd(a.hist + k, 1);
    }
    if (local) {
        __syncthreads();
        for (int b = threadIdx.x; b < a.nbins; b += 256) if (h[b]) atomicAdd(a.hist + b, h[b]);
    }
}

// exclusive scan of the cell counts by one workgroup of 256 threads (1024 until round 4: four waves per SIMD do not fit next to
// an integrator wave, and under load the launch waited for a CU without one): the counts are staged in LDS with coalesced
// loads (cells beyond the LDS window are handled in further rounds with a carry), every thread owns a contiguous run of 64
// cells — rows padded by one word so that the 64 lanes of a wave walk 64 different banks — and the 256 run totals are scanned
// once per round
constexpr int kCellScanThreads = 256;
constexpr int kCellScanRun = 64;
constexpr int kCellScanWin = kCellScanThreads * kCellScanRun;
// k_cell_scan's window lives in static LDS (66.6 KB): more than the 64 KB of the CDNA parts before gfx950 (160 KB per CU).
// This library is built for gfx950 only; tcr_ctx_create refuses a device whose workgroups cannot have that much.
constexpr size_t kCellScanLds = sizeof(int) * (kCellScanWin + 2 * kCellScanThreads + 1);
static_assert(kCellScanLds <= 160 * 1024, "k_cell_scan's LDS window exceeds a gfx950 CU's 160 KB");
__device__ __forceinline__ int cell_lds(int i) { return i + i / kCellScanRun; }
__global__ __launch_bounds__(kCellScanThreads) void k_cell_scan(CellOrderArgs a)
{
    __shared__ int v[kCellScanWin + kCellScanThreads];
    __shared__ int s[kCellScanThreads];
    __shared__ int n_long;
    if (threadIdx.x == 0) n_long = 0;
    int carry = 0;
    for (int base = 0; base < a.nbins; base += kCellScanWin) {
        // (loads in batches of sixteen before the LDS stores: one memory round trip per batch, not per word)
        for (int i0 = threadIdx.x; i0 < kCellScanWin; i0 += 16 * kCellScanThreads) {
            int t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int i = i0 + u * kCellScanThreads; t[u] = (i < kCellScanWin && base + i < a.nbins) ? a.hist[base + i] : 0; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int i = i0 + u * kCellScanThreads; if (i < kCellScanWin) v[cell_lds(i)] = t[u]; }
        }
        __syncthreads();
        int sum = 0;
        const int row = threadIdx.x * (kCellScanRun + 1);
#pragma unroll 8
        for (int j = 0; j < kCellScanRun; ++j) sum += v[row + j];
        s[threadIdx.x] = sum;
        __syncthreads();
        for (int off = 1; off < kCellScanThreads; off <<= 1) {
            const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        int run = carry + s[threadIdx.x] - sum;          // exclusive offset of this thread's first cell
#pragma unroll 8
        for (int j = 0; j < kCellScanRun; ++j) {
            const int c = v[row + j];
            if (c > kCellLong) a.long_cells[1 + atomicAdd(&n_long, 1)] = base + threadIdx.x * kCellScanRun + j;
            v[row + j] = run; run += c;
        }
        const int total = s[kCellScanThreads - 1];
        __syncthreads();
        for (int i = threadIdx.x; i < kCellScanWin; i += kCellScanThreads)
            if (base + i < a.nbins) { const int o = v[cell_lds(i)]; a.hist[base + i] = o; a.start[base + i] = o; }
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { a.hist[a.nbins] = carry; a.start[a.nbins] = carry; a.long_cells[0] = n_long; }
}

__global__ __launch_bounds__(256) void k_cell_scatter(CellOrderArgs a)
{
    __shared__ int h[kCellLdsBins];
    const bool local = a.nbins <= kCellLdsBins;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = i < cell_n(a);
    const int k = on ? a.key[i] : 0;
    int pos;
    if (local) {
        // rank inside the workgroup from an LDS counter, one global cursor update per non-empty cell and workgroup
        for (int b = threadIdx.x; b < a.nbins; b += 256) h[b] = 0;
        __syncthreads();
        const int r = on ? atomicAdd(&h[k], 1) : 0;
        __syncthreads();
        for (int b = threadIdx.x; b < a.nbins; b += 256) { const int c = h[b]; if (c) h[b] = atomicAdd(a.hist + b, c); }
        __syncthreads();
        pos = h[k] + r;
    } else {
        pos = on ? atomicAdd(a.hist + k, 1) : 0;
    }
    if (!on) return;
    a.tmp[pos] = a.idx_in[i];
    a.tmp_key[pos] = k;
}

// The atomic scatter leaves a cell's entries in arbitrary order.  One thread per entry ranks it inside its cell's (short)
// segment — independent loads of neighbouring words — and writes it to its final place, so that the result is THE stable
// sort: the dense order, and with it every row-by-row comparison between two runs, is deterministic.  The same launch
// zeroes the cell counters for the next call (nothing reads them here).
// Cells with more than kCellLong entries (listed by k_cell_scan) are sorted as segments instead, by the first workgroups of
// the launch in turn: the entries of a cell are distinct candidate indices, and their stable order is their ascending
// order.  A segment is sorted in chunks of 4 096 by a bitonic network in LDS; several chunks are merged by ranking every
// entry in the other (sorted) chunks with a binary search — O(L log^2 L) work per cell instead of O(L^2).
__device__ __forceinline__ void bitonic_sort_lds(int *x, int m)        // m: power of two >= 2, x[0 .. m) in LDS, all threads of the workgroup
{
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            // one compare-exchange per thread and turn: pair t is (i, i | j) with bit j of i clear
            for (int t = threadIdx.x; t < (m >> 1); t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const int xi = x[i], xl = x[l];
                if (((i & k) == 0) ? (xi > xl) : (xi < xl)) { x[i] = xl; x[l] = xi; }
            }
        }
    __syncthreads();
}

constexpr int kCellRankGroups = 64;     // workgroups of k_cell_rank that take the long cells in turn

__global__ __launch_bounds__(256) void k_cell_rank(CellOrderArgs a)
{
    __shared__ int x[kCellSortChunk];
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p <= a.nbins) a.hist[p] = 0;
    if (p < cell_n(a)) {
        const int k = a.tmp_key[p];
        const int lo = a.start[k], hi = a.start[k + 1];
        if (hi - lo <= kCellLong) {
            const int32_t v = a.tmp[p];
            int r = lo;
            for (int q = lo; q < hi; ++q) r += (a.tmp[q] < v) ? 1 : 0;
            a.idx_out[r] = v;
        }
    }
    if (blockIdx.x >= kCellRankGroups) return;
    const int n_long = a.long_cells[0];
    const int stride = gridDim.x < (unsigned)kCellRankGroups ? (int)gridDim.x : kCellRankGroups;
    for (int c = blockIdx.x; c < n_long; c += stride) {
        const int k = a.long_cells[1 + c];
        const int lo = a.start[k], L = a.start[k + 1] - lo;
        const int chunks = (L + kCellSortChunk - 1) / kCellSortChunk;
        int32_t *sorted = chunks == 1 ? a.idx_out + lo : a.idx_in + lo;       // (idx_in is free once the scatter has read it)
        for (int ch = 0; ch < chunks; ++ch) {
            const int c0 = ch * kCellSortChunk, len = (L - c0 < kCellSortChunk) ? L - c0 : kCellSortChunk;
            int m = 1;
            while (m < len) m <<= 1;
            __syncthreads();
            for (int i = threadIdx.x; i < m; i += 256) x[i] = i < len ? a.tmp[lo + c0 + i] : 0x7fffffff;
            bitonic_sort_lds(x, m);
            for (int i = threadIdx.x; i < len; i += 256) sorted[c0 + i] = x[i];
        }
        if (chunks == 1) continue;
        __threadfence_block();
        __syncthreads();
        // merge: the final place of an entry is its position in its own sorted chunk + the entries smaller than it in every
        // other chunk — one chunk at a time staged in LDS and searched there by every thread for its entries (searching the
        // chunks in global memory cost 12 dependent round trips per entry and chunk); the running ranks live in key[lo ..),
        // which nothing reads any more
        int32_t *rank = a.key + lo;
        for (int i = threadIdx.x; i < L; i += 256) rank[i] = i % kCellSortChunk;
        for (int ch = 0; ch < chunks; ++ch) {
            const int c0 = ch * kCellSortChunk, len = (L - c0 < kCellSortChunk) ? L - c0 : kCellSortChunk;
            __syncthreads();
            for (int i = threadIdx.x; i < len; i += 256) x[i] = sorted[c0 + i];
            __syncthreads();
            for (int i = threadIdx.x; i < L; i += 256) {
                if (i / kCellSortChunk == ch) continue;
                const int32_t v = sorted[i];
                int b = 0, e = len;                         // lower bound of v in the staged chunk
                while (b < e) { const int mid = (b + e) >> 1; if (x[mid] < v) b = mid + 1; else e = mid; }
                rank[i] += b;
            }
        }
        for (int i = threadIdx.x; i < L; i += 256) a.idx_out[lo + rank[i]] = sorted[i];
    }
}

// Sums over a finished batch (throughput accounting / round control): one atomic per workgroup.
// n_out counters are written: 6 (the original six sums), 8 (+ short batch, storms counted) or 10 (+ [8] storms whose step
// record overflowed, status -3; [9] storms the batch had no room for, max(*n_dev - n, 0)).
__global__ __launch_bounds__(256) void k_stats(int64_t n, const int64_t *__restrict__ n_dev, const int32_t *__restrict__ n_valid,
                                               const int32_t *__restrict__ nfev, const int32_t *__restrict__ flags,
                                               const int32_t *__restrict__ status, unsigned long long *__restrict__ out, int n_out)
{
    const bool is_short = n_dev && *n_dev < n;         // the batch holds fewer storms than it was sized for
    const int64_t dropped = (n_dev && *n_dev > n) ? *n_dev - n : 0;
    if (is_short) n = *n_dev;
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_out >= 8) {
        if (is_short) atomicAdd(out + 6, 1ull);
        atomicAdd(out + 7, (unsigned long long)(n > 0 ? n : 0));
        if (n_out >= 10 && dropped > 0) atomicAdd(out + 9, (unsigned long long)dropped);
    }
    if (n_out >= 10 && status) {
        unsigned long long bad = 0;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            bad += status[i] == TCR_STATUS_STEP_OVERFLOW ? 1 : 0;
        const unsigned long long any = __ballot(bad != 0);
        if (any) {                                  // rare: one atomic per lane that saw one
            if (bad) atomicAdd(out + 8, bad);
        }
    }
    __shared__ unsigned long long s[4][6];
    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int nv = n_valid[i];
        a0 += nv > 1 ? nv - 1 : 0;
        a1 += nfev[i];
        a2 += nv;
        a3 += (flags[i] & TCR_FLAG_ACCEPTED) ? 1 : 0;
        a4 += (flags[i] & TCR_FLAG_IS_TC) ? 1 : 0;
        a5 += (flags[i] & TCR_FLAG_IS_TC) ? nv : 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off); a2 += __shfl_down(a2, off); a3 += __shfl_down(a3, off);
        a4 += __shfl_down(a4, off); a5 += __shfl_down(a5, off);
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        s[w][0] = a0; s[w][1] = a1; s[w][2] = a2; s[w][3] = a3; s[w][4] = a4; s[w][5] = a5;
    }
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(out + threadIdx.x, s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x]);
}

// Pack selected tracks into fixed-size survivor records for the all-gather:
// packed[row] = { lon[ns], lat[ns], v[ns], m[ns], vmax[ns], envw[ns*4] }  (9*ns doubles)
template <typename R>
struct PackArgsT {
    const R *lon, *lat, *v, *m, *vmax, *envw;
    const int32_t *idx;
    const int64_t *count;       // device scalar written by k_compact
    int64_t cap;
    int ns;
    int64_t row_stride;         // doubles between packed rows (>= 9 * ns)
    double *packed;
    // optional meta columns 9*ns .. 9*ns + 2 of a record (row_stride >= 9*ns + 3): global candidate index, month (1..12),
    // genesis-basin index — what run_tracks keeps per track next to the rows (compute.py:206-208).  cand_idx maps a dense
    // batch row to its position in the round's candidate block (NULL: the identity), cand0 / key give the block's first index.
    const int32_t *cand_idx, *slot, *basin_idx;
    int64_t cand0;
    const RoundKey *key;
};

template <typename R>
__global__ __launch_bounds__(256) void k_pack_tracks(PackArgsT<R> a)
{
    const int64_t row = blockIdx.x;
    const int64_t cnt = *a.count < a.cap ? *a.count : a.cap;
    if (row >= cnt) return;
    const size_t j = (size_t)a.idx[row];
    const int ns = a.ns;
    double *dst = a.packed + (size_t)row * (size_t)a.row_stride;
    const R *planes[5] = {a.lon, a.lat, a.v, a.m, a.vmax};
    for (int p = 0; p < 5; ++p)
        for (int i = threadIdx.x; i < ns; i += blockDim.x) dst[(size_t)p * ns + i] = (double)planes[p][j * ns + i];
    for (int i = threadIdx.x; i < ns * 4; i += blockDim.x) dst[(size_t)5 * ns + i] = (double)a.envw[j * ns * 4 + i];
    if (a.slot && threadIdx.x == 0) {
        const int64_t cand0 = a.key ? a.key->cand0 : a.cand0;
        dst[(size_t)9 * ns] = (double)(cand0 + (a.cand_idx ? (int64_t)a.cand_idx[j] : (int64_t)j));
        dst[(size_t)9 * ns + 1] = (double)(a.slot[j] + 1);
        dst[(size_t)9 * ns + 2] = (double)a.basin_idx[j];
    }
}

// n_seeds of a round (compute.py:165-167): candidates that count (seed_flags bit 0), per (genesis basin, month), optionally
// only those whose global index is <= *cutoff (the candidate that completed the quota).  out[7 * 12] int64 is SET.
struct SeedHistArgs {
    const int32_t *seed_flags, *basin_idx, *slot;
    int64_t n, cand0;
    const RoundKey *key;
    const double *cutoff;       // device scalar (a candidate index held as a double, as the survivor records hold it) or NULL
    unsigned long long *out;
    unsigned int *partial;      // [gridDim.x][7 * 12] per-workgroup counts, then one ticket word (zero between launches)
};

// 256-thread workgroups (a 1024-thread one — four waves per SIMD — cannot be resident next to integrator waves, DESIGN.md
// section 9, round 4): each counts its share in LDS and publishes it; the workgroup that draws the last ticket adds
// the published counts up and SETS out[] — one launch, nothing to zero beforehand, no atomics on out[].
constexpr int kSeedHistBlocks = 64;
__global__ __launch_bounds__(256) void k_seed_hist(SeedHistArgs a)
{
    constexpr int NB = TCR_N_BASINS * 12;
    __shared__ unsigned int h[NB];
    __shared__ bool last;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int64_t cand0 = a.key ? a.key->cand0 : a.cand0;
    const double cut = a.cutoff ? *a.cutoff : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!(a.seed_flags[i] & 1)) continue;
        if (a.cutoff && !((double)(cand0 + i) <= cut)) continue;
        const int b = a.basin_idx[i], m = a.slot[i];
        if (b >= 0 && b < TCR_N_BASINS && m >= 0 && m < 12) atomicAdd(&h[b * 12 + m], 1u);
    }
    __syncthreads();
    unsigned int *ticket = a.partial + (size_t)gridDim.x * NB;
    for (int i = threadIdx.x; i < NB; i += blockDim.x)
        __hip_atomic_store(a.partial + (size_t)blockIdx.x * NB + i, h[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) {
        unsigned long long sum = 0;
        for (unsigned b = 0; b < gridDim.x; ++b) sum += __hip_atomic_load(a.partial + (size_t)b * NB + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.out[i] = sum;
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace tcr
