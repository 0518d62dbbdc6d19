// Multi-GPU exchange behind the C ABI (include/tcrisk_hip.h, "multi-GPU" section): RCCL collectives over xGMI.
//
// The reference collects the 9-tuples of its `run_tracks` workers by pickling them back through dask
// (util/compute.py:223-242).  Here one process owns one GPU, every rank integrates a block of the candidate index space
// (or whole years), and what travels is ONE fixed-shape ncclAllGather of survivor records per round / per run — xGMI is
// point to point (7 links per GPU), so one large ring all-gather is the shape that suits it, not one collective per storm
// or per variable.  Ragged contributions are sent padded to a common row count; tcr_concat_rows_dev then packs the
// received blocks in rank order (= candidate order, tropical_cyclone_risk_amd/distributed.py) on the device.
//
// RCCL is loaded at run time (dlopen): the library has no link-time dependency on it, single-GPU users never touch it,
// and inside a PyTorch process the copy PyTorch has already loaded is the one that is found.
#include <dlfcn.h>

namespace {

typedef struct { char internal[TCR_COMM_ID_BYTES]; } rccl_unique_id;      // ncclUniqueId (rccl.h: 128 bytes)
static_assert(TCR_COMM_ID_BYTES == 128, "ncclUniqueId is 128 bytes");
// ncclDataType_t / ncclRedOp_t values used (rccl.h): ncclInt64 = 4, ncclFloat64 = 8, ncclUint8 = 1; ncclSum = 0
constexpr int kNcclUint8 = 1, kNcclInt64 = 4, kNcclFloat64 = 8, kNcclSum = 0;

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(rccl_unique_id *) = nullptr;
    int (*CommInitRank)(void **, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};

Rccl *rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            R.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (R.h) break;
        }
        if (!R.h) { R.err = std::string("RCCL not found (librccl.so.1): ") + (dlerror() ? dlerror() : ""); return; }
        auto sym = [&](const char *s) { void *p = dlsym(R.h, s); if (!p && R.err.empty()) R.err = std::string("RCCL symbol missing: ") + s; return p; };
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
        R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &R;
}

// out[r0 + i] = gathered[r][i] for i < min(counts[r], cap), r0 = sum of the clipped counts of the ranks in front: one workgroup
// per (rank, row) pair of the padded block, rows copied 16 bytes per lane
__global__ __launch_bounds__(256) void k_concat_rows(const double *__restrict__ gathered, const int64_t *__restrict__ counts, int world,
                                                     int64_t cap, int64_t row_stride, double *__restrict__ out, int64_t out_cap,
                                                     int64_t *__restrict__ n_out)
{
    const int64_t item = blockIdx.x;
    const int r = (int)(item / cap);
    const int64_t i = item - (int64_t)r * cap;
    int64_t base = 0, total = 0;
    for (int q = 0; q < world; ++q) {
        const int64_t c = counts[q] < cap ? counts[q] : cap;
        if (q < r) base += c;
        total += c;
    }
    if (item == 0 && threadIdx.x == 0 && n_out) *n_out = total < out_cap ? total : out_cap;
    const int64_t mine = counts[r] < cap ? counts[r] : cap;
    if (i >= mine || base + i >= out_cap) return;
    const double *src = gathered + item * row_stride;
    double *dst = out + (base + i) * row_stride;
    if ((row_stride & 1) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {      // wave-uniform
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        for (int64_t k = threadIdx.x; k < row_stride / 2; k += blockDim.x) d2[k] = s2[k];
    } else {
        for (int64_t k = threadIdx.x; k < row_stride; k += blockDim.x) dst[k] = src[k];
    }
}

}  // namespace

struct tcr_comm {
    tcr_ctx *ctx = nullptr;
    void *nccl = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int tcr_comm_unique_id(uint8_t id[TCR_COMM_ID_BYTES])
{
    Rccl *R = rccl();
    if (!R->err.empty()) { g_create_error = R->err; return -1; }
    if (!id) { g_create_error = "tcr_comm_unique_id: NULL argument"; return -1; }
    rccl_unique_id u;
    const int rc = R->GetUniqueId(&u);
    if (rc != 0) { g_create_error = std::string("ncclGetUniqueId: ") + R->GetErrorString(rc); return -1; }
    memcpy(id, u.internal, TCR_COMM_ID_BYTES);
    return 0;
}

int tcr_comm_create(tcr_ctx *ctx, const uint8_t id[TCR_COMM_ID_BYTES], int32_t rank, int32_t world, tcr_comm **out)
{
    if (!ctx) return -1;
    if (!id || !out) return fail(ctx, "tcr_comm_create: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, "tcr_comm_create: need 0 <= rank < world");
    Rccl *R = rccl();
    if (!R->err.empty()) return fail(ctx, "tcr_comm_create: ", R->err.c_str());
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rccl_unique_id u;
    memcpy(u.internal, id, TCR_COMM_ID_BYTES);
    void *c = nullptr;
    const int rc = R->CommInitRank(&c, world, u, rank);
    if (rc != 0 || !c) return fail(ctx, "ncclCommInitRank: ", R->GetErrorString(rc));
    tcr_comm *cm = new tcr_comm;
    cm->ctx = ctx; cm->nccl = c; cm->rank = rank; cm->world = world;
    *out = cm;
    return 0;
}

int tcr_comm_destroy(tcr_comm *c)
{
    if (!c) return 0;
    int rc = 0;
    if (c->nccl) rc = rccl()->CommDestroy(c->nccl);
    delete c;
    return rc == 0 ? 0 : -1;
}

int tcr_comm_rank(const tcr_comm *c) { return c ? c->rank : -1; }
int tcr_comm_world(const tcr_comm *c) { return c ? c->world : -1; }

int tcr_allgather_dev(tcr_comm *c, const void *send_dev, void *recv_dev, int64_t bytes, void *stream_)
{
    if (!c) return -1;
    tcr_ctx *ctx = c->ctx;
    if (bytes < 0 || (bytes > 0 && (!send_dev || !recv_dev))) return fail(ctx, "tcr_allgather_dev: bad argument");
    if (bytes == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    // fp64 elements where the size allows it (what the records are), bytes otherwise: the transport is the same
    const bool f64 = (bytes % 8) == 0 && (reinterpret_cast<uintptr_t>(send_dev) % 8) == 0 && (reinterpret_cast<uintptr_t>(recv_dev) % 8) == 0;
    const int rc = rccl()->AllGather(send_dev, recv_dev, f64 ? (size_t)(bytes / 8) : (size_t)bytes, f64 ? kNcclFloat64 : kNcclUint8, c->nccl, st);
    if (rc != 0) return fail(ctx, "ncclAllGather: ", rccl()->GetErrorString(rc));
    return 0;
}

int tcr_allgather_rows_dev(tcr_comm *c, const double *rows_dev, int64_t n_rows, int64_t row_stride, double *gathered_dev, void *stream)
{
    if (!c) return -1;
    if (n_rows < 0 || row_stride <= 0) return fail(c->ctx, "tcr_allgather_rows_dev: bad shape");
    return tcr_allgather_dev(c, rows_dev, gathered_dev, n_rows * row_stride * (int64_t)sizeof(double), stream);
}

int tcr_allgather_counts_dev(tcr_comm *c, const int64_t *count_dev, int64_t *counts_dev, void *stream)
{
    return tcr_allgather_dev(c, count_dev, counts_dev, (int64_t)sizeof(int64_t), stream);
}

int tcr_allreduce_sum_i64_dev(tcr_comm *c, int64_t *buf_dev, int64_t n, void *stream_)
{
    if (!c) return -1;
    tcr_ctx *ctx = c->ctx;
    if (n < 0 || (n > 0 && !buf_dev)) return fail(ctx, "tcr_allreduce_sum_i64_dev: bad argument");
    if (n == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    const int rc = rccl()->AllReduce(buf_dev, buf_dev, (size_t)n, kNcclInt64, kNcclSum, c->nccl, st);
    if (rc != 0) return fail(ctx, "ncclAllReduce: ", rccl()->GetErrorString(rc));
    return 0;
}

int tcr_concat_rows_dev(tcr_ctx *ctx, int32_t n_blocks, const double *gathered_dev, const int64_t *counts_dev, int64_t cap, int64_t row_stride,
                        double *out_dev, int64_t out_cap, int64_t *n_out_dev, void *stream_)
{
    if (!ctx) return -1;
    if (n_blocks < 1 || cap < 0 || row_stride <= 0 || out_cap < 0 || !counts_dev) return fail(ctx, "tcr_concat_rows_dev: bad argument");
    if (cap * (int64_t)n_blocks > 0x7fffffffLL) return fail(ctx, "tcr_concat_rows_dev: cap x n_blocks exceeds the grid limit");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (cap == 0) {
        if (n_out_dev) HIPCHK(ctx, hipMemsetAsync(n_out_dev, 0, sizeof(int64_t), stream_ ? (hipStream_t)stream_ : ctx->stream));
        return 0;
    }
    if (!gathered_dev || !out_dev) return fail(ctx, "tcr_concat_rows_dev: NULL buffer");
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    hipLaunchKernelGGL(k_concat_rows, dim3((unsigned)(cap * n_blocks)), dim3(256), 0, st, gathered_dev, counts_dev, (int)n_blocks, cap, row_stride,
                       out_dev, out_cap, n_out_dev);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
