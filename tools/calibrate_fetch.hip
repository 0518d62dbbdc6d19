// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS project's access patterns (VERDICT r2 #7;
// MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern before trusting an absolute").
//
//   hipcc --offload-arch=gfx950 -O3 -o build_variants/calibrate_fetch tools/calibrate_fetch.hip
//   rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d out/f -o p -- ./build_variants/calibrate_fetch
//   rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d out/w -o p -- ./build_variants/calibrate_fetch
//   python tools/calibrate_fetch.py out > profiles/r03_fetch_calibration.json
//
// Every kernel touches a buffer of 1 GiB (4x the 256 MiB Infinity Cache) exactly once, so the true HBM byte count is
// known, and prints it; the buffer is flushed out of the caches between kernels by a 1 GiB fill of another buffer.
//   cal_stream_read    16 B per lane, fully coalesced                       (the guide's calibration case)
//   cal_gather_line    the integrator's / k_emit's pattern: a lane reads the 7 x 16 B of ONE 128-byte grid point of a
//                      [point][16] double array, points assigned to lanes by a bijective scramble (every point once,
//                      neighbouring lanes megabytes apart): true bytes = points x 128 B line fills (112 B requested)
//   cal_gather_16      a lane reads ONE 16-byte piece of a distinct 128-byte line (thermo / static / forcing-table style)
//   cal_stream_write   16 B per lane, fully coalesced store
//   cal_scatter_line   the step-record store pattern: a lane writes 7 x 16 B of one scrambled 128-byte line
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef double double2v __attribute__((ext_vector_type(2)));

constexpr size_t kBytes = 1ull << 30;
constexpr size_t kLines = kBytes / 128;          // 8 Mi lines / grid points
constexpr unsigned kScramble = 2654435761u;      // odd: i -> i * k mod 2^23 is a bijection on [0, 8 Mi)

__global__ __launch_bounds__(256) void cal_stream_read(const double2v *__restrict__ src, double *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // one 16-byte element per thread
    const double2v v = src[i];
    if (v[0] == 123.456) out[0] = v[1];
}

__global__ __launch_bounds__(256) void cal_gather_line(const double *__restrict__ src, double *__restrict__ out)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;                // one grid point per thread
    const size_t p = (size_t)((i * kScramble) & (unsigned)(kLines - 1));
    const double2v *q = reinterpret_cast<const double2v *>(src + p * 16);
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) { const double2v v = q[k]; acc += v[0] + v[1]; }
    if (acc == 123.456) out[0] = acc;
}

__global__ __launch_bounds__(256) void cal_gather_16(const double *__restrict__ src, double *__restrict__ out)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const size_t p = (size_t)((i * kScramble) & (unsigned)(kLines - 1));
    const double2v v = *reinterpret_cast<const double2v *>(src + p * 16 + 2 * (i & 7u));
    if (v[0] == 123.456) out[0] = v[1];
}

__global__ __launch_bounds__(256) void cal_stream_write(double2v *__restrict__ dst, double x)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double2v v; v[0] = x; v[1] = x + 1;
    dst[i] = v;
}

__global__ __launch_bounds__(256) void cal_scatter_line(double *__restrict__ dst, double x)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const size_t p = (size_t)((i * kScramble) & (unsigned)(kLines - 1));
    double2v *q = reinterpret_cast<double2v *>(dst + p * 16);
#pragma unroll
    for (int k = 0; k < 7; ++k) { double2v v; v[0] = x + k; v[1] = x; q[k] = v; }
}

__global__ __launch_bounds__(256) void cal_flush(double2v *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double2v v; v[0] = 1.0; v[1] = 2.0;
    dst[i] = v;
}

int main()
{
    double *a, *b, *out;
    CK(hipMalloc(&a, kBytes)); CK(hipMalloc(&b, kBytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, kBytes)); CK(hipMemset(b, 0, kBytes));
    const unsigned g16 = (unsigned)(kBytes / 16 / 256), gl = (unsigned)(kLines / 256);
    auto flush = [&]() { cal_flush<<<g16, 256>>>(reinterpret_cast<double2v *>(b)); CK(hipDeviceSynchronize()); };
    for (int rep = 0; rep < 3; ++rep) {
        flush(); cal_stream_read<<<g16, 256>>>(reinterpret_cast<const double2v *>(a), out); CK(hipDeviceSynchronize());
        flush(); cal_gather_line<<<gl, 256>>>(a, out); CK(hipDeviceSynchronize());
        flush(); cal_gather_16<<<gl, 256>>>(a, out); CK(hipDeviceSynchronize());
        flush(); cal_stream_write<<<g16, 256>>>(reinterpret_cast<double2v *>(a), 1.0 + rep); CK(hipDeviceSynchronize());
        flush(); cal_scatter_line<<<gl, 256>>>(a, 2.0 + rep); CK(hipDeviceSynchronize());
    }
    printf("true bytes: cal_stream_read %zu  cal_gather_line %zu (lines x 128 B; %zu requested)  cal_gather_16 %zu requested in %zu lines  "
           "cal_stream_write %zu  cal_scatter_line %zu written in %zu lines\n",
           kBytes, kLines * 128, kLines * 112, kLines * 16, kLines, kBytes, kLines * 112, kLines);
    return 0;
}
