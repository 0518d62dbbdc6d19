// Micro-benchmark behind DESIGN.md §9 (round 3): what does a wave64 pay for the integrator's access pattern —
// per lane, the four corners of a random cell of a [lat][lon][16]-double field slot, 7 x 16-byte loads per corner —
// as a function of occupancy, and how many such lane-requests per second does a CU's address / L1 path sustain?
// It decides whether an occupancy-first integrator WITHOUT the per-lane corner cache (44 gathers per evaluation
// instead of ~17) can beat the one-wave-per-SIMD kernel, or is bound by the CU's texture-address path.
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_bw tools/gather_bw.hip && ./gather_bw
//
// Modes: divergent (every lane its own cell, the real pattern), same-cell (all lanes of a wave one cell: one line per
// instruction), coherent (lanes of a wave within a 4x4-cell neighbourhood: what a locality-sorted batch would give).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef double double2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double2v ldg16(const double *p) { return *(const __attribute__((address_space(1))) double2v *)(p); }

template <int MODE, int NLOADS>
__global__ __launch_bounds__(64) void k_gather(const double *__restrict__ slots, int n_slots, int nlat, int nlon, int iters,
                                                double *__restrict__ out, int lds_pad_words)
{
    extern __shared__ double pad[];          // dynamic LDS only to cap the number of resident waves
    if (lds_pad_words < 0) pad[threadIdx.x] = 0;
    const int lane = threadIdx.x;
    unsigned s = (blockIdx.x * 64u + lane) * 2654435761u + 12345u;
    unsigned sw = blockIdx.x * 2246822519u + 777u;         // wave-uniform stream
    double acc0 = 0, acc1 = 0;
    const size_t slot_doubles = (size_t)nlat * nlon * 16;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        sw = sw * 1664525u + 1013904223u;
        unsigned r = (MODE == 1) ? sw : s;
        int month = (int)((r >> 8) % (unsigned)n_slots);
        int j = (int)((r >> 12) % (unsigned)(nlat - 1));
        int i = (int)((r >> 3) % (unsigned)(nlon - 1));
        if (MODE == 2) {                                   // coherent: wave-uniform base cell + per-lane offset in a 4x4 block
            month = (int)((sw >> 8) % (unsigned)n_slots);
            j = (int)((sw >> 12) % (unsigned)(nlat - 5)) + (int)((s >> 5) & 3);
            i = (int)((sw >> 3) % (unsigned)(nlon - 5)) + (int)((s >> 9) & 3);
        }
        const double *p00 = slots + (size_t)month * slot_doubles + ((size_t)j * nlon + i) * 16;
        const double *p01 = p00 + (size_t)nlon * 16;
        double2v v[NLOADS];
#pragma unroll
        for (int k = 0; k < NLOADS; ++k) {
            const int corner = k & 3, q = k >> 2;          // 4 corners x up to 7 sixteen-byte pieces
            const double *p = (corner & 1 ? p01 : p00) + (corner & 2 ? 16 : 0) + 2 * q;
            v[k] = ldg16(p);
        }
#pragma unroll
        for (int k = 0; k < NLOADS; ++k) { acc0 += v[k][0]; acc1 += v[k][1]; }
    }
    out[blockIdx.x * 64 + lane] = acc0 + acc1;
}

template <int MODE, int NLOADS>
double run(const double *d_slots, int n_slots, int nlat, int nlon, int waves, int iters, double *d_out, int waves_per_simd, int cus)
{
    // LDS per workgroup so that at most `waves_per_simd * 4` single-wave workgroups fit a CU (160 KB)
    size_t lds = waves_per_simd >= 8 ? 0 : (size_t)(160 * 1024 / (waves_per_simd * 4)) - 512;
    if (lds > 65536) lds = 65536;               // (1 wave per SIMD cannot be forced by LDS alone: 64 KB max => 2 per CU... use grid size)
    CK(hipFuncSetAttribute((const void *)k_gather<MODE, NLOADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_gather<MODE, NLOADS><<<waves, 64, lds>>>(d_slots, n_slots, nlat, nlon, 2, d_out, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_gather<MODE, NLOADS><<<waves, 64, lds>>>(d_slots, n_slots, nlat, nlon, iters, d_out, 0);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, mhz);
    const int nlat = 181, nlon = 360;
    for (int n_slots : {1, 12}) {
        const size_t n = (size_t)n_slots * nlat * nlon * 16;
        std::vector<double> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (double)(i % 1000) * 1e-3;
        double *d_slots, *d_out;
        CK(hipMalloc(&d_slots, n * 8));
        CK(hipMemcpy(d_slots, h.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_out, (size_t)cus * 4 * 16 * 64 * 8));
        printf("\n== %d month slot(s): %.1f MB of [lat][lon][16] doubles\n", n_slots, n * 8 / 1e6);
        printf("%-10s %5s %6s %9s %12s %14s %12s\n", "mode", "loads", "w/SIMD", "ms", "Greq/s", "clk/instr/CU", "TB/s (16 B)");
        const int iters = 2000;
        for (int wps : {1, 2, 4, 8}) {
            const int waves = cus * 4 * wps;             // exactly one residency's worth: no tail
#define ROW(MODE, NAME, NL) { double ms = run<MODE, NL>(d_slots, n_slots, nlat, nlon, waves, iters, d_out, wps, cus); \
            const double req = (double)waves * 64 * iters * NL; \
            const double instr_per_cu = (double)waves * iters * NL / cus; \
            printf("%-10s %5d %6d %9.3f %12.1f %14.1f %12.2f\n", NAME, NL, wps, ms, req / ms / 1e6, ms * 1e-3 * mhz * 1e6 / instr_per_cu, req * 16 / ms / 1e9); }
            ROW(0, "divergent", 28)
            ROW(2, "coherent", 28)
            ROW(1, "same-cell", 28)
            ROW(0, "divergent", 8)
        }
        CK(hipFree(d_slots)); CK(hipFree(d_out));
    }
    return 0;
}
