// Can a small-register kernel run NEXT to a wave that owns 424 of its SIMD's 512 registers (the integrator's shape:
// 256 VGPRs + 164 AGPRs, one wave per SIMD on every SIMD of the chip)?  DESIGN.md §9 (round 3).
//   hipcc --offload-arch=gfx950 -O3 -o build_variants/coresidency tools/coresidency.hip && ./build_variants/coresidency
// Kernel A: one wave per SIMD on all SIMDs, spins on dependent fp64 FMAs for a fixed number of shader clocks; its register
// footprint is forced by inline asm that touches v255 and a163.  Kernel B: many small workgroups, 2 waves each, <= 80 (or 96)
// VGPRs, a fixed amount of fp64 FMAs.  B is launched on a second stream while A runs; we report when B's first / last
// workgroup started relative to A's start and end (device wall clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64, 1) void k_big(unsigned long long clocks, unsigned long long *t, double *out)
{
    asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a163, v255" ::: "v255", "a163");
    const unsigned long long w0 = wall_clock64();
    const unsigned long long c0 = clock64();
    double x = threadIdx.x * 1e-3, y = 1.0000001;
    while (clock64() - c0 < clocks) { for (int i = 0; i < 64; ++i) x = x * y + 1e-9; }
    if (threadIdx.x == 0) { atomicMin(t + 0, w0); atomicMax(t + 1, wall_clock64()); }
    if (x == 123.456) out[0] = x;
}

template <int NREG>
__global__ __launch_bounds__(128) void k_small(int iters, unsigned long long *t, double *out)
{
    if (NREG > 80) asm volatile("v_mov_b32 v95, 0" ::: "v95"); else asm volatile("v_mov_b32 v79, 0" ::: "v79");
    const unsigned long long w0 = wall_clock64();
    double x = threadIdx.x * 1e-3, y = 1.0000001;
    for (int i = 0; i < iters; ++i) x = x * y + 1e-9;
    if (threadIdx.x == 0) { atomicMin(t + 2, w0); atomicMax(t + 3, w0); atomicMax(t + 4, wall_clock64()); }
    if (x == 123.456) out[0] = x;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4;
    unsigned long long *t; double *out;
    CK(hipMalloc(&t, 64)); CK(hipMalloc(&out, 64));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int variant = 0; variant < 3; ++variant) {
        unsigned long long init[5] = {~0ull, 0, ~0ull, 0, 0};
        CK(hipMemcpy(t, init, sizeof init, hipMemcpyHostToDevice));
        const unsigned long long clocks = 4000000ull;           // ~1.7 ms at 2.4 GHz
        if (variant < 2) k_big<<<simds, 64, 0, s1>>>(clocks, t, out);
        // give A time to occupy every SIMD, then launch B
        CK(hipStreamSynchronize(s2));
        for (volatile int spin = 0; spin < 2000000; ++spin) {}
        if (variant == 1) k_small<96><<<8192, 128, 0, s2>>>(20000, t, out); else k_small<80><<<8192, 128, 0, s2>>>(20000, t, out);
        CK(hipDeviceSynchronize());
        unsigned long long h[5]; CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
        const double us = 1e6 / 100e6;                           // wall_clock64: 100 MHz
        const double a0 = variant < 2 ? h[0] * us : h[2] * us;
        printf("%s: A ran %.0f us; B (%s VGPRs) first wave started at +%.0f us, last wave started at +%.0f us, B ended at +%.0f us\n",
               variant == 0 ? "A + B<=80" : variant == 1 ? "A + B<=96" : "B alone  ",
               variant < 2 ? (h[1] - h[0]) * us : 0.0, variant == 1 ? "96" : "80", h[2] * us - a0, h[3] * us - a0, h[4] * us - a0);
    }
    return 0;
}
