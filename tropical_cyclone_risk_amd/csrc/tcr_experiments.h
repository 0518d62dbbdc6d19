// Experiment hooks of the launch path — NOT part of the product build (python tools/build_variant.py NAME -DTCR_EXPERIMENTS).
// They read the environment at every call, which the product library never does (tcr_tune, read once in tcr_ctx_create):
//   TCR_PASS_FILL=p       passes of k_integrate's chain after the first launch only p % as many lanes as they have parked storms
//   TCR_FS_WGS=n          workgroups of the forcing-table launch
//   TCR_DUMMY_LAUNCHES=k  k empty kernels per round, TCR_DUMMY_MODE=0..3 their shape; TCR_DUMMY_SPIN_US=t one wave spinning for
//                         t microseconds per round (what a dispatch costs under load, section 9, round 4, item 2)
//   TCR_GRAPH_DOT=file    dump a captured round as Graphviz
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

namespace tcr_exp {

inline int pass_fill_pct()
{
    if (const char *e = getenv("TCR_PASS_FILL")) { const long v = atol(e); if (v > 0 && v <= 100) return (int)v; }
    return 100;
}

inline int64_t fs_workgroups(int64_t want, int groups)
{
    if (const char *e = getenv("TCR_FS_WGS")) { const long v = atol(e); if (v > 0) return std::max<long>(1, v / groups); }
    return want;
}

__global__ void k_probe_empty() {}
__global__ void k_probe_store(unsigned long long *p) { if (threadIdx.x == 0) p[blockIdx.x * 16] = clock64(); }
__global__ void k_probe_spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

inline void dummy_launches(hipStream_t st, unsigned long long *scratch)
{
    if (const char *e = getenv("TCR_DUMMY_LAUNCHES")) {
        const char *m = getenv("TCR_DUMMY_MODE");
        const int mode = m ? atoi(m) : 0;       // 0: one empty wave; 1: 50 empty workgroups of 256; 2: one wave, one store; 3: 50 x 256, a store each
        for (long k = atol(e); k > 0; --k) {
            if (mode == 0) hipLaunchKernelGGL(k_probe_empty, dim3(1), dim3(64), 0, st);
            else if (mode == 1) hipLaunchKernelGGL(k_probe_empty, dim3(50), dim3(256), 0, st);
            else if (mode == 2) hipLaunchKernelGGL(k_probe_store, dim3(1), dim3(64), 0, st, scratch);
            else hipLaunchKernelGGL(k_probe_store, dim3(50), dim3(256), 0, st, scratch + 4096);
        }
    }
    if (const char *e = getenv("TCR_DUMMY_SPIN_US")) if (atol(e) > 0) hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, st, 100ll * atol(e));
}

inline void graph_dot(hipGraph_t graph)
{
    if (const char *dot = getenv("TCR_GRAPH_DOT")) (void)hipGraphDebugDotPrint(graph, dot, 0);
}

}  // namespace tcr_exp
