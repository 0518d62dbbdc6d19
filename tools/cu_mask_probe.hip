// Does a CU-masked stream (hipExtStreamCreateWithCUMask) confine a kernel's workgroups on MI355X, and how do the mask's bits map
// to (XCC, SE, CU)?  Every workgroup records HW_REG_XCC_ID and HW_REG_HW_ID; the host counts the distinct (xcc, se, cu) seen.
//   hipcc --offload-arch=gfx950 -O3 tools/cu_mask_probe.hip -o build_variants/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
__global__ void k(unsigned *out, long long spin)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;        // XCC_ID [3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);             // HW_ID: cu [11:8], sh [12], se [15:13] (gfx9)
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 24) | (hw & 0xffffff);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(4);                       // keep the CU busy so that workgroups spread out
}
static void run(const char *tag, hipStream_t st)
{
    const int n = 8192;
    unsigned *d; hipMalloc(&d, n * 4);
    k<<<n, 256, 0, st>>>(d, 2000);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; int per_xcc[16] = {0};
    std::set<unsigned> seen_x[16];
    for (int i = 0; i < n; ++i) {
        const unsigned xcc = h[i] >> 24, cu = (h[i] >> 8) & 15, sh = (h[i] >> 12) & 1, se = (h[i] >> 13) & 7;
        const unsigned id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        cus.insert(id); seen_x[xcc].insert(id);
    }
    printf("%-28s distinct (xcc, se, sh, cu): %zu;  per xcc:", tag, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %zu", seen_x[x].size());
    printf("\n");
    hipFree(d);
}
int main()
{
    hipStream_t s0; hipStreamCreate(&s0); run("no mask", s0);
    for (int bits : {64, 128, 192}) {
        uint32_t mask[8] = {0};
        for (int b = 0; b < bits; ++b) mask[b / 32] |= 1u << (b % 32);
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%d bits) failed: %s\n", bits, hipGetErrorString(e)); continue; }
        char tag[64]; snprintf(tag, sizeof tag, "first %d bits", bits); run(tag, s);
    }
    {   // the complement of the first 192: bits 192..255
        uint32_t mask[8] = {0};
        for (int b = 192; b < 256; ++b) mask[b / 32] |= 1u << (b % 32);
        hipStream_t s; if (hipExtStreamCreateWithCUMask(&s, 8, mask) == hipSuccess) run("bits 192..255", s);
    }
    {   // every fourth bit
        uint32_t mask[8] = {0};
        for (int b = 0; b < 256; b += 4) mask[b / 32] |= 1u << (b % 32);
        hipStream_t s; if (hipExtStreamCreateWithCUMask(&s, 8, mask) == hipSuccess) run("every 4th bit", s);
    }
    return 0;
}
