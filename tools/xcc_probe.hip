// Which XCD does a workgroup run on?  s_getreg HW_REG_XCC_ID (id 20 on gfx942 / gfx950), bits [3:0].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(int *out) {
    const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 15);
}
int main() {
    const int n = 4096;
    int *d; hipMalloc(&d, n * 4);
    k<<<n, 64>>>(d);
    std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int match = 0, hist[16] = {0};
    for (int i = 0; i < n; ++i) { hist[h[i]]++; match += (h[i] == i % 8); }
    printf("first 24:"); for (int i = 0; i < 24; ++i) printf(" %d", h[i]); printf("\n");
    printf("xcc histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\nblockIdx %% 8 == xcc for %d of %d\n", match, n);
    return 0;
}
