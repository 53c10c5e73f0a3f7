// What do v_permlane32_swap / v_permlane16_swap return?  x = 1000 + lane, y = 2000 + lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x, x = 1000 + l, y = 2000 + l;
    const u32x2 a = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    const u32x2 b = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    o[l] = a[0]; o[64 + l] = a[1]; o[128 + l] = b[0]; o[192 + l] = b[1];
}
int main() {
    unsigned h[256], *d;
    (void)hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane32_swap [0]", "permlane32_swap [1]", "permlane16_swap [0]", "permlane16_swap [1]"};
    for (int r = 0; r < 4; ++r) {
        printf("%s: lanes 0,16,32,48 ->", names[r]);
        for (int l = 0; l < 64; l += 16) printf(" %u", h[64 * r + l]);
        printf("\n");
    }
    return 0;
}
