// Probe (round 5): which XCD does workgroup b of a launch run on?  s_getreg_b32 HW_REG_XCC_ID against blockIdx.x.
//   hipcc --offload-arch=gfx950 -O3 exp/xcc_probe.hip -o exp/xcc_probe && exp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    for (int grid : {8, 64, 1024, 768}) {
        unsigned* d; (void)hipMalloc(&d, grid * 4);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(grid); (void)hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
        int cnt[16] = {0}, mism = 0;
        for (int b = 0; b < grid; ++b) { cnt[h[b] & 15]++; mism += ((h[b] & 15) != (unsigned)(b % 8)); }
        printf("grid %d: raw[0..9] =", grid); for (int b = 0; b < 10 && b < grid; ++b) printf(" %x", h[b]);
        printf("  | per-id counts:"); for (int i = 0; i < 16; ++i) if (cnt[i]) printf(" %d:%d", i, cnt[i]);
        printf("  | blocks with id != b %% 8: %d\n", mism);
        (void)hipFree(d);
    }
    return 0;
}
