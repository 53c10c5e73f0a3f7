// Which element of A / B does lane l supply to v_mfma_f32_16x16x4_f32, and where does D land?  (checks the operand-swap of the F(4x4) kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {   // A [16][4], B [4][16] row-major, D [16][16]
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
int main() {
    float hA[64], hB[64], hD[256], *dA, *dB, *dD;
    for (int i = 0; i < 64; ++i) { hA[i] = (float)(rand() % 17) - 8; hB[i] = (float)(rand() % 13) - 6; }
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j];
        if (std::fabs(s - hD[i * 16 + j]) > 1e-4) ++bad;
    }
    printf("A[l&15][l>>4], B[l>>4][l&15], D[4*(l>>4)+r][l&15]: %d of 256 wrong\n", bad);
    return 0;
}
