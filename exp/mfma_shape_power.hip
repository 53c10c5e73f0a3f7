// Micro-benchmark (round 5): the bf16 consumers' k-loop on v_mfma_f32_32x32x16_bf16 against the same FLOPs on v_mfma_f32_16x16x32_bf16, chip-wide
// (256 workgroups x 4 MFMA waves, one per SIMD: the power-limited regime conv3x3_bf16ws_kernel runs in at 2.0-2.1 GHz).  Per "tap": the A fragments of a
// 128-pixel x 32-channel block from LDS (ds_read_b128), hi and lo weight fragments of 64 output channels from L2, 2 terms x 128 x 64 x 32 MACs.
//   hipcc --offload-arch=gfx950 -O3 exp/mfma_shape_power.hip -o exp/mfma_shape_power && exp/mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ g, float* out, long long* cyc, int taps) {
    __shared__ float4 lds[4096];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = g[i];
    __syncthreads();
    const float4* gp = g + lane;
    float s = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (SHAPE == 32) {
        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        for (int t = 0; t < taps; ++t) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float4 a[4], b[2][2];
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = lds[(lane + 64 * m + 256 * ks + 512 * wid + (t & 3) * 64) & 4095];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int w = 0; w < 2; ++w) b[n][w] = gp[((t * 8 + ks * 4 + n * 2 + w) * 64) & 32767];
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, b[n][w]), acc[m][n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][7];
    } else {
        f32x4 acc[8][4];                                  // 8 x 4 blocks of 16 x 16: the same 128 pixels x 64 channels
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
        for (int t = 0; t < taps; ++t) {
            float4 a[8], b[4][2];
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] = lds[(lane + 64 * m + 512 * wid + (t & 3) * 64) & 4095];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int w = 0; w < 2; ++w) b[n][w] = gp[((t * 8 + n * 2 + w) * 64) & 32767];
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, b[n][w]), acc[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][3];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE> void run(const float4* g, float* out, long long* cyc, int blocks, int taps) {
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(256), 0, 0, g, out, cyc, taps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double flops = (double)blocks * 4 * taps * 2.0 * 2 * 128 * 64 * 32;
        printf("%dx%d: %4d workgroups  %.3f ms  %.0f TF/s  %.0f s_memtime ticks per tap (1024 MFMA cycles)\n", SHAPE, SHAPE, blocks, ms, flops / (ms * 1e-3) / 1e12, avg / taps);
    }
}

int main() {
    float4* g; float* out; long long* cyc;
    (void)hipMalloc(&g, 65536 * 16); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
    std::vector<float> h(65536 * 4);
    unsigned s = 1u; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)(int)(s >> 9) * (1.0f / 4194304.0f) - 1.0f; }
    (void)hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int taps = 20000;
    for (int blocks : {32, 256}) { run<32>(g, out, cyc, blocks, taps); run<16>(g, out, cyc, blocks, taps); }
    return 0;
}
