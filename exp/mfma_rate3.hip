// Micro-benchmark 3: the Winograd MFMA phase shape: NACC independent accumulators, each (k-step, xi) pair = one
// ds_read_b128 (A) + one global_load_dwordx4 (B) + 4 MFMAs; 1 vs 2 waves per SIMD (256 vs 128 accumulator registers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float4* __restrict__ g, float* out, long long* cyc, int iters) {
    __shared__ float4 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = g[i];
    __syncthreads();
    f32x16 acc[NACC];
    for (int m = 0; m < NACC; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    constexpr int PF = 4;
    float4 bq[PF], a0 = lds[lane];
    for (int p = 0; p < PF; ++p) bq[p] = g[lane + 64 * p];
    const float4* gp = g + lane;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < NACC; ++p) {
            float4 a1 = lds[(lane + 64 * (p + 1) + it * 64) & 2047];
            __builtin_amdgcn_sched_barrier(0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bq[p % PF].x, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bq[p % PF].y, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bq[p % PF].z, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bq[p % PF].w, acc[p], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bq[p % PF] = gp[((it * NACC + p + PF) * 64) & 65535];
            a0 = a1;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int m = 0; m < NACC; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int WPS> void run(const float4* g, float* out, long long* cyc, int blocks) {
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, WPS>), dim3(blocks), dim3(256), 0, 0, g, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) continue;
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double nm = (double)iters * NACC * 4;
        printf("NACC %2d waves/SIMD %d (blocks %4d): %.2f cycles/MFMA/wave -> %.2f per SIMD, %.1f TF/s\n", NACC, WPS, blocks,
               avg / nm, avg / nm / WPS, (double)blocks * 4 * nm * 4096.0 / (ms * 1e-3) / 1e12);
    }
}

int main() {
    float4* g; float* out; long long* cyc;
    hipMalloc(&g, 65536 * 16 + 65536); hipMemset(g, 0, 65536 * 16 + 65536);
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 8);
    run<16, 1>(g, out, cyc, 256);
    run<8, 1>(g, out, cyc, 256);
    run<8, 2>(g, out, cyc, 512);
    run<4, 2>(g, out, cyc, 512);
    run<4, 3>(g, out, cyc, 768);
    return 0;
}
