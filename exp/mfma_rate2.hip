// Micro-benchmark 2: register tile MT x NT of 32x32x2 f32 MFMAs per k-step, A from LDS (MT ds_read_b128), B from L2
// (NT global_load_dwordx4), ping-pong operand registers, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MT, int NT>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ g, float* out, long long* cyc, int iters) {
    __shared__ float4 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = g[i];
    __syncthreads();
    f32x16 acc[MT][NT];
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float4 a[2][MT], b[2][NT];
    for (int m = 0; m < MT; ++m) { a[0][m] = lds[lane + 64 * m]; a[1][m] = a[0][m]; }
    for (int n = 0; n < NT; ++n) { b[0][n] = g[lane + 64 * n]; b[1][n] = b[0][n]; }
    const float4* gp = g + lane;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
            for (int m = 0; m < MT; ++m) a[ph ^ 1][m] = lds[(lane + 64 * m + (it & 15) * 64) & 2047];
#pragma unroll
            for (int n = 0; n < NT; ++n) b[ph ^ 1][n] = gp[((it * 64 * NT + 64 * n) & 65535)];
            __builtin_amdgcn_sched_barrier(0);
#define STEP(comp)                                                                                              \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int n = 0; n < NT; ++n)               \
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ph][m].comp, b[ph][n].comp, acc[m][n], 0, 0, 0);
            STEP(x) STEP(y) STEP(z) STEP(w)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MT, int NT> void run(const float4* g, float* out, long long* cyc, int blocks) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MT, NT>), dim3(blocks), dim3(256), 0, 0, g, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) continue;
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double nm = (double)iters * MT * NT * 4;
        printf("MT %d NT %d blocks %4d: %.2f cycles/MFMA/wave, %.1f TF/s\n", MT, NT, blocks, avg / nm,
               (double)blocks * 4 * nm * 4096.0 / (ms * 1e-3) / 1e12);
    }
}

int main() {
    float4* g; float* out; long long* cyc;
    hipMalloc(&g, 65536 * 16 + 65536); hipMemset(g, 0, 65536 * 16 + 65536);
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 8);
    for (int blocks : {256, 512}) {
        run<2, 1>(g, out, cyc, blocks);
        run<2, 2>(g, out, cyc, blocks);
        run<4, 1>(g, out, cyc, blocks);
        run<2, 4>(g, out, cyc, blocks);
        run<4, 2>(g, out, cyc, blocks);
        if (blocks == 256) run<4, 4>(g, out, cyc, blocks);
    }
    return 0;
}
