// Micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 in loops shaped like the conv k-step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ g, float* out, long long* cyc, int iters) {
    __shared__ float4 lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = g[i];
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float4 a0 = lds[lane], a1 = lds[lane + 64], b0 = g[lane], b1 = g[lane + 64];
    const float4* gp = g + lane;
    const long long t0 = __builtin_amdgcn_s_memtime();
#define M(i, av, bv) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
#define GROUP(A0, A1, B0, B1)                                                               \
        if (V == 0) {                                                                       \
            for (int r = 0; r < 4; ++r) { M(0, A0.x, B0.x) M(1, A0.x, B0.x) M(2, A0.x, B0.x) M(3, A0.x, B0.x) } \
        } else {                                                                            \
            M(0, A0.x, B0.x) M(1, A0.x, B1.x) M(2, A1.x, B0.x) M(3, A1.x, B1.x)             \
            M(0, A0.y, B0.y) M(1, A0.y, B1.y) M(2, A1.y, B0.y) M(3, A1.y, B1.y)             \
            M(0, A0.z, B0.z) M(1, A0.z, B1.z) M(2, A1.z, B0.z) M(3, A1.z, B1.z)             \
            M(0, A0.w, B0.w) M(1, A0.w, B1.w) M(2, A1.w, B0.w) M(3, A1.w, B1.w)             \
        }
    float4 c0 = a0, c1 = a1, d0 = b0, d1 = b1;       // second register set (ping-pong: no moves, no early waits)
    for (int it = 0; it < iters; it += 2) {
        if (V >= 2) { c0 = lds[(lane + it * 64) & 1023]; c1 = lds[(lane + it * 64 + 512) & 1023]; }
        if (V >= 3) { d0 = gp[((it * 128) & 65535)]; d1 = gp[((it * 128 + 64) & 65535)]; }
        if (V == 4) { asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7"); }
        __builtin_amdgcn_sched_barrier(0);
        GROUP(a0, a1, b0, b1)
        __builtin_amdgcn_sched_barrier(0);
        if (V >= 2) { a0 = lds[(lane + it * 64 + 64) & 1023]; a1 = lds[(lane + it * 64 + 576) & 1023]; }
        if (V >= 3) { b0 = gp[((it * 128 + 128) & 65535)]; b1 = gp[((it * 128 + 192) & 65535)]; }
        if (V == 4) { asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7"); }
        __builtin_amdgcn_sched_barrier(0);
        GROUP(c0, c1, d0, d1)
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V> void run(const char* name, const float4* g, float* out, long long* cyc, int blocks) {
    const int iters = 4000;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, g, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, g, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double flops = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("%-34s blocks %4d: %.1f cycles/MFMA, %.1f TF/s (%.3f ms)\n", name, blocks, avg / (iters * 16.0), flops / (ms * 1e-3) / 1e12, ms);
}

int main() {
    float4* g; float* out; long long* cyc;
    hipMalloc(&g, 65536 * 16 + 65536); hipMemset(g, 0, 65536 * 16 + 65536);
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 8);
    for (int blocks : {256, 512}) {
        run<0>("V0 same operands", g, out, cyc, blocks);
        run<1>("V1 distinct operands", g, out, cyc, blocks);
        run<2>("V2 + 2 ds_read_b128 / 16 MFMA", g, out, cyc, blocks);
        run<3>("V3 + 2 global_load_dwordx4 / 16", g, out, cyc, blocks);
        run<4>("V4 32 s_nop cycles / 16 MFMA", g, out, cyc, blocks);
    }
    return 0;
}
