// Micro-benchmark 4: do f32 MFMAs (v_mfma_f32_32x32x2_f32) of one wave and VALU / LDS work of ANOTHER wave on the same SIMD
// overlap?  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run a bare MFMA loop with 8 accumulators; waves 4-7
// (their SIMD partners) run, per variant: nothing / a v_fma_f32 loop / a v_pk_fma_f32 loop / an LDS read-modify-write loop.
// Reported: cycles per MFMA of the MFMA waves, and how much partner work retired meanwhile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* cyc, long long* work, int iters) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 8192; i += 512) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    __shared__ volatile int stop;
    if (tid == 0) stop = 0;
    __syncthreads();
    if (wave < 4) {                                       // MFMA waves
        f32x16 acc[8];
        for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        float a = lds[lane] , b = lds[lane + 64];
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
        out[blockIdx.x * 512 + tid] = s;
        if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) stop = 1;
    } else {                                              // partner waves
        long long n = 0;
        float x0 = lds[lane], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
        v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
        const v2f c2 = {1.0001f, 0.9999f};
        if (MODE != 0) {
            while (!stop) {
                if (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
                        x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f);
                    }
                    n += 128;
                } else if (MODE == 2) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) { p0 = p0 * c2 + c2; p1 = p1 * c2 + c2; p2 = p2 * c2 + c2; p3 = p3 * c2 + c2; }
                    n += 64;
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        float4 v = *reinterpret_cast<float4*>(&lds[((lane + 64 * j + (int)n) & 2047) * 4]);
                        x0 += v.x + v.y + v.z + v.w;
                    }
                    n += 16;
                }
            }
        }
        out[blockIdx.x * 512 + tid] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y;
        if (lane == 0) work[blockIdx.x * 4 + wave - 4] = n;
    }
}

template <int MODE> void run(const char* name, float* out, long long* cyc, long long* work) {
    const int iters = 4000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(work, 0, blocks * 4 * 8);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, out, cyc, work, iters);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(blocks * 4), w(blocks * 4);
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(w.data(), work, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double avg = 0, wk = 0; for (auto v : h) avg += v; avg /= h.size(); for (auto v : w) wk += v; wk /= w.size();
    printf("%-28s MFMA wave: %.2f cycles/MFMA; partner retired %.0f instr (%.2f per MFMA, one per %.1f cycles)\n", name,
           avg / (iters * 8.0), wk, wk / (iters * 8.0), wk > 0 ? avg / wk : 0.0);
}

int main() {
    float* out; long long *cyc, *work;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 4 * 8); hipMalloc(&work, 256 * 4 * 8);
    run<0>("partner idle", out, cyc, work);
    run<1>("partner v_fma_f32 loop", out, cyc, work);
    run<2>("partner v_pk_fma_f32 loop", out, cyc, work);
    run<3>("partner ds_read_b128 loop", out, cyc, work);
    return 0;
}
