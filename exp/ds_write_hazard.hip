// Micro-benchmark (round 4): does a DS write see the result of a v_cvt_pk_bf16_f32 issued right in front of it, and may its
// data / address registers be overwritten right behind it?  Three round-4 kernel variants whose separable upsample stored
// 16-byte bf16 pieces (four v_cvt_pk_bf16_f32 + ds_write2_b64 in a tight unrolled loop, the registers reused at once) were
// correct against every oracle tolerance and NOT bit-repeatable (profiles/r04_ablation.md); this isolates the instruction pattern.
//   hipcc --offload-arch=gfx950 -O3 exp/ds_write_hazard.hip -o exp/ds_write_hazard && exp/ds_write_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: cvt x4, ds_write2_b64 directly behind, data and address registers overwritten directly behind that
// MODE 1: the same with s_nop 7 between the converts and the write and between the write and the overwrites
// MODE 2: ds_write_b128 instead of ds_write2_b64, no nops
// MODE 3: ds_write_b64 x2, no nops
template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, unsigned* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[8 * 256 * 4];
    const int tid = threadIdx.x;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const float* base = src + ((size_t)(blockIdx.x * iters + it) * 8) * 256 * 8 + tid * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float* p = base + (size_t)r * 256 * 8;
            const float f0 = p[0], f1 = p[1], f2 = p[2], f3 = p[3], f4 = p[4], f5 = p[5], f6 = p[6], f7 = p[7];
            const unsigned addr = (unsigned)((r * 256 + tid) * 16);
#define CVT4 "v_cvt_pk_bf16_f32 v20, %1, %2\n v_cvt_pk_bf16_f32 v21, %3, %4\n v_cvt_pk_bf16_f32 v22, %5, %6\n v_cvt_pk_bf16_f32 v23, %7, %8\n"
#define KILL "v_mov_b32 v20, -1\n v_mov_b32 v21, -1\n v_mov_b32 v22, -1\n v_mov_b32 v23, -1\n"
#define OPS :: "v"(addr), "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7) : "v20", "v21", "v22", "v23", "memory"
            if constexpr (MODE == 0) asm volatile(CVT4 "ds_write2_b64 %0, v[20:21], v[22:23] offset1:1\n" KILL OPS);
            if constexpr (MODE == 1) asm volatile(CVT4 "s_nop 7\n ds_write2_b64 %0, v[20:21], v[22:23] offset1:1\n s_nop 7\n" KILL OPS);
            if constexpr (MODE == 2) asm volatile(CVT4 "ds_write_b128 %0, v[20:23]\n" KILL OPS);
            if constexpr (MODE == 3) asm volatile(CVT4 "ds_write_b64 %0, v[20:21]\n ds_write_b64 %0, v[22:23] offset:8\n" KILL OPS);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float* p = base + (size_t)r * 256 * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned want = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){p[2 * q], p[2 * q + 1]}, bf16x2));
                bad += lds[(r * 256 + tid) * 4 + q] != want;
            }
        }
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = bad;
}

template <int MODE>
static void run(const char* what, const float* d_src, unsigned* d_out, int blocks, int iters) {
    hipMemset(d_out, 0, sizeof(unsigned) * blocks * 256);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_src, d_out, iters);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 256);
    hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long bad = 0;
    for (unsigned v : h) bad += v;
    printf("%-70s mismatching dwords: %llu of %llu\n", what, bad, (unsigned long long)blocks * 256 * iters * 32);
}

int main() {
    const int blocks = 1024, iters = 16;
    const size_t n = (size_t)blocks * iters * 8 * 256 * 8;
    std::vector<float> h(n);
    uint32_t s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    float* d_src; unsigned* d_out;
    hipMalloc(&d_src, n * 4); hipMalloc(&d_out, sizeof(unsigned) * blocks * 256);
    hipMemcpy(d_src, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("cvt x4 | ds_write2_b64 | registers overwritten at once", d_src, d_out, blocks, iters);
        run<1>("cvt x4 | s_nop 7 | ds_write2_b64 | s_nop 7 | overwritten", d_src, d_out, blocks, iters);
        run<2>("cvt x4 | ds_write_b128 | registers overwritten at once", d_src, d_out, blocks, iters);
        run<3>("cvt x4 | ds_write_b64 x2 | registers overwritten at once", d_src, d_out, blocks, iters);
    }
    return 0;
}
