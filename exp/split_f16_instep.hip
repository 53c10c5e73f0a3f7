// Micro-benchmark 6: would the EXISTING wave layout of the F(4x4) kernel (8 waves, a wave owns 16 tiles x 16 channels x 36
// frequencies = 144 accumulators, two waves per SIMD, all waves in step) pay on the f16 matrix pipe with two f16 pieces per
// operand, K-packed as [v0 | v0] x [u0 | u1] + [v1 | v1] x [u0 | u1] on v_mfma_f32_16x16x32_f16?  The B fragment keeps its 16 bytes
// per lane (one per frequency and 16-channel chunk, as in f32), the A operand becomes two ds_read_b128, four f32 MFMAs (128 cycles)
// become two f16 ones (32).  The two waves that own the same channels (tile halves 0 / 1) load the SAME B fragments: does the
// CU's L1 merge them (the weight stream was the bound of exp/split_mfma.hip)?  Stand-in for the input transform + split per chunk
// and thread: NVALU VALU ops, 30 ds_read_b64 of a patch, 36 ds_write_b32 of the two V planes.
//   hipcc -O3 --offload-arch=gfx950 exp/split_f16_instep.hip -o exp/split_f16_instep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: MFMA phase only; 1: + transform stand-in between barriers (all waves in step); F32: the same loop on v_mfma_f32_16x16x4_f32
template <int MODE, bool F32, int NVALU, int PF, bool SHAREB>
__global__ __launch_bounds__(512, 2) void k(const u32x4* __restrict__ wstream, float* out, long long* cyc, int nchunks, int cpl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PLANE = 32 * 32;                                       // bytes per frequency plane of one piece: 32 tiles x 16 halves
    unsigned char* const V = smem;                                       // [piece 2][36][PLANE] (f32: 36 planes of 32 tiles x 16 floats = 2 KB)
    float* const patch = reinterpret_cast<float*>(smem + 2 * 36 * PLANE + (F32 ? 36 * PLANE : 0));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int th = wid >> 2, cq = wid & 3, tg = lane >> 4, cl = lane & 15;
    for (int i = tid; i < (2 * 36 * PLANE + (F32 ? 36 * PLANE : 0) + 40960) / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (unsigned)(i * 2654435761u >> 21);
    __syncthreads();
    f32x4 acc[36];
#pragma unroll
    for (int f = 0; f < 36; ++f) acc[f] = f32x4{0, 0, 0, 0};
    const int cby = ((int)blockIdx.x >> 3) & 3;
    // stream per 16-channel block: [cb][chunk][freq 36][64 lanes] x 16 B; SHAREB: both tile halves read the same fragments (as in the kernel)
    const int cb = cby * 4 + cq;
    const u32x4* const bbase = wstream + (size_t)(SHAREB ? cb : (cb * 2 + th)) * cpl * 36 * 64 + lane;
    const int aoff = F32 ? ((16 * th + cl) * 64 + 16 * tg) : ((16 * th + cl) * 32 + 16 * (tg & 1));
    float x0 = (float)tid, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 * .5f, x5 = x1 * .5f, x6 = x2 * .5f, x7 = x3 * .5f;
    const int prd = (tid * 2) % 9000, pwr = (tid * 4) % (PLANE * 4);
    u32x4 bq[PF];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nchunks; ++c) {
        const u32x4* bp = bbase + (size_t)(c % cpl) * 36 * 64;
        if constexpr (MODE == 1) {
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 6; ++g) {
#pragma unroll
                for (int i = 0; i < 5; ++i) { const f32x2 t = *reinterpret_cast<const f32x2*>(&patch[prd + 2 * ((g * 5 + i) * 37 % 500)]); x0 += t.x; x4 += t.y; }
#pragma unroll
                for (int i = 0; i < NVALU / 48; ++i) {
                    x0 = fmaf(x0, 1.0001f, x4); x1 = fmaf(x1, 0.9999f, x5); x2 = fmaf(x2, 1.0001f, x6); x3 = fmaf(x3, 0.9999f, x7);
                    x4 = fmaf(x4, 1.0001f, x1); x5 = fmaf(x5, 0.9999f, x2); x6 = fmaf(x6, 1.0001f, x3); x7 = fmaf(x7, 0.9999f, x0);
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) *reinterpret_cast<float*>(V + ((g * 6 + i) * PLANE + pwr) % (2 * 36 * PLANE)) = x1 + (float)i;
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) bq[p] = bp[p * 64];
        u32x4 a0[3], a1[3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a0[q] = *reinterpret_cast<const u32x4*>(V + q * (F32 ? 2 * PLANE : PLANE) + aoff);
            if (!F32) a1[q] = *reinterpret_cast<const u32x4*>(V + 36 * PLANE + q * PLANE + aoff);
        }
#pragma unroll
        for (int f = 0; f < 36; ++f) {
            if (f + 2 < 36) {
                a0[(f + 2) % 3] = *reinterpret_cast<const u32x4*>(V + (f + 2) * (F32 ? 2 * PLANE : PLANE) + aoff);
                if (!F32) a1[(f + 2) % 3] = *reinterpret_cast<const u32x4*>(V + 36 * PLANE + (f + 2) * PLANE + aoff);
            }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 b = bq[f % PF];
            if constexpr (F32) {
                const f32x4 af = __builtin_bit_cast(f32x4, a0[f % 3]), bf = __builtin_bit_cast(f32x4, b);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc[f], 0, 0, 0);
            } else {
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0[f % 3]), __builtin_bit_cast(f16x8, b), acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1[f % 3]), __builtin_bit_cast(f16x8, b), acc[f], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (f + PF < 36) bq[f % PF] = bp[(f + PF) * 64];
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
#pragma unroll
    for (int f = 0; f < 36; ++f) s += acc[f][0] + acc[f][1] + acc[f][2] + acc[f][3];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int MODE, bool F32, int NVALU, int PF, bool SHAREB>
static void run(const char* name, const u32x4* ws, float* out, long long* cyc) {
    const int blocks = 256, cpl = 16, nchunks = 16 * 8;
    const int lds = 2 * 36 * 1024 + (F32 ? 36 * 1024 : 0) + 40960;
    auto kern = k<MODE, F32, NVALU, PF, SHAREB>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, ws, out, cyc, nchunks, cpl);
        hipEventRecord(e1, 0);
        CHECK(hipDeviceSynchronize());
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> hc(blocks * 8);
    CHECK(hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : hc) avg += v; avg /= hc.size();
    printf("%-78s %7.0f cycles/chunk  kernel %.3f ms -> %.2f GHz, %.2f us/chunk\n", name, avg / nchunks, ms, avg / (ms * 1e-3) / 1e9, ms * 1e3 / nchunks);
}

int main() {
    u32x4* ws; float* out; long long* cyc;
    const size_t wbytes = (size_t)32 * 16 * 36 * 64 * 16;
    CHECK(hipMalloc(&ws, wbytes)); CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    { std::vector<unsigned short> hw(wbytes / 2); for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 22) % 512); CHECK(hipMemcpy(ws, hw.data(), wbytes, hipMemcpyHostToDevice)); }
    printf("8 waves in step, 144 accumulators per wave, one chunk = 32 tiles x 64 channels x 16 input channels (today's phased f32 kernel: ~12,400 cycles = 5.4 us)\n");
    run<0, true, 0, 12, true>("f32 16x16x4, MFMA phase only (4 per frequency), B shared by the tile halves", ws, out, cyc);
    run<1, true, 336, 12, true>("f32 + transform stand-in (336 VALU, 30 rd, 36 wr) in step", ws, out, cyc);
    run<0, false, 0, 12, true>("f16x2 16x16x32, MFMA phase only (2 per frequency), B shared by the tile halves", ws, out, cyc);
    run<0, false, 0, 12, false>("f16x2, MFMA phase only, B NOT shared (every wave its own stream)", ws, out, cyc);
    run<1, false, 480, 12, true>("f16x2 + transform/split stand-in (480 VALU, 30 rd, 36 wr) in step, B shared", ws, out, cyc);
    run<1, false, 480, 12, false>("f16x2 + stand-in in step, B not shared", ws, out, cyc);
    run<1, false, 480, 9, true>("f16x2 + stand-in in step, B shared, ring 9", ws, out, cyc);
    return 0;
}
