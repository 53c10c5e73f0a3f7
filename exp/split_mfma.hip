// Micro-benchmark 5 (round 3, the "gated experiment"): the F(4x4) frequency-domain GEMMs on the bf16 / f16 matrix pipe with
// SPLIT f32 operands (v = v0 + v1 (+ v2), products of the pieces, f32 accumulate).
//
// Part A - numerics on the real instruction: M[32 x 32] = V[32 x K] U[K x 32], K = 256, per problem with
//   f32     v_mfma_f32_32x32x2_f32 chain (today's arithmetic)
//   bf16x3  six cross terms of order <= 2 on v_mfma_f32_32x32x16_bf16 (small terms first inside a 16-channel step)
//   f16x2   three cross terms on v_mfma_f32_32x32x16_f16, operands pre-scaled by powers of two (V by sv, U by su)
// against an f64 host reference; also f16x2 WITHOUT scaling on small-magnitude data (do f16 subnormal operands survive?).
//
// Part B - rate of the inner loop a split-operand F(4x4) kernel would have: 4 waves per CU (one per SIMD, 288 accumulator
// registers: 32 tiles x 32 channels x 18 frequencies each), per 16-channel chunk and frequency NP A fragments from LDS
// (ds_read_b128, conflict-free image), NP B fragments from a packed global stream (one 256 -> 256 layer's worth, L2 / MALL
// resident, every CU streams its channel block's share), 6 or 3 MFMAs; plus a stand-in for the chunk's input transform + split
// (VALU ops, ds_read_b64 of a patch, ds_write_b64 of V) either absent, serialised between two barriers, or interleaved with
// the MFMAs.  Reported: cycles per chunk per wave (today's phased f32 kernel: ~13,700 per chunk for the same 32 tiles x 64
// channels x 16 input channels, profiles/r02_wino4_stamps_final.md).
//
//   hipcc -O3 --offload-arch=gfx950 exp/split_mfma.hip -o exp/split_mfma && exp/split_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- splits (device): pairs of f32 -> packed pieces --------------------------------------------------------------------------
__device__ __forceinline__ void split_bf16x3(f32x2 v, unsigned& p0, unsigned& p1, unsigned& p2) {
    const bf16x2 h0 = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(h0, f32x2);
    const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(h1, f32x2);
    const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
    p0 = __builtin_bit_cast(unsigned, h0); p1 = __builtin_bit_cast(unsigned, h1); p2 = __builtin_bit_cast(unsigned, h2);
}
__device__ __forceinline__ void split_f16x2(f32x2 v, unsigned& p0, unsigned& p1) {
    const f16x2 h0 = __builtin_convertvector(v, f16x2);
    const f32x2 r1 = v - __builtin_convertvector(h0, f32x2);
    const f16x2 h1 = __builtin_convertvector(r1, f16x2);
    p0 = __builtin_bit_cast(unsigned, h0); p1 = __builtin_bit_cast(unsigned, h1);
}

// ---- Part A ---------------------------------------------------------------------------------------------------------------------
// V [P][32][K] f32, U [P][K][32] f32, out [3][P][32][32].  One wave per problem.
template <int K>
__global__ __launch_bounds__(64) void numerics_kernel(const float* V, const float* U, float* out, int P, float sv, float su) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const int r = lane & 31, h = lane >> 5;
    const float* v = V + (size_t)p * 32 * K;
    const float* u = U + (size_t)p * K * 32;
    f32x16 a32, ab, ah;
    for (int i = 0; i < 16; ++i) { a32[i] = 0.f; ab[i] = 0.f; ah[i] = 0.f; }
    for (int k = 0; k < K; k += 2) a32 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[r * K + k + h], u[(k + h) * 32 + r], a32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        u32x4 A0, A1, A2, B0, B1, B2, HA0, HA1, HB0, HB1;
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + 8 * h + 2 * j;
            unsigned x0, x1, x2;
            split_bf16x3(f32x2{v[r * K + k], v[r * K + k + 1]}, x0, x1, x2); A0[j] = x0; A1[j] = x1; A2[j] = x2;
            split_bf16x3(f32x2{u[k * 32 + r], u[(k + 1) * 32 + r]}, x0, x1, x2); B0[j] = x0; B1[j] = x1; B2[j] = x2;
            split_f16x2(f32x2{v[r * K + k] * sv, v[r * K + k + 1] * sv}, x0, x1); HA0[j] = x0; HA1[j] = x1;
            split_f16x2(f32x2{u[k * 32 + r] * su, u[(k + 1) * 32 + r] * su}, x0, x1); HB0[j] = x0; HB1[j] = x1;
        }
#define BM(a, b) ab = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), ab, 0, 0, 0)
        BM(A1, B1); BM(A0, B2); BM(A2, B0); BM(A0, B1); BM(A1, B0); BM(A0, B0);
#define HM(a, b) ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), ah, 0, 0, 0)
        HM(HA0, HB1); HM(HA1, HB0); HM(HA0, HB0);
    }
    const float inv = 1.0f / (sv * su);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        out[((size_t)(0 * P + p) * 32 + row) * 32 + r] = a32[i];
        out[((size_t)(1 * P + p) * 32 + row) * 32 + r] = ab[i];
        out[((size_t)(2 * P + p) * 32 + row) * 32 + r] = ah[i] * inv;
    }
}

static void part_a(float vscale, float sv, float su, const char* title) {
    const int P = 64, K = 256;
    std::vector<float> V((size_t)P * 32 * K), U((size_t)P * K * 32);
    srand(7);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    auto gauss = [&]() { float s = 0; for (int i = 0; i < 6; ++i) s += rnd(); return s * 0.7f; };
    for (auto& x : V) x = gauss() * std::exp(2.0f * rnd()) * vscale;         // wide-ish magnitude spread
    for (auto& x : U) x = gauss() * 0.08f * std::exp(1.5f * rnd());
    float *dV, *dU, *dO;
    CHECK(hipMalloc(&dV, V.size() * 4)); CHECK(hipMalloc(&dU, U.size() * 4)); CHECK(hipMalloc(&dO, (size_t)3 * P * 1024 * 4));
    CHECK(hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((numerics_kernel<K>), dim3(P), dim3(64), 0, 0, dV, dU, dO, P, sv, su);
    CHECK(hipDeviceSynchronize());
    std::vector<float> O((size_t)3 * P * 1024);
    CHECK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
    double rms = 0, err[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int p = 0; p < P; ++p)
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)V[((size_t)p * 32 + i) * K + k] * (double)U[((size_t)p * K + k) * 32 + j];
                rms += s * s;
                for (int m = 0; m < 3; ++m) {
                    const double d = (double)O[((size_t)(m * P + p) * 32 + i) * 32 + j] - s;
                    err[m] += d * d; if (std::fabs(d) > mx[m]) mx[m] = std::fabs(d);
                }
            }
    rms = std::sqrt(rms / (P * 1024.0));
    printf("%s (errors relative to rms(M) = %.3g)\n", title, rms);
    const char* names[3] = {"f32 32x32x2 chain", "bf16 x3, 6 terms", "f16 x2, 3 terms"};
    for (int m = 0; m < 3; ++m) printf("  %-20s rms %.3e  max %.3e\n", names[m], std::sqrt(err[m] / (P * 1024.0)) / rms, mx[m] / rms);
    hipFree(dV); hipFree(dU); hipFree(dO);
}

// ---- Part B ---------------------------------------------------------------------------------------------------------------------
// NP = pieces per operand (3: bf16, 6 MFMAs per frequency; 2: f16, 3 MFMAs).  XF: 0 no transform stand-in, 1 serialised between
// barriers, 2 interleaved into the MFMA loop.  NVALU / NRD / NWR: the stand-in's VALU ops, ds_read_b64 and ds_write_b64 per
// thread and chunk.
template <int NP, int XF, int NVALU, int NRD, int NWR, int RING>
__global__ __launch_bounds__(256, 1) void loop_kernel(const u32x4* __restrict__ wstream, float* out, long long* cyc, int nchunks, int chunks_per_layer) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VBYTES = 36 * NP * 1024;                 // [freq 36][piece NP][32 tiles x 32 B]
    unsigned char* const Vb = smem;
    float* const patch = reinterpret_cast<float*>(smem + VBYTES);          // 40 KB stand-in for the halo patch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = wid >> 1, cq = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < (VBYTES + 40960) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (unsigned)(i * 2654435761u >> 20);
    __syncthreads();
    f32x16 acc[18];
#pragma unroll
    for (int f = 0; f < 18; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
    const int cby = ((int)blockIdx.x >> 3) & 3;
    // stream: [cby 4][cq 2][fh 2][chunk][freq 18][piece NP][64 lanes] of 16 B
    const size_t per_wave = (size_t)chunks_per_layer * 18 * NP * 64;
    const u32x4* const bbase = wstream + ((size_t)(cby * 2 + cq) * 2 + fh) * per_wave + lane;
    const int aoff = fh * 18 * NP * 1024 + r * 32 + 16 * (h ^ ((r >> 3) & 1));
    float x0 = (float)tid, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 * 0.5f, x5 = x1 * 0.5f, x6 = x2 * 0.5f, x7 = x3 * 0.5f;
    const int prd = (tid * 8) % 9000, pwr = (tid * 8) % (VBYTES / 4 - 64 * NWR * 2);
    auto valu = [&](int n) {
#pragma unroll
        for (int i = 0; i < n / 8; ++i) {
            x0 = fmaf(x0, 1.0001f, x4); x1 = fmaf(x1, 0.9999f, x5); x2 = fmaf(x2, 1.0001f, x6); x3 = fmaf(x3, 0.9999f, x7);
            x4 = fmaf(x4, 1.0001f, x1); x5 = fmaf(x5, 0.9999f, x2); x6 = fmaf(x6, 1.0001f, x3); x7 = fmaf(x7, 0.9999f, x0);
        }
    };
    auto lds_rd = [&](int k, int n) {
#pragma unroll
        for (int i = 0; i < n; ++i) { const f32x2 t = *reinterpret_cast<const f32x2*>(&patch[prd + 2 * ((k + i) * 37 % 500)]); x0 += t.x; x4 += t.y; }
    };
    auto lds_wr = [&](int k, int n) {
#pragma unroll
        for (int i = 0; i < n; ++i) *reinterpret_cast<f32x2*>(reinterpret_cast<float*>(Vb) + pwr + 128 * (k + i)) = f32x2{x1, x5};
    };
    u32x4 bq[RING][NP];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nchunks; ++c) {
        const u32x4* bp = bbase + (size_t)(c % chunks_per_layer) * 18 * NP * 64;
        if constexpr (XF == 1) {
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 18; ++g) {                 // same work as XF == 2, in 18 register-light groups
                lds_rd(g * (NRD / 18), NRD / 18); valu(NVALU / 18); lds_wr(g * (NWR / 18), NWR / 18);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < RING; ++q)
#pragma unroll
            for (int p = 0; p < NP; ++p) bq[q][p] = __builtin_nontemporal_load(bp + (q * NP + p) * 64);
        u32x4 ar[3][NP];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int p = 0; p < NP; ++p) ar[q][p] = *reinterpret_cast<const u32x4*>(Vb + aoff + (q * NP + p) * 1024);
#pragma unroll
        for (int f = 0; f < 18; ++f) {
            if (f + 2 < 18) {
#pragma unroll
                for (int p = 0; p < NP; ++p) ar[(f + 2) % 3][p] = *reinterpret_cast<const u32x4*>(Vb + aoff + ((f + 2) * NP + p) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int s = f % RING;
            if constexpr (NP == 3) {
#define BMM(a, b) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ar[f % 3][a]), __builtin_bit_cast(bf16x8, bq[s][b]), acc[f], 0, 0, 0)
                BMM(1, 1); BMM(0, 2); BMM(2, 0); BMM(0, 1); BMM(1, 0); BMM(0, 0);
            } else {
#define HMM(a, b) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ar[f % 3][a]), __builtin_bit_cast(f16x8, bq[s][b]), acc[f], 0, 0, 0)
                HMM(0, 1); HMM(1, 0); HMM(0, 0);
            }
            if constexpr (XF == 2) {                        // this frequency's share of the next chunk's transform
                lds_rd(f * (NRD / 18), NRD / 18); valu(NVALU / 18); lds_wr(f * (NWR / 18), NWR / 18);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (f + RING < 18) {
#pragma unroll
                for (int p = 0; p < NP; ++p) bq[s][p] = __builtin_nontemporal_load(bp + ((f + RING) * NP + p) * 64);
            }
        }
        if constexpr (XF == 2) __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float sres = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
#pragma unroll
    for (int f = 0; f < 18; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) sres += acc[f][i];
    out[blockIdx.x * 256 + tid] = sres;
    if (lane == 0) cyc[blockIdx.x * 4 + wid] = t1 - t0;
}

template <int NP, int XF, int NVALU, int NRD, int NWR, int RING>
static void run_b(const char* name, const u32x4* ws, float* out, long long* cyc) {
    const int blocks = 256, cpl = 16, nchunks = 16 * 8;
    const int lds = 36 * NP * 1024 + 40960;
    auto kern = loop_kernel<NP, XF, NVALU, NRD, NWR, RING>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, ws, out, cyc, nchunks, cpl);
        hipEventRecord(e1, 0);
        CHECK(hipDeviceSynchronize());
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> hc(blocks * 4);
    CHECK(hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : hc) avg += v; avg /= hc.size();
    const double per_chunk = avg / nchunks;
    const double bbytes = 18.0 * NP * 1024 * 4;           // per CU and chunk
    printf("%-58s %8.0f cycles/chunk/wave  (%.1f cyc per MFMA slot; B stream %.1f B/clk/CU; kernel %.3f ms -> %.2f GHz)\n", name, per_chunk,
           per_chunk / (18.0 * (NP == 3 ? 6 : 3)), bbytes / per_chunk, ms, avg / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const bool only_a = argc > 1 && !strcmp(argv[1], "a"), only_b = argc > 1 && !strcmp(argv[1], "b");
    if (!only_b) {
        part_a(1.0f, 256.0f, 4096.0f, "Part A, V ~ O(1), f16 operands scaled by 2^8 (V) / 2^12 (U)");
        part_a(1.0f, 1.0f, 1.0f, "Part A, V ~ O(1), f16 operands unscaled");
        part_a(1e-3f, 1.0f, 1.0f, "Part A, V ~ 1e-3, f16 operands unscaled (v1 pieces are f16 subnormals)");
        part_a(1e-3f, 262144.0f, 4096.0f, "Part A, V ~ 1e-3, f16 operands scaled by 2^18 (V) / 2^12 (U)");
    }
    if (only_a) return 0;
    u32x4* ws; float* out; long long* cyc;
    const size_t wbytes = (size_t)4 * 2 * 2 * 16 * 18 * 3 * 64 * 16;
    CHECK(hipMalloc(&ws, wbytes)); CHECK(hipMalloc(&out, 256 * 256 * 4)); CHECK(hipMalloc(&cyc, 256 * 4 * 8));
    {
        std::vector<unsigned short> hw(wbytes / 2);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 22) % 512);   // f16 / bf16 of O(1)
        CHECK(hipMemcpy(ws, hw.data(), wbytes, hipMemcpyHostToDevice));
    }
    printf("Part B: 4 waves/CU, 288 accumulators/wave, per chunk = 32 tiles x 64 channels x 16 input channels\n");
    run_b<3, 0, 0, 0, 0, 6>("bf16x3  MFMA phase only, B ring 6", ws, out, cyc);
    run_b<3, 0, 0, 0, 0, 3>("bf16x3  MFMA phase only, B ring 3", ws, out, cyc);
    run_b<3, 1, 504, 72, 54, 6>("bf16x3  + transform stand-in serialised (504 VALU, 72 rd, 54 wr)", ws, out, cyc);
    run_b<3, 2, 504, 72, 54, 6>("bf16x3  + transform stand-in interleaved", ws, out, cyc);
    run_b<2, 0, 0, 0, 0, 6>("f16x2   MFMA phase only, B ring 6", ws, out, cyc);
    run_b<2, 0, 0, 0, 0, 9>("f16x2   MFMA phase only, B ring 9", ws, out, cyc);
    run_b<2, 1, 324, 72, 36, 6>("f16x2   + transform stand-in serialised (324 VALU, 72 rd, 36 wr)", ws, out, cyc);
    run_b<2, 2, 324, 72, 36, 6>("f16x2   + transform stand-in interleaved", ws, out, cyc);
    run_b<2, 2, 324, 72, 36, 9>("f16x2   + transform stand-in interleaved, B ring 9", ws, out, cyc);
    return 0;
}
