// Probe v2 (round 5; v1 = pk_coexec_probe.hip, the compiler-generated loop): the exact operand form, outside libpnpadmm.  The fault profiles/r05_race.md pins down in conv3x3_bf16ws_kernel's separable
// producers: a v_pk_fma_f32 whose ADDEND pair was written by a v_pk_mul_f32 two or three VALU slots earlier loses the LOW half of
// the addend in lanes 48-63 - sometimes - while the sibling wave of the SIMD streams bf16 MFMAs, LDS reads and buffer loads.
//
// The workgroup is the kernel's: 8 waves, two per SIMD.  Waves 0-3 ("consumers") run a k-loop of ds_read_b128 A fragments, L2-hot
// 16-byte weight loads and v_mfma_f32_32x32x16_bf16; waves 4-7 ("producers") run the separable interpolation loop of the kernel
// verbatim (same source expression, so hipcc emits the same packed sequences) on synthetic LDS contents and check every row
// against a second evaluation from opaque copies of the same registers.  Ingredient switches (bit mask `mode`):
//   1 consumers issue MFMAs   2 consumers read LDS   4 consumers load weights from global memory
//   hipcc --offload-arch=gfx950 -O3 exp/pk_coexec_probe.hip -o exp/pk_coexec_probe && exp/pk_coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TW = 32, TH = 16, PH = TH + 2, PW = TW + 2, CK = 32, CKP = 20, PPP = 8;
constexpr int LH = TH / 2 + 3, LW = TW / 2 + 3, CKL = CK + 4;
constexpr int PATCH = PH * PW * CKP;
constexpr int LDS_FLOATS = 2 * PATCH + LH * LW * CKL + 4 * (PH + PW);

__device__ __forceinline__ float hashf(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

template <int mode, int PROD, int MF>
__global__ __launch_bounds__(512) void probe(const float4* __restrict__ wts, unsigned* __restrict__ res, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) float patch[];
    float* const lowres = patch + 2 * PATCH;
    float* const tb = lowres + LH * LW * CKL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < LH * LW * CKL; i += 512) lowres[i] = hashf(i * 2654435761u + blockIdx.x);
    for (int i = tid; i < 2 * PATCH; i += 512) patch[i] = hashf(i * 40503u + 17u);
    if (tid < PH + PW) {
        float4 e;
        if (tid < PH) {                       // row entry: {weight of line s, weight of line s + 1}
            const float l = 0.5f + 0.5f * hashf(tid * 77u + 5u);
            e = make_float4(1.f - l * 0.5f, l * 0.5f, 0.f, 0.f);
        } else {                              // column entry: {offset of source column 0, 1; weight 0, 1}
            const int px = tid - PH, i0 = px / 2, i1 = i0 + (i0 < LW - 1 ? 1 : 0);
            const float l = 0.5f + 0.5f * hashf(px * 31u + 9u);
            e = make_float4(__int_as_float(i0 * CKL), __int_as_float(i1 * CKL), 1.f - l * 0.5f, l * 0.5f);
        }
        *reinterpret_cast<float4*>(&tb[4 * tid]) = e;
    }
    __syncthreads();

    if (wid < 4) {
        // ------------------------------------------------ consumers: the kernel's k-loop shape
        f32x16 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        const int hh = lane >> 5, li = lane & 31;
        int aoff[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { const int q = (wid * 4 + mt) * 32 + li; aoff[mt] = ((q / TW) * PW + (q % TW)) * CKP + 4 * hh; }
        float4 a0[4], b0[2][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a0[mt] = make_float4(1.f + lane, 2.f, 3.f, 4.f);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int w = 0; w < 2; ++w) b0[nt][w] = make_float4(0.5f, 0.25f, (float)wid, 1.f);
        for (int it = 0; it < iters / 8; ++it) {
            const float* pb = patch + (it & 1) * PATCH;
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                const int tap = ks / 2, s1 = ks % 2;
                const int off = ((tap / 3) * PW + (tap % 3)) * CKP + 8 * s1;
                if constexpr ((mode & 2) != 0) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) a0[mt] = *reinterpret_cast<const float4*>(&pb[aoff[mt] + off]);
                }
                if constexpr ((mode & 4) != 0) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int w = 0; w < 2; ++w) b0[nt][w] = wts[(it & 3) * (18 * 4 * 64) + ((ks * 2 + nt) * 2 + w) * 64 + lane];
                }
                if constexpr ((mode & 1) != 0) {
#pragma unroll
                    for (int w = 0; w < 2; ++w)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                if constexpr (MF == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[mt]), __builtin_bit_cast(bf16x8, b0[nt][w]), acc[mt][nt], 0, 0, 0);
                                else if constexpr (MF == 1) {   // f32 MFMAs (they hold the vector issue port)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt].x, b0[nt][w].x, acc[mt][nt], 0, 0, 0);
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt].y, b0[nt][w].y, acc[mt][nt], 0, 0, 0);
                                } else {                        // the F(4x4) kernels' shape: v_mfma_f32_16x16x4_f32 (four per 32 x 32 block and k-step)
                                    typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        f32x4 c = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                                        c = __builtin_amdgcn_mfma_f32_16x16x4f32(q & 1 ? a0[mt].y : a0[mt].x, q & 2 ? b0[nt][w].y : b0[nt][w].x, c, 0, 0, 0);
                                        acc[mt][nt][4 * q] = c[0]; acc[mt][nt][4 * q + 1] = c[1]; acc[mt][nt][4 * q + 2] = c[2]; acc[mt][nt][4 * q + 3] = c[3];
                                    }
                                }
                } else {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[mt][0][0] += a0[mt].x + b0[0][0].y + b0[1][1].z;
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) s += acc[mt][nt][0] + acc[mt][nt][9];
        sink[blockIdx.x * 256 + tid] = s;
        return;
    }

    // ---------------------------------------------------- producers: the operand form under test, in inline assembly
    // T = A * (W.hi, W.hi) as ONE packed multiply whose LOW result reads the HIGH register of W (op_sel:[0,1]) - the form hipcc emitted in
    // the failing builds - against the same products from v_mul_f32.  16 independent packed ops per iteration, operands changing every time.
    const int ptid = tid - 256;
    asm volatile("" ::: "v255");               // the kernel's register allocation: 256 per wave, the SIMD's file full with two waves
    unsigned nbad = 0, nzero = 0, badq = 0, badhalf = 0;
    f32x2 A[16], W[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        A[k] = (f32x2){hashf(ptid * 131u + k * 7u + blockIdx.x * 977u), hashf(ptid * 137u + k * 11u + 3u)};
        W[k] = (f32x2){0.25f + 0.5f * hashf(ptid * 139u + k * 13u + 1u), 0.75f + 0.125f * hashf(ptid * 149u + k * 17u + 2u)};
    }
    for (int it = 0; it < iters; ++it) {
        f32x2 T[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (PROD == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(T[k]) : "v"(A[k]), "v"(W[k]));
            else if constexpr (PROD == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(T[k]) : "v"(A[k]), "v"(W[k]));
            else if constexpr (PROD == 3) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(T[k]) : "v"(A[k]), "v"(W[k]));                       // straight
            else if constexpr (PROD == 4) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(T[k]) : "v"(A[k]), "v"(W[k]));        // HIGH result reads the LOW register
            else if constexpr (PROD == 5) asm volatile("v_pk_mov_b32 %0, %2, %1 op_sel:[1,0]" : "=v"(T[k]) : "v"(A[k]), "v"(W[k]));            // (W.hi, A.lo): a move whose LOW result reads a HIGH register
            else                          asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(T[k]) : "v"(A[k]), "v"(W[k]));
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float elo, ehi;
            if constexpr (PROD == 1) {
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(A[k].x), "v"(W[k].y));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(A[k].y), "v"(W[k].y));
            } else if constexpr (PROD == 3) {
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(A[k].x), "v"(W[k].x));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(A[k].y), "v"(W[k].y));
            } else if constexpr (PROD == 5) {
                elo = W[k].y; ehi = A[k].x;
            } else if constexpr (PROD == 6) {
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(elo) : "v"(A[k].x), "v"(W[k].y));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(A[k].y), "v"(W[k].y));
            } else if constexpr (PROD == 4) {
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(A[k].x), "v"(W[k].x));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(A[k].y), "v"(W[k].x));
            } else {                          // lo = A.lo * W.hi + A.lo ; hi = A.hi * W.lo + A.hi
                asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(elo) : "v"(A[k].x), "v"(W[k].y));
                asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(ehi) : "v"(A[k].y), "v"(W[k].x));
            }
            const bool blo = __float_as_uint(T[k].x) != __float_as_uint(elo), bhi = __float_as_uint(T[k].y) != __float_as_uint(ehi);
            if (blo || bhi) {
                ++nbad;
                badq |= 1u << (lane >> 4);
                badhalf |= (blo ? 1u : 0u) | (bhi ? 2u : 0u);
                // the value an operand read as zero would give: 0 (mul) / A.lo (fma)
                nzero += (PROD == 2 || PROD == 6 ? __float_as_uint(T[k].x) == __float_as_uint(A[k].x) : (__float_as_uint(T[k].x) << 1) == 0u) ? 1u : 0u;
            }
            A[k].x = A[k].x * 0.999f + 0.0011f; A[k].y = A[k].y * 1.001f - 0.0007f;     // new operands for the next iteration
        }
    }
    unsigned* r = res + (blockIdx.x * 256 + ptid) * 4;
    r[0] = nbad; r[1] = badq; r[2] = badhalf; r[3] = nzero;
}

struct Case { const void* fn; const char* what; };
template <int mode, int PROD, int MF> static Case mk(const char* what) { return Case{(const void*)&probe<mode, PROD, MF>, what}; }

int main() {
    const int blocks = 256; int iters = 20000;
    float4* d_w; unsigned* d_res; float* d_sink;
    std::vector<float> hw(4 * 18 * 2 * 2 * 64 * 4);
    unsigned s = 12345u;
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    (void)hipMalloc(&d_w, hw.size() * 4); (void)hipMalloc(&d_res, sizeof(unsigned) * blocks * 256 * 4); (void)hipMalloc(&d_sink, sizeof(float) * blocks * 256);
    (void)hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int bytes = LDS_FLOATS * 4;
    // sibling wave's mix: 1 MFMA, 2 LDS reads, 4 weight loads from global memory
    const Case cases[] = {
        mk<0, 1, 0>("pk_mul op_sel:[0,1] | sibling idle"),
        mk<1, 1, 0>("pk_mul op_sel:[0,1] | bf16 MFMA"),
        mk<2, 1, 0>("pk_mul op_sel:[0,1] | LDS reads"),
        mk<4, 1, 0>("pk_mul op_sel:[0,1] | global loads"),
        mk<6, 1, 0>("pk_mul op_sel:[0,1] | LDS reads + global loads"),
        mk<3, 1, 0>("pk_mul op_sel:[0,1] | bf16 MFMA + LDS reads"),
        mk<5, 1, 0>("pk_mul op_sel:[0,1] | bf16 MFMA + global loads"),
        mk<7, 1, 0>("pk_mul op_sel:[0,1] | bf16 MFMA + LDS reads + global loads"),
        mk<5, 2, 0>("pk_fma op_sel:[0,1,0] op_sel_hi:[1,0,1] | bf16 MFMA + global loads"),
        mk<5, 3, 0>("pk_mul (no op_sel) | bf16 MFMA + global loads"),
        mk<5, 4, 0>("pk_mul op_sel_hi:[1,0] (high reads low) | bf16 MFMA + global loads"),
        mk<7, 3, 0>("pk_mul (no op_sel) | bf16 MFMA + LDS reads + global loads"),
        mk<7, 4, 0>("pk_mul op_sel_hi:[1,0] (high reads low) | bf16 MFMA + LDS + global loads"),
        mk<5, 5, 0>("pk_mov_b32 op_sel:[1,0] (low reads high) | bf16 MFMA + global loads"),
        mk<5, 6, 0>("pk_add op_sel:[0,1] | bf16 MFMA + global loads"),
        mk<1, 1, 1>("pk_mul op_sel:[0,1] | f32 MFMA"),
        mk<5, 1, 1>("pk_mul op_sel:[0,1] | f32 MFMA + global loads"),
        mk<7, 1, 1>("pk_mul op_sel:[0,1] | f32 MFMA + LDS reads + global loads"),
        mk<5, 1, 2>("pk_mul op_sel:[0,1] | f32 16x16x4 MFMA + global loads"),
        mk<7, 1, 2>("pk_mul op_sel:[0,1] | f32 16x16x4 MFMA + LDS reads + global loads"),
        mk<5, 2, 2>("pk_fma op_sel:[0,1,0] op_sel_hi:[1,0,1] | f32 16x16x4 MFMA + global loads"),
    };
    for (auto& c : cases) (void)hipFuncSetAttribute(c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    for (int rep = 0; rep < 1; ++rep)
        for (auto& c : cases) {
            (void)hipMemset(d_res, 0, sizeof(unsigned) * blocks * 256 * 4);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            void* args[] = {(void*)&d_w, (void*)&d_res, (void*)&d_sink, (void*)&iters};
            (void)hipLaunchKernel(c.fn, dim3(blocks), dim3(512), args, bytes, 0);
            (void)hipEventRecord(e1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned> h(blocks * 256 * 4);
            (void)hipMemcpy(h.data(), d_res, h.size() * 4, hipMemcpyDeviceToHost);
            unsigned long long bad = 0, zero = 0; unsigned q = 0, hf = 0;
            for (size_t i = 0; i < h.size(); i += 4) { bad += h[i]; q |= h[i + 1]; hf |= h[i + 2]; zero += h[i + 3]; }
            printf("%-78s %7.2f ms  wrong %11llu of %llu  (as if the operand were 0: %llu)  lane quarters 0x%x  halves 0x%x\n",
                   c.what, ms, bad, (unsigned long long)blocks * 256 * 16 * iters, zero, q, hf);
        }
    return 0;
}
