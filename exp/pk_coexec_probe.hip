// Probe (round 5): reproduce, outside libpnpadmm, the fault profiles/r05_race.md pins down in conv3x3_bf16ws_kernel's separable
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

template <int mode>
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
        for (int it = 0; it < iters; ++it) {
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
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[mt]), __builtin_bit_cast(bf16x8, b0[nt][w]), acc[mt][nt], 0, 0, 0);
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

    // ---------------------------------------------------- producers: conv_bf16_kernels.hip's separable interpolation, packed form
    const int ptid = tid - 256;
    unsigned nbad = 0, badq = 0, badc = 0;     // mismatching rows; bit mask of lane quarters; bit mask of channels
    auto lerp1 = [](float wa, float a_, float wb, float b_) { return wa * a_ + wb * b_; };
    constexpr int NG = (PH + 5) / 6, TPG = PW * PPP, TASKS = NG * TPG, ROUNDS = (TASKS + 255) / 256;
    for (int it = 0; it < iters; ++it) {
        float* const buf = patch + ((it + 1) & 1) * PATCH;
#pragma unroll 1
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int T = ptid + 256 * rd;
            if (T < TASKS) {
                const int rg = T / TPG, rest = T - rg * TPG, px = rest / PPP, pt = rest % PPP;
                const float4 ct = *reinterpret_cast<const float4*>(&tb[4 * (PH + px)]);
                const float* l0 = &lowres[(3 * rg) * (LW * CKL) + pt * 4];
                const int c0 = __float_as_int(ct.x), c1 = __float_as_int(ct.y);
                float4 h[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 u = *reinterpret_cast<const float4*>(l0 + j * (LW * CKL) + c0);
                    const float4 v = *reinterpret_cast<const float4*>(l0 + j * (LW * CKL) + c1);
                    h[j] = make_float4(ct.z * u.x + ct.w * v.x, ct.z * u.y + ct.w * v.y, ct.z * u.z + ct.w * v.z, ct.z * u.w + ct.w * v.w);
                }
                float* const dst = &buf[((6 * rg) * PW + px) * CKP + pt * 2];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const bool rok = 6 * rg + i < PH;
                    const float2 rw = *reinterpret_cast<const float2*>(&tb[4 * (rok ? 6 * rg + i : PH - 1)]);
                    const float4 &ha = h[i >> 1], &hb = h[(i >> 1) + 1];
                    const float4 o = make_float4(lerp1(rw.x, ha.x, rw.y, hb.x), lerp1(rw.x, ha.y, rw.y, hb.y), lerp1(rw.x, ha.z, rw.y, hb.z), lerp1(rw.x, ha.w, rw.y, hb.w));
                    const bf16x2 lo = __builtin_convertvector((f32x2){o.x, o.y}, bf16x2);
                    const bf16x2 hi = __builtin_convertvector((f32x2){o.z, o.w}, bf16x2);
                    if (rok) *reinterpret_cast<uint2*>(dst + i * (PW * CKP)) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
                    if (rok) {
                        float4 ha2 = ha, hb2 = hb; float2 rw2 = rw;
                        asm volatile("" : "+v"(ha2.x), "+v"(ha2.y), "+v"(ha2.z), "+v"(ha2.w), "+v"(hb2.x), "+v"(hb2.y), "+v"(hb2.z), "+v"(hb2.w), "+v"(rw2.x), "+v"(rw2.y));
                        const float4 o2 = make_float4(lerp1(rw2.x, ha2.x, rw2.y, hb2.x), lerp1(rw2.x, ha2.y, rw2.y, hb2.y), lerp1(rw2.x, ha2.z, rw2.y, hb2.z), lerp1(rw2.x, ha2.w, rw2.y, hb2.w));
                        const unsigned e0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){o2.x, o2.y}, bf16x2));
                        const unsigned e1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){o2.z, o2.w}, bf16x2));
                        const unsigned s0 = __builtin_bit_cast(unsigned, lo), s1 = __builtin_bit_cast(unsigned, hi);
                        if (e0 != s0 || e1 != s1) {
                            ++nbad;
                            badq |= 1u << (lane >> 4);
                            badc |= ((e0 ^ s0) & 0xffffu ? 1u : 0u) | ((e0 ^ s0) >> 16 ? 2u : 0u) | ((e1 ^ s1) & 0xffffu ? 4u : 0u) | ((e1 ^ s1) >> 16 ? 8u : 0u);
                        }
                    }
                }
            }
        }
    }
    unsigned* r = res + (blockIdx.x * 256 + ptid) * 3;
    r[0] = nbad; r[1] = badq; r[2] = badc;
}

int main() {
    const int blocks = 256; int iters = 400;
    float4* d_w; unsigned* d_res; float* d_sink;
    std::vector<float> hw(4 * 18 * 2 * 2 * 64 * 4);
    unsigned s = 12345u;
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    hipMalloc(&d_w, hw.size() * 4); hipMalloc(&d_res, sizeof(unsigned) * blocks * 256 * 3); hipMalloc(&d_sink, sizeof(float) * blocks * 256);
    hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int bytes = LDS_FLOATS * 4;
    const void* fns[8] = {(const void*)&probe<0>, (const void*)&probe<1>, (const void*)&probe<2>, (const void*)&probe<3>, (const void*)&probe<4>, (const void*)&probe<5>, (const void*)&probe<6>, (const void*)&probe<7>};
    for (auto f : fns) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    const char* names[8] = {"producers alone", "+ MFMA", "+ LDS reads", "+ MFMA + LDS reads", "+ weight loads", "+ MFMA + weight loads", "+ LDS reads + weight loads", "+ MFMA + LDS reads + weight loads (the kernel's mix)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 8; ++mode) {
            hipMemset(d_res, 0, sizeof(unsigned) * blocks * 256 * 3);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            void* args[] = {(void*)&d_w, (void*)&d_res, (void*)&d_sink, (void*)&iters};
            hipLaunchKernel(fns[mode], dim3(blocks), dim3(512), args, bytes, 0);
            hipEventRecord(e1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned> h(blocks * 256 * 3);
            hipMemcpy(h.data(), d_res, h.size() * 4, hipMemcpyDeviceToHost);
            unsigned long long bad = 0; unsigned q = 0, c = 0;
            for (size_t i = 0; i < h.size(); i += 3) { bad += h[i]; q |= h[i + 1]; c |= h[i + 2]; }
            printf("mode %d %-55s %8.2f ms  mismatching lane-rows %llu  lane quarters 0x%x  channels 0x%x\n", mode, names[mode], ms, bad, q, c);
        }
    return 0;
}
