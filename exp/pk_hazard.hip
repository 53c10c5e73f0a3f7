// Probe (round 4): does the packed-FP32 sequence hipcc emitted for the separable upsample's vertical lerp (profiles/r04_ablation.md) give
// wrong results when a sibling wave on the SIMD issues MFMAs?  Waves 0-3 of a workgroup run an MFMA loop; waves 4-7 run the sequence
//   v_pk_mul / v_pk_mul / v_pk_fma (op_sel swizzles) / v_pk_fma with its destination pair = its src1 pair / 2 x v_cvt_pk_bf16_f32 /
//   ds_write_b64 / ds_read_b64 into the registers just stored
// on known inputs and compare what landed in LDS with the same arithmetic in non-packed instructions.
//   hipcc --offload-arch=gfx950 -O3 exp/pk_hazard.hip -o exp/pk_hazard && exp/pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool WITH_MFMA>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, unsigned* __restrict__ bad, float* __restrict__ sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4 * 64 * 2 + 64];
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    if (wid < 4) {
        if (!WITH_MFMA) return;
        f32x16 acc = {};
        const float4 a = make_float4((float)lane, 1.f, 2.f, 3.f), b = make_float4(1.f, (float)wid, 0.5f, 0.25f);
        for (int it = 0; it < iters * 8; ++it)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        sink[blockIdx.x * 256 + tid] = acc[0] + acc[7];
        return;
    }
    unsigned nbad = 0;
    const unsigned addr = (unsigned)(((wid - 4) * 64 + lane) * 8);      // this lane's 8 bytes
    unsigned* wts = &lds[4 * 64 * 2];                                     // a few row-weight pairs to read "the next row's" from
    if (tid < 256 + 64) wts[lane] = __float_as_uint(0.25f + 0.001f * (float)lane);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned waddr = (unsigned)((4 * 64 * 2 + (lane & 31) * 2) * 4);
    for (int it = 0; it < iters; ++it) {
        const float* p = in + ((size_t)(blockIdx.x * iters + it) * 256 + (tid - 256)) * 10;
        const float ha0 = p[0], hb0 = p[1], ha1 = p[2], hb1 = p[3], ha2 = p[4], hb2 = p[5], ha3 = p[6], hb3 = p[7], wa = p[8], wb = p[9];
        // expected, non-packed: o_c = fma(wb, hb_c, wa * ha_c)
        float e0, e1, e2, e3, t;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(wa), "v"(ha0)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(wb), "v"(hb0), "v"(t));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(wa), "v"(ha1)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(wb), "v"(hb1), "v"(t));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(wa), "v"(ha2)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e2) : "v"(wb), "v"(hb2), "v"(t));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(wa), "v"(ha3)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e3) : "v"(wb), "v"(hb3), "v"(t));
        const unsigned want0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e0, e1}, bf16x2));
        const unsigned want1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e2, e3}, bf16x2));
        // the emitted sequence, registers as in the kernel's ISA: W = v[80:81] = (wa, wb); pairs (ha_c, hb_c) in v[112:119], v[128:129]
        //   o0: v34 = ha0*wa (pk_mul lo of (ha0, hb0)*(wa, wb)) ...  here the simpler equivalent pairing the compiler used:
        //   v[116:117] = (ha2*wa, ha3*wb')...  we reproduce its exact operand pattern:
        asm volatile(
            "v_mov_b32 v80, %2\n v_mov_b32 v81, %3\n"                 // W = (wa, wb)
            "v_mov_b32 v114, %8\n v_mov_b32 v115, %9\n"               // (ha2, ha3) x (wa, wb)?  no: the compiler's pairs are (x.lo lane, x.hi lane):
            "v_mov_b32 v112, %4\n v_mov_b32 v113, %5\n"               // v[112:113] = (ha0, hb1)   [lo lane * wa, hi lane * wb]
            "v_mov_b32 v118, %6\n v_mov_b32 v119, %7\n"               // v[118:119] = (hb0, ha1)   [op_sel: lo * wb, hi * wa]
            "v_mov_b32 v128, %10\n v_mov_b32 v129, %11\n"             // v[128:129] = (hb2, ha3)
            "v_pk_mul_f32 v[116:117], v[114:115], v[80:81]\n"          // (ha2*wa, hb3*wb)
            "v_pk_mul_f32 v[34:35], v[112:113], v[80:81]\n"            // (ha0*wa, hb1*wb)
            "v_pk_fma_f32 v[34:35], v[118:119], v[80:81], v[34:35] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"     // (hb0*wb + ., ha1*wa + .)
            "v_pk_fma_f32 v[80:81], v[128:129], v[80:81], v[116:117] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"   // (hb2*wb + ., ha3*wa + .), dst = W
            "v_cvt_pk_bf16_f32 v34, v34, v35\n"
            "v_cvt_pk_bf16_f32 v35, v80, v81\n"
            "ds_write_b64 %0, v[34:35]\n"
            "ds_read_b64 v[34:35], %1\n"
            "s_waitcnt lgkmcnt(0)\n"
            :: "v"(addr), "v"(waddr), "v"(wa), "v"(wb), "v"(ha0), "v"(hb1), "v"(hb0), "v"(ha1), "v"(ha2), "v"(hb3), "v"(hb2), "v"(ha3)
            : "v34", "v35", "v80", "v81", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v128", "v129", "memory");
        const unsigned got0 = lds[((wid - 4) * 64 + lane) * 2], got1 = lds[((wid - 4) * 64 + lane) * 2 + 1];
        // the sequence computes o0 = fma(hb0, wb, ha0*wa), o1 = fma(ha1, wa, hb1*wb), o2 = fma(hb2, wb, ha2*wa), o3 = fma(ha3, wa, hb3*wb):
        float f1, f3;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(wb), "v"(hb1)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(wa), "v"(ha1), "v"(t));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(wb), "v"(hb3)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f3) : "v"(wa), "v"(ha3), "v"(t));
        const unsigned w0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e0, f1}, bf16x2));
        const unsigned w1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e2, f3}, bf16x2));
        (void)want0; (void)want1;
        nbad += (got0 != w0) + (got1 != w1);
    }
    bad[blockIdx.x * 256 + (tid - 256)] = nbad;
}

template <bool M>
static void run(const char* what, const float* d_in, unsigned* d_bad, float* d_sink, int blocks, int iters) {
    hipMemset(d_bad, 0, sizeof(unsigned) * blocks * 256);
    hipLaunchKernelGGL(probe<M>, dim3(blocks), dim3(512), 0, 0, d_in, d_bad, d_sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 256);
    hipMemcpy(h.data(), d_bad, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long bad = 0;
    for (unsigned v : h) bad += v;
    printf("%-60s mismatching dwords: %llu of %llu\n", what, bad, (unsigned long long)blocks * 256 * iters * 2);
}

int main() {
    const int blocks = 512, iters = 256;
    const size_t n = (size_t)blocks * iters * 256 * 10;
    std::vector<float> h(n);
    unsigned s = 777u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    float *d_in, *d_sink; unsigned* d_bad;
    hipMalloc(&d_in, n * 4); hipMalloc(&d_bad, sizeof(unsigned) * blocks * 256); hipMalloc(&d_sink, sizeof(float) * blocks * 256);
    hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        run<false>("packed sequence, no MFMA traffic on the SIMD", d_in, d_bad, d_sink, blocks, iters);
        run<true>("packed sequence, sibling wave issuing MFMAs", d_in, d_bad, d_sink, blocks, iters);
    }
    return 0;
}
