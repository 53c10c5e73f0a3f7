// Staging helpers shared by the conv kernels (direct and Winograd): how one 16-byte piece of a conv's logical input
// tensor is fetched and transformed (MaxPool2d(2), bilinear x2 + concat) while the LDS patch is filled.
#pragma once
#include "pnp_internal.h"

namespace pnp {

static constexpr float kLeaky = 0.2f;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4max(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
// Vector arithmetic that hipcc must NOT pack (MI355X, measured in round 5 - profiles/r05_race.md, exp/pk_opsel_probe.hip): a
// v_pk_{mul,fma,add}_f32 whose LOW result reads the HIGH register of a source pair (op_sel bit = 1: hipcc's SLP vectoriser forms these
// freely, e.g. to broadcast the second weight of a loaded {w0, w1} pair) reads ZERO for that operand in lanes 48-63, about once in 200
// such instructions, while the other wave of the SIMD runs bf16 MFMAs fed by global loads.  f32 MFMAs (they hold the vector issue port) and
// kernels without MFMAs do not trigger it.  Every kernel with bf16 MFMAs therefore does its interpolation arithmetic through these two
// (plain v_mul_f32 / v_fma_f32 the compiler cannot re-pack), and tools/isa_audit.py fails the build if such an operand form appears in one.
__device__ __forceinline__ float mul_np(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fma_np(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// w0 * a + w1 * b, evaluated as fma(w1, b, w0 * a)
__device__ __forceinline__ float lerp_np(float w0, float a, float w1, float b) { return fma_np(w1, b, mul_np(w0, a)); }

template <bool NP = false>
__device__ __forceinline__ float4 f4lerp2(float4 p00, float4 p01, float4 p10, float4 p11, float wx0, float wx1,
                                          float wy0, float wy1) {
    // ATen upsample_bilinear2d: wy0*(wx0*p00 + wx1*p01) + wy1*(wx0*p10 + wx1*p11)
    float4 r;
    if constexpr (NP) {                    // kernels with bf16 MFMAs (above)
        r.x = lerp_np(wy0, lerp_np(wx0, p00.x, wx1, p01.x), wy1, lerp_np(wx0, p10.x, wx1, p11.x));
        r.y = lerp_np(wy0, lerp_np(wx0, p00.y, wx1, p01.y), wy1, lerp_np(wx0, p10.y, wx1, p11.y));
        r.z = lerp_np(wy0, lerp_np(wx0, p00.z, wx1, p01.z), wy1, lerp_np(wx0, p10.z, wx1, p11.z));
        r.w = lerp_np(wy0, lerp_np(wx0, p00.w, wx1, p01.w), wy1, lerp_np(wx0, p10.w, wx1, p11.w));
        return r;
    }
    r.x = wy0 * (wx0 * p00.x + wx1 * p01.x) + wy1 * (wx0 * p10.x + wx1 * p11.x);
    r.y = wy0 * (wx0 * p00.y + wx1 * p01.y) + wy1 * (wx0 * p10.y + wx1 * p11.y);
    r.z = wy0 * (wx0 * p00.z + wx1 * p01.z) + wy1 * (wx0 * p10.z + wx1 * p11.z);
    r.w = wy0 * (wx0 * p00.w + wx1 * p01.w) + wy1 * (wx0 * p10.w + wx1 * p11.w);
    return r;
}

// One 16-byte piece (4 channels starting at concatenated channel c0) of input pixel (n, gy, gx) of the conv's
// logical input tensor is produced in two halves so that the loads can stay in flight across a barrier:
// issue_piece() only LOADS the raw operands (1 float4, or the 4 float4 of a 2x2 max / bilinear footprint);
// finish_piece() applies the stage's input transform to them.  (gy, gx) is in bounds.
template <int SRC> struct RawPiece { float4 v[SRC == SRC_PLAIN ? 1 : 4]; };

template <int SRC>
__device__ __forceinline__ void issue_piece(const ConvArgs& a, int n, int gy, int gx, int c0, RawPiece<SRC>& r) {
    if constexpr (SRC == SRC_PLAIN) {
        r.v[0] = *reinterpret_cast<const float4*>(a.src0 + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + c0);
    } else if constexpr (SRC == SRC_POOL) {
        const int W2 = 2 * a.W;
        const float* p = a.src0 + (((size_t)n * (2 * a.H) + 2 * gy) * W2 + 2 * gx) * a.Cin + c0;
        r.v[0] = *reinterpret_cast<const float4*>(p);
        r.v[1] = *reinterpret_cast<const float4*>(p + a.Cin);
        r.v[2] = *reinterpret_cast<const float4*>(p + (size_t)W2 * a.Cin);
        r.v[3] = *reinterpret_cast<const float4*>(p + (size_t)W2 * a.Cin + a.Cin);
    } else {  // SRC_UPCAT
        if (c0 < a.Cskip) {
            r.v[0] = *reinterpret_cast<const float4*>(a.src0 + (((size_t)n * a.H + gy) * a.W + gx) * a.Cskip + c0);
        } else {
            const int Cup = a.Cin - a.Cskip, Hs = a.H >> 1, Ws = a.W >> 1;
            const int y0 = (int)(a.rh * (float)gy), x0 = (int)(a.rw * (float)gx);
            const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
            const float* base = a.src1 + (size_t)n * Hs * Ws * Cup + (c0 - a.Cskip);
            r.v[0] = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Ws + x0) * Cup);
            r.v[1] = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Ws + x1) * Cup);
            r.v[2] = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Ws + x0) * Cup);
            r.v[3] = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Ws + x1) * Cup);
        }
    }
}

template <int SRC>
__device__ __forceinline__ float4 finish_piece(const ConvArgs& a, int gy, int gx, int c0, const RawPiece<SRC>& r) {
    if constexpr (SRC == SRC_PLAIN) {
        return r.v[0];
    } else if constexpr (SRC == SRC_POOL) {
        return f4max(f4max(r.v[0], r.v[1]), f4max(r.v[2], r.v[3]));
    } else {
        if (c0 < a.Cskip) return r.v[0];
        const float sy = a.rh * (float)gy, sx = a.rw * (float)gx;
        const float ly = fminf(fmaxf(sy - (float)(int)sy, 0.f), 1.f), lx = fminf(fmaxf(sx - (float)(int)sx, 0.f), 1.f);
        return f4lerp2(r.v[0], r.v[1], r.v[2], r.v[3], 1.f - lx, lx, 1.f - ly, ly);
    }
}


}  // namespace pnp
