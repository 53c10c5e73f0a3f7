// Denoiser convolutions for gfx950 (MI355X): the plug-in regulariser of the PnP-ADMM loop.
//
// Replaces the ATen conv2d / max_pool2d / upsample_bilinear2d / cat sequence that the reference's
// UNet.forward issues (/root/reference/evaluation/noise.py:119-133; ConvBlock :88-98).
//
// conv3x3_mfma_kernel: implicit GEMM  out[pixel][cout] = sum_{tap,cin} patch[pixel+tap][cin] * w[cout][cin][tap]
// on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, so parity with
// the f32 reference holds to rounding-order effects only).
//   * activations are NHWC f32: one pixel's 32-channel chunk is one 128-B line
//   * a workgroup (4 waves) owns a TH x TW pixel tile of one slice and BN output channels; per 32-channel
//     chunk it stages the (TH+2) x (TW+2) halo patch in LDS ONCE and reuses it for all 9 taps
//   * the input transform of the stage's first conv is applied while staging, so the pooled /
//     upsampled / concatenated tensors of the reference are never materialised:
//       SRC_POOL  : 2x2 max of the 2H x 2W source                (noise.py:22-25 MaxPool2d(2))
//       SRC_UPCAT : chunks < Cskip from the skip tensor, the rest bilinear x2 (align_corners=True)
//                   of the low-res tensor                          (noise.py:39,46,59)
//   * A fragments come from LDS with one ds_read_b128 per 4 MFMAs (pixel stride 36 floats keeps the
//     b128 lane groups on distinct banks); B fragments stream straight from L2 in the pre-packed
//     per-lane order (1 KiB contiguous per wave-load), prefetched one k-step ahead
//   * epilogue: + bias, LeakyReLU(0.2), NHWC store (two full 128-B lines per store instruction)
#include "pnp_internal.h"

namespace pnp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int CK = 32;    // channels per staged chunk
static constexpr int CKP = 36;   // padded pixel stride in LDS (floats)
static constexpr float kLeaky = 0.2f;

const LayerSpec kLayers[N_LAYERS] = {
    {2, 32, 3, 0, SRC_SIGMA, 0},    {32, 32, 3, 0, SRC_PLAIN, 0},   {32, 32, 3, 0, SRC_PLAIN, 0},
    {32, 64, 3, 1, SRC_POOL, 0},    {64, 64, 3, 1, SRC_PLAIN, 0},   {64, 64, 3, 1, SRC_PLAIN, 0},
    {64, 128, 3, 2, SRC_POOL, 0},   {128, 128, 3, 2, SRC_PLAIN, 0}, {128, 128, 3, 2, SRC_PLAIN, 0},
    {128, 256, 3, 3, SRC_POOL, 0},  {256, 256, 3, 3, SRC_PLAIN, 0}, {256, 256, 3, 3, SRC_PLAIN, 0},
    {256, 512, 3, 4, SRC_POOL, 0},  {512, 512, 3, 4, SRC_PLAIN, 0}, {512, 512, 3, 4, SRC_PLAIN, 0},
    {768, 256, 3, 3, SRC_UPCAT, 256}, {256, 256, 3, 3, SRC_PLAIN, 0}, {256, 256, 3, 3, SRC_PLAIN, 0},
    {384, 128, 3, 2, SRC_UPCAT, 128}, {128, 128, 3, 2, SRC_PLAIN, 0}, {128, 128, 3, 2, SRC_PLAIN, 0},
    {192, 64, 3, 1, SRC_UPCAT, 64},  {64, 64, 3, 1, SRC_PLAIN, 0},   {64, 64, 3, 1, SRC_PLAIN, 0},
    {96, 32, 3, 0, SRC_UPCAT, 32},   {32, 32, 3, 0, SRC_PLAIN, 0},   {32, 32, 3, 0, SRC_PLAIN, 0},
    {32, 1, 1, 0, SRC_PLAIN, 0},
};

// ------------------------------------------------------------------------------------------------
// Weight pack: float4 units [cout/32][cin/32][tap 9][s 4][lane 64]; lane l (n = l&31, hh = l>>5) holds
// W[cout = 32*cb + n][cin = 32*chunk + 8*s + 4*hh + j][ky][kx], j = 0..3 - the B operand of the j-th
// MFMA of k-step (chunk, tap, s).  Two extra k-steps of zeros pad the tail for the prefetch.
size_t conv3x3_pack_floats(int cin, int cout) {
    return (size_t)(cout / 32) * (cin / 32) * 36 * 256 + 512;
}

void pack_conv3x3_weights(const float* oihw, int cin, int cout, float* dst) {
    const int nch = cin / 32;
    size_t o = 0;
    for (int cb = 0; cb < cout / 32; ++cb)
        for (int ch = 0; ch < nch; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int s = 0; s < 4; ++s)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 4; ++j) {
                            const int co = 32 * cb + (l & 31);
                            const int ci = 32 * ch + 8 * s + 4 * (l >> 5) + j;
                            dst[o++] = oihw[((size_t)co * cin + ci) * 9 + tap];
                        }
    for (int i = 0; i < 512; ++i) dst[o++] = 0.f;
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4max(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float4 f4lerp2(float4 p00, float4 p01, float4 p10, float4 p11, float wx0, float wx1,
                                          float wy0, float wy1) {
    // ATen upsample_bilinear2d: wy0*(wx0*p00 + wx1*p01) + wy1*(wx0*p10 + wx1*p11)
    float4 r;
    r.x = wy0 * (wx0 * p00.x + wx1 * p01.x) + wy1 * (wx0 * p10.x + wx1 * p11.x);
    r.y = wy0 * (wx0 * p00.y + wx1 * p01.y) + wy1 * (wx0 * p10.y + wx1 * p11.y);
    r.z = wy0 * (wx0 * p00.z + wx1 * p01.z) + wy1 * (wx0 * p10.z + wx1 * p11.z);
    r.w = wy0 * (wx0 * p00.w + wx1 * p01.w) + wy1 * (wx0 * p10.w + wx1 * p11.w);
    return r;
}

// One 16-byte piece (4 channels starting at concatenated channel c0) of input pixel (n, gy, gx) of the
// conv's logical input tensor, after the stage's input transform.  (gy, gx) is in bounds.
template <int SRC>
__device__ __forceinline__ float4 load_input_piece(const ConvArgs& a, int n, int gy, int gx, int c0) {
    if constexpr (SRC == SRC_PLAIN) {
        return *reinterpret_cast<const float4*>(a.src0 + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + c0);
    } else if constexpr (SRC == SRC_POOL) {
        const int W2 = 2 * a.W;
        const float* p = a.src0 + (((size_t)n * (2 * a.H) + 2 * gy) * W2 + 2 * gx) * a.Cin + c0;
        const float4 v00 = *reinterpret_cast<const float4*>(p);
        const float4 v01 = *reinterpret_cast<const float4*>(p + a.Cin);
        const float4 v10 = *reinterpret_cast<const float4*>(p + (size_t)W2 * a.Cin);
        const float4 v11 = *reinterpret_cast<const float4*>(p + (size_t)W2 * a.Cin + a.Cin);
        return f4max(f4max(v00, v01), f4max(v10, v11));
    } else {  // SRC_UPCAT
        if (c0 < a.Cskip) {
            return *reinterpret_cast<const float4*>(a.src0 + (((size_t)n * a.H + gy) * a.W + gx) * a.Cskip + c0);
        }
        const int Cup = a.Cin - a.Cskip, Hs = a.H >> 1, Ws = a.W >> 1;
        const float sy = a.rh * (float)gy, sx = a.rw * (float)gx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float ly = fminf(fmaxf(sy - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
        const float* base = a.src1 + (size_t)n * Hs * Ws * Cup + (c0 - a.Cskip);
        const float4 p00 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Ws + x0) * Cup);
        const float4 p01 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Ws + x1) * Cup);
        const float4 p10 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Ws + x0) * Cup);
        const float4 p11 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Ws + x1) * Cup);
        return f4lerp2(p00, p01, p10, p11, 1.f - lx, lx, 1.f - ly, ly);
    }
}

template <int TW, int MT, int NT, int WM, int WN, int SRC>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const ConvArgs a) {
    constexpr int BM = WM * MT * 32;   // pixels per workgroup
    constexpr int TH = BM / TW;
    constexpr int PWL = TW + 2;        // patch width (pixels)
    constexpr int PH = TH + 2;
    constexpr int PW = PWL;            // LDS row stride (pixels)
    constexpr int ITEMS = PH * PWL * 8;          // 16-byte pieces per chunk
    constexpr int NIT = (ITEMS + 255) / 256;
    constexpr int LB = SRC == SRC_PLAIN ? NIT : (SRC == SRC_POOL ? 4 : 3);   // staging loads in flight per thread
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM % TW == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) float patch[PH * PW * CKP];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int hh = lane >> 5, li = lane & 31;

    int bt = blockIdx.x;
    const int tx0 = (bt % a.tilesX) * TW;
    bt /= a.tilesX;
    const int ty0 = (bt % a.tilesY) * TH;
    const int n = bt / a.tilesY;
    if (a.tact != nullptr && a.tact[n] > 0.5f) return;   // slice is done: leave its planes untouched

    // LDS float offset of this lane's A row for each of its M-blocks (tap (0,0), channel 4*hh)
    int aoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wm * MT + mt) * 32 + li;
        aoff[mt] = ((q / TW) * PW + (q % TW)) * CKP + 4 * hh;
    }

    const int nchunks = a.Cin / CK;
    const float4* bptr[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cb = blockIdx.y * (WN * NT) + wn * NT + nt;
        bptr[nt] = reinterpret_cast<const float4*>(a.wpack) + (size_t)cb * nchunks * 36 * 64 + lane;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // B fragments run two k-steps ahead of the MFMAs that consume them (L2 latency ~ one k-step of MFMA time);
    // the tail of the packed stream is zero-padded by two k-steps so the prefetch never leaves the buffer.
    float4 b0[NT], b1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        b0[nt] = bptr[nt][0];
        b1[nt] = bptr[nt][64];
    }

    for (int c = 0; c < nchunks; ++c) {
        // ---- stage the halo patch of chunk c (zero outside the image = the conv's zero padding) ----
        // Loads are issued LB pieces at a time so the pool / bilinear transforms (4 loads per piece) stay
        // within the register budget; the barrier that retires the previous chunk's readers sits after
        // the first batch of loads has been issued.
#pragma unroll 1
        for (int it0 = 0; it0 < NIT; it0 += LB) {
            float4 stg[LB];
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                const int idx = tid + (it0 + k) * 256;
                const int part = idx & 7, pp = idx >> 3;
                const int py = pp / PWL, px = pp % PWL;
                const int gy = ty0 + py - 1, gx = tx0 + px - 1;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = load_input_piece<SRC>(a, n, gy, gx, c * CK + part * 4);
                stg[k] = v;
            }
            if (it0 == 0 && c > 0) __syncthreads();   // all waves finished reading the previous chunk's patch
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                const int idx = tid + (it0 + k) * 256;
                const int part = idx & 7, pp = idx >> 3;
                const int py = pp / PWL, px = pp % PWL;
                if (idx < ITEMS) *reinterpret_cast<float4*>(&patch[(py * PW + px) * CKP + part * 4]) = stg[k];
            }
        }
        __syncthreads();

        // ---- 9 taps x 4 k-steps of 8 channels, software-pipelined: while the MFMAs of k-step ks issue, the A
        // fragments of ks+1 (LDS) and the B fragments of ks+2 (L2) are in flight.  The sched_barriers keep hipcc
        // from sinking the loads down to their first use (it does, and then every k-step waits on L2).
        const float4* bp[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bp[nt] = bptr[nt] + (size_t)c * 36 * 64;
        float4 a0[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a0[mt] = *reinterpret_cast<const float4*>(&patch[aoff[mt]]);
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) {
            float4 b2[NT], a1[MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b2[nt] = bp[nt][(ks + 2) * 64];
            if (ks + 1 < 36) {
                const int tap1 = (ks + 1) >> 2, s1 = (ks + 1) & 3;
                const int off1 = ((tap1 / 3) * PW + (tap1 % 3)) * CKP + 8 * s1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a1[mt] = *reinterpret_cast<const float4*>(&patch[aoff[mt] + off1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt].x, b0[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt].y, b0[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt].z, b0[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt].w, b0[nt].w, acc[mt][nt], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { b0[nt] = b1[nt]; b1[nt] = b2[nt]; }
            if (ks + 1 < 36) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a0[mt] = a1[mt];
            }
        }
    }

    // ---- epilogue: bias + LeakyReLU(0.2), NHWC store ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (blockIdx.y * (WN * NT) + wn * NT + nt) * 32 + li;
        const float bias = a.bias[co];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int q = (wm * MT + mt) * 32 + row;
                const int gy = ty0 + q / TW, gx = tx0 + q % TW;
                if (gy < a.H && gx < a.W) {
                    float v = acc[mt][nt][r] + bias;
                    v = v > 0.f ? v : kLeaky * v;
                    a.dst[(((size_t)n * a.H + gy) * a.W + gx) * a.Cout + co] = v;
                }
            }
        }
    }
}

template <int TW, int MT, int NT, int WM, int WN>
static hipError_t launch_cfg(const ConvArgs& a0, int src_mode, hipStream_t s) {
    constexpr int BM = WM * MT * 32, TH = BM / TW, BN = WN * NT * 32;
    ConvArgs a = a0;
    a.tilesX = (a.W + TW - 1) / TW;
    a.tilesY = (a.H + TH - 1) / TH;
    dim3 grid((unsigned)(a.tilesX * a.tilesY * a.N), (unsigned)(a.Cout / BN));
    dim3 block(256);
    switch (src_mode) {
        case SRC_PLAIN: hipLaunchKernelGGL((conv3x3_mfma_kernel<TW, MT, NT, WM, WN, SRC_PLAIN>), grid, block, 0, s, a); break;
        case SRC_POOL:  hipLaunchKernelGGL((conv3x3_mfma_kernel<TW, MT, NT, WM, WN, SRC_POOL>), grid, block, 0, s, a); break;
        case SRC_UPCAT: hipLaunchKernelGGL((conv3x3_mfma_kernel<TW, MT, NT, WM, WN, SRC_UPCAT>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int TW>
static hipError_t launch_tw(const ConvArgs& a, int src_mode, hipStream_t s) {
    if (a.Cout == 32) return launch_cfg<TW, 2, 1, 4, 1>(a, src_mode, s);
    if (a.Cout == 64) return launch_cfg<TW, 2, 1, 2, 2>(a, src_mode, s);
    return launch_cfg<TW, 2, 2, 2, 2>(a, src_mode, s);   // Cout multiple of 128
}

hipError_t launch_conv3x3(const ConvArgs& a, int src_mode, hipStream_t s) {
    if (a.Cin % 32 != 0 || a.Cout % 32 != 0 || (a.Cout > 64 && a.Cout % 128 != 0)) return hipErrorInvalidValue;
    if (a.W >= 32) return launch_tw<32>(a, src_mode, s);
    if (a.W >= 16) return launch_tw<16>(a, src_mode, s);
    return launch_tw<8>(a, src_mode, s);
}

// ------------------------------------------------------------------------------------------------
// First conv: Cin = 2 (image, sigma plane), K = 18.  Direct f32 FMA; 8 lanes share a pixel, 4 couts each,
// so a wave stores 8 pixels x 128 B contiguous.  The sigma plane is never materialised; like every
// conv input it is ZERO in the padding halo (noise.py:161-162 cat, then conv with padding=1).
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ ximg, const float2* __restrict__ z,
                                                         const float2* __restrict__ u, const float* __restrict__ sigma,
                                                         const float* __restrict__ tact, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ dst,
                                                         int N, int H, int W) {
    const int cg = threadIdx.x & 7;
    float wr[4][18];
    float br[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        br[j] = bias[cg * 4 + j];
#pragma unroll
        for (int k = 0; k < 18; ++k) wr[j][k] = w[(cg * 4 + j) * 18 + k];
    }
    const size_t total = (size_t)N * H * W;
    for (size_t p = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); p < total; p += (size_t)gridDim.x * 32) {
        const int gx = (int)(p % W);
        const int gy = (int)((p / W) % H);
        const int n = (int)(p / ((size_t)W * H));
        if (tact != nullptr && tact[n] > 0.5f) continue;
        const float sg = sigma[n];
        float acc[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = gy + ky - 1, xx = gx + kx - 1;
                float d = 0.f, sv = 0.f;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const size_t q = ((size_t)n * H + yy) * W + xx;
                    d = ximg != nullptr ? ximg[q] : (z[q].x - u[q].x);
                    sv = sg;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = fmaf(wr[j][ky * 3 + kx], d, acc[j]);
                    acc[j] = fmaf(wr[j][9 + ky * 3 + kx], sv, acc[j]);
                }
            }
        float4 o;
        o.x = acc[0] > 0.f ? acc[0] : kLeaky * acc[0];
        o.y = acc[1] > 0.f ? acc[1] : kLeaky * acc[1];
        o.z = acc[2] > 0.f ? acc[2] : kLeaky * acc[2];
        o.w = acc[3] > 0.f ? acc[3] : kLeaky * acc[3];
        *reinterpret_cast<float4*>(dst + p * 32 + cg * 4) = o;
    }
}

hipError_t launch_conv_first(const float* ximg, const float2* z, const float2* u, const float* sigma,
                             const float* tact, const float* w, const float* bias, float* dst, int N, int H, int W,
                             hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    unsigned blocks = (unsigned)((total + 31) / 32);
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipLaunchKernelGGL(conv_first_kernel, dim3(blocks), dim3(256), 0, s, ximg, z, u, sigma, tact, w, bias, dst, N, H, W);
    return hipGetLastError();
}

// Last conv: 1x1, 32 -> 1, + residual on the image channel + clamp to [0,1]
// (noise.py:67,130-133 `noisy_img[:, :C] + residual`, :164 clamp).
__global__ __launch_bounds__(256) void conv_last_kernel(const float* __restrict__ act, const float* __restrict__ ximg,
                                                        const float2* __restrict__ z, const float2* __restrict__ u,
                                                        const float* __restrict__ tact, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int N, int H, int W) {
    const int cg = threadIdx.x & 7;
    const float4 wv = *reinterpret_cast<const float4*>(w + cg * 4);
    const float b = bias[0];
    const size_t hw = (size_t)H * W;
    const size_t total = (size_t)N * hw;
    for (size_t p = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); p < total; p += (size_t)gridDim.x * 32) {
        const int n = (int)(p / hw);
        if (tact != nullptr && tact[n] > 0.5f) continue;
        const float4 v = *reinterpret_cast<const float4*>(act + p * 32 + cg * 4);
        float d = v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        d += __shfl_xor(d, 4);
        if (cg == 0) {
            const float img = ximg != nullptr ? ximg[p] : (z[p].x - u[p].x);
            out[p] = fminf(fmaxf(img + (d + b), 0.f), 1.f);
        }
    }
}

hipError_t launch_conv_last(const float* act, const float* ximg, const float2* z, const float2* u, const float* tact,
                            const float* w, const float* bias, float* out, int N, int H, int W, hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    unsigned blocks = (unsigned)((total + 31) / 32);
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipLaunchKernelGGL(conv_last_kernel, dim3(blocks), dim3(256), 0, s, act, ximg, z, u, tact, w, bias, out, N, H, W);
    return hipGetLastError();
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W) {
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t p = i / C;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const int n = (int)(p / H);
        dst[(((size_t)n * C + c) * H + y) * W + x] = src[i];
    }
}

hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s) {
    const size_t total = (size_t)N * C * H * W;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks), dim3(256), 0, s, src, dst, N, C, H, W);
    return hipGetLastError();
}

}  // namespace pnp
