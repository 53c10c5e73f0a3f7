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
#include "conv_staging.h"
#include <cstring>
#include <cstdlib>

namespace pnp {



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
// Tile plan of a conv3x3 launch.  Measured on MI355X (exp/mfma_rate2.hip): a wave that issues a 32x32x2 f32 MFMA
// cannot hide its own LDS/global loads under it, so the k-loop rate is set by loads per MFMA: a 2x1 register
// tile (MT x NT blocks of 32x32) runs at 74 cycles/MFMA, 2x2 at 69, 4x1 at 67.6, 4x2 at 66.2, 4x4 at 65.0 (64 is
// the pipe).  So every config uses MT = 4 and the widest NT that Cout and the CU count allow.
ConvPlan conv3x3_plan(int N, int H, int W, int Cin, int Cout, bool bf16, int src_mode, bool allow_ws) {
    ConvPlan p{};
    p.tw = W >= 32 ? 32 : (W >= 16 ? 16 : 8);
    const long pixels = (long)N * H * W;
    p.mt = 4;
    p.splitk = 1;
    if (Cout == 32) { p.mt = 2; p.nt = 1; p.wm = 4; p.wn = 1; p.ck = 32; }    // short K, 128-B pixels: small tile, 3 workgroups/CU
    else if (Cout == 64) { p.nt = 2; p.wm = 4; p.wn = 1; p.ck = 16; }
    else if (!bf16 && Cout % 256 == 0 && pixels / 256 * (Cout / 256) >= 256) { p.nt = 4; p.wm = 2; p.wn = 2; p.ck = 32; }
    else { p.nt = 2; p.wm = 2; p.wn = 2; p.ck = 32; }
    auto finish = [&]() {
        p.bm = p.wm * p.mt * 32;
        p.bn = p.wn * p.nt * 32;
        p.th = p.bm / p.tw;
        p.tiles_x = (W + p.tw - 1) / p.tw;
        p.tiles_y = (H + p.th - 1) / p.th;
        return (long)p.tiles_x * p.tiles_y * N * (Cout / p.bn);
    };
    long blocks = finish();
    // Small problems (the reference's own batch-1 128x128 case): the big tiles leave most CUs idle behind a serial
    // K loop.  Use narrow tiles and cut K into `splitk` ranges of whole chunks, one workgroup each; the partial sums
    // are combined in a fixed order by splitk_reduce_kernel (bit-reproducible, unlike float atomics).
    // Round 4: what bounds such a launch is ONE WAVE's chain of MFMAs - a 32-channel chunk on a 2 x 1 register tile is 288
    // v_mfma_f32_32x32x2_f32 = 7.7 us - on a chip with most SIMDs idle, so the f32 plan cuts finer: one 32-pixel M-block per
    // wave and 16-channel chunks (72 MFMAs per chunk, twice the tiles, twice the K ranges): 20 -> 14 us per layer at
    // 1 x 128 x 128 (profiles/r04_ablation.md).  (bf16 k-steps are sixteen times shorter; that mode keeps the round-3 tiles.)
    // The upsample + concat layers keep the round-3 tile and its 384-workgroup rule (their chain is the two-stage staging, not the
    // MFMAs: no gain measured).
    static const bool coarse = getenv("PNP_SPLITK_COARSE") != nullptr;       // (A/B: the round-3 tiles)
    const bool fine = !coarse && !bf16;
    if (blocks < 128 && Cout >= 64) {
        const bool f = fine && src_mode != SRC_UPCAT;
        p.mt = f ? 1 : 2; p.nt = 1; p.ck = f ? 16 : 32;
        if (Cout % 128 == 0) { p.wm = 1; p.wn = 4; } else { p.wm = 2; p.wn = 2; }
        blocks = finish();
        const int nchunks = Cin / p.ck;
        int sk = (int)((384 + blocks - 1) / blocks);
        if (sk > nchunks) sk = nchunks;
        while (nchunks % sk != 0) --sk;             // equal ranges
        if (f) {
            // K ranges by a makespan model fitted to per-layer timings (MI355X, 1 ... 8 slices of 128 x 128 ... 512 x 512): a workgroup
            // alone on its CU takes ~5.2 us per 16-channel chunk (one wave's dependent MFMA chain + an exposed staging round trip),
            // k co-resident ones ~3.3 us each per chunk (the pipe is shared); every K range writes and re-reads one plane (~5 TB/s,
            // mostly L2 / MALL hits)
            const float plane_us = (float)pixels * (float)Cout * 4.f * 2.f / 5.0e6f;
            float best = 1e30f;
            for (int d = 1; d <= nchunks; ++d) {
                if (nchunks % d != 0) continue;
                const float k = (float)((blocks * d + 255) / 256);
                const float cost = (float)(nchunks / d) * fmaxf(5.2f, 3.3f * k) + (float)d * plane_us;
                if (cost < best * 0.97f) { best = cost; sk = d; }
            }
        }
        p.splitk = sk;
    } else if (blocks <= 128 && Cout == 32 && fine) {
        p.mt = 1;                                   // 128-pixel tiles (LDS epilogue, pooled copy and fused last layer as on the 256-pixel tile)
        blocks = finish();
    }
    // bf16 mode, chip-filling problems: the producer / consumer kernel (conv_bf16_kernels.hip) takes the layers with
    // Cout >= 64 (sources PLAIN, UPCAT; POOL on the 256 x 128 tile), 512-pixel x 64-channel tiles for Cout = 64,
    // 256 x 128 otherwise, always 32-channel chunks; it is persistent (one 8-wave workgroup per CU), so it wants >= 192 tiles
    if (bf16 && allow_ws && p.splitk == 1 && p.mt == 4 && p.tw >= 16 && Cin % 32 == 0 && (Cout == 64 || Cout % 128 == 0) &&
        (src_mode == SRC_PLAIN || src_mode == SRC_UPCAT || (src_mode == SRC_POOL && Cout != 64))) {
        ConvPlan q = p;
        q.nt = 2; q.ck = 32;
        if (Cout == 64) { q.wm = 4; q.wn = 1; } else { q.wm = 2; q.wn = 2; }
        const ConvPlan keep = p;
        p = q;
        if (finish() >= 192) { p.ws = 1; return p; }
        p = keep;
    }
    // ... and the plain 32-channel level-0 layers on its 2 x 1 register tile (same tile as here: 256 pixels x 32 channels); these
    // run at the HBM roofline of their bf16 tensors either way (measured 0.139 / 0.155 / 0.138 ms against 0.152 / 0.167 / 0.149
    // here), while up4.conv-0 - two interpolated chunks per 1152-cycle k-loop: producer-bound - and the fused last layer are
    // faster on this file's kernel (0.52 vs 0.66 ms, 0.14 vs 0.18), so they stay (pnp_capi.hip clears ws for layer 26)
    if (bf16 && allow_ws && p.splitk == 1 && Cout == 32 && p.tw == 32 && Cin % 32 == 0 && src_mode == SRC_PLAIN && blocks >= 192)
        p.ws = 1;
    return p;
}

// Weight pack: float4 units [cout/32][cin/ck][tap 9][s ck/8][lane 64]; lane l (n = l&31, hh = l>>5) holds
// W[cout = 32*cb + n][cin = ck*chunk + 8*s + 4*hh + j][ky][kx], j = 0..3 - the B operand of the j-th MFMA of
// k-step (chunk, tap, s).  Four extra k-steps of zeros pad the tail for the prefetch.
size_t conv3x3_pack_floats(int cin, int cout) { return (size_t)(cout / 32) * (cin / 8) * 9 * 256 + 1024; }

void pack_conv3x3_weights(const float* oihw, int cin, int cout, int ck, float* dst) {
    size_t o = 0;
    for (int cb = 0; cb < cout / 32; ++cb)
        for (int ch = 0; ch < cin / ck; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int s = 0; s < ck / 8; ++s)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 4; ++j) {
                            const int co = 32 * cb + (l & 31);
                            const int ci = ck * ch + 8 * s + 4 * (l >> 5) + j;
                            dst[o++] = oihw[((size_t)co * cin + ci) * 9 + tap];
                        }
    for (int i = 0; i < 1024; ++i) dst[o++] = 0.f;
}

// bf16 variant (PNP_FLAG_BF16_CONVS): 16-byte units [cout/32][cin/ck][tap 9][s ck/16][term][lane 64]; lane l (n = l&31, h = l>>5)
// holds W[32*cb + n][ck*chunk + 16*s + 8*h + j][ky][kx], j = 0..7: the B operand of v_mfma_f32_32x32x16_bf16 for k-step
// (chunk, tap, s).  terms = 1: the weight rounded to bf16 (nearest even).  terms = 2 (the mode's default): TWO bf16 terms per
// weight, hi = bf16(w) and lo = bf16(w - hi), 1 KiB each per k-step, hi first - the k-loop multiplies the same activation fragment
// by both and accumulates both products in f32, i.e. it convolves with the 16-bit-mantissa weight hi + lo: the fixed 2^-8
// perturbation of the network that one-term weights are - 0.015 dB of PSNR drift over configs[4]'s 50 iterations - becomes 2^-16.
static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
constexpr int kBf16TailSteps = 16;                        // zero k-steps behind the last stream: the fragment rings read ahead (PFD <= 9)
size_t conv3x3_pack_floats_bf16(int cin, int cout, int terms) {
    return ((size_t)(cout / 32) * (cin / 16) * 9 + kBf16TailSteps) * 256 * terms;
}
void pack_conv3x3_weights_bf16(const float* oihw, int cin, int cout, int ck, int terms, float* dst_f) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
    size_t o = 0;
    for (int cb = 0; cb < cout / 32; ++cb)
        for (int ch = 0; ch < cin / ck; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int s = 0; s < ck / 16; ++s)
                    for (int term = 0; term < terms; ++term)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const int co = 32 * cb + (l & 31);
                                const int ci = ck * ch + 16 * s + 8 * (l >> 5) + j;
                                const float w = oihw[((size_t)co * cin + ci) * 9 + tap];
                                const uint16_t hi = f32_to_bf16_rne(w);
                                dst[o++] = term == 0 ? hi : f32_to_bf16_rne(w - bf16_to_f32(hi));   // (w - hi is exact in f32)
                            }
    for (size_t i = 0; i < (size_t)kBf16TailSteps * 512 * terms; ++i) dst[o++] = 0;
}

// One workgroup (4 waves) = one TH x TW pixel tile of one slice x BN output channels; wave (wm, wn) owns a
// MT x NT register tile of 32x32 MFMA blocks (32*MT pixels x 32*NT channels).  Per CK-channel chunk:
//     barrier | registers -> LDS patch (input transform applied) | barrier | issue next chunk's loads | k-loop
// The next chunk's global loads are issued BEFORE the k-loop and consumed after it, so HBM latency sits under
// thousands of MFMA cycles; 2-3 workgroups per CU cover each other's barriers, LDS writes and epilogues.
//
// BF16 = true (PNP_FLAG_BF16_CONVS, BASELINE configs[4]): the patch is rounded to bf16 as it is committed to LDS and the
// k-loop runs on v_mfma_f32_32x32x16_bf16 (f32 accumulate): one MFMA per (M-block, N-block) and 16-channel k-step instead
// of four per 8 channels.  Activations in HBM, input transforms, bias and epilogue stay f32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

//
// A16 = true (bf16 mode, the 32-channel level-0 plan): the PLAIN source / the UPCAT skip tensor hold bf16 (written so by the
// producing launch, rounded once with the same round-to-nearest-even the staging would apply - the patch gets the same
// bits, half the HBM bytes) and so does dst unless a.act16 says f32 (the layer in front of the unfused 1x1 conv).  The
// pooled copy and the low-res source of the upsample stay f32: their consumers round AFTER max / interpolation.
template <int TW, int MT, int NT, int WM, int WN, int CK, int SRC, int WPS, bool SPLITK, int BF16, bool A16>
__global__ __launch_bounds__(256, WPS) void conv3x3_mfma_kernel(const ConvArgs a) {
    constexpr int NW = BF16 == 2 ? 2 : 1;   // bf16 terms per weight (pack_conv3x3_weights_bf16): NW MFMAs per (M-block, N-block, k-step)
    static_assert(!A16 || (BF16 && !SPLITK && MT == 2 && NT == 1 && WM == 4 && CK == 32 && SRC != SRC_POOL), "bf16 activations: level-0 plan only");
    // padded pixel stride in LDS (floats): b128 lane groups hit 16 distinct slots (f32: CK + 4; bf16: CK + 8 halves)
    constexpr int CKP = BF16 ? (CK + 8) / 2 : CK + 4;
    constexpr int KCH = BF16 ? 16 : 8; // channels per k-step
    constexpr int KS = 9 * (CK / KCH); // k-steps per chunk
    // B fragments in flight (k-steps ahead).  PFD must divide KS (18 or 36 / 9 or 18) so that a chunk's k-step j always sits
    // in slot j % PFD; a bf16 k-step is only MT*NT*32 cycles of MFMA, so the small tiles look 9 k-steps ahead.
    // (two-term weights: a slot holds two fragments and a k-step is twice as long - 6 slots cover the latency 9 did and the
    // three-workgroups-per-CU variants stop spilling)
    // (the upsample + concat variant on bf16 activations - up4.conv-0 - runs three workgroups per CU since late round 4: its staging is a
    // chain of LDS round trips that only more waves hide, 0.618 -> 0.536 ms at 16 x 512 x 512; three slots keep it at 157 registers)
    constexpr int PFD = BF16 ? (MT * NT <= 2 ? (NW == 2 ? ((SRC == SRC_UPCAT && A16) ? 3 : 6) : 9) : 3) : 2;
    static_assert(KS % PFD == 0, "slot rotation must line up at chunk boundaries");
    constexpr int PPP = CK / 4;        // 16-byte pieces per pixel
    constexpr int BM = WM * MT * 32;   // pixels per workgroup tile
    constexpr int TH = BM / TW;
    constexpr int PWL = TW + 2;        // patch width (pixels)
    constexpr int PH = TH + 2;
    constexpr int PW = PWL;            // LDS row stride (pixels)
    constexpr int ITEMS = PH * PWL * PPP;        // 16-byte f32 pieces per chunk
    constexpr int NIT = (ITEMS + 255) / 256;
    constexpr int PPS = A16 ? CK / 8 : PPP;      // pieces per pixel / per chunk of a chunk loaded from src0 (A16: 8 bf16 channels each)
    constexpr int ITEMS_S = PH * PWL * PPS;
    constexpr int NIT_S = (ITEMS_S + 255) / 256;
    // PLAIN chunks: the raw loads of the next chunk are held in registers across the k-loop.  UPCAT chunks that come from
    // the bilinear x2 upsample (4 source pixels per patch pixel) do not fit in registers that way; instead the LOW-RES
    // source region of the tile, (TH/2+3) x (TW/2+3) pixels, is prefetched like a PLAIN chunk, parked in LDS, and the
    // patch is interpolated LDS -> LDS (same scheme as the Winograd kernel).  POOL sources (only used when no pooled copy
    // exists) stage synchronously in batches.
    constexpr bool UP2 = SRC == SRC_UPCAT;
    constexpr bool PREFETCH = SRC == SRC_PLAIN || UP2;
    constexpr int LB = PREFETCH ? NIT_S : (NIT < 3 ? NIT : 3);  // pieces per batch when staging synchronously
    constexpr int LH = TH / 2 + 3, LW = TW / 2 + 3;            // low-res region bound (rows, cols)
    constexpr int CKL = CK + 4;                                // its pixel stride (floats, always f32)
    constexpr int LITEMS = LH * LW * PPP;
    constexpr int NITL = (LITEMS + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM % TW == 0, "tile shape");

    constexpr int BN_ = WN * NT * 32;
    constexpr bool LDS_EPI = !SPLITK && (size_t)BM * (BN_ + 4) <= (size_t)PH * PW * (CK + 4);   // output tile fits in the f32 patch space
    constexpr int LOWRES_FLOATS = UP2 ? ((LH * LW * CKL + 3) & ~3) : 0;
    // the output tile of the LDS epilogue may run on into the low-res region behind the patch (dead by then): with the bf16
    // patch a 32-channel UPCAT workgroup stays under 48 KB
    constexpr int EPI_FLOATS = LDS_EPI ? BM * (BN_ + 4) : 0;
    constexpr int PATCH_FLOATS = PH * PW * CKP + LOWRES_FLOATS >= EPI_FLOATS ? PH * PW * CKP : EPI_FLOATS - LOWRES_FLOATS;
    static_assert(PATCH_FLOATS % 4 == 0, "low-res region stays 16-byte aligned");
    constexpr int LDS_FLOATS = PATCH_FLOATS + LOWRES_FLOATS + (UP2 ? 4 * (PH + PW) : 0);
    __shared__ __attribute__((aligned(16))) float patch[LDS_FLOATS];
    float* const lowres = patch + PATCH_FLOATS;            // UPCAT: [LH][LW][CKL] low-res source region
    float* const rowT = lowres + LOWRES_FLOATS;            // UPCAT: per patch row / column {offsets of the two source lines in the low-res
    float* const colT = rowT + 4 * PH;                     // region (int bits), their two weights}; zero weights outside the image

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
    const int cbt = blockIdx.y;
    if (a.tact != nullptr && a.tact[n] > 0.5f) return;   // slice is done: leave its planes untouched

    const int nchunks_all = a.Cin / CK;
    const int cpb = SPLITK ? nchunks_all / (int)gridDim.z : nchunks_all;   // chunks per workgroup
    const int c_begin = SPLITK ? (int)blockIdx.z * cpb : 0;
    const int c_end = c_begin + cpb;

    // ---- staging helpers ------------------------------------------------------------------------------------
    auto store_patch = [&](int py, int px, int part, float4 v) {
        if constexpr (BF16) {                             // round to nearest even, 4 channels = 8 bytes
            const bf16x2 lo = __builtin_convertvector((f32x2){v.x, v.y}, bf16x2);
            const bf16x2 hi = __builtin_convertvector((f32x2){v.z, v.w}, bf16x2);
            *reinterpret_cast<uint2*>(&patch[(py * PW + px) * CKP + part * 2]) =
                make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
        } else {
            *reinterpret_cast<float4*>(&patch[(py * PW + px) * CKP + part * 4]) = v;
        }
    };
    // UPCAT geometry of this tile: low-res rows/cols [ylo, ylo+LH) x [xlo, xlo+LW) cover every source pixel of the patch
    const int Hs = a.H >> 1, Ws = a.W >> 1;
    // SEP (bf16 operands, round 5): the upsampled chunks interpolate separably, as the producers of conv3x3_bf16ws_kernel do - a task's four
    // horizontal lerps once, one vertical lerp of two COMPILE-TIME lines per patch row (upsample_lines_regular(), checked by pnp_create);
    // the region then starts at line ty0 / 2 - 1 (a zero line above the image)
    constexpr bool SEP = UP2 && BF16 != 0;
    const int ylo = UP2 ? (SEP ? ty0 / 2 - 1 : (int)(a.rh * (float)(ty0 > 0 ? ty0 - 1 : 0))) : 0;
    const int xlo = UP2 ? (int)(a.rw * (float)(tx0 > 0 ? tx0 - 1 : 0)) : 0;
    const int nskip = UP2 ? a.Cskip / CK : 0;             // leading chunks that come straight from the skip tensor

    constexpr int LBS = NIT_S < 4 ? NIT_S : 4;           // skip-chunk pieces per synchronous batch (UPCAT)
    RawPiece<SRC> raw[UP2 ? 1 : (PREFETCH ? NIT_S : LB)];
    // UPCAT: the low-res region of an upsampled chunk (always prefetched) or the pieces of a skip chunk - prefetched whole
    // under the previous k-loop where the accumulator tile leaves the registers (32 accumulators), else staged in
    // synchronous batches of LBS
    constexpr bool SKIP_PF = UP2 && MT * NT <= 2;
    constexpr int RAWU_SKIP = SKIP_PF ? NIT_S : LBS;
    float4 rawu[UP2 ? (NITL > RAWU_SKIP ? NITL : RAWU_SKIP) : 1];
    auto issue = [&](int c, int it0, int cnt) {           // loads of pieces [it0, it0+cnt) of chunk c -> raw[0..cnt)
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            const int idx = tid + (it0 + k) * 256;
            const int part = idx % PPS, pp = idx / PPS;
            const int py = pp / PWL, px = pp % PWL;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            if (idx < ITEMS_S && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const size_t pix = ((size_t)n * a.H + gy) * a.W + gx;
                if constexpr (A16) {                      // 8 bf16 channels of the PLAIN source / the skip tensor
                    const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const uint16_t*>(a.src0) +
                                                                      pix * (UP2 ? a.Cskip : a.Cin) + c * CK + part * 8);
                    if constexpr (UP2) rawu[k] = v; else raw[k].v[0] = v;
                } else if constexpr (UP2) {               // a skip chunk: one plain 16-byte piece
                    rawu[k] = *reinterpret_cast<const float4*>(a.src0 + pix * a.Cskip + c * CK + part * 4);
                } else {
                    issue_piece<SRC>(a, n, gy, gx, c * CK + part * 4, raw[k]);
                }
            }
        }
    };
    auto commit = [&](int c, int it0, int cnt) {          // transform raw[0..cnt) and write the LDS patch
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            const int idx = tid + (it0 + k) * 256;
            const int part = idx % PPS, pp = idx / PPS;
            const int py = pp / PWL, px = pp % PWL;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            if (idx < ITEMS_S) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);      // zero outside the image = the conv's zero padding
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                    if constexpr (UP2) v = rawu[k];
                    else if constexpr (A16) v = raw[k].v[0];
                    else v = finish_piece<SRC>(a, gy, gx, c * CK + part * 4, raw[k]);
                }
                if constexpr (A16) *reinterpret_cast<float4*>(&patch[(py * PW + px) * CKP + part * 4]) = v;   // bf16 already
                else store_patch(py, px, part, v);
            }
        }
    };
    auto issue_lo = [&](int c) {                          // UPCAT: low-res region of upsampled chunk c -> rawu
        const int Cup = a.Cin - a.Cskip;
        const float* base = a.src1 + (size_t)n * Hs * Ws * Cup + (c * CK - a.Cskip);
#pragma unroll
        for (int k = 0; k < NITL; ++k) {
            const int idx = tid + k * 256;
            const int part = idx % PPP, pp = idx / PPP;
            const int sy = ylo + pp / LW, sx = xlo + pp % LW;
            if (idx < LITEMS && sy >= 0 && sy < Hs && sx < Ws)
                rawu[k] = *reinterpret_cast<const float4*>(base + ((size_t)sy * Ws + sx) * Cup + part * 4);
            else if constexpr (SEP) rawu[k] = make_float4(0.f, 0.f, 0.f, 0.f);   // (lines the separable form multiplies by a zero weight must be finite)
        }
    };
    auto commit_lo = [&]() {
        // park the low-res region in LDS, then interpolate the patch from it (ATen upsample_bilinear2d,
        // align_corners=True: src = dst * (in-1)/(out-1), weights (1-l, l), noise.py:39,46)
#pragma unroll
        for (int k = 0; k < NITL; ++k) {
            const int idx = tid + k * 256;
            if (idx < LITEMS) *reinterpret_cast<float4*>(&lowres[(idx / PPP) * CKL + (idx % PPP) * 4]) = rawu[k];
        }
        __syncthreads();
        if constexpr (SEP) {
            // task = (group of 6 patch rows, patch column, 4-channel part): ~7 vector instructions per stored piece instead of 27.  All
            // arithmetic through lerp_np (conv_staging.h): this kernel's other waves run bf16 MFMAs (profiles/r05_race.md).
            constexpr int NG = (PH + 5) / 6, TPG = PWL * PPP, TASKS = NG * TPG, ROUNDS = (TASKS + 255) / 256;
            static_assert(3 * (NG - 1) + 3 < LH, "a group's four source lines are inside the parked region");
#pragma unroll 1
            for (int rd = 0; rd < ROUNDS; ++rd) {
                const int T = tid + 256 * rd;
                if (T < TASKS) {
                    const int rg = T / TPG, rest = T - rg * TPG, px = rest / PPP, pt = rest % PPP;
                    const float4 ct = *reinterpret_cast<const float4*>(&colT[4 * px]);
                    const float* l0 = &lowres[(3 * rg) * (LW * CKL) + pt * 4];
                    const int c0 = __float_as_int(ct.x), c1 = __float_as_int(ct.y);
                    float4 h[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 u = *reinterpret_cast<const float4*>(l0 + j * (LW * CKL) + c0);
                        const float4 v = *reinterpret_cast<const float4*>(l0 + j * (LW * CKL) + c1);
                        h[j] = make_float4(lerp_np(ct.z, u.x, ct.w, v.x), lerp_np(ct.z, u.y, ct.w, v.y), lerp_np(ct.z, u.z, ct.w, v.z), lerp_np(ct.z, u.w, ct.w, v.w));
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const int py = 6 * rg + i;
                        if (py < PH) {
                            const float2 rw = *reinterpret_cast<const float2*>(&rowT[4 * py]);
                            const float4 &ha = h[i >> 1], &hb = h[(i >> 1) + 1];
                            store_patch(py, px, pt, make_float4(lerp_np(rw.x, ha.x, rw.y, hb.x), lerp_np(rw.x, ha.y, rw.y, hb.y),
                                                                lerp_np(rw.x, ha.z, rw.y, hb.z), lerp_np(rw.x, ha.w, rw.y, hb.w)));
                        }
                    }
                }
            }
            return;
        }
        // the interpolation's coordinates are the same for every chunk: one table entry per patch row and column (built once,
        // below); a pixel outside the image has zero weights = the conv's zero padding
#pragma unroll 2
        for (int k = 0; k < NIT; ++k) {
            const int idx = tid + k * 256;
            const int part = idx % PPP, pp = idx / PPP;
            const int py = pp / PWL, px = pp % PWL;
            if (idx < ITEMS) {
                const float4 rt = *reinterpret_cast<const float4*>(&rowT[4 * py]), ct = *reinterpret_cast<const float4*>(&colT[4 * px]);
                const float* l0 = &lowres[__float_as_int(rt.x) + part * 4];
                const float* l1 = &lowres[__float_as_int(rt.y) + part * 4];
                const int c0 = __float_as_int(ct.x), c1 = __float_as_int(ct.y);
                store_patch(py, px, part,
                            f4lerp2<BF16 != 0>(*reinterpret_cast<const float4*>(l0 + c0), *reinterpret_cast<const float4*>(l0 + c1),
                                    *reinterpret_cast<const float4*>(l1 + c0), *reinterpret_cast<const float4*>(l1 + c1),
                                    ct.z, ct.w, rt.z, rt.w));
            }
        }
    };
    if constexpr (UP2) {
        if (tid < PH + PW) {                               // (first read behind the barrier inside commit_lo)
            const bool isrow = tid < PH;
            const int pq = isrow ? tid : tid - PH;
            const int g = (isrow ? ty0 : tx0) + pq - 1;
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g >= 0 && g < (isrow ? a.H : a.W)) {
                const float sc = (isrow ? a.rh : a.rw) * (float)g;
                const int i0 = (int)sc;
                const int i1 = i0 + (i0 < (isrow ? Hs : Ws) - 1 ? 1 : 0);
                const float l = fminf(fmaxf(sc - (float)i0, 0.f), 1.f);
                const int lo = isrow ? ylo : xlo, mul = isrow ? LW * CKL : CKL;
                e = make_float4(__int_as_float((i0 - lo) * mul), __int_as_float((i1 - lo) * mul), 1.f - l, l);
                if (SEP && isrow) {                       // {weight of line s0, weight of line s0 + 1}, s0 = ylo + (row >> 1)
                    const int s0 = ylo + (pq >> 1);
                    e = make_float4((i0 == s0 ? 1.f - l : 0.f) + (i1 == s0 ? l : 0.f), (i0 == s0 + 1 ? 1.f - l : 0.f) + (i1 == s0 + 1 ? l : 0.f), 0.f, 0.f);
                }
            }
            *reinterpret_cast<float4*>(&(isrow ? rowT : colT)[4 * pq]) = e;
        }
    }

    if constexpr (UP2) { if (c_begin >= nskip) issue_lo(c_begin); else if (SKIP_PF) issue(c_begin, 0, NIT_S); }
    else if (PREFETCH) issue(c_begin, 0, NIT_S);

    // LDS float offset of this lane's A row for each of its M-blocks (tap (0,0), channel 4*hh)
    int aoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wm * MT + mt) * 32 + li;
        aoff[mt] = ((q / TW) * PW + (q % TW)) * CKP + 4 * hh;
    }

    const float4* bptr[NT];
    f32x16 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cb = cbt * (WN * NT) + wn * NT + nt;
        bptr[nt] = reinterpret_cast<const float4*>(a.wpack) + ((size_t)cb * nchunks_all + c_begin) * KS * 64 * NW + lane;
        const float bias = SPLITK ? 0.f : a.bias[cb * 32 + li];    // accumulators start at the bias
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bias;
    }

    // B fragments (weights, pre-packed per lane, L2-resident) run PFD k-steps ahead: slot ks % PFD is refilled with
    // k-step ks + PFD right after the MFMAs of k-step ks; an f32 k-step is 1000-4000 cycles of MFMA issue.
    float4 bq[PFD][NT][NW];
#pragma unroll
    for (int p = 0; p < PFD; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int w = 0; w < NW; ++w) bq[p][nt][w] = bptr[nt][(p * NW + w) * 64];

    for (int c = c_begin; c < c_end; ++c) {
        if (c > c_begin) __syncthreads();      // every wave is done reading the previous chunk's patch
        if constexpr (UP2) {
            if (c >= nskip) {
                commit_lo();
            } else if constexpr (SKIP_PF) {
                commit(c, 0, NIT_S);
            } else {
#pragma unroll 1
                for (int it0 = 0; it0 < NIT_S; it0 += LBS) { issue(c, it0, LBS); commit(c, it0, LBS); }
            }
        } else if (PREFETCH) {
            commit(c, 0, NIT_S);
        } else {
#pragma unroll 1
            for (int it0 = 0; it0 < NIT; it0 += LB) { issue(c, it0, LB); commit(c, it0, LB); }
        }
        __syncthreads();
        if constexpr (UP2) { if (c + 1 < c_end) { if (c + 1 >= nskip) issue_lo(c + 1); else if (SKIP_PF) issue(c + 1, 0, NIT_S); } }
        else if (PREFETCH && c + 1 < c_end) issue(c + 1, 0, NIT_S);

        const float4* bp[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bp[nt] = bptr[nt] + (size_t)(c - c_begin) * KS * 64 * NW;
        float4 a0[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a0[mt] = *reinterpret_cast<const float4*>(&patch[aoff[mt]]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float4 a1[MT];
            if (ks + 1 < KS) {                                       // A fragments of the next k-step (LDS)
                const int tap1 = (ks + 1) / (CK / KCH), s1 = (ks + 1) % (CK / KCH);
                const int off1 = ((tap1 / 3) * PW + (tap1 % 3)) * CKP + 8 * s1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a1[mt] = *reinterpret_cast<const float4*>(&patch[aoff[mt] + off1]);
            }
            // the sched_barriers keep hipcc from sinking the loads down to their first use
            __builtin_amdgcn_sched_barrier(0);
#define PNP_MFMA_ROUND(comp)                                                                                     \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)          \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt].comp, bq[ks % PFD][nt][0].comp, acc[mt][nt], 0, 0, 0);
            if constexpr (BF16 != 0) {
#pragma unroll
                for (int w = 0; w < NW; ++w)               // (hi, then lo: the same accumulator comes round again MT * NT MFMAs later)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[mt]),
                                                                                  __builtin_bit_cast(bf16x8, bq[ks % PFD][nt][w]), acc[mt][nt], 0, 0, 0);
            } else {
            PNP_MFMA_ROUND(x)
            PNP_MFMA_ROUND(y)
            PNP_MFMA_ROUND(z)
            PNP_MFMA_ROUND(w)
            }
#undef PNP_MFMA_ROUND
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int w = 0; w < NW; ++w) bq[ks % PFD][nt][w] = bp[nt][((ks + PFD) * NW + w) * 64];   // refill the slot just read
            if (ks + 1 < KS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a0[mt] = a1[mt];
            }
        }
    }

    // ---- epilogue: LeakyReLU(0.2) and NHWC store.
    constexpr int BN = WN * NT * 32;
    constexpr int OSTR = BN + 4;
    if constexpr (LDS_EPI) {
        // Through LDS, so the global stores are 16 B per lane over whole pixels (a 4-B-per-lane store tail is
        // store-issue-bound) and the 2x2 max-pooled copy for the next stage can be written from the same tile.
        __syncthreads();                                   // every wave is done reading the last patch
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;   // tile pixel (row-major TH x TW)
                    patch[q * OSTR + (wn * NT + nt) * 32 + li] = fmaxf(acc[mt][nt][r], kLeaky * acc[mt][nt][r]);
                }
        __syncthreads();
        constexpr int V4 = BN / 4;
        const int cbase = cbt * BN;
        if constexpr (BN == 32 && BM <= 256) {
            if (a.last_w != nullptr) {
                // Fused last layer (noise.py:67,130-133,164): 1x1 conv 32 -> 1, + image channel, clamp; this conv's own
                // 32-channel output is never written.  One pixel per thread.
                const int gy = ty0 + tid / TW, gx = tx0 + tid % TW;
                if (tid < BM && gy < a.H && gx < a.W) {
                    float dsum = a.last_b[0];
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4) {
                        const float4 v = *reinterpret_cast<const float4*>(&patch[tid * OSTR + 4 * c4]);
                        const float4 wv = *reinterpret_cast<const float4*>(a.last_w + 4 * c4);
                        dsum += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
                    }
                    const size_t q = ((size_t)n * a.H + gy) * a.W + gx;
                    const float img = a.last_ximg != nullptr ? a.last_ximg[q] : (a.last_z[q].x - a.last_u[q].x);
                    a.last_out[q] = fminf(fmaxf(img + dsum, 0.f), 1.f);
                }
                return;
            }
        }
        if (A16 && (a.act16 & 2)) {                        // dst holds bf16: 8 channels (16 B) per lane, 64 B per pixel
            uint16_t* const d16 = reinterpret_cast<uint16_t*>(a.dst);
#pragma unroll 4
            for (int f = tid; f < BM * (BN / 8); f += 256) {
                const int p = f / (BN / 8), c8 = f % (BN / 8);
                const int gy = ty0 + p / TW, gx = tx0 + p % TW;
                if (gy < a.H && gx < a.W) {
                    const float4 v0 = *reinterpret_cast<const float4*>(&patch[p * OSTR + 8 * c8]);
                    const float4 v1 = *reinterpret_cast<const float4*>(&patch[p * OSTR + 8 * c8 + 4]);
                    uint4 o;
                    o.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0.x, v0.y}, bf16x2));
                    o.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0.z, v0.w}, bf16x2));
                    o.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v1.x, v1.y}, bf16x2));
                    o.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v1.z, v1.w}, bf16x2));
                    *reinterpret_cast<uint4*>(d16 + (((size_t)n * a.H + gy) * a.W + gx) * a.Cout + cbase + 8 * c8) = o;
                }
            }
        } else {
#pragma unroll 4
            for (int f = tid; f < BM * V4; f += 256) {
                const int p = f / V4, c4 = f % V4;
                const int gy = ty0 + p / TW, gx = tx0 + p % TW;
                if (gy < a.H && gx < a.W)
                    *reinterpret_cast<float4*>(a.dst + (((size_t)n * a.H + gy) * a.W + gx) * a.Cout + cbase + 4 * c4) =
                        *reinterpret_cast<const float4*>(&patch[p * OSTR + 4 * c4]);
            }
        }
        if (a.pooled != nullptr) {                         // MaxPool2d(2) of this tile for the next stage
            const int Hp = a.H >> 1, Wp = a.W >> 1;
            for (int f = tid; f < (BM / 4) * V4; f += 256) {
                const int q = f / V4, c4 = f % V4;
                const int qy = q / (TW / 2), qx = q % (TW / 2);
                const int gy = (ty0 >> 1) + qy, gx = (tx0 >> 1) + qx;
                if (gy < Hp && gx < Wp) {
                    const float* o = &patch[((2 * qy) * TW + 2 * qx) * OSTR + 4 * c4];
                    const float4 m = f4max(f4max(*reinterpret_cast<const float4*>(o), *reinterpret_cast<const float4*>(o + OSTR)),
                                           f4max(*reinterpret_cast<const float4*>(o + TW * OSTR), *reinterpret_cast<const float4*>(o + TW * OSTR + OSTR)));
                    *reinterpret_cast<float4*>(a.pooled + (((size_t)n * Hp + gy) * Wp + gx) * a.Cout + cbase + 4 * c4) = m;
                }
            }
        }
    } else {
        // Two full 128-B lines per store instruction; per-slot address part in the scalar offset of a buffer store,
        // per-lane part in one VGPR per N-block.  Split-K workgroups store raw partial sums to plane blockIdx.z, combined either by
        // splitk_reduce_kernel or (a.arrive != nullptr) in this launch by the LAST of the tile's gridDim.z workgroups to arrive.
        // The hand-over uses no fence: a device-scope fence (`__threadfence()` = L2 write-back + invalidate of the workgroup's XCD)
        // per workgroup cost more than the 21 launches it saved (round 4: 0.90 against 0.546 ms per step at 1 x 128 x 128,
        // profiles/r04_ablation.md).  Instead every access to the shared data is itself agent-scope (sc1: stores write through the
        // XCD's L2, loads do not hit in it): partial sums stored sc1 - vmcnt(0) - workgroup barrier - one relaxed agent-scope
        // fetch_add on the tile's arrival counter - the last arrival loads all planes sc1, in plane order (bit-identical to the kernel).
        const size_t plane = (size_t)a.N * a.H * a.W * a.Cout;
        float* const obuf = SPLITK ? a.partial + (size_t)blockIdx.z * plane : a.dst;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)obuf, 0, (int)(plane * sizeof(float)), 0x00020000);
        const bool inlaunch = SPLITK && a.arrive != nullptr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = (cbt * (WN * NT) + wn * NT + nt) * 32 + li;
            const unsigned obase = ((unsigned)(((size_t)n * a.H + ty0) * a.W + tx0 + 4 * hh) * (unsigned)a.Cout + (unsigned)co) * 4u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qc = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2);   // tile pixel index, lane-independent part
                    const int gy = ty0 + qc / TW, gx = tx0 + qc % TW + 4 * hh;
                    const int soff = __builtin_amdgcn_readfirstlane(((qc / TW) * a.W + qc % TW) * a.Cout * 4);
                    const float v = SPLITK ? acc[mt][nt][r] : fmaxf(acc[mt][nt][r], kLeaky * acc[mt][nt][r]);
                    if (gy < a.H && gx < a.W) {
                        if (inlaunch) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, obase, soff, 16);   // sc1
                        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, obase, soff, 0);
                    }
                }
        }
        if constexpr (SPLITK) {
            if (!inlaunch) return;
            __shared__ unsigned arrived;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial sums have been written through
            __syncthreads();
            unsigned* const cnt = a.arrive + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            if (tid == 0) arrived = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (arrived + 1 != gridDim.z) return;           // (uniform)
            if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            const __amdgpu_buffer_rsrc_t prsrc =
                __builtin_amdgcn_make_buffer_rsrc((void*)a.partial, 0, (int)(plane * gridDim.z * sizeof(float)), 0x00020000);
            constexpr int V4 = BN_ / 4;
            const int cbase = cbt * BN_;
            // four tile positions x four planes in flight per thread: sixteen independent 16-byte loads per round trip (one load
            // at a time, 8 x gridDim.z dependent round trips of ~0.25 us, this combine cost 2 us per plane)
            const unsigned nz = gridDim.z, pstride = (unsigned)plane * 4u;
#pragma unroll 1
            for (int f0 = tid; f0 < BM * V4; f0 += 4 * 256) {
                unsigned e[4];
                bool ok[4];
                float4 sum[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = f0 + 256 * j, p = f / V4, c4 = f % V4;
                    const int gy = ty0 + p / TW, gx = tx0 + p % TW;
                    ok[j] = f < BM * V4 && gy < a.H && gx < a.W;
                    e[j] = ok[j] ? (unsigned)(((size_t)n * a.H + gy) * a.W + gx) * (unsigned)a.Cout + (unsigned)(cbase + 4 * c4) : 0u;
                    sum[j] = *reinterpret_cast<const float4*>(a.bias + cbase + 4 * c4);
                }
#pragma unroll 1
                for (unsigned z0 = 0; z0 < nz; z0 += 4) {
                    float4 q[4][4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned z = z0 + k < nz ? z0 + k : nz - 1;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            q[k][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, e[j] * 4u, (int)(z * pstride), 16));
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (z0 + k < nz) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) { sum[j].x += q[k][j].x; sum[j].y += q[k][j].y; sum[j].z += q[k][j].z; sum[j].w += q[k][j].w; }
                        }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!ok[j]) continue;
                    float4 v = sum[j];
                    v.x = fmaxf(v.x, kLeaky * v.x); v.y = fmaxf(v.y, kLeaky * v.y);
                    v.z = fmaxf(v.z, kLeaky * v.z); v.w = fmaxf(v.w, kLeaky * v.w);
                    *reinterpret_cast<float4*>(a.dst + e[j]) = v;
                }
            }
        }
    }
}

// out = LeakyReLU(bias + sum_z partial[z]) in a fixed order: the split-K combine.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                            float* __restrict__ dst, const float* __restrict__ tact,
                                                            size_t plane4, int splitk, int cout4, size_t slice4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane4; i += (size_t)gridDim.x * 256) {
        if (tact != nullptr && tact[i / slice4] > 0.5f) continue;
        float4 s = reinterpret_cast<const float4*>(bias)[i % cout4];
        for (int z0 = 0; z0 < splitk; z0 += 8) {          // eight planes in flight, added in plane order
            float4 p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = reinterpret_cast<const float4*>(partial)[(size_t)(z0 + k < splitk ? z0 + k : splitk - 1) * plane4 + i];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (z0 + k < splitk) { s.x += p[k].x; s.y += p[k].y; s.z += p[k].z; s.w += p[k].w; }
        }
        s.x = fmaxf(s.x, kLeaky * s.x); s.y = fmaxf(s.y, kLeaky * s.y);
        s.z = fmaxf(s.z, kLeaky * s.z); s.w = fmaxf(s.w, kLeaky * s.w);
        reinterpret_cast<float4*>(dst)[i] = s;
    }
}

template <int TW, int MT, int NT, int WM, int WN, int CK, int SRC, bool SPLITK, int BF16>
static hipError_t launch_inst(const ConvArgs& a, const ConvPlan& p, hipStream_t s) {
    // registers: 16*MT*NT accumulators + operands + staging: 32 acc -> 3 waves per SIMD, 128 -> 2, 256 -> 1
    // (the UPCAT variant also holds the low-res source region in LDS and a prefetched skip chunk in registers: two
    // workgroups per CU; three with the bf16 patch measured 8 % slower on up4.conv-0 in round 2, when it spilled 54 registers - on
    // bf16 activations and a three-slot weight ring it fits 157 and is 13 % FASTER: WPS16)
    constexpr int WPS = (MT * NT <= 2 && SRC != SRC_UPCAT) ? 3 : (MT * NT <= 8 ? 2 : 1);
    constexpr int WPS16 = (MT * NT <= 2 && BF16 == 2) ? 3 : WPS;      // bf16 activations, two-term weights: the UPCAT variant too
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * a.N), (unsigned)(a.Cout / p.bn), (unsigned)p.splitk);
    constexpr bool CAN16 = BF16 != 0 && !SPLITK && MT == 2 && NT == 1 && WM == 4 && CK == 32 && SRC != SRC_POOL;
    if constexpr (CAN16) {
        if (a.act16 & 1) {
            hipLaunchKernelGGL((conv3x3_mfma_kernel<TW, MT, NT, WM, WN, CK, SRC, WPS16, SPLITK, BF16, true>), grid, dim3(256), 0, s, a);
            return hipGetLastError();
        }
    }
    if (a.act16 != 0) return hipErrorInvalidValue;        // bf16 activations reach only the plan that has them
    hipLaunchKernelGGL((conv3x3_mfma_kernel<TW, MT, NT, WM, WN, CK, SRC, WPS, SPLITK, BF16, false>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int TW, int MT, int NT, int WM, int WN, int CK, bool SPLITK, int BF16>
static hipError_t launch_cfg(const ConvArgs& a, const ConvPlan& p, int src_mode, hipStream_t s) {
    switch (src_mode) {
        case SRC_PLAIN: return launch_inst<TW, MT, NT, WM, WN, CK, SRC_PLAIN, SPLITK, BF16>(a, p, s);
        case SRC_POOL:  return launch_inst<TW, MT, NT, WM, WN, CK, SRC_POOL, SPLITK, BF16>(a, p, s);
        case SRC_UPCAT: return launch_inst<TW, MT, NT, WM, WN, CK, SRC_UPCAT, SPLITK, BF16>(a, p, s);
        default: return hipErrorInvalidValue;
    }
}

template <int TW, int BF16>
static hipError_t launch_tw(const ConvArgs& a, const ConvPlan& p, int src_mode, hipStream_t s) {
    if constexpr (BF16 == 0) {
        if (p.mt == 1 && p.wn == 4) return launch_cfg<TW, 1, 1, 1, 4, 16, true, 0>(a, p, src_mode, s);
        if (p.mt == 1 && p.wn == 2) return launch_cfg<TW, 1, 1, 2, 2, 16, true, 0>(a, p, src_mode, s);
    }
    if (p.mt == 2 && p.wn == 4) return launch_cfg<TW, 2, 1, 1, 4, 32, true, BF16>(a, p, src_mode, s);    // small problems
    if (p.mt == 2 && p.wn == 2) return launch_cfg<TW, 2, 1, 2, 2, 32, true, BF16>(a, p, src_mode, s);
    if constexpr (BF16 == 0) {
        if (p.nt == 1 && p.mt == 1) return launch_cfg<TW, 1, 1, 4, 1, 32, false, 0>(a, p, src_mode, s);      // small problems, Cout = 32
    }
    if (p.nt == 1) return launch_cfg<TW, 2, 1, 4, 1, 32, false, BF16>(a, p, src_mode, s);
    if (p.nt == 2 && p.wn == 1) return launch_cfg<TW, 4, 2, 4, 1, 16, false, BF16>(a, p, src_mode, s);
    if (p.nt == 2) return launch_cfg<TW, 4, 2, 2, 2, 32, false, BF16>(a, p, src_mode, s);
    if constexpr (BF16 != 0) return hipErrorInvalidValue;  // the bf16 plan never picks the 256-accumulator tile
    else return launch_cfg<TW, 4, 4, 2, 2, 32, false, 0>(a, p, src_mode, s);
}

hipError_t launch_conv3x3(const ConvArgs& a0, const ConvPlan& p, int src_mode, hipStream_t s) {
    if (a0.Cin % 32 != 0 || a0.Cout % 32 != 0 || (a0.Cout > 64 && a0.Cout % 128 != 0)) return hipErrorInvalidValue;
    if (!conv3x3_tensor_fits(a0.N, a0.H, a0.W, a0.Cin, a0.Cout)) return hipErrorInvalidValue;
    if (p.ws) return launch_conv3x3_bf16ws(a0, p, src_mode, s);
    ConvArgs a = a0;
    a.tilesX = p.tiles_x;
    a.tilesY = p.tiles_y;
    const bool split = p.mt <= 2 && p.wn >= 2;          // the small-problem configs always go through the workspace
    if (split && a.partial == nullptr) return hipErrorInvalidValue;
    if (!split || (size_t)p.tiles_x * p.tiles_y * a.N * (a.Cout / p.bn) > 4096) a.arrive = nullptr;   // (the counters the engine allocates)
    hipError_t e;
    if (a.bf16 == 2) {                                    // bf16 operands, two-term weights (the mode's default)
        if (p.tw == 32) e = launch_tw<32, 2>(a, p, src_mode, s);
        else if (p.tw == 16) e = launch_tw<16, 2>(a, p, src_mode, s);
        else e = launch_tw<8, 2>(a, p, src_mode, s);
    } else if (a.bf16 == 1) {                             // PNP_BF16_W1: one-term weights (round-3 arithmetic, ablation)
        if (p.tw == 32) e = launch_tw<32, 1>(a, p, src_mode, s);
        else if (p.tw == 16) e = launch_tw<16, 1>(a, p, src_mode, s);
        else e = launch_tw<8, 1>(a, p, src_mode, s);
    } else {
        if (p.tw == 32) e = launch_tw<32, 0>(a, p, src_mode, s);
        else if (p.tw == 16) e = launch_tw<16, 0>(a, p, src_mode, s);
        else e = launch_tw<8, 0>(a, p, src_mode, s);
    }
    if (e != hipSuccess || !split || a.arrive != nullptr) return e;
    const size_t plane4 = (size_t)a.N * a.H * a.W * a.Cout / 4;
    unsigned blocks = (unsigned)((plane4 + 255) / 256);
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, a.partial, a.bias, a.dst, a.tact, plane4,
                       p.splitk, a.Cout / 4, (size_t)a.H * a.W * a.Cout / 4);
    return hipGetLastError();
}

// The direct kernels address a whole activation tensor through ONE buffer descriptor (32-bit num_records, 32-bit byte offsets):
// an OUTPUT tensor of 2 GiB or more would wrap them - refused here and, with a message, by pnp_create.  (The F(4x4) kernels use a
// descriptor per slice.)
bool conv3x3_tensor_fits(int N, int H, int W, int Cin, int Cout) {
    (void)Cin;                                             // (sources: per-slice descriptors / 64-bit pointers)
    return (size_t)N * H * W * (size_t)Cout * 4 < ((size_t)1 << 31);
}

// The direct kernel writes the pooled copy only from its LDS epilogue (Cout = 32 plan on a large problem).
bool conv3x3_pooled_output_ok(const ConvPlan& p) {   // also: can fuse the last layer
    return p.splitk == 1 && p.mt <= 2 && p.nt == 1 && p.wm == 4;
}

size_t conv3x3_partial_floats(const ConvPlan& p, int N, int H, int W, int Cout) {
    return (p.mt <= 2 && p.wn >= 2) ? (size_t)p.splitk * N * H * W * Cout : 0;
}

Tuning tuning_from_env() {
    Tuning t;
    if (const char* v = getenv("PNP_WINO_MIN_CIN")) t.wino_min_cin = atoi(v);
    if (const char* v = getenv("PNP_WINO_MIN_BLOCKS")) t.wino_min_blocks = atol(v);
    t.wino_big = getenv("PNP_WINO_BIG_GROUPS") != nullptr;
    t.wino_small = getenv("PNP_WINO_SMALL_GROUPS") != nullptr;
    t.no_wino = getenv("PNP_NO_WINOGRAD") != nullptr;
    if (const char* v = getenv("PNP_WINO_F4_MIN_CIN")) t.f4_min_cin = atoi(v);
    t.no_f4 = getenv("PNP_NO_WINO_F4") != nullptr;
    t.no_f4_phased = getenv("PNP_NO_WINO_F4_PHASED") != nullptr;
    t.no_f4_fused_last = getenv("PNP_NO_F4_FUSED_LAST") != nullptr;
    t.no_f4_fused_first = getenv("PNP_NO_F4_FUSED_FIRST") != nullptr;
    if (const char* v = getenv("PNP_WINO_F4_MT16")) t.f4_mt16 = atoi(v);
    if (const char* v = getenv("PNP_WINO_F4_ORDER")) t.f4_order = atoi(v) != 0;
    t.bf16_f32_acts = getenv("PNP_BF16_F32_ACTS") != nullptr;
    t.bf16_no_ws = getenv("PNP_BF16_NO_WS") != nullptr;
    t.bf16_w1 = getenv("PNP_BF16_W1") != nullptr;
    t.bf16_no_holdhi = getenv("PNP_BF16_NO_HOLDHI") != nullptr;
    t.fft_xcd = getenv("PNP_FFT_XCD") != nullptr && atoi(getenv("PNP_FFT_XCD")) != 0;
    if (const char* v = getenv("PNP_SPLITK_INLAUNCH")) t.splitk_inlaunch = atoi(v);
    if (const char* v = getenv("PNP_SLICE128_MIN_N")) t.slice128_min_n = atoi(v);
    return t;
}

hipError_t raise_lds_cap(const void* fn, int bytes, DeviceOnce& once) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (once.mask.load(std::memory_order_relaxed) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    once.mask.fetch_or(bit, std::memory_order_relaxed);
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// First conv: Cin = 2 (image, sigma plane), K = 18: too thin for MFMA, direct f32 FMA.  A workgroup owns an 8 x 32
// pixel tile: the image channel d = ximg (or Re(z - u)) and the in-image mask of the halo are staged in LDS once;
// 8 lanes share a pixel (4 output channels each, weights in registers), walk down the tile's 8 rows with a sliding
// 3-row window (6 LDS reads per pixel instead of 18 global loads) and store 8 pixels x 128 B contiguous per wave.
// The sigma plane is never materialised; like every conv input it is ZERO in the padding halo (noise.py:161-162 cat,
// then conv with padding=1), which is what the mask plane encodes.
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ ximg, const float2* __restrict__ z,
                                                         const float2* __restrict__ u, const float* __restrict__ sigma,
                                                         const float* __restrict__ tact, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ dst,
                                                         int N, int H, int W, int tilesX, int tilesY, int TYG, int dst16) {
    constexpr int TH = 8, TW = 32, PW = TW + 2, PH = TH + 2;
    // TYG (8 on chip-filling problems, fewer on small ones): tiles (of 8 rows) a workgroup walks down: the 76 weight / bias
    __shared__ float dt[2][PH][PW + 1];                        // registers of a thread are loaded once per 64 rows, and the next
    __shared__ float mt[2][PH][PW + 1];                        // tile's halo is staged (double buffer) under this tile's FMAs
    const int groupsY = (tilesY + TYG - 1) / TYG;
    int bt = blockIdx.x;
    const int tx0 = (bt % tilesX) * TW;
    bt /= tilesX;
    const int tyg = bt % groupsY;
    const int n = bt / groupsY;
    if (tact != nullptr && tact[n] > 0.5f) return;
    auto stage = [&](int buf, int ty0) {
        for (int i = threadIdx.x; i < PH * PW; i += 256) {
            const int py = i / PW, px = i % PW;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            float d = 0.f, m = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const size_t q = ((size_t)n * H + gy) * W + gx;
                d = ximg != nullptr ? ximg[q] : (z[q].x - u[q].x);
                m = 1.f;
            }
            dt[buf][py][px] = d;
            mt[buf][py][px] = m;
        }
    };
    const int t_first = tyg * TYG, t_last = min(t_first + TYG, tilesY);
    stage(0, t_first * TH);
    const int cg = threadIdx.x & 7, col = threadIdx.x >> 3;
    float wd[4][9], ws[4][9], br[4], brs[4];
    const float sg = sigma[n];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        br[j] = bias[cg * 4 + j];
        brs[j] = br[j];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            wd[j][k] = w[(cg * 4 + j) * 18 + k];
            ws[j][k] = w[(cg * 4 + j) * 18 + 9 + k] * sg;      // sigma folded into the second channel's taps
            brs[j] += ws[j][k];                                // interior tiles: the whole sigma plane folded into the bias (below)
        }
    }
    const int gx = tx0 + col;
    for (int t = t_first; t < t_last; ++t) {
        const int buf = (t - t_first) & 1, ty0 = t * TH;
        const bool interior = tx0 > 0 && tx0 + TW < W && ty0 > 0 && ty0 + TH < H;   // (workgroup-uniform)
        __syncthreads();                                       // this tile's halo is staged; the other buffer is free again
        if (t + 1 < t_last) stage(buf ^ 1, ty0 + TH);
        float d[3][3], m[3][3];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { d[r + 1][c] = dt[buf][r][col + c]; m[r + 1][c] = mt[buf][r][col + c]; }
#pragma unroll
        for (int row = 0; row < TH; ++row) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {                      // slide the window one row down
                d[0][c] = d[1][c]; d[1][c] = d[2][c]; d[2][c] = dt[buf][row + 2][col + c];
                m[0][c] = m[1][c]; m[1][c] = m[2][c]; m[2][c] = mt[buf][row + 2][col + c];
            }
            float acc[4];
            if (interior) {
                // every tap of every pixel of this tile lies inside the image: the sigma channel (a constant plane) contributes
                // sum_k w_sigma[k] * sigma to every output - 18 FMAs per output instead of 36.  The kernel is bound by exactly that
                // vector work (36 FMAs x 32 channels x 4.2 Mpx = 0.123 ms at the chip's FMA rate against 0.119 measured at
                // 16 x 512 x 512), the F(4x4) path's fused first layer folds the same way (Fusions)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = brs[j];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = fmaf(wd[j][ky * 3 + kx], d[ky][kx], acc[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = br[j];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[j] = fmaf(wd[j][ky * 3 + kx], d[ky][kx], acc[j]);
                            acc[j] = fmaf(ws[j][ky * 3 + kx], m[ky][kx], acc[j]);
                        }
            }
            const int gy = ty0 + row;
            if (gy < H && gx < W) {
                float4 o;
                o.x = fmaxf(acc[0], kLeaky * acc[0]);
                o.y = fmaxf(acc[1], kLeaky * acc[1]);
                o.z = fmaxf(acc[2], kLeaky * acc[2]);
                o.w = fmaxf(acc[3], kLeaky * acc[3]);
                if (dst16) {                                  // bf16 activations (see conv3x3_mfma_kernel, A16)
                    uint2 o16;
                    o16.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){o.x, o.y}, bf16x2));
                    o16.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){o.z, o.w}, bf16x2));
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(dst) + (((size_t)n * H + gy) * W + gx) * 32 + cg * 4) = o16;
                } else {
                    *reinterpret_cast<float4*>(dst + (((size_t)n * H + gy) * W + gx) * 32 + cg * 4) = o;
                }
            }
        }
    }
}

hipError_t launch_conv_first(const float* ximg, const float2* z, const float2* u, const float* sigma,
                             const float* tact, const float* w, const float* bias, float* dst, int N, int H, int W,
                             hipStream_t s, bool dst_bf16) {
    const int tilesX = (W + 31) / 32, tilesY = (H + 7) / 8;
    // a workgroup walks TYG tiles down a column of tiles (weights loaded once per walk); small problems walk fewer tiles so that
    // the grid still covers the chip (one 128 x 128 slice: 8 workgroups of 8 tiles took 27 us, 64 of one tile take a third)
    int tyg = 8;
    while (tyg > 1 && (long)tilesX * ((tilesY + tyg - 1) / tyg) * N < 512) tyg >>= 1;
    const int groupsY = (tilesY + tyg - 1) / tyg;
    hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)(tilesX * groupsY * N)), dim3(256), 0, s, ximg, z, u, sigma, tact, w,
                       bias, dst, N, H, W, tilesX, tilesY, tyg, dst_bf16 ? 1 : 0);
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
