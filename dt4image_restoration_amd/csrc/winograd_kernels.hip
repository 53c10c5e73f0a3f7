// Winograd F(2x2, 3x3) convolution for the K-heavy denoiser layers on gfx950 (MI355X).
//
// Same operator as conv3x3_mfma_kernel (conv3x3 s1 p1 + bias + LeakyReLU(0.2), /root/reference/evaluation/noise.py:75-98,
// with the stage's MaxPool / bilinear-upsample+concat input transform applied while staging), computed with the
// minimal-filtering identity  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  per 2x2 output tile: 16 multiplies per tile and
// (cin, cout) pair instead of 36, i.e. 2.25x fewer MFMAs, in exact f32 MFMA arithmetic (coefficients 1, 1/2, 1/4).
// The f32 matrix pipe is the bound of the direct kernel (124 TF/s = 79 % of peak), so fewer MFMAs is the lever left.
//
// Workgroup = 2 frequency halves x WM groups of 32 tiles x WN groups of 32 output channels (4 or 8 waves; two waves
// per SIMD, either of one 8-wave workgroup or of two resident 4-wave workgroups).  Per CK-channel chunk:
//   1. stage the (TH+2) x (TW+2) halo patch in LDS (the loads of chunk c+1 are in flight during chunk c's MFMAs)
//   2. input transform: thread (tile, 4 channels, frequency half) reads its 4x4 window from the patch and writes the
//      8 frequency planes V[xi][tile][channel] its own wave pair consumes
//   3. 16 independent GEMMs, 8 per wave: acc[xi] += V[xi] (A fragment, ds_read_b128) x U[xi] (B fragment, transformed
//      weights streamed from L2 in pre-packed per-lane order, 4 pairs ahead)
// The output transform Y = A^T M A is lane-local along columns and exchanges 2 floats per element with the partner wave
// (other frequency half) through LDS along rows; bias, LeakyReLU, the pooled copy and the NHWC stores follow from LDS.
#include "pnp_internal.h"
#include "conv_staging.h"
#include <cstdlib>

namespace pnp {

// Diagnostic build only (make STAMPS=1 -> libpnpadmm_stamps.so, read by tools/wino_stamps.py): wave 0 of 1024 mid-grid
// workgroups per launch records s_memtime at its phase boundaries.
#ifdef PNP_STAMPS
#define PNP_STAMP_SLOTS 32
#define PNP_STAMP_WGS 1024
#define PNP_STAMP_N 112
__device__ unsigned long long g_wino_stamps[PNP_STAMP_SLOTS * PNP_STAMP_WGS * PNP_STAMP_N];
static int g_stamp_slot = 0;
#define STAMP() do { if (ns < PNP_STAMP_N) st[ns] = __builtin_amdgcn_s_memtime(); ++ns; } while (0)
#else
#define STAMP() do { } while (0)
#endif

// ---- host: U = G g G^T, packed [cout/32][xi half][chunk][ks][k 8][lane][4]: one contiguous B stream per wave ---------
size_t winograd_pack_floats(int cin, int cout) { return (size_t)(cout / 32) * ((size_t)(cin / 8) * 16 * 256 + 2 * 4 * 256); }

void pack_winograd_weights(const float* oihw, int cin, int cout, int ck, float* dst) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> U((size_t)cout * cin * 16);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = oihw + ((size_t)co * cin + ci) * 9;
            double t[4][3];
            for (int i = 0; i < 4; ++i)
                for (int kx = 0; kx < 3; ++kx) t[i][kx] = G[i][0] * g[kx] + G[i][1] * g[3 + kx] + G[i][2] * g[6 + kx];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    U[((size_t)co * cin + ci) * 16 + 4 * i + j] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
    size_t o = 0;
    for (int cb = 0; cb < cout / 32; ++cb)
        for (int half = 0; half < 2; ++half) {             // frequencies xi = 8*half + k: rows {0,1} / {2,3} of the 4x4
            for (int ch = 0; ch < cin / ck; ++ch)
                for (int ks = 0; ks < ck / 8; ++ks)
                    for (int k = 0; k < 8; ++k)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 4; ++j) {
                                const int co = 32 * cb + (l & 31);
                                const int ci = ck * ch + 8 * ks + 4 * (l >> 5) + j;
                                dst[o++] = U[((size_t)co * cin + ci) * 16 + 8 * half + k];
                            }
            for (int i = 0; i < 4 * 256; ++i) dst[o++] = 0.f;   // prefetch tail of this stream
        }
}


// Eligibility + tile plan.  The transforms and the 16-accumulator epilogue are per-item overhead that long K amortises
// best (1.65x over the direct kernel at Cin >= 128), but even the K = 288 layers gain ~9 %; small problems (too few
// workgroups) stay on the direct kernel with split-K.
WinoPlan winograd_plan(int N, int H, int W, int Cin, int Cout, int src_mode, const Tuning& t) {
    WinoPlan p{};
    p.use = false;
    p.algo = 1;
    if (t.no_wino || Cin < t.wino_min_cin || Cin % 32 || Cout % 32) return p;
    // F(4x4,3x3) (winograd4_kernels.hip): 1.78x fewer MFMAs again where the matrix pipe is the bound - Cin >= 64 (measured: +10 % at 64, +21..30 % at 128..768),
    // 64-channel output blocks, 32 tiles of 4x4 pixels per workgroup (16 x 32 or 32 x 16 pixels), so the image must be at
    // least that large in the tile's long direction - or 16 x 16 exactly, where two slices are stacked into one workgroup.
    if (!t.no_f4 && Cin >= t.f4_min_cin && (Cout % 64 == 0 || Cout == 32) && Cin % 16 == 0 && (src_mode == SRC_PLAIN || src_mode == SRC_UPCAT) &&
        W >= 16 && H >= 16 && (W >= 32 || H >= 32 || (W == 16 && H == 16 && src_mode == SRC_PLAIN && N >= 2)) &&
        W % 4 == 0 &&   // (its epilogue stores 4-wide tiles whole in x, wino4_epilogue: other widths - 34, 18 - take F(2x2) / direct)
        (src_mode != SRC_UPCAT || upsample_lines_regular(H))) {   // (its interpolation: two compile-time source lines per patch row)
        WinoPlan f{};
        f.algo = 4;
        // tile order: an XCD walks a contiguous range of spatial tiles (halo pixels shared through its L2: -1...2.7 % on the 256 / 128
        // pixel levels) except on 16 x 16 images, whose weight-stream-bound layers measured 2.5 % slower that way
        f.order = (t.f4_order && (long)H * W >= 32 * 32) ? 1 : 0;
        f.tw = W >= 32 ? 32 : 16;
        f.th = f.tw == 32 ? 16 : 32;
        f.bn = 64; f.wm = 1; f.wn = 2; f.ck = 16;
        if (Cout == 32) { f.bn = 32; f.wn = 1; f.ck = 8; }  // 4-wave workgroups (one 32-channel group), two per CU
        f.tiles_x = (W + f.tw - 1) / f.tw;
        f.tiles_y = (H + f.th - 1) / f.th;
        f.stack = (W == 16 && H == 16) ? 1 : 0;             // 16 x 16 images: two slices per 32-tile workgroup
        // (stacked 16 x 16 slices have no 32-channel variant: such a layer skips F(4x4) only and is planned as F(2x2) below)
        if (!(f.stack && Cout == 32)) {
            f.mt = 32;
            // 16-tile M-blocks (two independent workgroups per CU): measured faster than the lockstep schedules on the upsample + concat
            // layers (-2..3.5 %) and on the short-K plain layers (Cin <= 64: -3..5 %), equal at 128, slower from 256 channels on
            const bool mt16 = t.f4_mt16 == 0 ? (src_mode == SRC_UPCAT || Cin <= 64) : (t.f4_mt16 == 1 ? false : (t.f4_mt16 == 2 ? src_mode == SRC_UPCAT : true));
            if (f.bn == 64 && mt16 && (f.tw == 32 ? H >= 8 : H >= 16)) {
                f.mt = 16;                                      // 8 x 32 or 16 x 16 pixels per workgroup
                f.th = f.tw == 32 ? 8 : 16;
                f.tiles_y = (H + f.th - 1) / f.th;
                f.stack = 0;
            }
            f.phased = (f.mt == 32 && f.bn == 64 && src_mode == SRC_PLAIN && !t.no_f4_phased) ? 1 : 0;
            const long blocks = (long)f.tiles_x * f.tiles_y * (f.stack ? (N + 1) / 2 : N) * (Cout / f.bn);
            f.use = blocks >= t.wino_min_blocks;
            if (f.use) return f;
        }
    }
    p.tw = W >= 32 ? 32 : (W >= 16 ? 16 : 8);
    // Cout >= 128: one 8-wave workgroup per CU (32 tiles x 128 channels, 32-channel chunks).  Cout = 64 / 32: 4-wave
    // workgroups (32 tiles x 64 channels / 64 tiles x 32 channels) small enough for TWO per CU, so one workgroup's
    // barriers, transforms, prologue loads and epilogue run under the other's MFMAs - their chunks are short (64 / 32
    // MFMAs per wave), which made per-chunk overhead the bound with a single resident workgroup.
    const bool big = t.wino_big;                                        // experiments: the 8-wave plans everywhere
    if (Cout % 128 == 0 && !t.wino_small) { p.wm = 1; p.wn = 4; p.ck = 32; }
    else if (Cout % 64 == 0) { if (big) { p.wm = 2; p.wn = 2; } else { p.wm = 1; p.wn = 2; } p.ck = 16; }
    else { if (big) { p.wm = 4; p.wn = 1; } else { p.wm = 2; p.wn = 1; } p.ck = 8; }
    const int tc = p.tw / 2, tr = 32 / tc;
    p.th = p.wm * 2 * tr;
    p.bn = p.wn * 32;
    p.tiles_x = (W + p.tw - 1) / p.tw;
    p.tiles_y = (H + p.th - 1) / p.th;
    const long blocks = (long)p.tiles_x * p.tiles_y * N * (Cout / p.bn);
    p.use = blocks >= t.wino_min_blocks;
    return p;
}

template <int TW, int WM, int WN, int CK, int SRC>
__global__ __launch_bounds__(WM * WN * 128, 2) void conv3x3_winograd_kernel(const ConvArgs a) {
    constexpr int GW = WM * WN;                // (tile group, channel group) pairs; each is served by TWO waves
    constexpr int NT_ = GW * 128;              // threads: 2 frequency halves x WM tile groups x WN channel groups waves
    constexpr int CKP = CK + 4;
    constexpr int PPP = CK / 4;
    constexpr int TC = TW / 2, TR = 32 / TC;   // tiles per row / rows of tiles in one 32-tile M-block
    constexpr int TH = WM * 2 * TR;
    constexpr int NTILES = WM * 32;
    constexpr int PWL = TW + 2, PH = TH + 2, PW = PWL;
    constexpr int ITEMS = PH * PWL * PPP;
    constexpr int NIT = (ITEMS + NT_ - 1) / NT_;
    // PLAIN chunks: the raw loads of the next chunk are held in registers across the MFMA phase.  UPCAT chunks that come
    // from the bilinear x2 upsample (4 source pixels per patch pixel) do not fit in registers that way and staged
    // synchronously they cost 30 % of the kernel.  Instead the LOW-RES source region of the tile, (TH/2+3) x (TW/2+3)
    // pixels - 13x fewer loads - is prefetched like a PLAIN chunk, parked in LDS, and the patch is interpolated
    // LDS -> LDS.  POOL sources (only used when no pooled copy exists) stage synchronously in batches.
    constexpr bool UP2 = SRC == SRC_UPCAT;
    constexpr bool PREFETCH = SRC == SRC_PLAIN || UP2;
    constexpr int LH = TH / 2 + 3, LW = TW / 2 + 3;        // low-res region bound (rows, cols)
    constexpr int LITEMS = LH * LW * PPP;
    constexpr int NITL = (LITEMS + NT_ - 1) / NT_;
    constexpr int NRAW = PREFETCH ? (UP2 && NITL > NIT ? NITL : NIT) : (NIT < 3 ? NIT : 3);
    constexpr int LB = PREFETCH ? NIT : (NIT < 3 ? NIT : 3);
    constexpr int KSC = CK / 8;                // k-steps per chunk
    constexpr int PAIRS = 8 * KSC;             // (k-step, frequency) pairs per chunk and wave, 4 MFMAs each
    constexpr int PF = 4;                      // B fragments in flight
    static_assert((GW == 4 || GW == 2) && NTILES * PPP * 2 == NT_, "one (tile, 4-channel, frequency-half) transform item per thread");

#ifdef PNP_STAMPS
    unsigned long long st[PNP_STAMP_N];
    int ns = 0;
    STAMP();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const patch = smem;                             // [PH][PW][CKP]
    float* const V = smem + PH * PW * CKP;                 // [16][NTILES][CKP]
    float* const lowres = V + 16 * NTILES * CKP;           // UPCAT: [LH][LW][CKP] low-res source region

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int xh = wid / GW;                               // frequency half: V rows {2xh, 2xh+1}, xi = 8*xh + k
    const int wm = (wid % GW) / WN, wn = (wid % GW) % WN;
    const int hh = lane >> 5, li = lane & 31;

    int bt = blockIdx.x;
    const int tx0 = (bt % a.tilesX) * TW;
    bt /= a.tilesX;
    const int ty0 = (bt % a.tilesY) * TH;
    const int n = bt / a.tilesY;
    const int cb = blockIdx.y * WN + wn;                   // this wave's 32-channel block
    if (a.tact != nullptr && a.tact[n] > 0.5f) return;
    const int nchunks = a.Cin / CK;

    // UPCAT geometry of this tile: low-res rows/cols [ylo, ylo+LH) x [xlo, xlo+LW) cover every source pixel of the patch
    const int Hs = a.H >> 1, Ws = a.W >> 1;
    const int ylo = UP2 ? (int)(a.rh * (float)(ty0 > 0 ? ty0 - 1 : 0)) : 0;
    const int xlo = UP2 ? (int)(a.rw * (float)(tx0 > 0 ? tx0 - 1 : 0)) : 0;
    const int nskip = UP2 ? a.Cskip / CK : 0;              // leading chunks that come straight from the skip tensor

    float4 raw[NRAW][(SRC == SRC_POOL) ? 4 : 1];
    auto issue = [&](int c, int it0, int cnt) {
        if (UP2 && c >= nskip) {                           // low-res region of an upsampled chunk
            const int Cup = a.Cin - a.Cskip;
            const float* base = a.src1 + (size_t)n * Hs * Ws * Cup + (c * CK - a.Cskip);
#pragma unroll
            for (int k = 0; k < NITL; ++k) {
                const int idx = tid + k * NT_;
                const int part = idx % PPP, pp = idx / PPP;
                const int sy = ylo + pp / LW, sx = xlo + pp % LW;
                if (idx < LITEMS && sy < Hs && sx < Ws)
                    raw[k][0] = *reinterpret_cast<const float4*>(base + ((size_t)sy * Ws + sx) * Cup + part * 4);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            const int idx = tid + (it0 + k) * NT_;
            const int part = idx % PPP, pp = idx / PPP;
            const int py = pp / PWL, px = pp % PWL;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            if (idx < ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                if constexpr (SRC == SRC_POOL) {
                    RawPiece<SRC_POOL> r;
                    issue_piece<SRC_POOL>(a, n, gy, gx, c * CK + part * 4, r);
                    raw[k][0] = r.v[0]; raw[k][1] = r.v[1]; raw[k][2] = r.v[2]; raw[k][3] = r.v[3];
                } else {                                   // PLAIN, or a skip chunk of UPCAT
                    const int cs = UP2 ? a.Cskip : a.Cin;
                    raw[k][0] = *reinterpret_cast<const float4*>(a.src0 + (((size_t)n * a.H + gy) * a.W + gx) * cs + c * CK + part * 4);
                }
            }
        }
    };
    auto commit = [&](int c, int it0, int cnt) {
        if (UP2 && c >= nskip) {
            // park the low-res region in LDS, then interpolate the patch from it (ATen upsample_bilinear2d,
            // align_corners=True: src = dst * (in-1)/(out-1), weights (1-l, l), noise.py:39,46)
#pragma unroll
            for (int k = 0; k < NITL; ++k) {
                const int idx = tid + k * NT_;
                if (idx < LITEMS) *reinterpret_cast<float4*>(&lowres[(idx / PPP) * CKP + (idx % PPP) * 4]) = raw[k][0];
            }
            __syncthreads();
#pragma unroll                                             // (kept unrolled: as a real loop it would not spill 6-20 registers, but
            for (int k = 0; k < NIT; ++k) {                // measured 4 % slower on up4.conv-0: 1.502 vs 1.445 ms)
                const int idx = tid + k * NT_;
                const int part = idx % PPP, pp = idx / PPP;
                const int py = pp / PWL, px = pp % PWL;
                const int gy = ty0 + py - 1, gx = tx0 + px - 1;
                if (idx < ITEMS) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                        const float sy = a.rh * (float)gy, sx = a.rw * (float)gx;
                        const int y0 = (int)sy, x0 = (int)sx;
                        const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
                        const float ly = fminf(fmaxf(sy - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
                        const float* r0 = &lowres[((y0 - ylo) * LW - xlo) * CKP + part * 4];
                        const float* r1 = &lowres[((y1 - ylo) * LW - xlo) * CKP + part * 4];
                        v = f4lerp2(*reinterpret_cast<const float4*>(r0 + x0 * CKP), *reinterpret_cast<const float4*>(r0 + x1 * CKP),
                                    *reinterpret_cast<const float4*>(r1 + x0 * CKP), *reinterpret_cast<const float4*>(r1 + x1 * CKP),
                                    1.f - lx, lx, 1.f - ly, ly);
                    }
                    *reinterpret_cast<float4*>(&patch[(py * PW + px) * CKP + part * 4]) = v;
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            const int idx = tid + (it0 + k) * NT_;
            const int part = idx % PPP, pp = idx / PPP;
            const int py = pp / PWL, px = pp % PWL;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            if (idx < ITEMS) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);      // zero outside the image = the conv's zero padding
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                    if constexpr (SRC == SRC_POOL) v = f4max(f4max(raw[k][0], raw[k][1]), f4max(raw[k][2], raw[k][3]));
                    else v = raw[k][0];
                }
                *reinterpret_cast<float4*>(&patch[(py * PW + px) * CKP + part * 4]) = v;
            }
        }
    };
    if (PREFETCH) issue(0, 0, NIT);
    STAMP();                                               // [1] first loads issued

    // this thread's transform item: tile tq of the workgroup's tile grid (TC wide), channels [4*tg, 4*tg+4), and the
    // two V rows of frequency half xh (the half this thread's own wave consumes)
    const int t8 = tid % (NT_ / 2);
    const int tg = t8 % PPP, tq = t8 / PPP;
    const int win = ((2 * (tq / TC)) * PW + 2 * (tq % TC)) * CKP + 4 * tg;     // top-left of the 4x4 input window
    const int vout = tq * CKP + 4 * tg;

    // 8 accumulators: local k -> frequency xi = 8*xh + k = (row i = 2*xh + k/4, column j = k%4).  The bias rides in on
    // (1,1), which enters all four outputs with coefficient +1: local k = 5 of half 0.
    f32x16 acc[8];
    const float bias = xh == 0 ? a.bias[cb * 32 + li] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = k == 5 ? bias : 0.f;

    const size_t stream = (size_t)nchunks * PAIRS * 64 + 4 * 64;                   // float4 per (cb, half) stream
    const float4* bptr = reinterpret_cast<const float4*>(a.wpack) + ((size_t)cb * 2 + xh) * stream + lane;
    float4 bq[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) bq[p] = bptr[p * 64];
    const int aoff = (8 * xh * NTILES + wm * 32 + li) * CKP + 4 * hh;              // this lane's row of V[8*xh]

    for (int c = 0; c < nchunks; ++c) {
        STAMP();                                           // [2+4c] chunk top
        if (c > 0) __syncthreads();                        // MFMA phase of the previous chunk is done with V
        if (PREFETCH) {
            commit(c, 0, NIT);
        } else {
#pragma unroll 1
            for (int it0 = 0; it0 < NIT; it0 += LB) { issue(c, it0, LB); commit(c, it0, LB); }
        }
        __syncthreads();
        STAMP();                                           // [3+4c] patch committed
        if (PREFETCH && c + 1 < nchunks) issue(c + 1, 0, NIT);

        // ---- input transform V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: rows 2xh, 2xh+1 -----------
        {
            // half 0: t0 = d0 - d2, t1 = d1 + d2;  half 1: t2 = d2 - d1, t3 = d1 - d3   (d_y = window row y)
            float4 ta[4], tb[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float4 d1 = *reinterpret_cast<const float4*>(&patch[win + (1 * PW + x) * CKP]);
                const float4 d2 = *reinterpret_cast<const float4*>(&patch[win + (2 * PW + x) * CKP]);
                const float4 de = *reinterpret_cast<const float4*>(&patch[win + ((xh == 0 ? 0 : 3) * PW + x) * CKP]);
                if (xh == 0) {
                    ta[x] = make_float4(de.x - d2.x, de.y - d2.y, de.z - d2.z, de.w - d2.w);
                    tb[x] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
                } else {
                    ta[x] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
                    tb[x] = make_float4(d1.x - de.x, d1.y - de.y, d1.z - de.z, d1.w - de.w);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4* t = i == 0 ? ta : tb;
                const float4 v0 = make_float4(t[0].x - t[2].x, t[0].y - t[2].y, t[0].z - t[2].z, t[0].w - t[2].w);
                const float4 v1 = make_float4(t[1].x + t[2].x, t[1].y + t[2].y, t[1].z + t[2].z, t[1].w + t[2].w);
                const float4 v2 = make_float4(t[2].x - t[1].x, t[2].y - t[1].y, t[2].z - t[1].z, t[2].w - t[1].w);
                const float4 v3 = make_float4(t[1].x - t[3].x, t[1].y - t[3].y, t[1].z - t[3].z, t[1].w - t[3].w);
                float* vrow = &V[(8 * xh + 4 * i) * NTILES * CKP + vout];
                *reinterpret_cast<float4*>(vrow + 0 * NTILES * CKP) = v0;
                *reinterpret_cast<float4*>(vrow + 1 * NTILES * CKP) = v1;
                *reinterpret_cast<float4*>(vrow + 2 * NTILES * CKP) = v2;
                *reinterpret_cast<float4*>(vrow + 3 * NTILES * CKP) = v3;
            }
        }
        __syncthreads();
        STAMP();                                           // [4+4c] transformed

        // ---- 8 GEMMs per wave: pair p = (k-step, k); A from V (LDS), B from the packed U stream (L2), 4 MFMAs per pair
        const float4* bp = bptr + (size_t)c * PAIRS * 64;
        float4 a0 = *reinterpret_cast<const float4*>(&V[aoff]);
#pragma unroll
        for (int p = 0; p < PAIRS; ++p) {
            float4 a1;
            if (p + 1 < PAIRS) {
                const int ks1 = (p + 1) / 8, k1 = (p + 1) % 8;
                a1 = *reinterpret_cast<const float4*>(&V[k1 * NTILES * CKP + aoff + 8 * ks1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int k = p % 8;
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bq[p % PF].x, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bq[p % PF].y, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bq[p % PF].z, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bq[p % PF].w, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bq[p % PF] = bp[(p + PF) * 64];                // refill the slot just read (tail zero-padded)
            if (p + 1 < PAIRS) a0 = a1;
        }
        STAMP();                                           // [5+4c] MFMA phase issued
    }

    // ---- output transform Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]).  Column direction lane-local:
    //   T_i[b] = M_i0 + M_i1 + M_i2 (b = 0),  M_i1 - M_i2 - M_i3 (b = 1).
    // Row direction across the two frequency halves: Y[0][b] = T_0 + T_1 + T_2,  Y[1][b] = T_1 - T_2 - T_3: half 0 owns
    // output row 0 and needs T_2 from its partner wave, half 1 owns output row 1 and needs T_1: 2 floats per element
    // each way through LDS (the V/patch space is free now).  Results go through LDS again so the global stores are
    // 16 B per lane over whole pixels (a 4-B-per-lane store tail is store-issue-bound).
    constexpr int BN = WN * 32;
    constexpr int OSTR = BN + 4;                           // floats per pixel row of the LDS output tile
    float* const otile = smem;                             // [TH*TW][OSTR]
    float* const exch = smem;                              // [2*GW waves][16 r][2 b][64 lanes]
    float ta0[16], ta1[16], tb0[16], tb1[16];              // T of this half's first / second row, b = 0 / 1
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        ta0[r] = acc[0][r] + acc[1][r] + acc[2][r];
        ta1[r] = acc[1][r] - acc[2][r] - acc[3][r];
        tb0[r] = acc[4][r] + acc[5][r] + acc[6][r];
        tb1[r] = acc[5][r] - acc[6][r] - acc[7][r];
    }
    __syncthreads();                                       // every wave has left its last MFMA phase: V is free
    STAMP();                                               // [E0] MFMAs drained, column transform done
#pragma unroll
    for (int r = 0; r < 16; ++r) {                         // gift: half 0 sends T_1 (its second row), half 1 sends T_2 (its first)
        exch[((wid * 16 + r) * 2 + 0) * 64 + lane] = xh == 0 ? tb0[r] : ta0[r];
        exch[((wid * 16 + r) * 2 + 1) * 64 + lane] = xh == 0 ? tb1[r] : ta1[r];
    }
    __syncthreads();
    STAMP();                                               // [Ea] gifts written
    float y0[16], y1[16];                                  // this half's output row, columns b = 0 / 1
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float g0 = exch[(((wid ^ GW) * 16 + r) * 2 + 0) * 64 + lane];
        const float g1 = exch[(((wid ^ GW) * 16 + r) * 2 + 1) * 64 + lane];
        if (xh == 0) { y0[r] = ta0[r] + tb0[r] + g0; y1[r] = ta1[r] + tb1[r] + g1; }      // T_0 + T_1 + T_2
        else         { y0[r] = g0 - ta0[r] - tb0[r]; y1[r] = g1 - ta1[r] - tb1[r]; }      // T_1 - T_2 - T_3
    }
    __syncthreads();                                       // the gifts have been read: the space becomes the output tile
    STAMP();                                               // [Eb] gifts read
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = (r & 3) + 8 * (r >> 2) + 4 * hh;     // tile within the M-block (hh is per lane)
        const int py = wm * 2 * TR + 2 * (t / TC) + xh, px = 2 * (t % TC);
        otile[(py * TW + px) * OSTR + wn * 32 + li] = fmaxf(y0[r], kLeaky * y0[r]);
        otile[(py * TW + px + 1) * OSTR + wn * 32 + li] = fmaxf(y1[r], kLeaky * y1[r]);
    }
    __syncthreads();
    STAMP();                                               // [E1] output tile in LDS
    constexpr int V4 = BN / 4;                             // float4 per pixel
    const int cbase = blockIdx.y * BN;
    bool fused_last = false;
    if constexpr (BN == 32) {
        if (a.last_w != nullptr) {
            // Fused last layer (noise.py:67,130-133,164): 1x1 conv 32 -> 1, + image channel, clamp; this conv's own
            // 32-channel output is never written.
            for (int p = tid; p < TH * TW; p += NT_) {
                const int gy = ty0 + p / TW, gx = tx0 + p % TW;
                if (gy < a.H && gx < a.W) {
                    float dsum = a.last_b[0];
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4) {
                        const float4 v = *reinterpret_cast<const float4*>(&otile[p * OSTR + 4 * c4]);
                        const float4 wv = *reinterpret_cast<const float4*>(a.last_w + 4 * c4);
                        dsum += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
                    }
                    const size_t q = ((size_t)n * a.H + gy) * a.W + gx;
                    const float img = a.last_ximg != nullptr ? a.last_ximg[q] : (a.last_z[q].x - a.last_u[q].x);
                    a.last_out[q] = fminf(fmaxf(img + dsum, 0.f), 1.f);
                }
            }
            fused_last = true;
        }
    }
    if (!fused_last) {
#pragma unroll 4
    for (int f = tid; f < TH * TW * V4; f += NT_) {
        const int p = f / V4, c4 = f % V4;
        const int gy = ty0 + p / TW, gx = tx0 + p % TW;
        if (gy < a.H && gx < a.W)
            *reinterpret_cast<float4*>(a.dst + (((size_t)n * a.H + gy) * a.W + gx) * a.Cout + cbase + 4 * c4) =
                *reinterpret_cast<const float4*>(&otile[p * OSTR + 4 * c4]);
    }
    if (a.pooled != nullptr) {                             // MaxPool2d(2) of this tile for the next stage (noise.py:22-25)
        const int Hp = a.H >> 1, Wp = a.W >> 1;
        for (int f = tid; f < (TH / 2) * (TW / 2) * V4; f += NT_) {
            const int q = f / V4, c4 = f % V4;
            const int qy = q / (TW / 2), qx = q % (TW / 2);
            const int gy = (ty0 >> 1) + qy, gx = (tx0 >> 1) + qx;
            if (gy < Hp && gx < Wp) {
                const float* o = &otile[((2 * qy) * TW + 2 * qx) * OSTR + 4 * c4];
                const float4 m = f4max(f4max(*reinterpret_cast<const float4*>(o), *reinterpret_cast<const float4*>(o + OSTR)),
                                       f4max(*reinterpret_cast<const float4*>(o + TW * OSTR), *reinterpret_cast<const float4*>(o + TW * OSTR + OSTR)));
                *reinterpret_cast<float4*>(a.pooled + (((size_t)n * Hp + gy) * Wp + gx) * a.Cout + cbase + 4 * c4) = m;
            }
        }
    }
    }
#ifdef PNP_STAMPS
    STAMP();                                               // [E2] stores issued
    {
        const int w = (int)blockIdx.x - (int)(gridDim.x / 2);
        if (tid == 0 && blockIdx.y == 0 && w >= 0 && w < PNP_STAMP_WGS && a.stamp_slot < PNP_STAMP_SLOTS) {
            unsigned long long* o = g_wino_stamps + ((size_t)a.stamp_slot * PNP_STAMP_WGS + w) * PNP_STAMP_N;
            for (int i = 0; i < PNP_STAMP_N; ++i) o[i] = i < ns ? st[i] : 0ull;
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            o[PNP_STAMP_N - 1] = ((unsigned long long)ns << 32) | hwid;
        }
    }
#endif
}

template <int TW, int WM, int WN, int CK, int SRC>
static hipError_t launch_wino_inst(const ConvArgs& a, const WinoPlan& p, hipStream_t s) {
    constexpr int TC = TW / 2, TR = 32 / TC, TH = WM * 2 * TR;
    constexpr size_t lds_main = ((size_t)(TH + 2) * (TW + 2) + 16 * WM * 32 + (SRC == SRC_UPCAT ? (TH / 2 + 3) * (TW / 2 + 3) : 0)) * (CK + 4) * sizeof(float);
    constexpr size_t lds_out = (size_t)TH * TW * (WN * 32 + 4) * sizeof(float);
    constexpr size_t lds_x = (size_t)(2 * WM * WN) * 16 * 2 * 64 * sizeof(float);   // cross-wave exchange of the output transform
    constexpr size_t lds = (lds_main > lds_out ? lds_main : lds_out) > lds_x ? (lds_main > lds_out ? lds_main : lds_out) : lds_x;
    auto kern = conv3x3_winograd_kernel<TW, WM, WN, CK, SRC>;
    static DeviceOnce cap;
    if (hipError_t e = raise_lds_cap((const void*)kern, (int)lds, cap); e != hipSuccess) return e;
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * a.N), (unsigned)(a.Cout / p.bn));
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 128), lds, s, a);
    return hipGetLastError();
}

template <int TW, int WM, int WN, int CK>
static hipError_t launch_wino_cfg(const ConvArgs& a, const WinoPlan& p, int src_mode, hipStream_t s) {
    switch (src_mode) {
        case SRC_PLAIN: return launch_wino_inst<TW, WM, WN, CK, SRC_PLAIN>(a, p, s);
        case SRC_POOL:  return launch_wino_inst<TW, WM, WN, CK, SRC_POOL>(a, p, s);
        case SRC_UPCAT: return launch_wino_inst<TW, WM, WN, CK, SRC_UPCAT>(a, p, s);
        default: return hipErrorInvalidValue;
    }
}

template <int TW>
static hipError_t launch_wino_tw(const ConvArgs& a, const WinoPlan& p, int src_mode, hipStream_t s) {
    if (p.wm == 1 && p.wn == 4) return launch_wino_cfg<TW, 1, 4, 32>(a, p, src_mode, s);
    if (p.wm == 1 && p.wn == 2) return launch_wino_cfg<TW, 1, 2, 16>(a, p, src_mode, s);
    if (p.wm == 2 && p.wn == 2) return launch_wino_cfg<TW, 2, 2, 16>(a, p, src_mode, s);
    if (p.wm == 2 && p.wn == 1) return launch_wino_cfg<TW, 2, 1, 8>(a, p, src_mode, s);
    return launch_wino_cfg<TW, 4, 1, 8>(a, p, src_mode, s);
}

// `a.wpack` must be the Winograd pack (pack_winograd_weights with the plan's ck).
hipError_t launch_conv3x3_winograd(const ConvArgs& a0, const WinoPlan& p, int src_mode, hipStream_t s) {
    if (!p.use) return hipErrorInvalidValue;
    ConvArgs a = a0;
    a.tilesX = p.tiles_x;
    a.tilesY = p.tiles_y;
#ifdef PNP_STAMPS
    a.stamp_slot = g_stamp_slot++;
#endif
    if (p.tw == 32) return launch_wino_tw<32>(a, p, src_mode, s);
    if (p.tw == 16) return launch_wino_tw<16>(a, p, src_mode, s);
    return launch_wino_tw<8>(a, p, src_mode, s);
}

}  // namespace pnp

#ifdef PNP_STAMPS
extern "C" int pnp_debug_stamps_reset(void) { pnp::g_stamp_slot = 0; return 0; }
extern "C" int pnp_debug_stamps_read(unsigned long long* dst, int slots) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pnp::g_wino_stamps),
                                    (size_t)slots * PNP_STAMP_WGS * PNP_STAMP_N * sizeof(unsigned long long));
}
#endif
