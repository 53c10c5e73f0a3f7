// bf16-operand conv3x3 for gfx950, producer / consumer form (PNP_FLAG_BF16_CONVS, BASELINE configs[4]; on chip-filling problems
// every layer but up4.conv-0 and the one carrying the fused last layer - conv3x3_plan() in conv_kernels.hip decides).
// Same arithmetic as conv3x3_mfma_kernel<.., BF16 = true> (conv_kernels.hip): operands
// rounded to bf16 (round to nearest even), v_mfma_f32_32x32x16_bf16 with f32 accumulate, f32 bias / LeakyReLU / activations
// (/root/reference/evaluation/noise.py:88-98 ConvBlock; :22-25 MaxPool2d when the source is pooled while staging).
//
// Why another kernel: a bf16 k-step of the 4 x 2 register tile is 8 MFMAs = 256 cycles, sixteen times shorter than the f32
// one, and the all-waves-do-everything schedule of conv_kernels.hip stops working at that rate (measured, r03: MFMA pipe
// busy 0.15-0.26, WAIT_ANY 0.5-0.6):
//   * vector-memory loads return IN ORDER per wave, so a weight-fragment load issued behind the next chunk's patch loads
//     (HBM, microseconds) waits for them however deep the fragment ring is;
//   * patch registers + 128 accumulators + a ring deep enough for L2 latency do not fit 256 VGPRs (61-73 spilled).
// Here a workgroup is 8 waves, two per SIMD: waves 0-3 (consumers) only run the k-loop - accumulators, A fragments from LDS,
// a 6-deep ring of weight fragments whose loads have nothing slow in front of them - and the epilogue; waves 4-7 (producers)
// only fetch, transform and round the next chunk's patch into the other half of a double-buffered LDS patch.  One
// workgroup barrier per chunk hands a buffer over in each direction:
//     interval k :  consumers  k-loop(item k-1) from buffer (k-1)&1
//                   producers  wait loads(k) - write buffer k&1 - issue loads(k+2) into the register set just freed
//     barrier #k :  buffer k&1 is complete; buffer (k-1)&1 is free
// An "item" is one (tile, 32-channel chunk); a workgroup walks tiles blockIdx.x, + gridDim.x, ... (persistent, one per CU),
// so the producers are already fetching the next tile while the consumers store this one.  (On the 32 -> 32 layers the producers
// store the tiles as well - OFFLOAD below - and the consumers hold the layer's weights in registers - HOLDHI.)
#include "pnp_internal.h"
#include "conv_staging.h"
// Diagnostic build (-DPNP_WS_STAMPS, `make stamps`; tools/ws_stamps.py): s_memtime of consumer wave 0 and producer wave 4 of one
// workgroup at every hand-over, per launch.
// Diagnostic build (-DPNP_WS_RACE_DBG, tools/race_dbg.py; round 5): the separable producers check every patch row they store -
// a second evaluation from the same registers, the bytes that are in LDS, the row weights re-read - and record the first mismatches.
#ifdef PNP_WS_RACE_DBG
#define RACE_REC 32                       // 32-bit words per record
__device__ unsigned g_race_dbg[4 + 256 * RACE_REC];   // [0] mismatching rows, [1] rows checked (low 32 bits), then 256 records
#endif
#ifdef PNP_WS_STAMPS
__device__ unsigned long long g_ws_stamps[64 * 2 * 128];
static int g_ws_slot = 0;
#define WS_STAMP(role) do { if (blockIdx.x == 17 && lane == 0 && (wid == 0 || wid == 4) && nst < 128) { g_ws_stamps[(a.order * 2 + (role)) * 128 + nst] = __builtin_amdgcn_s_memtime(); } ++nst; } while (0)
#else
#define WS_STAMP(role) do { } while (0)
#endif

namespace pnp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS floats of a workgroup: two patches of (TH+2) x (TW+2) pixels x 80 B; UPCAT adds the f32 low-res source region of one
// upsampled chunk ((TH/2+3) x (TW/2+3) pixels x 36 floats) and two copies (tile parity) of the interpolation table.
// The 32-channel tile (NT = 1) adds its epilogue's transpose space, 36 floats per tile pixel; the others the staging space of their
// bf16 epilogue, one M-block (32 pixels x NT * 64 + 16 bytes) per consumer wave.
template <int TW, int WM, int MT, int NT, int SRC>
constexpr int bf16ws_lds_floats() {
    constexpr int TH = WM * MT * 32 / TW, PH = TH + 2, PW = TW + 2;
    return 2 * PH * PW * 20 + (SRC == SRC_UPCAT ? (TH / 2 + 3) * (TW / 2 + 3) * 36 + 2 * 4 * (PH + PW) : 0) +
           (NT == 1 ? 2 * WM * MT * 32 * 36 : 4 * 32 * (SRC == SRC_UPCAT ? NT * 16 + 4 : NT * 32 + 4)) + 256;   // (NT = 1: two output tiles, see OFFLOAD;
                                                             // else the epilogue's staging slices, f32-sized where the source leaves the LDS for it) + the stopped-tile flags
}

// A16S: src0 (the PLAIN / POOL source, the UPCAT skip tensor) holds bf16 - written so by the launch that produced it, rounded
// once with the rounding this staging would apply (same patch bits, half the HBM bytes; ConvArgs.act16 bit 0); dst holds
// bf16 when a.act16 has bit 1.  The low-res source of an upsample stays f32: its consumer rounds AFTER interpolating.
//
// Register tiles (MT x NT blocks of 32 x 32 per consumer wave): 4 x 2 for Cout >= 64 (256 pixels x 128 channels per workgroup,
// or 512 x 64 for Cout = 64), 2 x 1 for the 32-channel level-0 layers (256 x 32; HBM-bound even on bf16 tensors, so its
// epilogue goes through LDS for 16-B stores, the pooled copy of the next stage and the fused last layer, as in
// conv_kernels.hip).
//
// NW: bf16 terms per weight.  2 (the mode's default): every k-step multiplies its A fragments by the weight's hi AND lo term
// (pack_conv3x3_weights_bf16: hi = bf16(w), lo = bf16(w - hi), 1 KiB each per k-step and N-block, hi first), 2 x MT x NT MFMAs into
// the same accumulators - the convolution with the 16-bit-mantissa weight hi + lo.  1: PNP_BF16_W1, the round-3 arithmetic.
// HOLDHI (32 -> 32 layers on the 2 x 1 tile under two-term weights: one chunk, one output-channel block - every tile multiplies by the SAME
// 18 k-steps of weights): a consumer wave keeps BOTH fragments of all 18 k-steps in registers (144) for the life of the workgroup and
// streams no weights at all.  The four consumer waves of that tile fetched identical fragments, two per 128 MFMA cycles each = 64 B/clk per
// CU, which is what the L1 delivers: the tile's k-loop ran at 0.67 of the matrix rate (`profiles/r04_bf16ws_stamps.txt`).  hi held: -10 % on
// inc.conv-1 / inc.conv-2 / up4.conv-1, both held: -18 % (`profiles/r04_ablation.md`; PNP_BF16_NO_HOLDHI switches it off).
template <int TW, int WM, int WN, int MT, int NT, int SRC, bool A16S, int NW, bool HOLDHI = false>
__global__ __launch_bounds__(512) void conv3x3_bf16ws_kernel(const ConvArgs a) {
    static_assert(!HOLDHI || (NT == 1 && NW == 2 && SRC == SRC_PLAIN), "held hi fragments: the 32 -> 32 tile");
    constexpr int CK = 32;
    constexpr int KS = 9 * (CK / 16);   // k-steps (16 channels) per chunk
    constexpr int KB = NW * 1024;       // bytes of one k-step of one N-block in the weight stream
    // weight fragments in flight, k-steps ahead (divides KS: k-step j sits in slot j % PFD): 1536 MFMA cycles for the 4 x 2 tile
    // (6 k-steps of 8 MFMAs, 3 of 16 with two-term weights); the 2 x 1 tile's k-step is 64 (128) cycles, it keeps a whole
    // chunk (half a chunk) ahead
    constexpr int PFD = (MT * NT >= 8 ? 6 : 18) / NW;
    static_assert(KS % PFD == 0 && WM * WN == 4, "ring / wave grid");
    static_assert(NT == 2 || (NT == 1 && WN == 1 && TW == 32 && MT == 2), "32-channel tile: a wave owns two whole tile rows");
    constexpr int CKP = (CK + 8) / 2;   // patch pixel stride in floats (CK + 8 halves: b128 lane groups on distinct banks)
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int TH = BM / TW, PH = TH + 2, PW = TW + 2;
    constexpr int PPP = CK / 4;         // 16-byte f32 pieces per pixel
    constexpr int ITEMS = PH * PW * PPP;
    constexpr int NIT = (ITEMS + 255) / 256;
    constexpr int PATCH = PH * PW * CKP;
    constexpr int NRAW = SRC == SRC_POOL ? 4 : 1;
    static_assert(SRC == SRC_PLAIN || SRC == SRC_POOL || SRC == SRC_UPCAT, "sources staged by this kernel");
    constexpr bool UP2 = SRC == SRC_UPCAT;
    // UPCAT (noise.py:39,46,59): chunks < nskip come straight from the skip tensor; the others are the bilinear x2 upsample
    // (align_corners=True) of the low-res tensor: the producers park the chunk's low-res region [ylo, ylo+LH) x [xlo, xlo+LW)
    // in LDS (f32), meet at an extra workgroup barrier X - the consumers pass it early in the k-loop of the item before - and
    // interpolate the patch LDS -> LDS in f32 before rounding, exactly as conv_kernels.hip does
    constexpr int LH = TH / 2 + 3, LW = TW / 2 + 3, CKL = CK + 4;
    constexpr int NITL = (LH * LW * PPP + 255) / 256;
    constexpr int XK = 2;               // the consumers' k-step in front of which barrier X sits
    static_assert(PPP == 8, "256 producer lanes = 32 pixels x 8 f32 pieces per pass");
    // src0 pieces: 16 B = 4 f32 or 8 bf16 channels; the 256 producer lanes cover PXP pixels x PPS pieces per pass
    constexpr int ESZ = A16S ? 2 : 4;
    constexpr int PPS = CK * ESZ / 16, PXP = 256 / PPS;
    constexpr int NITS = (PH * PW + PXP - 1) / PXP;
    constexpr int NR = (UP2 && NITL > NITS) ? NITL : NITS;   // a raw set holds either kind of item
    extern __shared__ __attribute__((aligned(16))) float patch[];   // bf16ws_lds_floats(): 54-130 KB, dynamic
    float* const lowres = patch + 2 * PATCH;                        // UPCAT: [LH * LW][CKL]
    float* const tabs = lowres + (UP2 ? LH * LW * CKL : 0);         // UPCAT: [2][PH + PW] x {offset of source line 0, 1; weight 0, 1}
    float* const epi = tabs + (UP2 ? 2 * 4 * (PH + PW) : 0);        // NT = 1: [BM][36] output tile, a wave's rows private to it;
                                                                    // else [4 consumer waves][32][NT * 16 + 4] bf16-pair staging
    const int nskip = UP2 ? a.Cskip / CK : 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = a.tilesX * a.tilesY * a.N;
    const int total = tilesM * (a.Cout / BN);
    const int nchunks = a.Cin / CK;
    // tile t: output-channel block cb = t / tilesM is the slow index, so the chip streams one block's weights at a time
    // (tile indices stay far below 2^23: quotient from the float reciprocal, one correction step either way - a dozen instructions
    // instead of the ~35 of a 32-bit division, and a tile decode has five of them)
    auto fdiv = [](int x, int d, float rd) {
        int q = (int)((float)x * rd);
        const int r = x - q * d;
        q += r >= d ? 1 : 0;
        q -= r < 0 ? 1 : 0;
        return q;
    };
    const int tilesXY = a.tilesX * a.tilesY;
    const float r_tilesM = 1.f / (float)tilesM, r_tilesXY = 1.f / (float)tilesXY, r_tilesX = 1.f / (float)a.tilesX;
    struct TileAt { int cb, n, ty0, tx0; };
    auto tile_at = [&](int tt) {
        TileAt o;
        o.cb = fdiv(tt, tilesM, r_tilesM);
        const int m = tt - o.cb * tilesM;
        o.n = fdiv(m, tilesXY, r_tilesXY);
        const int mm = m - o.n * tilesXY;
        const int ty = fdiv(mm, a.tilesX, r_tilesX);
        o.ty0 = ty * TH;
        o.tx0 = (mm - ty * a.tilesX) * TW;
        return o;
    };
    auto slice_of = [&](int t) { return tile_at(t).n; };
    // Stopped slices (tact[n] > 0.5) are skipped tile by tile.  The flags of this workgroup's tiles - blockIdx.x, + gridDim.x,
    // ... - are fetched once into LDS: a read of tact[] per tile is a memory round trip in both roles' critical paths, and
    // a 32-channel tile is only ~2 us of work.
    constexpr int MAXLIVE = 256;
    int* const stopped = reinterpret_cast<int*>(epi + (NT == 1 ? 2 * BM * 36 : 4 * 32 * (SRC == SRC_UPCAT ? NT * 16 + 4 : NT * 32 + 4)));
    const int mine = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup
    const bool cached = a.tact != nullptr && mine <= MAXLIVE;
    if (cached) {
        if (tid < mine) stopped[tid] = a.tact[slice_of((int)blockIdx.x + tid * (int)gridDim.x)] > 0.5f ? 1 : 0;
        __syncthreads();
    }
    auto next_live = [&](int t) {       // first tile >= t (stride gridDim.x) whose slice is still running
        if (a.tact == nullptr) return t;
        if (cached) {
            int i = (t - (int)blockIdx.x) / (int)gridDim.x;
            while (i < mine && stopped[i] != 0) ++i;
            return (int)blockIdx.x + i * (int)gridDim.x;
        }
        while (t < total && a.tact[slice_of(t)] > 0.5f) t += (int)gridDim.x;
        return t;
    };
#ifdef PNP_WS_STAMPS
    int nst = 0;
#endif
    // Staggered start.  Persistent workgroups with equal work run in lockstep: the whole chip reads patches, then the whole chip
    // stores tiles (a store burst at the HBM write peak, `profiles/r03_bf16ws_stamps.txt`).  Sixteen start phases 1536 cycles apart
    // (blocks b and b + 8 share an XCD: the phase is (b / 8) % 16, so every XCD holds all of them) mix the two kinds of traffic
    // and leave the NEXT launch's workgroups out of step too; launches of fewer than four tiles per workgroup are too short
    // to win the skew back (measured: 1-2-tile layers +5 us, the others -3...-8 %, the step -3 %).
    if (mine >= 4) {
        const int phase = ((int)blockIdx.x >> 3) & 15;
        for (int i = 0; i < 3 * phase; ++i) __builtin_amdgcn_s_sleep(8);
    }
    WS_STAMP(wid >= 4);
    int t = next_live((int)blockIdx.x);
    if (t >= total) return;             // (both roles agree: no barrier is ever reached)
    WS_STAMP(wid >= 4);

    // OFFLOAD (the 32 -> 32 layers, one chunk per tile: item g IS tile g): the consumers only write their LeakyReLU'd f32 tile into
    // epi[g & 1] and go on to the next tile's k-loop; the tile's read-back, converts, 16-byte stores and pooled copy - 1100-1900 of the
    // 5300 cycles of a tile on the one consumer wave of a SIMD (`profiles/r04_bf16ws_stamps_final.txt`) - are done by the PRODUCER wave of
    // the same SIMD in the next interval (it has ~4000 idle cycles per item on these layers): epi[g & 1] is complete at barrier #(g + 1),
    // read before barrier #(g + 2), rewritten after it.
    constexpr bool OFFLOAD = HOLDHI;
    auto store_tile = [&](const float* et, int w4, int n, int ty0, int tx0) __attribute__((always_inline)) {   // wave w4's two rows of the tile at et
        constexpr int OSTR = 36, WPX = MT * 32;
        auto act4 = [](float4 v) {                         // LeakyReLU(0.2) of the raw sums the consumers left
            return make_float4(fmaxf(v.x, kLeaky * v.x), fmaxf(v.y, kLeaky * v.y), fmaxf(v.z, kLeaky * v.z), fmaxf(v.w, kLeaky * v.w));
        };
        const float* const ew = et + w4 * WPX * OSTR;
        const int wy0 = ty0 + w4 * (WPX / TW);
        if (a.last_w != nullptr) {
            // fused last layer (noise.py:67,130-133,164): 1x1 conv 32 -> 1 + image residual + clamp; this conv's own output is never
            // written.  One pixel per lane, the sums in the order of the consumers' form of it (below) and of conv_kernels.hip.
            const int gy = wy0 + lane / TW, gx = tx0 + lane % TW;
            if (gy < a.H && gx < a.W) {
                float dsum = a.last_b[0];
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) {
                    const float4 v = act4(*reinterpret_cast<const float4*>(&ew[lane * OSTR + 4 * c4]));
                    const float4 wv = *reinterpret_cast<const float4*>(a.last_w + 4 * c4);
                    dsum += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
                }
                const size_t q = ((size_t)n * a.H + gy) * a.W + gx;
                const float img = a.last_ximg != nullptr ? a.last_ximg[q] : (a.last_z[q].x - a.last_u[q].x);
                a.last_out[q] = fminf(fmaxf(img + dsum, 0.f), 1.f);
            }
            return;
        }
        if (a.act16 & 2) {                                  // bf16 dst: 8 channels (16 B) per lane, 64 B per pixel
            uint16_t* const d16 = reinterpret_cast<uint16_t*>(a.dst);
#pragma unroll
            for (int j = 0; j < WPX * 4 / 64; ++j) {
                const int f = lane + 64 * j, px = f / 4, c8 = f % 4;
                const int gy = wy0 + px / TW, gx = tx0 + px % TW;
                if (gy < a.H && gx < a.W) {
                    const float4 v0 = act4(*reinterpret_cast<const float4*>(&ew[px * OSTR + 8 * c8]));
                    const float4 v1 = act4(*reinterpret_cast<const float4*>(&ew[px * OSTR + 8 * c8 + 4]));
                    uint4 o;
                    o.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0.x, v0.y}, bf16x2));
                    o.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0.z, v0.w}, bf16x2));
                    o.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v1.x, v1.y}, bf16x2));
                    o.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v1.z, v1.w}, bf16x2));
                    *reinterpret_cast<uint4*>(d16 + (((size_t)n * a.H + gy) * a.W + gx) * 32 + 8 * c8) = o;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < WPX * 8 / 64; ++j) {
                const int f = lane + 64 * j, px = f / 8, c4 = f % 8;
                const int gy = wy0 + px / TW, gx = tx0 + px % TW;
                if (gy < a.H && gx < a.W)
                    *reinterpret_cast<float4*>(a.dst + (((size_t)n * a.H + gy) * a.W + gx) * 32 + 4 * c4) =
                        act4(*reinterpret_cast<const float4*>(&ew[px * OSTR + 4 * c4]));
            }
        }
        if (a.pooled != nullptr) {                         // MaxPool2d(2) of the wave's two rows, f32 (its reader rounds after)
            const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
            for (int j = 0; j < (TW / 2) * 8 / 64; ++j) {
                const int f = lane + 64 * j, qx = f / 8, c4 = f % 8;
                const int gy = (wy0 >> 1), gx = (tx0 >> 1) + qx;
                if (gy < Hp && gx < Wp) {
                    const float* o = &ew[(2 * qx) * OSTR + 4 * c4];
                    const float4 m = act4(f4max(f4max(*reinterpret_cast<const float4*>(o), *reinterpret_cast<const float4*>(o + OSTR)),        // (monotonic:
                                                f4max(*reinterpret_cast<const float4*>(o + TW * OSTR), *reinterpret_cast<const float4*>(o + TW * OSTR + OSTR))));   // = max of the activations)
                    if (a.act16 & 4) {                     // bf16 pooled copy (rounding is monotonic too: its reader would round these very values)
                        uint2 o16;
                        o16.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){m.x, m.y}, bf16x2));
                        o16.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){m.z, m.w}, bf16x2));
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.pooled) + (((size_t)n * Hp + gy) * Wp + gx) * 32 + 4 * c4) = o16;
                    } else
                    *reinterpret_cast<float4*>(a.pooled + (((size_t)n * Hp + gy) * Wp + gx) * 32 + 4 * c4) = m;
                }
            }
        }
    };

    if (wid >= 4) {
        // ---------------------------------------------------------------- producers ----------------------------
        // Staging has to be cheap in INSTRUCTIONS: a chunk's k-loop is only 4608 MFMA cycles, and NIT pieces per lane with
        // div / mod / 64-bit address arithmetic each (what conv_kernels.hip does, amortised there over a 16x longer k-loop)
        // cost more vector issue than that.  A lane's pieces are pixel p0 + 32 k (k = 0..NIT-1), part tid % 8 - so the
        // patch coordinates (once per kernel), the byte offset of each piece in its slice with 0x80000000 for "outside the
        // image" (once per tile; the buffer load's range check then returns the conv's zero padding) and LDS addresses that
        // differ by constants are all the state; per chunk a piece costs one buffer load, two converts and one LDS store.
        const int ptid = tid - 256;
        const int part = ptid % PPP, p0 = ptid / PPP;          // f32 geometry: low-res pieces, interpolated pieces
        const int partS = ptid % PPS, p0S = ptid / PPS;        // src0 geometry
        constexpr int SRCMUL = SRC == SRC_POOL ? 2 : 1;        // POOL: the source is 2H x 2W (noise.py:22-25 MaxPool2d(2))
        const int Hs = SRCMUL * a.H, Ws = SRCMUL * a.W;
        const int C0 = UP2 ? a.Cskip : a.Cin;                  // channels of the tensor behind src0
        int pyx[UP2 ? NIT : 1];                                // UPCAT: (py << 16) | px of the patch pixel of interpolated piece k
        int lyx[UP2 ? NITL : 1];                               // same for the low-res region's pieces; -1: no such piece
        if constexpr (UP2) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int pp = p0 + 32 * k;
                pyx[k] = pp < PH * PW ? (((pp / PW) << 16) | (pp % PW)) : -1;
            }
#pragma unroll
            for (int k = 0; k < NITL; ++k) {
                const int lp = p0 + 32 * k;
                lyx[k] = lp < LH * LW ? (((lp / LW) << 16) | (lp % LW)) : -1;
            }
        }
        unsigned goff[NITS];
        unsigned goffL[UP2 ? NITL : 1];
        __amdgpu_buffer_rsrc_t rsrcL;
        int tile_par = 1;                                      // parity of the tile decoded last (UPCAT tables)
        typedef float4 RawSet[NR][NRAW];
        RawSet raw0, raw1;                                     // items alternate between the two sets: two items' loads in flight
        __amdgpu_buffer_rsrc_t rsrc;
        auto decode = [&](int tt) {
            const TileAt ta = tile_at(tt);
            const int tx0 = ta.tx0, ty0 = ta.ty0, n = ta.n;
            const size_t slice = (size_t)Hs * Ws * C0;
            rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.src0) + (size_t)n * slice * ESZ), 0,
                                                     (int)(slice * ESZ), 0x00020000);
#pragma unroll
            for (int k = 0; k < NITS; ++k) {
                const int pp = p0S + PXP * k;
                const int gy = ty0 + pp / PW - 1, gx = tx0 + pp % PW - 1;
                const bool in = pp < PH * PW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                goff[k] = in ? (unsigned)(((SRCMUL * gy) * Ws + SRCMUL * gx) * C0 * ESZ + partS * 16) : 0x80000000u;
            }
            if constexpr (UP2) {
                const int Hl = a.H >> 1, Wl = a.W >> 1, Cup = a.Cin - a.Cskip;
#ifndef PNP_WS_NO_SEP
                const int ylo = ty0 / 2 - 1;                   // separable form: patch row pq interpolates lines ylo + (pq >> 1) and the next one
                const int xlo = (int)(a.rw * (float)(tx0 > 0 ? tx0 - 1 : 0));
#else
                const int ylo = (int)(a.rh * (float)(ty0 > 0 ? ty0 - 1 : 0)), xlo = (int)(a.rw * (float)(tx0 > 0 ? tx0 - 1 : 0));
#endif
                const size_t sl = (size_t)Hl * Wl * Cup;
                rsrcL = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 + (size_t)n * sl), 0, (int)(sl * sizeof(float)), 0x00020000);
#pragma unroll
                for (int k = 0; k < NITL; ++k) {
                    const int sy = ylo + (lyx[k] >> 16), sx = xlo + (lyx[k] & 0xffff);
                    goffL[k] = (lyx[k] >= 0 && sy >= 0 && sy < Hl && sx < Wl) ? (unsigned)((sy * Wl + sx) * Cup + part * 4) * 4u : 0x80000000u;
                }
                // this tile's interpolation table (ATen upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1),
                // weights (1-l, l)); a row / column outside the image gets zero weights on line 0 = the conv's zero padding.
                // Copy tile_par: its previous user is two tiles back, every item of which is committed (>= 3 chunks per tile)
                tile_par ^= 1;
                if (ptid < PH + PW) {
                    const bool isrow = ptid < PH;
                    const int pq = isrow ? ptid : ptid - PH;
                    const int gq = (isrow ? ty0 : tx0) + pq - 1;
                    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gq >= 0 && gq < (isrow ? a.H : a.W)) {
                        const float sc = (isrow ? a.rh : a.rw) * (float)gq;
                        const int i0 = (int)sc;
                        const int i1 = i0 + (i0 < (isrow ? Hl : Wl) - 1 ? 1 : 0);
                        const float l = fminf(fmaxf(sc - (float)i0, 0.f), 1.f);
                        const int lo = isrow ? ylo : xlo, mul = isrow ? LW * CKL : CKL;
                        e = make_float4(__int_as_float((i0 - lo) * mul), __int_as_float((i1 - lo) * mul), 1.f - l, l);
#ifndef PNP_WS_NO_SEP
                        if (isrow) {                           // {weight of line s0, weight of line s0 + 1}: upsample_lines_regular(), as in the F(4x4) kernels
                            const int s0 = ylo + (pq >> 1);
                            e = make_float4((i0 == s0 ? 1.f - l : 0.f) + (i1 == s0 ? l : 0.f), (i0 == s0 + 1 ? 1.f - l : 0.f) + (i1 == s0 + 1 ? l : 0.f), 0.f, 0.f);
                        }
#endif
                    }
                    *reinterpret_cast<float4*>(&tabs[tile_par * 4 * (PH + PW) + 4 * ptid]) = e;
                }
            }
        };
        auto issue = [&](int c, RawSet& raw) {
            if constexpr (UP2) {
                // ONE load sequence for both kinds of chunk - the skip tensor's pieces or the low-res region of an upsampled chunk - with the
                // descriptor, the scalar offset and each piece's byte offset selected by the (wave-uniform) kind; a piece the kind does not have
                // carries the out-of-range offset and costs an issue slot, no memory traffic.  (Rounds 3-4 had one loop per kind in the two arms
                // of a branch: hipcc sank the arms' last stores into a store through a PHI of the two raw-set elements' addresses, which kept
                // four float4 of the raw sets in scratch memory - a scratch_store right behind the loads, i.e. `s_waitcnt vmcnt(0)` at ISSUE time
                // in every UPCAT instantiation: the producers waited for the loads they had just sent out two items ahead, 48-80 B of
                // scratch, `profiles/r05_ablation.md`.)
                const bool up = c >= nskip;
                const int so = up ? (c - nskip) * CK * 4 : c * CK * ESZ;
                const __amdgpu_buffer_rsrc_t rs = up ? rsrcL : rsrc;
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const unsigned o = up ? (k < NITL ? goffL[k < NITL ? k : 0] : 0x80000000u) : (k < NITS ? goff[k < NITS ? k : 0] : 0x80000000u);
                    raw[k][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, so, 0));
                }
                return;
            }
            const int so = c * CK * ESZ;
#pragma unroll
            for (int k = 0; k < NITS; ++k) {
                raw[k][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[k], so, 0));
                if constexpr (SRC == SRC_POOL) {
                    raw[k][NRAW > 1 ? 1 : 0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[k], so + C0 * ESZ, 0));
                    raw[k][NRAW > 1 ? 2 : 0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[k], so + Ws * C0 * ESZ, 0));
                    raw[k][NRAW > 1 ? 3 : 0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[k], so + (Ws + 1) * C0 * ESZ, 0));
                }
            }
        };
        auto store_bf16 = [&](float* buf, int k, float4 v) {   // piece k of the f32 geometry: 4 channels, rounded to nearest even
            const bf16x2 lo = __builtin_convertvector((f32x2){v.x, v.y}, bf16x2);
            const bf16x2 hi = __builtin_convertvector((f32x2){v.z, v.w}, bf16x2);
            *reinterpret_cast<uint2*>(&buf[(p0 + 32 * k) * CKP + part * 2]) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
        };
        auto max16 = [](unsigned x, unsigned y) {              // max of two packed bf16 pairs (exact: bf16 -> f32 is a shift)
            const float lo = fmaxf(__uint_as_float(x << 16), __uint_as_float(y << 16));
            const float hi = fmaxf(__uint_as_float(x & 0xffff0000u), __uint_as_float(y & 0xffff0000u));
            return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
        };
        auto f4max16 = [&](float4 x, float4 y) {
            return make_float4(__uint_as_float(max16(__float_as_uint(x.x), __float_as_uint(y.x))), __uint_as_float(max16(__float_as_uint(x.y), __float_as_uint(y.y))),
                               __uint_as_float(max16(__float_as_uint(x.z), __float_as_uint(y.z))), __uint_as_float(max16(__float_as_uint(x.w), __float_as_uint(y.w))));
        };
        auto commit = [&](float* buf, const RawSet& raw, int meta) {   // meta (UPCAT): bit 0 = upsampled chunk, bit 1 = its tile's parity
            if constexpr (UP2) {
                if (meta & 1) {
#pragma unroll
                    for (int k = 0; k < NITL; ++k)
                        if (p0 + 32 * k < LH * LW) *reinterpret_cast<float4*>(&lowres[(p0 + 32 * k) * CKL + part * 4]) = raw[k][0];
                    __syncthreads();                           // barrier X: the region is parked (its readers are these four waves)
                    const float* tb = tabs + (meta >> 1) * 4 * (PH + PW);
#ifndef PNP_WS_NO_SEP
                    {
                        // Separable form (late round 4; the producers' interpolation is bound by vector issue, `profiles/r04_ablation.md`): a task is
                        // (group of 6 patch rows, patch column, 4-channel part); the horizontal lerps of the four source lines the group touches
                        // are done once, in registers, and each row is then ONE vertical lerp of two of them - a third of the 4-tap form's
                        // instructions.  Needs the regular line structure of upsample_lines_regular() (pnp_create keeps other heights off this kernel).
                        //
                        // The vertical lerp goes through lerp_np() (conv_staging.h): plain v_mul_f32 / v_fma_f32 that hipcc cannot re-pack.  What it
                        // makes of the plain expression is v_pk_mul_f32 / v_pk_fma_f32 that broadcast the second row weight of the loaded {w0, w1}
                        // pair with op_sel = 1 - the LOW result reads the HIGH register - and on MI355X that operand reads ZERO in lanes 48-63 about
                        // once in 200 such instructions while the consumer wave of the SIMD runs bf16 MFMAs fed by buffer loads (round 5:
                        // profiles/r05_race.md; stand-alone reproducer exp/pk_opsel_probe.hip; `-DPNP_WS_LERP_PACKED` rebuilds the faulty form:
                        // ~2000 wrong patch values per denoiser pass, 1/3 to all passes differing).  Round 4 met this as "non-repeatable"
                        // variants and kept the inline assembly without knowing why it helped.  The horizontal lerps below pack into forms whose
                        // HIGH result reads the LOW register (op_sel_hi = 0), which the probe shows clean; tools/isa_audit.py (make audit, the
                        // CPU test suite) fails if a low-reads-high packed operand appears in any kernel with bf16 MFMAs.
                        auto lerp1 = [](float wa, float a_, float wb, float b_) {
#if defined(PNP_WS_LERP_PACKED)
                            return wa * a_ + wb * b_;
#else
                            return lerp_np(wa, a_, wb, b_);
#endif
                        };
                        constexpr int NG = (PH + 5) / 6;               // row groups of six (the last one may be short)
                        static_assert(3 * (NG - 1) + 3 < LH, "a group's four source lines are inside the parked region");
                        constexpr int TPG = PW * PPP, TASKS = NG * TPG, ROUNDS = (TASKS + 255) / 256;
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
#ifdef PNP_WS_RACE_DBG
                                    if (rok) {
                                        // (1) the same expression once more from opaque copies of the same registers
                                        float4 ha2 = ha, hb2 = hb; float2 rw2 = rw;
                                        asm volatile("" : "+v"(ha2.x), "+v"(ha2.y), "+v"(ha2.z), "+v"(ha2.w), "+v"(hb2.x), "+v"(hb2.y), "+v"(hb2.z), "+v"(hb2.w), "+v"(rw2.x), "+v"(rw2.y));
                                        const float4 o2 = make_float4(lerp1(rw2.x, ha2.x, rw2.y, hb2.x), lerp1(rw2.x, ha2.y, rw2.y, hb2.y), lerp1(rw2.x, ha2.z, rw2.y, hb2.z), lerp1(rw2.x, ha2.w, rw2.y, hb2.w));
                                        const unsigned e0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){o2.x, o2.y}, bf16x2));
                                        const unsigned e1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){o2.z, o2.w}, bf16x2));
                                        // (2) what LDS holds now (same wave, in order behind the store), (3) the row weights re-read
                                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                        const u32x2 got_ = *reinterpret_cast<volatile u32x2*>(dst + i * (PW * CKP));
                                        const u32x2 rw3_ = *reinterpret_cast<volatile u32x2*>(const_cast<float*>(&tb[4 * (6 * rg + i)]));
                                        const uint2 got = make_uint2(got_.x, got_.y);
                                        const float2 rw3 = make_float2(__uint_as_float(rw3_.x), __uint_as_float(rw3_.y));
                                        const unsigned s0 = __builtin_bit_cast(unsigned, lo), s1 = __builtin_bit_cast(unsigned, hi);
                                        const int kind = (got.x != s0 || got.y != s1 ? 1 : 0) | (e0 != s0 || e1 != s1 ? 2 : 0) |
                                                         (__float_as_uint(rw3.x) != __float_as_uint(rw.x) || __float_as_uint(rw3.y) != __float_as_uint(rw.y) ? 4 : 0);
                                        if (lane == 0) atomicAdd(&g_race_dbg[1], 1u);
                                        if (kind) {
                                            const unsigned slot = atomicAdd(&g_race_dbg[0], 1u);
                                            if (slot < 256) {
                                                unsigned* r = &g_race_dbg[4 + slot * RACE_REC];
                                                r[0] = kind; r[1] = blockIdx.x; r[2] = (wid << 8) | lane; r[3] = (rd << 16) | (rg << 8) | i;
                                                r[4] = __float_as_uint(rw.x); r[5] = __float_as_uint(rw.y);
                                                r[6] = __float_as_uint(ha.x); r[7] = __float_as_uint(ha.y); r[8] = __float_as_uint(ha.z); r[9] = __float_as_uint(ha.w);
                                                r[10] = __float_as_uint(hb.x); r[11] = __float_as_uint(hb.y); r[12] = __float_as_uint(hb.z); r[13] = __float_as_uint(hb.w);
                                                r[14] = s0; r[15] = s1; r[16] = e0; r[17] = e1; r[18] = got.x; r[19] = got.y;
                                                r[20] = __float_as_uint(rw3.x); r[21] = __float_as_uint(rw3.y); r[22] = (unsigned)__builtin_amdgcn_s_memtime(); r[23] = a.H;
                                                r[24] = __float_as_uint(o.x); r[25] = __float_as_uint(o.y); r[26] = __float_as_uint(o.z); r[27] = __float_as_uint(o.w);
                                                r[28] = __float_as_uint(o2.x); r[29] = __float_as_uint(o2.y); r[30] = __float_as_uint(o2.z); r[31] = __float_as_uint(o2.w);
                                            }
                                        }
                                    }
#endif
                                }
                            }
                        }
                        return;
                    }
#endif
                    // The 4-tap form of rounds 3-4 (reached only in a -DPNP_WS_NO_SEP build: the A/B baseline of the separable form above).
                    // Pieces in groups of G: all table entries, then all source pieces, then the arithmetic and the stores - a piece is
                    // two dependent LDS round trips (~350 cycles each under the consumers' load), and one piece at a time is 8200
                    // cycles per item (`profiles/r03_bf16ws_stamps.txt`), nearly twice the consumers' k-loop.  (An LDS store between
                    // two pieces' reads would serialise them again: the compiler cannot prove patch and region disjoint.)  Pairs
                    // on the 256-pixel tile (up1/up2.conv-0 -6 %; groups of four: hipcc reuses the registers and gains nothing
                    // more); the 512-pixel tile has no registers to spare next to its 20-piece raw set.
                    constexpr int G = NIT <= 12 ? 2 : 1;
                    constexpr int UNR = (NIT <= 12 || A16S) ? NIT : 2;   // (512-pixel tile, f32 skip tensor: fully unrolled it spills)
#pragma unroll UNR
                    for (int k0 = 0; k0 < NIT; k0 += G) {
                        float4 rt[G], ct[G], q[G][4];
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const int k = k0 + j < NIT ? k0 + j : NIT - 1;
                            const int yx = pyx[k] >= 0 ? pyx[k] : 0;
                            rt[j] = *reinterpret_cast<const float4*>(&tb[4 * (yx >> 16)]);
                            ct[j] = *reinterpret_cast<const float4*>(&tb[4 * (PH + (yx & 0xffff))]);
                        }
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const float* l0 = &lowres[__float_as_int(rt[j].x) + part * 4];
                            const float* l1 = &lowres[__float_as_int(rt[j].y) + part * 4];
                            const int c0 = __float_as_int(ct[j].x), c1 = __float_as_int(ct[j].y);
                            q[j][0] = *reinterpret_cast<const float4*>(l0 + c0);
                            q[j][1] = *reinterpret_cast<const float4*>(l0 + c1);
                            q[j][2] = *reinterpret_cast<const float4*>(l1 + c0);
                            q[j][3] = *reinterpret_cast<const float4*>(l1 + c1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            if (k0 + j < NIT && p0 + 32 * (k0 + j) < PH * PW)
                                store_bf16(buf, k0 + j, f4lerp2<true>(q[j][0], q[j][1], q[j][2], q[j][3], ct[j].z, ct[j].w, rt[j].z, rt[j].w));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    return;
                }
            }
#pragma unroll
            for (int k = 0; k < NITS; ++k) {
                if (p0S + PXP * k < PH * PW) {                   // (out-of-image pieces loaded zeros: the conv's zero padding)
                    float4 v = raw[k][0];
                    if constexpr (A16S) {
                        if constexpr (SRC == SRC_POOL)
                            v = f4max16(f4max16(raw[k][0], raw[k][NRAW > 1 ? 1 : 0]), f4max16(raw[k][NRAW > 1 ? 2 : 0], raw[k][NRAW > 1 ? 3 : 0]));
                        *reinterpret_cast<float4*>(&buf[(p0S + PXP * k) * CKP + partS * 4]) = v;      // bf16 already
                    } else {
                        if constexpr (SRC == SRC_POOL)
                            v = f4max(f4max(raw[k][0], raw[k][NRAW > 1 ? 1 : 0]), f4max(raw[k][NRAW > 1 ? 2 : 0], raw[k][NRAW > 1 ? 3 : 0]));
                        store_bf16(buf, k, v);
                    }
                }
            }
        };
        // The producers run TWO items ahead of the consumers: item g is committed from raw set g & 1 into patch buffer g & 1,
        // and the loads of item g + 2 go out into the freed set right away - a whole k-loop interval before they are needed.
        int c = 0;                                             // (t, c): the item issued last
        int pending = 0;                                       // items issued, not yet committed
        bool dry = false;                                      // no further item to issue
        auto issue_next = [&](RawSet& raw, int& meta, bool first) {
            if (dry) return;
            int t1 = t, c1 = c + 1;
            if (first) { c1 = 0; }
            else if (c1 == nchunks) { c1 = 0; t1 = next_live(t + (int)gridDim.x); }
            if (t1 >= total) { dry = true; return; }
            if (first || t1 != t) decode(t1);
            issue(c1, raw);
            meta = (c1 >= nskip && UP2 ? 1 : 0) | (tile_par << 1);
            t = t1; c = c1; ++pending;
        };
        int meta0 = 0, meta1 = 0;
        int gp = 0;                                            // items committed = barriers passed
        int ts = t;                                            // OFFLOAD: the tile whose output the consumers finish next (their sequence)
        auto store_prev = [&]() __attribute__((always_inline)) {     // OFFLOAD, after barrier #gp' (gp' = gp - 1 >= 1): tile gp' - 1 is complete in epi
            const TileAt ta = tile_at(ts);
            store_tile(epi + ((gp - 2) & 1) * (BM * 36), wid - 4, ta.n, ta.ty0, ta.tx0);
            ts = next_live(ts + (int)gridDim.x);
        };
        auto step = [&](RawSet& raw, int& meta, float* buf) {
            commit(buf, raw, meta);
            WS_STAMP(1);
            --pending;
            issue_next(raw, meta, false);
            WS_STAMP(1);
            __syncthreads();                                   // barrier #g
            WS_STAMP(1);
            ++gp;
            if constexpr (OFFLOAD) { if (gp >= 2) store_prev(); }
            return pending > 0;
        };
        issue_next(raw0, meta0, true);
        WS_STAMP(1);
        if constexpr (NRAW == 1 && !(UP2 && NR > 12)) {
            issue_next(raw1, meta1, false);
            for (;;) {
                if (!step(raw0, meta0, patch)) break;
                if (!step(raw1, meta1, patch + PATCH)) break;
            }
        } else {                                               // POOL (four loads per piece), UPCAT on the 512-pixel tile: one item ahead (registers)
            for (;;) {
                if (!step(raw0, meta0, patch)) break;
                if (!step(raw0, meta0, patch + PATCH)) break;
            }
        }
        if constexpr (OFFLOAD) {
            __syncthreads();                                   // the consumers' last tile is in epi
            ++gp;
            store_prev();
        }
        return;
    }

    // -------------------------------------------------------------------- consumers ----------------------------
    const int wm = wid / WN, wn = wid % WN;
    const int hh = lane >> 5, li = lane & 31;
    int aoff[MT];                       // LDS float offset of this lane's A row per M-block (tap (0,0), k-half hh)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wm * MT + mt) * 32 + li;
        aoff[mt] = ((q / TW) * PW + (q % TW)) * CKP + 4 * hh;
    }
    const size_t plane = (size_t)a.N * a.H * a.W * a.Cout;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, (int)(plane * sizeof(float)), 0x00020000);
    // What a tile needs from memory besides its patches is requested ahead, so that a tile boundary costs the consumers no
    // round trip: the next live tile (a tact[] read) at the start of this one, its bias values and the first PFD k-steps
    // of its weights right after this tile's last k-loop - they land under the epilogue's stores.  (The ring's refills of a
    // tile's last PFD k-steps run on past its weights and are thrown away: keeping the chunk loop free of a "last chunk"
    // case is what keeps it free of spills - hipcc unswitches the 144-MFMA body on it and then runs out of registers.)
    // Weight fragments come through a buffer descriptor: the lane part of the address (lane * 16 B) is one VGPR for good and
    // everything else - tile, N-block, chunk, k-step - is scalar arithmetic in the instruction's soffset.
    const size_t wbytes = ((size_t)(a.Cout / 32) * nchunks * KS + 16) * KB;        // the pack ends in 16 zero k-steps
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, (int)wbytes, 0x00020000);
    const int wlane = lane * 16;
    const int wstride = nchunks * KS * KB;                                         // bytes between N-blocks
    auto wbase = [&](int tt) {          // byte offset of the weight stream of tile tt, this wave's N-block 0
        return (fdiv(tt, tilesM, r_tilesM) * (WN * NT) + wn * NT) * wstride;
    };
    auto wload = [&](int soff) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, soff, 0)); };
    float4 bq[PFD][NT][NW];
    float bias_v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bias_v[nt] = a.bias[(fdiv(t, tilesM, r_tilesM) * (WN * NT) + wn * NT + nt) * 32 + li];
#pragma unroll
        for (int p = 0; p < PFD; ++p)
#pragma unroll
            for (int w = 0; w < NW; ++w) bq[p][nt][w] = wload(wbase(t) + nt * wstride + p * KB + w * 1024);
    }
    float4 bhi[HOLDHI ? KS : 1], blo[HOLDHI ? KS : 1];                            // HOLDHI: both fragments of every k-step (Cin = 32: one chunk)
    if constexpr (HOLDHI) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { bhi[ks] = wload(ks * KB); blo[ks] = wload(ks * KB + 1024); }
    }
    int g = 0;
    WS_STAMP(0);
    // 32-channel tile: the NEXT tile's decode (five reciprocal divisions, ~500 cycles of a 4500-cycle tile at the top of the loop) is
    // computed inside this tile's k-loop, in the shadow of its MFMAs
    constexpr bool DECODE_AHEAD = NT == 1;       // (on the 4 x 2 tiles: +0.2...0.5 %, inside the noise, for 5-20 more spilled SGPRs - not taken)
    TileAt ta_c = tile_at(t), ta_nx = ta_c;
    int wt_c = wbase(t), wt_nx = wt_c;
    while (t < total) {
        const TileAt ta = DECODE_AHEAD ? ta_c : tile_at(t);
        const int cbt = ta.cb, tx0 = ta.tx0, ty0 = ta.ty0, n = ta.n;
        const int t_next = next_live(t + (int)gridDim.x);      // (its loads fly under this tile's k-loops)
        const int wtile = DECODE_AHEAD ? wt_c : wbase(t);
        f32x16 acc[MT][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bias_v[nt];   // accumulators start at the bias
        }

        for (int c = 0; c < nchunks; ++c, ++g) {
            WS_STAMP(0);
            __syncthreads();                                   // barrier #g: this item's patch is complete
            WS_STAMP(0);
            const bool park = UP2 && c + 1 < nchunks && c + 1 >= nskip;   // the producers park a low-res region during this k-loop
            const float* pb = patch + (g & 1) * PATCH;
            const int wc = wtile + c * (KS * KB);              // this chunk's k-step 0
            // the 2 x 1 tile's ring runs on past the tile's last chunk into the next tile's first, refilled in place (few registers
            // here, and its short epilogue would not cover a re-prime); the 4 x 2 tile's runs on into whatever follows the chunk
            // and is re-primed after the tile's last k-loop
            const int wfollow = (NT == 1 && c + 1 == nchunks) ? wbase(t_next < total ? t_next : t) : wc + KS * KB;
            float4 a0[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a0[mt] = *reinterpret_cast<const float4*>(&pb[aoff[mt]]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (UP2 && ks == XK && park) __syncthreads();  // barrier X (see the producers' commit)
                if constexpr (DECODE_AHEAD) {
                    if (ks == 2 && c == 0) {
                        const int tn = t_next < total ? t_next : t;
                        const TileAt q = tile_at(tn);
                        ta_nx.cb = __builtin_amdgcn_readfirstlane(q.cb); ta_nx.n = __builtin_amdgcn_readfirstlane(q.n);
                        ta_nx.ty0 = __builtin_amdgcn_readfirstlane(q.ty0); ta_nx.tx0 = __builtin_amdgcn_readfirstlane(q.tx0);
                        wt_nx = __builtin_amdgcn_readfirstlane(wbase(tn));
                    }
                }
                float4 a1[MT];
                if (ks + 1 < KS) {                             // A fragments of the next k-step
                    const int tap1 = (ks + 1) / (CK / 16), s1 = (ks + 1) % (CK / 16);
                    const int off1 = ((tap1 / 3) * PW + (tap1 % 3)) * CKP + 8 * s1;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a1[mt] = *reinterpret_cast<const float4*>(&pb[aoff[mt] + off1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = 0; w < NW; ++w)                   // hi, then lo: an accumulator comes round again MT * NT MFMAs later
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[mt]),
                                                                                  __builtin_bit_cast(bf16x8, HOLDHI ? (w == 0 ? bhi[HOLDHI ? ks : 0] : blo[HOLDHI ? ks : 0]) : bq[ks % PFD][nt][w]), acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // refill the slot just read with k-step ks + PFD of the stream (this chunk's, or the following chunk's first ones)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int w = HOLDHI ? NW : 0; w < NW; ++w)
                        bq[ks % PFD][nt][w] = wload((ks + PFD < KS ? wc + (ks + PFD) * KB : wfollow + (ks + PFD - KS) * KB) + nt * wstride + w * 1024);
                if (ks + 1 < KS) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a0[mt] = a1[mt];
                }
            }
        }

        WS_STAMP(0);                                           // (diagnostic build: end of the tile's k-loops)
        if (t_next < total) {                                  // the next tile's bias and first weight fragments
            const int wn0 = wbase(t_next);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bias_v[nt] = a.bias[(fdiv(t_next, tilesM, r_tilesM) * (WN * NT) + wn * NT + nt) * 32 + li];
                if constexpr (NT != 1) {
#pragma unroll
                    for (int p = 0; p < PFD; ++p)
#pragma unroll
                        for (int w = 0; w < NW; ++w) bq[p][nt][w] = wload(wn0 + nt * wstride + p * KB + w * 1024);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NT == 1) {
            // 32-channel tile: LeakyReLU(0.2), then through the wave's own rows of the LDS output tile - no barrier: a wave reads
            // back only what it wrote - so that global stores are 16 B per lane over whole pixels, and the 2x2 max-pooled copy for
            // the next stage (noise.py:22-25) or the fused last layer (1x1 conv 32 -> 1 + image residual + clamp, noise.py:67,
            // 130-133,164; this conv's own output is then never written) come from the same tile
            constexpr int OSTR = 36, WPX = MT * 32;            // a wave's pixels: WPX / TW = 2 whole tile rows
            float* const ew = epi + (OFFLOAD ? ((g - 1) & 1) * (BM * 36) : 0) + wid * WPX * OSTR;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ew[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * OSTR + li] =
                        OFFLOAD ? acc[mt][0][r] : fmaxf(acc[mt][0][r], kLeaky * acc[mt][0][r]);   // (OFFLOAD: the activation is applied by store_tile)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            WS_STAMP(0);                                       // (diagnostic build: output tile written to LDS)
            const int wy0 = ty0 + wid * (WPX / TW);            // the wave's first image row
            if constexpr (OFFLOAD) {
                // stored by producer wave wid + 4 after the next workgroup barrier (store_tile above)
            } else
            if (a.last_w != nullptr) {
                const int gy = wy0 + lane / TW, gx = tx0 + lane % TW;          // one pixel per lane
                if (gy < a.H && gx < a.W) {
                    float dsum = a.last_b[0];
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4) {
                        const float4 v = *reinterpret_cast<const float4*>(&ew[lane * OSTR + 4 * c4]);
                        const float4 wv = *reinterpret_cast<const float4*>(a.last_w + 4 * c4);
                        dsum += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
                    }
                    const size_t q = ((size_t)n * a.H + gy) * a.W + gx;
                    const float img = a.last_ximg != nullptr ? a.last_ximg[q] : (a.last_z[q].x - a.last_u[q].x);
                    a.last_out[q] = fminf(fmaxf(img + dsum, 0.f), 1.f);
                }
            } else {
                if (a.act16 & 2) {                              // bf16 dst: 8 channels (16 B) per lane, 64 B per pixel
                    uint16_t* const d16 = reinterpret_cast<uint16_t*>(a.dst);
#pragma unroll
                    for (int j = 0; j < WPX * 4 / 64; ++j) {
                        const int f = lane + 64 * j, px = f / 4, c8 = f % 4;
                        const int gy = wy0 + px / TW, gx = tx0 + px % TW;
                        if (gy < a.H && gx < a.W) {
                            const float4 v0 = *reinterpret_cast<const float4*>(&ew[px * OSTR + 8 * c8]);
                            const float4 v1 = *reinterpret_cast<const float4*>(&ew[px * OSTR + 8 * c8 + 4]);
                            uint4 o;
                            o.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0.x, v0.y}, bf16x2));
                            o.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0.z, v0.w}, bf16x2));
                            o.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v1.x, v1.y}, bf16x2));
                            o.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v1.z, v1.w}, bf16x2));
                            *reinterpret_cast<uint4*>(d16 + (((size_t)n * a.H + gy) * a.W + gx) * 32 + 8 * c8) = o;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < WPX * 8 / 64; ++j) {
                        const int f = lane + 64 * j, px = f / 8, c4 = f % 8;
                        const int gy = wy0 + px / TW, gx = tx0 + px % TW;
                        if (gy < a.H && gx < a.W)
                            *reinterpret_cast<float4*>(a.dst + (((size_t)n * a.H + gy) * a.W + gx) * 32 + 4 * c4) =
                                *reinterpret_cast<const float4*>(&ew[px * OSTR + 4 * c4]);
                    }
                }
                if (a.pooled != nullptr) {                     // MaxPool2d(2) of the wave's two rows, f32 (its reader rounds after)
                    const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
                    for (int j = 0; j < (TW / 2) * 8 / 64; ++j) {
                        const int f = lane + 64 * j, qx = f / 8, c4 = f % 8;
                        const int gy = (wy0 >> 1), gx = (tx0 >> 1) + qx;
                        if (gy < Hp && gx < Wp) {
                            const float* o = &ew[(2 * qx) * OSTR + 4 * c4];
                            const float4 m = f4max(f4max(*reinterpret_cast<const float4*>(o), *reinterpret_cast<const float4*>(o + OSTR)),
                                                   f4max(*reinterpret_cast<const float4*>(o + TW * OSTR), *reinterpret_cast<const float4*>(o + TW * OSTR + OSTR)));
                            *reinterpret_cast<float4*>(a.pooled + (((size_t)n * Hp + gy) * Wp + gx) * 32 + 4 * c4) = m;
                        }
                    }
                }
            }
            // (the next tile's epilogue rewrites ew only after this wave's reads: same wave, program order)
        } else
        // epilogue: LeakyReLU(0.2), NHWC store; per-slot address part in the scalar offset of a buffer store, per-lane part in
        // one VGPR per N-block
        if (a.act16 & 2) {
            // bf16 dst.  Lanes (2j, 2j+1) hold channels (2j, 2j+1) of the same pixels; they swap one value per pixel PAIR (DPP quad_perm
            // [1,0,3,2]) so that the even lane holds pixel r's channel pair and the odd lane pixel r+1's (4 B each).  Round 4: the pairs
            // go through LDS - one M-block (32 pixels x NT x 32 channels) at a time into THIS WAVE's staging slice - and leave as
            // 16-byte stores over whole 64 / 128-byte channel runs: 4 x MT store instructions per lane, every one full lines,
            // instead of 8 x MT x NT four-byte stores on 64-byte fragments, which took a third of a level-1 tile (~10000 cycles,
            // `profiles/r04_bf16ws_stamps.txt`) and held up the next tile's first weight fragments behind them.
            // (The slices are LDS of their own.  The first form of this epilogue staged in the patch buffer the tile's last k-loop
            // had just read - free of the PRODUCERS until the next barrier, but not of the other consumer waves, which read the same
            // patch and are not in step: a wave one M-block's conversion ahead overwrote A fragments a slower one had yet to
            // read.  It passed every parity test; the repeatability test caught it once, late in the round.)
            constexpr int SSTR = NT * 16 + 4;                  // staged pixel: NT * 64 bytes of channels + 16 bytes of skew, in floats
            constexpr int CPP = NT * 4;                        // 16-byte pieces per staged pixel
            static_assert((32 * CPP) % 64 == 0, "whole rounds of 64 lanes");
            unsigned* const st = reinterpret_cast<unsigned*>(epi) + wid * (32 * SSTR);
            const bool odd = (li & 1) != 0;
            uint16_t* const d16 = reinterpret_cast<uint16_t*>(a.dst);
            const int cw0 = (cbt * (WN * NT) + wn * NT) * 32;  // this wave's first output channel
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int pl = (r & 3) + 8 * (r >> 2) + 4 * hh + (odd ? 1 : 0);   // pixel of the M-block this lane's pair belongs to
                        const float x = fmaxf(acc[mt][nt][r], kLeaky * acc[mt][nt][r]);
                        const float y = fmaxf(acc[mt][nt][r + 1], kLeaky * acc[mt][nt][r + 1]);
                        const float got = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(odd ? x : y), 0xB1, 0xf, 0xf, false));
                        const bf16x2 pk = __builtin_convertvector((f32x2){odd ? got : x, odd ? y : got}, bf16x2);   // round to nearest even
                        st[pl * SSTR + nt * 16 + (li >> 1)] = __builtin_bit_cast(unsigned, pk);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int i = 0; i < 32 * CPP / 64; ++i) {
                    const int f = lane + 64 * i, pl = f / CPP, c8 = f % CPP;
                    const int q = (wm * MT + mt) * 32 + pl;
                    const int gy = ty0 + q / TW, gx = tx0 + q % TW;
                    const uint4 v = *reinterpret_cast<const uint4*>(&st[pl * SSTR + c8 * 4]);
                    if (gy < a.H && gx < a.W)
                        *reinterpret_cast<uint4*>(d16 + (((size_t)n * a.H + gy) * a.W + gx) * a.Cout + cw0 + 8 * c8) = v;
                }
                __builtin_amdgcn_sched_barrier(0);             // (the next M-block's LDS writes follow these reads in program order)
            }
        } else if (SRC != SRC_UPCAT) {
            // f32 dst (the layers whose output is the low-res source of an upsample: it stays f32, its reader rounds after interpolating), the
            // same way: one M-block at a time through this wave's staging slice, out as 16-byte stores over whole 256-byte channel runs
            // (4-byte stores on 128-byte fragments made this epilogue 12800 cycles against the bf16 one's 6900, `r04_bf16ws_stamps_final.txt`)
            constexpr int SSTRF = NT * 32 + 4;                 // staged pixel: NT * 128 bytes of channels + 16 bytes of skew, in floats
            constexpr int CPPF = NT * 8;                       // 16-byte pieces per staged pixel
            float* const stf = epi + wid * (32 * SSTRF);
            const int cw0 = (cbt * (WN * NT) + wn * NT) * 32;  // this wave's first output channel
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stf[((r & 3) + 8 * (r >> 2) + 4 * hh) * SSTRF + nt * 32 + li] = fmaxf(acc[mt][nt][r], kLeaky * acc[mt][nt][r]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int i = 0; i < 32 * CPPF / 64; ++i) {
                    const int f = lane + 64 * i, pl = f / CPPF, c4 = f % CPPF;
                    const int q = (wm * MT + mt) * 32 + pl;
                    const int gy = ty0 + q / TW, gx = tx0 + q % TW;
                    const float4 v = *reinterpret_cast<const float4*>(&stf[pl * SSTRF + c4 * 4]);
                    if (gy < a.H && gx < a.W)
                        *reinterpret_cast<float4*>(a.dst + (((size_t)n * a.H + gy) * a.W + gx) * a.Cout + cw0 + 4 * c4) = v;
                }
                __builtin_amdgcn_sched_barrier(0);             // (the next M-block's LDS writes follow these reads in program order)
            }
        } else {
            // (UPCAT instantiations, f32 dst: only the PNP_BF16_F32_ACTS ablation comes here.)  The 64 per-register scalar offsets below are
            // loop invariants: hipcc hoisted them out of the tile loop and spilled 160 SGPRs to keep them - in every UPCAT instantiation,
            // whether this path runs or not.  The two factors pass through an opaque asm inside the loop, so each offset is two scalar
            // multiplies at its store.
            int rowb = a.W * a.Cout * 4, pixb = a.Cout * 4;
            asm volatile("" : "+s"(rowb), "+s"(pixb));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = (cbt * (WN * NT) + wn * NT + nt) * 32 + li;
                const unsigned obase = ((unsigned)(((size_t)n * a.H + ty0) * a.W + tx0 + 4 * hh) * (unsigned)a.Cout + (unsigned)co) * 4u;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qc = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2);   // tile pixel index, lane-independent part
                        const int gy = ty0 + qc / TW, gx = tx0 + qc % TW + 4 * hh;
                        const int soff = (qc / TW) * rowb + (qc % TW) * pixb;
                        const float v = fmaxf(acc[mt][nt][r], kLeaky * acc[mt][nt][r]);
                        if (gy < a.H && gx < a.W) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, obase, soff, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);         // one 32 x 32 block at a time: the weight ring stays in registers
                }
            }
        }
        WS_STAMP(0);
        t = t_next;
        ta_c = ta_nx; wt_c = wt_nx;
    }
    if constexpr (OFFLOAD) __syncthreads();                    // hands the last tile to the producers
}

template <int TW, int WM, int WN, int MT, int NT, int SRC, bool A16S, int NW, bool HOLDHI = false>
static hipError_t launch_k(const ConvArgs& a, unsigned grid, hipStream_t s) {
    constexpr int BYTES = bf16ws_lds_floats<TW, WM, MT, NT, SRC>() * 4;
    static DeviceOnce once;
    const hipError_t e = raise_lds_cap(reinterpret_cast<const void*>(&conv3x3_bf16ws_kernel<TW, WM, WN, MT, NT, SRC, A16S, NW, HOLDHI>), BYTES, once);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((conv3x3_bf16ws_kernel<TW, WM, WN, MT, NT, SRC, A16S, NW, HOLDHI>), dim3(grid), dim3(512), BYTES, s, a);
    return hipGetLastError();
}

template <int TW, int WM, int WN, int MT, int NT, int SRC>
static hipError_t launch_one(const ConvArgs& a, unsigned grid, hipStream_t s, bool holdhi) {
    if constexpr (NT == 1 && SRC == SRC_PLAIN) {           // 32 -> 32 layers: the hi fragments stay in registers (HOLDHI; the handle's plan decides)
        if (a.bf16 == 2 && a.Cin == 32 && a.Cout == 32 && holdhi)
            return (a.act16 & 1) ? launch_k<TW, WM, WN, MT, NT, SRC, true, 2, true>(a, grid, s) : launch_k<TW, WM, WN, MT, NT, SRC, false, 2, true>(a, grid, s);
    }
    if (a.bf16 == 2)
        return (a.act16 & 1) ? launch_k<TW, WM, WN, MT, NT, SRC, true, 2>(a, grid, s) : launch_k<TW, WM, WN, MT, NT, SRC, false, 2>(a, grid, s);
    return (a.act16 & 1) ? launch_k<TW, WM, WN, MT, NT, SRC, true, 1>(a, grid, s) : launch_k<TW, WM, WN, MT, NT, SRC, false, 1>(a, grid, s);
}

template <int TW, int WM, int WN, int MT, int NT>
static hipError_t launch_src(const ConvArgs& a, int src_mode, unsigned grid, hipStream_t s, bool holdhi) {
    if (src_mode == SRC_PLAIN) return launch_one<TW, WM, WN, MT, NT, SRC_PLAIN>(a, grid, s, holdhi);
    if constexpr (WN == 2) {
        if (src_mode == SRC_POOL) return launch_one<TW, WM, WN, MT, NT, SRC_POOL>(a, grid, s, holdhi);
    }
    if (src_mode == SRC_UPCAT && a.Cskip % 32 == 0 && a.Cskip >= 32 && a.Cin - a.Cskip >= 32) return launch_one<TW, WM, WN, MT, NT, SRC_UPCAT>(a, grid, s, holdhi);
    return hipErrorInvalidValue;
}

hipError_t launch_conv3x3_bf16ws(const ConvArgs& a0, const ConvPlan& p, int src_mode, hipStream_t s) {
    if (!p.ws || p.ck != 32 || a0.Cin % 32 != 0 || a0.Cout % p.bn != 0 || !a0.bf16 || !conv3x3_tensor_fits(a0.N, a0.H, a0.W, a0.Cin, a0.Cout) ||
        ((a0.pooled != nullptr || a0.last_w != nullptr) && p.nt != 1))
        return hipErrorInvalidValue;
    ConvArgs a = a0;
    a.tilesX = p.tiles_x;
    a.tilesY = p.tiles_y;
#ifdef PNP_WS_STAMPS
    a.order = g_ws_slot < 64 ? g_ws_slot++ : 63;
#endif
    const long total = (long)p.tiles_x * p.tiles_y * a.N * (a.Cout / p.bn);
    const unsigned grid = (unsigned)(total < 256 ? total : 256);     // persistent: one workgroup (8 waves) per CU
    if (p.nt == 1) {
        if (p.tw == 32 && p.mt == 2 && p.wm == 4) return launch_src<32, 4, 1, 2, 1>(a, src_mode, grid, s, p.holdhi != 0);
    } else if (p.wn == 2) {
        if (p.tw == 32) return launch_src<32, 2, 2, 4, 2>(a, src_mode, grid, s, p.holdhi != 0);
        if (p.tw == 16) return launch_src<16, 2, 2, 4, 2>(a, src_mode, grid, s, p.holdhi != 0);
    } else {
        if (p.tw == 32) return launch_src<32, 4, 1, 4, 2>(a, src_mode, grid, s, p.holdhi != 0);
        if (p.tw == 16) return launch_src<16, 4, 1, 4, 2>(a, src_mode, grid, s, p.holdhi != 0);
    }
    return hipErrorInvalidValue;
}

}  // namespace pnp

#ifdef PNP_WS_RACE_DBG
extern "C" int pnp_debug_race_reset(void) { static unsigned z[4 + 256 * RACE_REC]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_race_dbg), z, sizeof(z)); }
extern "C" int pnp_debug_race_read(unsigned* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_race_dbg), sizeof(unsigned) * (4 + 256 * RACE_REC)); }
#endif
#ifdef PNP_WS_STAMPS
extern "C" int pnp_debug_ws_stamps_reset(void) { g_ws_slot = 0; return 0; }
extern "C" int pnp_debug_ws_stamps_read(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ws_stamps), sizeof(unsigned long long) * 64 * 2 * 128);
}
#endif
