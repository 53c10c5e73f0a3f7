// Winograd F(4x4, 3x3) convolution for the K-heavy denoiser layers (Cin >= 128) on gfx950 (MI355X).
//
// Same operator as conv3x3_winograd_kernel / conv3x3_mfma_kernel (conv3x3 s1 p1 + bias + LeakyReLU(0.2),
// /root/reference/evaluation/noise.py:75-98, the stage's bilinear-upsample+concat input transform applied while staging),
// computed with  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  per 4x4 output tile: 36 multiplies per 16 outputs and (cin, cout)
// pair instead of 144 (direct) or 64 (F(2x2)) - 1.78x fewer MFMAs than F(2x2) on layers whose bound is the f32 matrix pipe.
//
// Interpolation points (0, +-3/4, +-3/2, inf) instead of the textbook (0, +-1, +-2, inf): every transform coefficient stays
// a dyadic rational (exact in f32) and the f32 error of a 256-channel layer drops from 9.8e-6 to 1.9e-6 of the output scale
// (3x the F(2x2) kernel's, 6x the direct sum's; tools/wino_points.py scans the candidates).  All arithmetic is f32; the
// weight transform G g G^T is done once on the host in f64.
//
// Workgroup = 8 waves = 4 frequency quadrants (3x3 blocks of the 6x6 frequency grid) x 2 groups of 32 output channels,
// working on 32 tiles (512 output pixels) x 64 channels; every wave holds 9 accumulators of 32 tiles x 32 channels
// (144 registers), two waves per SIMD.  Per 16-channel chunk:
//   1. stage the (TH+2) x (TW+2) halo patch in LDS (loads of chunk c+1 in flight during chunk c's MFMAs; upsampled chunks
//      come from the tile's low-res source region parked in LDS, like the F(2x2) kernel)
//   2. input transform: thread (tile, 2 channels, frequency-column group) reads the 6x5 part of the 6x6 window its three
//      frequency columns need and writes their 18 frequency planes V[xi][tile][channel] (no value is computed twice)
//   3. 36 independent GEMMs, 9 per wave: acc[xi] += V[xi] (A fragment, ds_read_b128) x U[xi] (B fragment, transformed
//      weights streamed from L2 in pre-packed per-lane order)
// Output transform: every wave turns its 3x3 block of M into a PARTIAL 4x4 output tile (lane-local), the four partials
// meet in LDS and are summed in a fixed order (bit-reproducible) by threads that own (2x2 pixel window, 4 channels):
// + bias, LeakyReLU, 16-byte NHWC stores and the 2x2 max-pooled copy for the next stage from the same registers.
#include "pnp_internal.h"
#include "conv_staging.h"

namespace pnp {

// Diagnostic build only (make stamps -> libpnpadmm_stamps.so, read by tools/wino4_stamps.py): wave 0 of up to 1024 mid-grid
// workgroups per launch accumulates s_memtime differences per phase.
#ifdef PNP_STAMPS
#define W4_SLOTS 32
#define W4_WGS 1024
#define W4_N 16
__device__ unsigned long long g_w4_stamps[W4_SLOTS * W4_WGS * W4_N];
static int g_w4_slot = 0;
#define W4T() __builtin_amdgcn_s_memtime()
#endif

namespace {
// points (0, +-a, +-b, inf), a = 3/4, b = 3/2
constexpr double kA = 0.75, kB = 1.5;
constexpr float fA = 0.75f, fB = 1.5f, fA2 = 0.5625f, fB2 = 2.25f, fA3 = 0.421875f, fB3 = 3.375f;
constexpr float fK0 = 1.265625f;      // a^2 b^2
constexpr float fK1 = 2.8125f;        // a^2 + b^2
constexpr float fAB2 = 1.6875f;       // a b^2
constexpr float fA2B = 0.84375f;      // a^2 b

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
typedef float v2f __attribute__((ext_vector_type(2)));

// One-dimensional input transform, three of the six outputs.  B^T rows (ascending powers of the window index):
//   0: [a^2 b^2, 0, -(a^2+b^2), 0, 1, 0]   1,2: [0, -+a b^2, -b^2, +-a, 1, 0]
//   3,4: [0, -+a^2 b, -a^2, +-b, 1, 0]     5: [0, a^2 b^2, 0, -(a^2+b^2), 0, 1]
// G = 0: outputs 0,1,2 from d[0..4];  G = 1: outputs 3,4,5 from d[1..5] (d is indexed by window position either way).
template <int G>
__device__ __forceinline__ void bt3(const v2f* d, v2f* o) {
    if constexpr (G == 0) {
        o[0] = fK0 * d[0] + (d[4] - fK1 * d[2]);
        const v2f e = d[4] - fB2 * d[2];
        const v2f od = fA * (d[3] - fB2 * d[1]);
        o[1] = e + od;
        o[2] = e - od;
    } else {
        const v2f e = d[4] - fA2 * d[2];
        const v2f od = fB * (d[3] - fA2 * d[1]);
        o[0] = e + od;
        o[1] = e - od;
        o[2] = fK0 * d[1] + (d[5] - fK1 * d[3]);
    }
}

// V = B^T d B for one (tile, 2 channels) item and ONE column group (QJ = 0: frequency columns 0..2 from window positions
// 0..4; QJ = 1: columns 3..5 from positions 1..5), all six frequency rows: the column transform of each of the six window
// rows (three outputs), then the full six-point row transform of each of the three columns.  Nothing is computed twice
// (84 packed VALU instructions and 30 + 18 LDS accesses per item; splitting the 36 frequencies by quadrant instead cost
// 130 + 68 per thread because the two quadrant rows both transformed window rows 1..4).
template <int QJ, int PW, int CKP, int PLANE>
__device__ __forceinline__ void wino4_input_transform(const float* w, float* v) {
    constexpr int X0 = QJ == 0 ? 0 : 1;
    v2f c[6][3];
#pragma unroll
    for (int y = 0; y < 6; ++y) {
        v2f d[6];
        __builtin_amdgcn_sched_barrier(0);                 // one window row in flight at a time (registers)
#pragma unroll
        for (int x = 0; x < 5; ++x) d[X0 + x] = *reinterpret_cast<const v2f*>(w + (y * PW + X0 + x) * CKP);
        bt3<QJ>(d, c[y]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const v2f r0 = c[0][j], r1 = c[1][j], r2 = c[2][j], r3 = c[3][j], r4 = c[4][j], r5 = c[5][j];
        float* vj = v + (3 * QJ + j) * PLANE;
        *reinterpret_cast<v2f*>(vj + 0 * 6 * PLANE) = fK0 * r0 + (r4 - fK1 * r2);
        const v2f e1 = r4 - fB2 * r2, o1 = fA * (r3 - fB2 * r1);
        *reinterpret_cast<v2f*>(vj + 1 * 6 * PLANE) = e1 + o1;
        *reinterpret_cast<v2f*>(vj + 2 * 6 * PLANE) = e1 - o1;
        const v2f e2 = r4 - fA2 * r2, o2 = fB * (r3 - fA2 * r1);
        *reinterpret_cast<v2f*>(vj + 3 * 6 * PLANE) = e2 + o2;
        *reinterpret_cast<v2f*>(vj + 4 * 6 * PLANE) = e2 - o2;
        *reinterpret_cast<v2f*>(vj + 5 * 6 * PLANE) = fK0 * r1 + (r5 - fK1 * r3);
    }
}
}  // namespace

// ---- host: U = G g G^T (6x6 per (cout, cin)), packed [cout/32][quadrant 4][chunk][ks 2][k 9][lane 64][4] -----------------
size_t winograd4_pack_floats(int cin, int cout) { return (size_t)(cout / 32) * ((size_t)(cin / 8) * 36 * 256 + 4 * 4 * 256); }

void pack_winograd4_weights(const float* oihw, int cin, int cout, int ck, float* dst) {
    const int CK = ck;                                  // 16 (64-channel workgroups) or 8 (32-channel workgroups)
    // G rows: p^k / N_p for the finite points, [0 0 1] for infinity; N_p = prod_{q != p} (p - q)
    const double pts[5] = {0.0, kA, -kA, kB, -kB};
    double G[6][3];
    for (int j = 0; j < 5; ++j) {
        double nrm = 1.0;
        for (int l = 0; l < 5; ++l) if (l != j) nrm *= pts[j] - pts[l];
        G[j][0] = 1.0 / nrm; G[j][1] = pts[j] / nrm; G[j][2] = pts[j] * pts[j] / nrm;
    }
    G[5][0] = 0.0; G[5][1] = 0.0; G[5][2] = 1.0;
    std::vector<float> U((size_t)cout * cin * 36);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = oihw + ((size_t)co * cin + ci) * 9;
            double t[6][3];
            for (int i = 0; i < 6; ++i)
                for (int kx = 0; kx < 3; ++kx) t[i][kx] = G[i][0] * g[kx] + G[i][1] * g[3 + kx] + G[i][2] * g[6 + kx];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j)
                    U[((size_t)co * cin + ci) * 36 + 6 * i + j] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
    size_t o = 0;
    for (int cb = 0; cb < cout / 32; ++cb)
        for (int q = 0; q < 4; ++q) {                       // quadrant (qi, qj): frequency rows 3qi..3qi+2, columns 3qj..3qj+2
            const int qi = q >> 1, qj = q & 1;
            for (int ch = 0; ch < cin / CK; ++ch)
                for (int ks = 0; ks < CK / 8; ++ks)
                    for (int k = 0; k < 9; ++k)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 4; ++j) {
                                const int co = 32 * cb + (l & 31);
                                const int ci = CK * ch + 8 * ks + 4 * (l >> 5) + j;
                                const int xi = (3 * qi + k / 3) * 6 + 3 * qj + k % 3;
                                dst[o++] = U[((size_t)co * cin + ci) * 36 + xi];
                            }
            for (int i = 0; i < 4 * 256; ++i) dst[o++] = 0.f;   // prefetch tail of this stream
        }
}

// STK (TW = 16, SRC_PLAIN, 16 x 16 images only - the bottom level of a 256 x 256 slice): the workgroup's 32 tiles are the
// 4 x 4 tiles of TWO consecutive slices stacked (tile rows 0..3 -> slice n0, 4..7 -> slice n0 + 1; two 18-row halo patches
// one above the other), so the 32-tile M-block is full instead of half padding.
// WN = 2: 8 waves, 64 output channels, 16-channel chunks, one workgroup per CU (the layout described above).
// WN = 1 (Cout = 32, the full-resolution layers): 4 waves = the 4 quadrants of ONE 32-channel group, 8-channel chunks, and the
// LDS cut to 79.8 KB (patch pixel stride 10 floats instead of 12) so that TWO workgroups share a CU.
template <int TW, int SRC, bool STK = false, int WN = 2>
__global__ __launch_bounds__(256 * WN, 2) void conv3x3_wino4_kernel(const ConvArgs a) {
    constexpr int CK = 8 * WN, CKP = CK + 4, PPP = CK / 4, NT_ = 256 * WN;
    constexpr int CKQ = WN == 2 ? CK + 4 : CK + 2;     // patch pixel stride (floats): b64 window reads conflict-free for both
    constexpr int HCN = CK / 2;                        // 2-channel units per pixel
    constexpr int TC = TW / 4, TR = 32 / TC;           // tiles per row / rows of tiles in the 32-tile M-block
    constexpr int TH = 4 * TR;
    constexpr int SUBH = STK ? TH / 2 + 2 : 0;         // rows of one slice's halo patch in the stacked layout
    constexpr int PH = STK ? 2 * SUBH : TH + 2, PW = TW + 2;
    constexpr int ITEMS = PH * PW * PPP;
    static_assert(!STK || (TW == 16 && SRC == SRC_PLAIN && WN == 2), "stacked slices: 16-wide tiles, plain source");
    constexpr int NIT = (ITEMS + NT_ - 1) / NT_;
    constexpr bool UP2 = SRC == SRC_UPCAT;
    constexpr int LH = TH / 2 + 3, LW = TW / 2 + 3;    // low-res region bound (rows, cols)
    constexpr int LITEMS = LH * LW * PPP;
    constexpr int NITL = (LITEMS + NT_ - 1) / NT_;
    constexpr int NRAW = (UP2 && NITL > NIT) ? NITL : NIT;
    constexpr int KSC = CK / 8;                        // k-steps per chunk
    constexpr int PAIRS = 9 * KSC;                     // (k-step, frequency) pairs per chunk and wave, 4 MFMAs each
    constexpr int PF = 3;                              // B fragments in flight (must divide PAIRS: slots line up across chunks)
    constexpr int PLANE = 32 * CKP;                    // floats per frequency plane of V
    static_assert(SRC == SRC_PLAIN || SRC == SRC_UPCAT, "pooled sources go through the pooled copy");
    static_assert(PAIRS % PF == 0 && 32 * HCN * 2 == NT_, "one (tile, 2-channel, column group) transform item per thread");
    static_assert(LH * LW * CKP <= 36 * PLANE, "the low-res region is parked in V's space");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const patch = smem;                             // [PH][PW][CKQ]
    float* const V = smem + ((PH * PW * CKQ + 3) & ~3);    // [36][32 tiles][CKP]
    float* const lowres = V;                               // UPCAT: [LH][LW][CKP] low-res source region; V is dead while a
                                                           // chunk is staged (between the loop-top barrier and the transform)
#ifdef PNP_STAMPS
    unsigned long long st_t0 = W4T(), st_setup = 0, st_loop0 = 0, st_commit = 0, st_trans = 0, st_mfma = 0, st_loop1 = 0, st_ew = 0, st_er = 0, st_tmp = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wid / WN, cg = wid % WN;                 // frequency quadrant, 32-channel group of this wave
    const int qi = q >> 1, qj = q & 1;
    const int hh = lane >> 5, li = lane & 31;

    // XCD-aware decode: the channel groups of one spatial tile run back to back on the same XCD (blocks b and b + 8 share
    // an XCD under round-robin placement - speed only), so the patch they all read is fetched into that L2 once.  (The
    // opposite affinity - one channel block per XCD, so that its share of the transformed weights, 9.4 MB per 256 -> 256
    // layer in all, stays L2-resident - measured 4-6 % SLOWER on every layer: the weights stream well from the Infinity Cache.)
    const int ny = a.Cout / (32 * WN);
    const int bid = blockIdx.x;
    const int grp = bid / (8 * ny), rem = bid % (8 * ny);
    const int cby = rem >> 3;                              // this workgroup's block of 32 * WN channels
    int bt = grp * 8 + (rem & 7);                          // spatial tile index
    if (bt >= a.tilesX * a.tilesY * (STK ? (a.N + 1) / 2 : a.N)) return;
    const int tx0 = (bt % a.tilesX) * TW;
    bt /= a.tilesX;
    const int ty0 = (bt % a.tilesY) * TH;
    const int n = STK ? 2 * (bt / a.tilesY) : bt / a.tilesY;               // first (or only) slice of this workgroup
    const int nsl = STK ? (n + 1 < a.N ? 2 : 1) : 1;                       // slices this workgroup holds
    const bool live0 = !(a.tact != nullptr && a.tact[n] > 0.5f);
    const bool live1 = STK && nsl == 2 && !(a.tact != nullptr && a.tact[n + 1] > 0.5f);
    if (!live0 && !live1) return;
    const int cb = cby * WN + cg;                          // this wave's 32-channel block
    const int nchunks = a.Cin / CK;

    const int Hs = a.H >> 1, Ws = a.W >> 1;
    const int ylo = UP2 ? (int)(a.rh * (float)(ty0 > 0 ? ty0 - 1 : 0)) : 0;
    const int xlo = UP2 ? (int)(a.rw * (float)(tx0 > 0 ? tx0 - 1 : 0)) : 0;
    const int nskip = UP2 ? a.Cskip / CK : 0;              // leading chunks that come straight from the skip tensor

    // Staging addresses are computed ONCE: every thread owns the same NIT patch items (pixel, 4 channels) in every chunk, so
    // per chunk only the channel offset changes - a scalar (soffset of a buffer load).  Pixels outside the image (the conv's
    // zero padding) and items past the patch get an out-of-range offset: the buffer's bounds check returns zeros, no
    // per-item branches.  (f32 MFMAs run on the SIMD's vector issue port - exp/mfma_valu_coexec.hip - so every VALU
    // instruction of the chunk loop is matrix time lost; the per-chunk index arithmetic used to cost as much as the input
    // transform.)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned OOB = 0x80000000u;
    const int cs = UP2 ? a.Cskip : a.Cin;
    const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.src0 + (size_t)n * a.H * a.W * cs), 0, nsl * a.H * a.W * cs * 4, 0x00020000);
    unsigned voff[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int idx = tid + k * NT_;
        const int part = idx % PPP, pp = idx / PPP;
        const int py = pp / PW, px = pp % PW;
        const int sub = STK ? py / SUBH : 0;                               // stacked: which of the two slices
        const int gy = STK ? py % SUBH - 1 : ty0 + py - 1, gx = tx0 + px - 1;
        voff[k] = (idx < ITEMS && sub < nsl && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                      ? (unsigned)((((sub * a.H + gy) * a.W + gx) * cs + part * 4) * 4) : OOB;
    }
    const int Cup = a.Cin - a.Cskip;
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(UP2 ? a.src1 + (size_t)n * Hs * Ws * Cup : a.src0), 0, UP2 ? Hs * Ws * Cup * 4 : 0, 0x00020000);
    unsigned voffL[UP2 ? NITL : 1];
    if constexpr (UP2) {
#pragma unroll
        for (int k = 0; k < NITL; ++k) {
            const int idx = tid + k * NT_;
            const int part = idx % PPP, pp = idx / PPP;
            const int sy = ylo + pp / LW, sx = xlo + pp % LW;
            voffL[k] = (idx < LITEMS && sy < Hs && sx < Ws) ? (unsigned)(((sy * Ws + sx) * Cup + part * 4) * 4) : OOB;
        }
    }
    const int ldst = ((tid / PPP) * CKQ + (tid % PPP) * 4);      // patch slot of item 0; item k is k * (NT_ / PPP) pixels further
    const int ldstL = ((tid / PPP) * CKP + (tid % PPP) * 4);     // the same in the low-res region (pixel stride CKP)
    auto put_patch = [&](int off, float4 v) {              // pixel stride 10 floats (WN = 1) is only 8-byte aligned
        if constexpr (CKQ % 4 == 0) *reinterpret_cast<float4*>(&patch[off]) = v;
        else { *reinterpret_cast<float2*>(&patch[off]) = make_float2(v.x, v.y); *reinterpret_cast<float2*>(&patch[off + 2]) = make_float2(v.z, v.w); }
    };

    float4 raw[NRAW];
    auto issue = [&](int c) {
        if (UP2 && c >= nskip) {                           // low-res region of an upsampled chunk
            const int soff = (c * CK - a.Cskip) * 4;
#pragma unroll
            for (int k = 0; k < NITL; ++k)
                raw[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc1, voffL[k], soff, 0));
            return;
        }
        const int soff = c * CK * 4;
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            raw[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc0, voff[k], soff, 0));
    };
    auto commit = [&](int c) {
        if (UP2 && c >= nskip) {
            // park the low-res region in LDS, then interpolate the patch from it (ATen upsample_bilinear2d,
            // align_corners=True: src = dst * (in-1)/(out-1), weights (1-l, l), noise.py:39,46)
#pragma unroll
            for (int k = 0; k < NITL; ++k) {
                const int idx = tid + k * NT_;
                if (idx < LITEMS) *reinterpret_cast<float4*>(&lowres[ldstL + k * (NT_ / PPP) * CKP]) = raw[k];
            }
            __syncthreads();
#pragma unroll 1                                           // LDS -> LDS, no prefetch registers involved: keep it a loop (registers)
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * NT_;
                const int part = idx % PPP, pp = idx / PPP;
                const int py = pp / PW, px = pp % PW;
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
                    put_patch((py * PW + px) * CKQ + part * 4, v);
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int idx = tid + k * NT_;
            if (idx < ITEMS) put_patch(ldst + k * (NT_ / PPP) * CKQ, raw[k]);
        }
    };
    issue(0);

    // this thread's transform item: tile tq of the workgroup's 32 tiles (TC per row), channels [2*hc, 2*hc+2), frequency
    // column group tj (first half of the waves: columns 0..2, second half: columns 3..5; all six rows).  HCN lanes cover a
    // tile's channels and a 32-lane ds_read_b64 group covers 4 tiles 80 floats apart (WN = 2) or 8 tiles 40 floats apart
    // (WN = 1): all 64 banks, conflict-free.
    const int hc = tid % HCN, tq = (tid / HCN) & 31;
    const int tj = wid / (2 * WN);
    const int trow = tq / TC;                                                  // tile row; stacked: rows 0..3 / 4..7 = slice 0 / 1
    const int wrow = STK ? (trow / (TR / 2)) * SUBH + 4 * (trow % (TR / 2)) : 4 * trow;
    const int win = (wrow * PW + 4 * (tq % TC)) * CKQ + 2 * hc;                // top-left of tile tq's 6x6 input window
    const int vout = tq * CKP + 2 * hc;                                        // this item's slot in every frequency plane

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    const size_t stream = (size_t)nchunks * PAIRS * 64 + 4 * 64;                   // float4 per (cb, quadrant) stream
    const float4* bptr = reinterpret_cast<const float4*>(a.wpack) + ((size_t)cb * 4 + q) * stream + lane;
    float4 bq[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) bq[p] = bptr[p * 64];
    const int aoff = ((3 * qi) * 6 + 3 * qj) * PLANE + li * CKP + 4 * hh;          // this lane's row of V[(3qi, 3qj)]

#ifdef PNP_STAMPS
    st_setup = W4T();
    st_loop0 = st_setup;
#endif
    for (int c = 0; c < nchunks; ++c) {
#ifdef PNP_STAMPS
        st_tmp = W4T();
#endif
        if (c > 0) __syncthreads();                        // MFMA phase of the previous chunk is done with V
        commit(c);
        __syncthreads();
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_commit += t - st_tmp; st_tmp = t; }
#endif

        // ---- input transform V = B^T d B: all six frequency rows of this thread's column group ------------------------------
        // WN = 1 (32 -> 32 layers: 4 short chunks, 36 MFMAs each): the next chunk's loads go out BEFORE the transform - a
        // chunk's MFMA phase alone (~2 us) is shorter than an HBM round trip under load
        if constexpr (WN == 1) { if (c + 1 < nchunks) issue(c + 1); }
        __builtin_amdgcn_sched_barrier(0);                 // WN = 2: keep the next chunk's loads (20 registers) behind the transform
        if (tj == 0) wino4_input_transform<0, PW, CKQ, PLANE>(patch + win, V + vout);
        else wino4_input_transform<1, PW, CKQ, PLANE>(patch + win, V + vout);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WN == 2) { if (c + 1 < nchunks) issue(c + 1); }   // next chunk's loads fly under this chunk's MFMAs
        __syncthreads();
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_trans += t - st_tmp; st_tmp = t; }
#endif

        // ---- 9 GEMMs per wave: pair p = (k-step, k); A from V (LDS), B from the packed U stream (L2), 4 MFMAs per pair -----
        const float4* bp = bptr + (size_t)c * PAIRS * 64;
        float4 a0 = *reinterpret_cast<const float4*>(&V[aoff]);
#pragma unroll
        for (int p = 0; p < PAIRS; ++p) {
            float4 a1;
            if (p + 1 < PAIRS) {
                const int ks1 = (p + 1) / 9, k1 = (p + 1) % 9;
                a1 = *reinterpret_cast<const float4*>(&V[((k1 / 3) * 6 + k1 % 3) * PLANE + aoff + 8 * ks1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int k = p % 9;
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bq[p % PF].x, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bq[p % PF].y, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bq[p % PF].z, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bq[p % PF].w, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bq[p % PF] = bp[(p + PF) * 64];                // refill the slot just read (tail zero-padded)
            if (p + 1 < PAIRS) a0 = a1;
        }
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_mfma += t - st_tmp; }
#endif
    }
#ifdef PNP_STAMPS
    st_loop1 = W4T();
#endif

    // ---- output transform Y = A^T M A,  A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a^2 a^2 b^2 b^2 0; 0 a^3 -a^3 b^3 -b^3 1].
    // Wave (qi, qj) holds M[3qi..3qi+2][3qj..3qj+2] and forms its partial tile  Yp = A^T[:, 3qi..] M_q A[3qj.., :]  (lane-local).
    // Four rounds of 4 accumulator registers (8 tiles): partials -> P[cg][q][rr][16 outputs][64 lanes] in LDS, barrier, then
    // thread (cg, rr, hh, 2x2 window, 4 channels) sums the four quadrants in the order q = 0..3 and finishes the pixels.
    float* const P = smem;
    constexpr int V4 = 8;                                   // float4 per 32-channel group
    constexpr int CGB = WN == 2 ? 1 : 0;                   // bits of the channel-group field in the reducer's thread index
    const int r_c4 = tid & 7, r_cg = (tid >> 3) & (WN - 1), r_hh = (tid >> (3 + CGB)) & 1, r_w = (tid >> (4 + CGB)) & 3,
              r_rr = tid >> (6 + CGB);
    const int cout0 = cby * (32 * WN) + r_cg * 32 + 4 * r_c4;
    const float4 bias4 = *reinterpret_cast<const float4*>(a.bias + cout0);
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    // one-dimensional partial output transform of three values (the quadrant's rows or columns) into four
    auto at4 = [&](int g, float m0, float m1, float m2, float* t) {
        if (g == 0) {            // frequencies 0, a, -a
            const float s = m1 + m2, d = m1 - m2;
            t[0] = m0 + s; t[1] = fA * d; t[2] = fA2 * s; t[3] = fA3 * d;
        } else {                 // frequencies b, -b, inf
            const float s = m0 + m1, d = m0 - m1;
            t[0] = s; t[1] = fB * d; t[2] = fB2 * s; t[3] = fmaf(fB3, d, m2);
        }
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#ifdef PNP_STAMPS
        st_tmp = W4T();
#endif
        __syncthreads();                                   // V (first round) / the previous round's partials are no longer read
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            float t[3][4];                                 // [kj][a']: rows transformed, per frequency column of the quadrant
#pragma unroll
            for (int kj = 0; kj < 3; ++kj) {
                float m0, m1, m2;
                m0 = acc[0 + kj][4 * g + rr]; m1 = acc[3 + kj][4 * g + rr]; m2 = acc[6 + kj][4 * g + rr];
                at4(qi, m0, m1, m2, t[kj]);
            }
            float* pw = P + ((((cg * 4 + q) * 4 + rr) * 16) * 64) + lane;
#pragma unroll
            for (int ay = 0; ay < 4; ++ay) {
                float y[4];
                at4(qj, t[0][ay], t[1][ay], t[2][ay], y);
#pragma unroll
                for (int bx = 0; bx < 4; ++bx) pw[(ay * 4 + bx) * 64] = y[bx];
            }
        }
        __syncthreads();
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_ew += t - st_tmp; st_tmp = t; }
#endif
        // reduce + finish: tile t = rr + 8g + 4hh of the M-block, window (wy, wx) of its 4x4 pixels
        {
            const int t = r_rr + 8 * g + 4 * r_hh;
            const int tr = t / TC;
            const int sub = STK ? tr / (TR / 2) : 0;                        // stacked: the slice this tile belongs to
            const bool live = STK ? (sub == 0 ? live0 : live1) : true;
            const int py = (STK ? 4 * (tr % (TR / 2)) : 4 * tr) + 2 * (r_w >> 1), px = 4 * (t % TC) + 2 * (r_w & 1);
            const float* pr = P + (((r_cg * 4) * 4 + r_rr) * 16) * 64 + r_hh * 32 + 4 * r_c4;
            float4 o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int oidx = (2 * (r_w >> 1) + (e >> 1)) * 4 + 2 * (r_w & 1) + (e & 1);
                float4 s = *reinterpret_cast<const float4*>(pr + oidx * 64);
#pragma unroll
                for (int qq = 1; qq < 4; ++qq) s = f4add(s, *reinterpret_cast<const float4*>(pr + (qq * 4 * 16 + oidx) * 64));
                s = f4add(s, bias4);
                o[e] = make_float4(fmaxf(s.x, kLeaky * s.x), fmaxf(s.y, kLeaky * s.y), fmaxf(s.z, kLeaky * s.z), fmaxf(s.w, kLeaky * s.w));
                const int gy = ty0 + py + (e >> 1), gx = tx0 + px + (e & 1);
                if constexpr (WN == 1) {
                    if (a.last_w != nullptr) {
                        // Fused last layer (noise.py:67,130-133,164): 1x1 conv 32 -> 1, + image channel, clamp.  The eight
                        // threads c4 = 0..7 of a pixel hold its 32 channels: partial dot products, xor-shuffled together; this
                        // conv's own 32-channel output is never written.
                        const float4 wv = *reinterpret_cast<const float4*>(a.last_w + 4 * r_c4);
                        float dsum = o[e].x * wv.x + o[e].y * wv.y + o[e].z * wv.z + o[e].w * wv.w;
                        dsum += __shfl_xor(dsum, 1); dsum += __shfl_xor(dsum, 2); dsum += __shfl_xor(dsum, 4);
                        if (r_c4 == 0 && gy < a.H && gx < a.W) {
                            const size_t qx = ((size_t)n * a.H + gy) * a.W + gx;
                            const float img = a.last_ximg != nullptr ? a.last_ximg[qx] : (a.last_z[qx].x - a.last_u[qx].x);
                            a.last_out[qx] = fminf(fmaxf(img + dsum + a.last_b[0], 0.f), 1.f);
                        }
                        continue;
                    }
                }
                if (live && gy < a.H && gx < a.W)
                    *reinterpret_cast<float4*>(a.dst + (((size_t)(n + sub) * a.H + gy) * a.W + gx) * a.Cout + cout0) = o[e];
            }
            if (a.pooled != nullptr) {                     // MaxPool2d(2) of this window for the next stage (noise.py:22-25)
                const int gy = (ty0 + py) >> 1, gx = (tx0 + px) >> 1;
                if (live && gy < Hp && gx < Wp)
                    *reinterpret_cast<float4*>(a.pooled + (((size_t)(n + sub) * Hp + gy) * Wp + gx) * a.Cout + cout0) =
                        f4max(f4max(o[0], o[1]), f4max(o[2], o[3]));
            }
        }
#ifdef PNP_STAMPS
        st_er += W4T() - st_tmp;
#endif
    }
    (void)V4;
#ifdef PNP_STAMPS
    {
        const int w = (int)blockIdx.x - (int)(gridDim.x / 2);
        if (tid == 0 && w >= 0 && w < W4_WGS && a.stamp_slot < W4_SLOTS) {
            unsigned long long* o = g_w4_stamps + ((size_t)a.stamp_slot * W4_WGS + w) * W4_N;
            const unsigned long long te = W4T();
            o[0] = 1; o[1] = st_setup - st_t0; o[2] = st_commit; o[3] = st_trans; o[4] = st_mfma; o[5] = st_loop1 - st_loop0;
            o[6] = st_ew; o[7] = st_er; o[8] = te - st_loop1; o[9] = te - st_t0; o[10] = (unsigned long long)nchunks;
        }
    }
#endif
}

template <int TW, int SRC, bool STK = false, int WN = 2>
static hipError_t launch_wino4_inst(const ConvArgs& a, const WinoPlan& p, hipStream_t s) {
    constexpr int CK = 8 * WN, CKP = CK + 4, CKQ = WN == 2 ? CK + 4 : CK + 2, TC = TW / 4, TR = 32 / TC, TH = 4 * TR;
    constexpr size_t patch_f = (((size_t)(STK ? TH + 4 : TH + 2) * (TW + 2) * CKQ + 3) / 4) * 4;
    constexpr size_t lds_main = (patch_f + (size_t)36 * 32 * CKP) * sizeof(float);
    constexpr size_t lds_out = (size_t)WN * 4 * 4 * 16 * 64 * sizeof(float);
    constexpr size_t lds = lds_main > lds_out ? lds_main : lds_out;
    static_assert(lds <= (WN == 2 ? 160 : 80) * 1024, "one (WN = 2) / two (WN = 1) workgroups per CU");
    auto kern = conv3x3_wino4_kernel<TW, SRC, STK, WN>;
    static DeviceOnce cap;
    if (hipError_t e = raise_lds_cap((const void*)kern, (int)lds, cap); e != hipSuccess) return e;
    const int ntiles = p.tiles_x * p.tiles_y * (STK ? (a.N + 1) / 2 : a.N), ny = a.Cout / (32 * WN);
    dim3 grid((unsigned)(((ntiles + 7) / 8) * 8 * ny));
    hipLaunchKernelGGL(kern, grid, dim3(256 * WN), lds, s, a);
    return hipGetLastError();
}

// `a.wpack` must be the F(4x4) pack (pack_winograd4_weights).
hipError_t launch_conv3x3_winograd4(const ConvArgs& a0, const WinoPlan& p, int src_mode, hipStream_t s) {
    if (!p.use || p.algo != 4 || a0.Cout % p.bn || a0.Cin % p.ck || (src_mode == SRC_UPCAT && a0.Cskip % p.ck)) return hipErrorInvalidValue;
    ConvArgs a = a0;
    a.tilesX = p.tiles_x;
    a.tilesY = p.tiles_y;
#ifdef PNP_STAMPS
    a.stamp_slot = g_w4_slot++;
#endif
    if (p.bn == 32) {                                      // Cout = 32: 4-wave workgroups, 8-channel chunks, two per CU
        if (p.ck != 8 || p.stack) return hipErrorInvalidValue;
        if (p.tw == 32 && src_mode == SRC_PLAIN) return launch_wino4_inst<32, SRC_PLAIN, false, 1>(a, p, s);
        if (p.tw == 32 && src_mode == SRC_UPCAT) return launch_wino4_inst<32, SRC_UPCAT, false, 1>(a, p, s);
        if (p.tw == 16 && src_mode == SRC_PLAIN) return launch_wino4_inst<16, SRC_PLAIN, false, 1>(a, p, s);
        if (p.tw == 16 && src_mode == SRC_UPCAT) return launch_wino4_inst<16, SRC_UPCAT, false, 1>(a, p, s);
        return hipErrorInvalidValue;
    }
    if (a.last_w != nullptr || p.ck != 16) return hipErrorInvalidValue;
    if (p.tw == 32) {
        if (src_mode == SRC_PLAIN) return launch_wino4_inst<32, SRC_PLAIN>(a, p, s);
        if (src_mode == SRC_UPCAT) return launch_wino4_inst<32, SRC_UPCAT>(a, p, s);
    } else if (p.tw == 16) {
        if (p.stack) return src_mode == SRC_PLAIN && a.H == 16 && a.W == 16 ? launch_wino4_inst<16, SRC_PLAIN, true>(a, p, s) : hipErrorInvalidValue;
        if (src_mode == SRC_PLAIN) return launch_wino4_inst<16, SRC_PLAIN>(a, p, s);
        if (src_mode == SRC_UPCAT) return launch_wino4_inst<16, SRC_UPCAT>(a, p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace pnp

#ifdef PNP_STAMPS
extern "C" int pnp_debug_stamps4_reset(void) { pnp::g_w4_slot = 0; return 0; }
extern "C" int pnp_debug_stamps4_read(unsigned long long* dst, int slots) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pnp::g_w4_stamps), (size_t)slots * W4_WGS * W4_N * sizeof(unsigned long long));
}
#endif
