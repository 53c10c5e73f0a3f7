// Winograd F(4x4, 3x3) convolution for the denoiser layers with Cin >= 32 (all 26 conv3x3 layers at the headline size) on gfx950
// (MI355X); the first of them also evaluates the 2 -> 32 input layer in its staging (SRC_FIRST), the last carries the 1x1 output
// layer in its epilogue.
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
// Workgroup = 8 waves on 32 tiles (512 output pixels) x 64 output channels; wave (th, cq) OWNS 16 tiles x 16 channels for
// ALL 36 frequencies: 36 accumulators of v_mfma_f32_16x16x4_f32 (144 registers), two waves per SIMD.  Per 16-channel chunk:
//   1. stage the (TH+2) x (TW+2) halo patch in LDS (loads of chunk c+1 in flight during chunk c's MFMAs; upsampled chunks
//      come from the tile's low-res source region parked in LDS, like the F(2x2) kernel)
//   2. input transform: thread (tile, 2 channels, frequency-column group) reads the 6x5 part of the 6x6 window its three
//      frequency columns need and writes their 18 frequency planes V[xi][tile][channel] (no value is computed twice)
//   3. 36 independent GEMMs per wave: acc[xi] += V[xi] (A fragment, one ds_read_b128 per frequency, swizzled so that the
//      16-lane read groups are conflict-free without padding) x U[xi] (B fragment, transformed weights streamed from L2 in
//      pre-packed per-lane order)
// Output transform Y = A^T M A is LANE-LOCAL: a lane holds all 36 frequencies of its 4 tiles x 1 channel, so there is no
// exchange through LDS and no barrier after the MFMA loop (the previous generation split the frequencies over four waves and
// met in LDS: 256 ds_write_b32 per lane, ~15k cycles per workgroup - 40 % of a 32 -> 32 layer's lifetime,
// profiles/r02_wino4_stamps.md).  + bias, LeakyReLU, NHWC stores (16 consecutive channels = 64 bytes per pixel and wave, the
// neighbouring wave writes the other half of the line) and the 2x2 max-pooled copy for the next stage from the same registers.
//
// Three schedules of the same arithmetic (winograd_plan picks per layer):
//   conv3x3_wino4_kernel<.., WN = 2, MT = 32>  8 waves in step (barrier - transform - barrier - MFMAs), one workgroup per CU
//   conv3x3_wino4p_kernel                      8 waves, the two tile halves half a chunk apart: one wave of every SIMD is in its
//                                              MFMA phase while the other transforms (plain layers with Cin >= 128)
//   conv3x3_wino4_kernel<.., MT = 16 / WN = 1> 4-wave workgroups (16 tiles x 64 channels / 32 tiles x 32 channels), two
//                                              independent ones per CU (upsample + concat layers, Cin <= 64, Cout = 32)
#include "pnp_internal.h"
#include "conv_staging.h"
#include <type_traits>

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

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// ---- host: U = G g G^T (6x6 per (cout, cin)), packed [cout/16][chunk][xi 36][lane 64][CK/4] (+ a prefetch tail per stream) ---
// Lane l of a v_mfma_f32_16x16x4_f32 B operand holds U[xi][cin = CK*chunk + (CK/4)*(l/16) + s][cout = 16*cb + l%16] for the
// MFMA number s = 0..CK/4-1 of the frequency.
constexpr int kW4Tail = 8 * 64 * 4;                     // floats: the B ring reads up to 6 fragments past the last chunk
size_t winograd4_pack_floats(int cin, int cout) { return (size_t)cout * cin * 36 + (size_t)(cout / 16) * kW4Tail; }

void pack_winograd4_weights(const float* oihw, int cin, int cout, int ck, float* dst) {
    const int CK = ck, S = CK / 4;                      // 16 (64-channel workgroups) or 8 (32-channel workgroups)
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
    for (int cb = 0; cb < cout / 16; ++cb) {
        for (int ch = 0; ch < cin / CK; ++ch)
            for (int xi = 0; xi < 36; ++xi)
                for (int l = 0; l < 64; ++l)
                    for (int sidx = 0; sidx < S; ++sidx) {
                        const int co = 16 * cb + (l & 15);
                        const int ci = CK * ch + S * (l >> 4) + sidx;
                        dst[o++] = U[((size_t)co * cin + ci) * 36 + xi];
                    }
        for (int i = 0; i < kW4Tail; ++i) dst[o++] = 0.f;       // prefetch tail of this stream
    }
}

// The F(4x4) kernels' interpolation assumes that output row g of an exact x2 bilinear upsample (align_corners=True) to H rows reads
// the source lines floor((g - 1) / 2) and the next one (clamped), with a non-zero weight on no other line - true in exact arithmetic;
// here it is checked against the float32 product rh * g that ATen (and these kernels' weight tables) evaluate.  True for every even
// H up to 1024 (tools: python check in DESIGN.md); winograd_plan plans an upsample + concat layer on F(4x4) only where it holds.
bool upsample_lines_regular(int H) {
    if (H < 2 || H % 2) return false;
    const int Hs = H / 2;
    const float rh = (float)(Hs - 1) / (float)(H - 1);
    for (int g = 0; g < H; ++g) {
        const float sc = rh * (float)g;
        const int i0 = (int)sc;
        const int i1 = i0 + (i0 < Hs - 1 ? 1 : 0);
        const float l = sc - (float)i0;
        const int s0 = g >= 1 ? (g - 1) / 2 : -1;
        if (l < 1.f && i0 != s0 && i0 != s0 + 1) return false;       // weight 1 - l on line i0
        if (l > 0.f && i1 != s0 && i1 != s0 + 1) return false;       // weight l on line i1
    }
    return true;
}

// ---- blockIdx -> (channel block, spatial tile).  Blocks b and b + 8 share an XCD under round-robin placement (speed only, never
// correctness).  Both orders run the channel blocks of one spatial tile back to back on ONE XCD, so the patch they all read comes
// into that L2 once.  order 0 (rounds 1-2): XCD k takes the spatial tiles k, k + 8, k + 16, ... - a tile's neighbours sit on other
// XCDs and every halo row / column is fetched from the fabric twice.  order 1: XCD k takes the CONTIGUOUS range of tiles
// [k * ceil(ntiles / 8), ...), in image order, so consecutive workgroups of an XCD are neighbours in the image and their shared halo
// pixels are L2 hits.
__device__ __forceinline__ void wino4_decode(int bid, int ny, int ntiles, int order, int& cby, int& bt) {
    if (order == 0) {
        const int grp = bid / (8 * ny), rem = bid % (8 * ny);
        cby = rem >> 3;
        bt = grp * 8 + (rem & 7);
    } else {
        const int xcd = bid & 7, slot = bid >> 3;          // slot: position in this XCD's queue
        const int tpx = (ntiles + 7) >> 3;                 // tiles per XCD
        const int lt = slot / ny;
        cby = slot - lt * ny;
        bt = lt < tpx ? xcd * tpx + lt : ntiles;           // (grid padding)
    }
}

// ---- output transform Y = A^T M A,  A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a^2 a^2 b^2 b^2 0; 0 a^3 -a^3 b^3 -b^3 1],
// lane-local: accumulator register r of acc[6i + j] is M[i][j] of tile 16 th + 4 tg + r, channel 16 cb + cl.  Two tiles at a
// time in packed f32 (registers (0,1) and (2,3) of an accumulator are aligned pairs).  + bias, LeakyReLU, NHWC stores (16
// consecutive channels = 64 bytes per pixel and wave), the 2x2 max-pooled copy for the next stage from the same registers.
// FUSE (the 32-channel variant, 4 waves = 2 tile halves x 2 channel groups) and a.last_w set: the denoiser's last layer (1x1 conv
// 32 -> 1, + image channel, clamp; noise.py:67,130-133,164) is applied here and this conv's own 32-channel output is never
// written.  A pixel's 32 channels sit in the 16 lanes of a DPP row (x 2 waves): reduce-scatter over the row - four steps of
// "keep the half of the values my lane bit selects, add the partner's" with row_mirror / row_half_mirror / quad_perm, after
// which lane cl holds pixel cl of each of its 4 tiles - then the two channel-group waves meet in LDS (`xch`, 2 KB).
template <int CTRL>
__device__ __forceinline__ float dpp_xchg(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int H, int CTRL>
__device__ __forceinline__ void row_reduce_scatter_step(v2f* P, bool bit) {
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const v2f lo = P[i], hi = P[i + H];
        const v2f keep = bit ? hi : lo, send = bit ? lo : hi;
        P[i] = v2f{keep.x + dpp_xchg<CTRL>(send.x), keep.y + dpp_xchg<CTRL>(send.y)};
    }
}

template <int TW, bool STK, int MT = 32, bool FUSE = false>
__device__ __forceinline__ void wino4_epilogue(const ConvArgs& a, const f32x4 (&acc)[36], int n, int th, int tg, int cl, int cb,
                                               int tx0, int ty0, bool live, float* xch = nullptr) {
    constexpr int TC = TW / 4, TR = MT / TC;
    constexpr unsigned OOB = 0x80000000u;
    const int cout0 = cb * 16 + cl;
    const float bias = a.bias[cout0];
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    auto at6 = [&](v2f m0, v2f m1, v2f m2, v2f m3, v2f m4, v2f m5, v2f* t) {
        const v2f s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
        t[0] = m0 + s1 + s2;
        t[1] = fA * d1 + fB * d2;
        t[2] = fA2 * s1 + fB2 * s2;
        t[3] = fA3 * d1 + fB3 * d2 + m5;
    };
    if (!live) return;                                     // stacked: wave th works on slice n + th, which may have stopped
    const bool fused = FUSE && a.last_w != nullptr;        // (uniform over the workgroup)
#ifdef PNP_DIAG
    const bool nostore = (a.diag & 1) != 0;
#else
    constexpr bool nostore = false;
#endif
    v2f part[2];                                           // fused: pixel cl of tiles 4 tg + (0,1), (2,3), summed over this wave's channels
    const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.dst + (size_t)(n + (STK ? th : 0)) * a.H * a.W * a.Cout), 0, nostore ? 0 : a.H * a.W * a.Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rpool = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.pooled != nullptr ? a.pooled + (size_t)(n + (STK ? th : 0)) * Hp * Wp * a.Cout : a.dst), 0,
        (a.pooled != nullptr && !nostore) ? Hp * Wp * a.Cout * 4 : 0, 0x00020000);
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        v2f T[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            v2f m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = v2f{acc[6 * i + j][2 * rp], acc[6 * i + j][2 * rp + 1]};
            v2f t[4];
            at6(m[0], m[1], m[2], m[3], m[4], m[5], t);
#pragma unroll
            for (int ay = 0; ay < 4; ++ay) T[ay][j] = t[ay];
        }
        v2f Y[4][4];
#pragma unroll
        for (int ay = 0; ay < 4; ++ay) {
            at6(T[ay][0], T[ay][1], T[ay][2], T[ay][3], T[ay][4], T[ay][5], Y[ay]);
#pragma unroll
            for (int ax = 0; ax < 4; ++ax) {
                const v2f sv = Y[ay][ax] + bias;
                Y[ay][ax] = v2f{fmaxf(sv.x, kLeaky * sv.x), fmaxf(sv.y, kLeaky * sv.y)};
            }
        }
        if constexpr (FUSE) {
            if (fused) {
                const float wl = a.last_w[cout0];
                v2f P[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) P[i] = Y[i >> 2][i & 3] * wl;
                row_reduce_scatter_step<8, 0x140>(P, (cl & 8) != 0);           // row_mirror: lane ^ 15
                row_reduce_scatter_step<4, 0x141>(P, (cl & 4) != 0);           // row_half_mirror: lane ^ 7
                row_reduce_scatter_step<2, 0x1B>(P, (cl & 2) != 0);            // quad_perm [3,2,1,0]: lane ^ 3
                row_reduce_scatter_step<1, 0xB1>(P, (cl & 1) != 0);            // quad_perm [1,0,3,2]: lane ^ 1
                part[rp] = P[0];
                continue;
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = 16 * th + 4 * tg + 2 * rp + e;                       // tile of the M-block
            const int tr = t / TC;
            const int py = STK ? 4 * (tr % (TR / 2)) : 4 * tr, px = 4 * (t % TC);
            const int gy = ty0 + py, gx = tx0 + px;                            // top-left pixel of the tile
            // Bounds without per-store compares: W is a multiple of 4 (winograd_plan admits no other width: the column step is a
            // scalar offset the range check does not see), so a 4-wide tile is inside or outside the image as a WHOLE in
            // x (one select per tile); its rows travel in the vector offset, so a row past H is past the descriptor's num_records
            // (one slice) and the store is dropped by the range check; only the column step is a scalar offset.
            const unsigned base = (gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.Cout + cout0) * 4) : OOB;
            const unsigned rowb = (unsigned)(a.W * a.Cout * 4);
#pragma unroll
            for (int ay = 0; ay < 4; ++ay)
#pragma unroll
                for (int ax = 0; ax < 4; ++ax) {
                    const float val = e == 0 ? Y[ay][ax].x : Y[ay][ax].y;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rdst, base + ay * rowb, ax * a.Cout * 4, 0);
                }
            if (a.pooled != nullptr) {                 // MaxPool2d(2) of the tile's four windows (noise.py:22-25)
                const int qy = gy >> 1, qx = gx >> 1;
                const unsigned pbase = (qx < Wp) ? (unsigned)(((qy * Wp + qx) * a.Cout + cout0) * 4) : OOB;
                const unsigned prowb = (unsigned)(Wp * a.Cout * 4);
#pragma unroll
                for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                    for (int wx = 0; wx < 2; ++wx) {
                        const v2f mx = __builtin_elementwise_max(__builtin_elementwise_max(Y[2 * wy][2 * wx], Y[2 * wy][2 * wx + 1]),
                                                                 __builtin_elementwise_max(Y[2 * wy + 1][2 * wx], Y[2 * wy + 1][2 * wx + 1]));
                        const float val = e == 0 ? mx.x : mx.y;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rpool, pbase + wy * prowb, wx * a.Cout * 4, 0);
                    }
            }
        }
    }
    if constexpr (FUSE) {
        if (fused) {
            const int cq = cb & 1, lane = 16 * tg + cl;
            float4* const x4 = reinterpret_cast<float4*>(xch) + th * 64 + lane;
            if (cq == 1) *x4 = make_float4(part[0].x, part[0].y, part[1].x, part[1].y);
            __syncthreads();                               // (patch space: nobody reads it after the last chunk's transform)
            if (cq == 0) {
                const float4 o = *x4;
                const float sum[4] = {part[0].x + o.x, part[0].y + o.y, part[1].x + o.z, part[1].y + o.w};
                const float lb = a.last_b[0];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 16 * th + 4 * tg + r;
                    const int gy = ty0 + 4 * (t / TC) + (cl >> 2), gx = tx0 + 4 * (t % TC) + (cl & 3);
                    if (gy < a.H && gx < a.W && !nostore) {
                        const size_t qx = ((size_t)n * a.H + gy) * a.W + gx;
                        const float img = a.last_ximg != nullptr ? a.last_ximg[qx] : (a.last_z[qx].x - a.last_u[qx].x);
                        a.last_out[qx] = fminf(fmaxf(img + sum[r] + lb, 0.f), 1.f);
                    }
                }
            }
        }
    }
}

// STK (TW = 16, SRC_PLAIN, 16 x 16 images only - the bottom level of a 256 x 256 slice): the workgroup's 32 tiles are the
// 4 x 4 tiles of TWO consecutive slices stacked (tile rows 0..3 -> slice n0, 4..7 -> slice n0 + 1; two 18-row halo patches
// one above the other), so the 32-tile M-block is full instead of half padding.
// WN = 2: 8 waves, 64 output channels, 16-channel chunks, one workgroup per CU (the layout described above).
// WN = 1 (Cout = 32, the full-resolution layers): 4 waves = 2 tile halves x 2 groups of 16 channels, 8-channel chunks (two
// MFMAs per frequency, b64 fragments), 61 KB of LDS so that TWO workgroups share a CU.
// MT = 16 (with WN = 2): the M-block is 16 tiles - 4 waves, one per 16-channel group, 64-75 KB of LDS - so that TWO independent
// workgroups share a CU and one's transforms, barriers, interpolation and epilogue run under the other's MFMAs without any
// lockstep between them (the upsample + concat layers; 16 x 16 images need no stacking: 4 x 4 tiles are one slice).
template <int TW, int SRC, bool STK = false, int WN = 2, int MT = 32>
__global__ __launch_bounds__(8 * MT * WN, 2) void conv3x3_wino4_kernel(const ConvArgs a) {
    constexpr int CK = 8 * WN, CKP = CK, LP = CK, PPP = CK / 4, NT_ = 8 * MT * WN;   // CKP / LP: pixel stride of V / the low-res region (no
                                                       // padding: neighbouring output pixels interpolate from the same source pixels - broadcasts)
    constexpr int CKQ = WN == 2 ? CK + 4 : CK + 2;     // patch pixel stride (floats): b64 window reads conflict-free for both
    constexpr int HCN = CK / 2;                        // 2-channel units per pixel
    constexpr int TC = TW / 4, TR = MT / TC;           // tiles per row / rows of tiles in the MT-tile M-block
    constexpr int TH = 4 * TR;
    constexpr int SUBH = STK ? TH / 2 + 2 : 0;         // rows of one slice's halo patch in the stacked layout
    constexpr int PH = STK ? 2 * SUBH : TH + 2, PW = TW + 2;
    constexpr int ITEMS = PH * PW * PPP;
    static_assert(!STK || (TW == 16 && SRC == SRC_PLAIN && WN == 2 && MT == 32), "stacked slices: 16-wide tiles, plain source");
    static_assert(MT == 32 || (MT == 16 && WN == 2), "16-tile M-blocks: 64-channel workgroups only");
    constexpr int NIT = (ITEMS + NT_ - 1) / NT_;
    constexpr bool UP2 = SRC == SRC_UPCAT;
    constexpr bool FIRST = SRC == SRC_FIRST;           // the patch is the denoiser's FIRST layer, computed here (see first_patch)
    constexpr int LH = TH / 2 + 3, LW = TW / 2 + 3;    // low-res region (rows, cols): rows [ty0 / 2 - 1, ...) - see interpolate() -, cols from xlo
    constexpr int LITEMS = LH * LW * PPP;
    constexpr int NITL = (LITEMS + NT_ - 1) / NT_;
    // SPLIT (64-channel upsample + concat variants): a skip chunk's NIT staging loads go out in TWO half-batches - the first at the
    // start of the previous chunk's MFMA phase, parked half way through it, where the second goes out - so that only NIT / 2
    // staging registers are live beside the 144 accumulators and the B ring (the whole batch held across the phase spilled 22
    // registers into scratch inside the chunk loop)
    constexpr bool SPLIT = UP2 && WN == 2;
    constexpr int NH0 = SPLIT ? (NIT + 1) / 2 : NIT;       // items of the first half-batch (all of them without SPLIT)
    constexpr int NRAW = SPLIT ? (NH0 > NITL ? NH0 : NITL) : ((UP2 && NITL > NIT) ? NITL : NIT);
    constexpr int S = CK / 4;                          // MFMAs (4 channels each) per frequency and chunk
    constexpr int PF = WN == 2 ? 9 : 12;              // B fragments in flight in the MFMA loop (must divide 36: slots line up
                                                       // across chunks; the 16-tile upsample variant spilled 22 registers into scratch
                                                       // INSIDE its chunk loop with a ring of 9) ...
    constexpr int KEEP = WN == 1 ? 6 : 3;              // ... of which only the first KEEP are loaded across the chunk boundary (the
                                                       // input transform needs the registers; the 32-channel variant's 8-byte fragments
                                                       // leave room for 6: -1 % on the level-0 layers); the rest go out after the transform
    constexpr int PLANE = MT * CKP;                    // floats per frequency plane of V
    static_assert(SRC == SRC_PLAIN || SRC == SRC_UPCAT || SRC == SRC_FIRST, "pooled sources go through the pooled copy");
    static_assert(!FIRST || (WN == 1 && MT == 32), "fused first layer: the 32-channel variant");
    static_assert(36 % PF == 0 && MT * HCN * 2 == NT_, "one (tile, 2-channel, column group) transform item per thread");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const patch = smem;                             // [PH][PW][CKQ]
    float* const V = smem + ((PH * PW * CKQ + 3) & ~3);    // [36][MT tiles][CKP]
    float* const lowres = V + 36 * PLANE;                  // UPCAT: [LH][LW][LP] low-res source region of the chunk being staged
    float* const rowT = lowres + LH * LW * LP;             // UPCAT: per patch row the weights of its three candidate source lines; per patch
    float* const colT = rowT + 4 * PH;                     // column {offsets of its two source columns in the region (int), their weights}
    // FIRST: image halo tile (2 pixels around the patch... 1 around each patch pixel's 3x3 window), its in-image mask, and the
    // first layer's weights [18 taps][32 channels] (sigma taps pre-scaled by sigma), bias, and the sigma taps' per-channel sum
    constexpr int IW = PW + 3;                             // row stride of the image tile (PW + 2 used)
    float* const dimg = V + 36 * PLANE;                    // [PH + 2][IW]
    float* const mimg = dimg + (PH + 2) * IW;
    float* const wl = mimg + (PH + 2) * IW;                // [18][32]
    float* const bl = wl + 18 * 32;                        // [32] bias, [32] bias + sigma-plane constant (interior pixels)
#ifdef PNP_STAMPS
    unsigned long long st_t0 = W4T(), st_setup = 0, st_loop0 = 0, st_commit = 0, st_trans = 0, st_mfma = 0, st_loop1 = 0, st_ew = 0, st_er = 0, st_tmp = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cq = wid / (MT / 16), th = wid % (MT / 16);  // this wave's 16-channel group and 16-tile part of the M-block
    const int tg = lane >> 4, cl = lane & 15;              // MFMA lane = (k / row group tg, column / row cl)

    // XCD-aware decode: the channel groups of one spatial tile run back to back on the same XCD (blocks b and b + 8 share
    // an XCD under round-robin placement - speed only), so the patch they all read is fetched into that L2 once.  (The
    // opposite affinity - one channel block per XCD, so that its share of the transformed weights, 9.4 MB per 256 -> 256
    // layer in all, stays L2-resident - measured 4-6 % SLOWER on every layer: the weights stream well from the Infinity Cache.)
    const int ny = a.Cout / (32 * WN);
    const int bid = blockIdx.x;
    const int ntl = a.tilesX * a.tilesY * (STK ? (a.N + 1) / 2 : a.N);
    int cby, bt;                                           // this workgroup's block of 32 * WN channels, its spatial tile index
    wino4_decode(bid, ny, ntl, a.order, cby, bt);
    if (bt >= ntl) return;
    const int tx0 = (bt % a.tilesX) * TW;
    bt /= a.tilesX;
    const int ty0 = (bt % a.tilesY) * TH;
    const int n = STK ? 2 * (bt / a.tilesY) : bt / a.tilesY;               // first (or only) slice of this workgroup
    const int nsl = STK ? (n + 1 < a.N ? 2 : 1) : 1;                       // slices this workgroup holds
    const bool live0 = !(a.tact != nullptr && a.tact[n] > 0.5f);
    const bool live1 = STK && nsl == 2 && !(a.tact != nullptr && a.tact[n + 1] > 0.5f);
    if (!live0 && !live1) return;
    const int cb = cby * (2 * WN) + cq;                    // this wave's 16-channel block
    const int nchunks = a.Cin / CK;

    const int Hs = a.H >> 1, Ws = a.W >> 1;
    const int ylo = UP2 ? ty0 / 2 - 1 : 0;                 // first low-res row of the parked region (-1 at the top of the image: a zero row)
    const int xlo = UP2 ? (int)(a.rw * (float)(tx0 > 0 ? tx0 - 1 : 0)) : 0;
    const int nskip = UP2 ? a.Cskip / CK : 0;              // leading chunks that come straight from the skip tensor

    // Staging addresses are computed ONCE: every thread owns the same NIT patch items (pixel, 4 channels) in every chunk, so
    // per chunk only the channel offset changes - a scalar (soffset of a buffer load).  Pixels outside the image (the conv's
    // zero padding) and items past the patch get an out-of-range offset: the buffer's bounds check returns zeros, no
    // per-item branches.  (f32 MFMAs run on the SIMD's vector issue port - exp/mfma_valu_coexec.hip - so every VALU
    // instruction of the chunk loop is matrix time lost; the per-chunk index arithmetic used to cost as much as the input
    // transform.)
    constexpr unsigned OOB = 0x80000000u;
    const int cs = UP2 ? a.Cskip : a.Cin;
#ifdef PNP_DIAG
    const bool hot = (a.diag & 2) != 0;                    // timing only: every workgroup stages tile 0 of slice 0
#else
    constexpr bool hot = false;
#endif
    const int ly0 = hot ? 0 : ty0, lx0 = hot ? 0 : tx0, ln = hot ? 0 : n;
    const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.src0 + (size_t)ln * a.H * a.W * cs), 0, nsl * a.H * a.W * cs * 4, 0x00020000);
    unsigned voff[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int idx = tid + k * NT_;
        const int part = idx % PPP, pp = idx / PPP;
        const int py = pp / PW, px = pp % PW;
        const int sub = STK ? py / SUBH : 0;                               // stacked: which of the two slices
        const int gy = STK ? py % SUBH - 1 : ly0 + py - 1, gx = lx0 + px - 1;
        voff[k] = (idx < ITEMS && sub < nsl && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                      ? (unsigned)((((sub * a.H + gy) * a.W + gx) * cs + part * 4) * 4) : OOB;
    }
    const int Cup = a.Cin - a.Cskip;
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(UP2 ? a.src1 + (size_t)ln * Hs * Ws * Cup : a.src0), 0, UP2 ? Hs * Ws * Cup * 4 : 0, 0x00020000);
    unsigned voffL[UP2 ? NITL : 1];
    if constexpr (UP2) {
#pragma unroll
        for (int k = 0; k < NITL; ++k) {
            const int idx = tid + k * NT_;
            const int part = idx % PPP, pp = idx / PPP;
            const int sy = (hot ? -1 : ylo) + pp / LW, sx = (hot ? 0 : xlo) + pp % LW;
            voffL[k] = (idx < LITEMS && sy >= 0 && sy < Hs && sx < Ws) ? (unsigned)(((sy * Ws + sx) * Cup + part * 4) * 4) : OOB;
        }
    }
    const int ldst = ((tid / PPP) * CKQ + (tid % PPP) * 4);      // patch slot of item 0; item k is k * (NT_ / PPP) pixels further
    const int ldstL = ((tid / PPP) * LP + (tid % PPP) * 4);      // the same in the low-res region (pixel stride LP)
    auto put_patch = [&](int off, float4 v) {              // pixel stride 10 floats (WN = 1) is only 8-byte aligned
        if constexpr (CKQ % 4 == 0) *reinterpret_cast<float4*>(&patch[off]) = v;
        else { *reinterpret_cast<float2*>(&patch[off]) = make_float2(v.x, v.y); *reinterpret_cast<float2*>(&patch[off + 2]) = make_float2(v.z, v.w); }
    };

    float4 raw[NRAW];
    // half: 0 = first half-batch (without SPLIT: everything), 1 = second half-batch of a skip chunk (SPLIT only)
    auto issue = [&](int c, int half = 0) {
        if (UP2 && c >= nskip) {                           // low-res region of an upsampled chunk (one batch, NITL <= NRAW items)
            if (half != 0) return;
            const int soff = (c * CK - a.Cskip) * 4;
#pragma unroll
            for (int k = 0; k < NITL; ++k)
                raw[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc1, voffL[k], soff, 0));
            return;
        }
        const int soff = c * CK * 4;
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < NH0; ++k)
                raw[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc0, voff[k], soff, 0));
        } else {
#pragma unroll
            for (int k = NH0; k < NIT; ++k)
                raw[k - NH0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc0, voff[k], soff, 0));
        }
    };
    // registers -> LDS: the patch itself (plain chunks) or the low-res source region (upsampled chunks).  Runs right after a
    // wave's MFMA phase of the previous chunk - the patch and the low-res region are dead from the post-transform barrier on -
    // so it overlaps the partner waves' MFMAs and needs no barrier of its own.
    auto park = [&](int c, int half = 0) {
        if (UP2 && c >= nskip) {
            if (half != 0) return;
#pragma unroll
            for (int k = 0; k < NITL; ++k) {
                const int idx = tid + k * NT_;
                if (idx < LITEMS) *reinterpret_cast<float4*>(&lowres[ldstL + k * (NT_ / PPP) * LP]) = raw[k];
            }
            return;
        }
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < NH0; ++k) {
                const int idx = tid + k * NT_;
                if (idx < ITEMS) put_patch(ldst + k * (NT_ / PPP) * CKQ, raw[k]);
            }
        } else {
#pragma unroll
            for (int k = NH0; k < NIT; ++k) {
                const int idx = tid + k * NT_;
                if (idx < ITEMS) put_patch(ldst + k * (NT_ / PPP) * CKQ, raw[k - NH0]);
            }
        }
    };
    // upsampled chunk: interpolate the patch from the parked low-res region (ATen upsample_bilinear2d, align_corners=True:
    // src = dst * (in-1)/(out-1), weights (1-l, l), noise.py:39,46): out = wy0 (wx0 p[i0][c0] + wx1 p[i0][c1]) + wy1 (... p[i1] ...).
    // SEPARABLE (round 4): a lane owns one (patch column, 4-channel piece) and the rows of its row group.  It first forms
    // t[k] = wx0 p[k][c0] + wx1 p[k][c1] for the low-res lines its rows touch - every read at a compile-time offset from one
    // address, all in flight together - and then each patch row as a weighted sum of two of them: for an exact x2 upsample the
    // source lines of output row g >= 1 are floor((g - 1) / 2) and the next (winograd_plan checks that the float product
    // rh * g ATen evaluates agrees for every row of this image height - it does for every even height up to 1024), so patch row y
    // of a tile whose first row is even touches lines s, s + 1 with s = ty0 / 2 - 1 + (y >> 1): two compile-time registers; the
    // per-row table holds their weights, filled from the float formula.  Same products and sums per output value as
    // interpolating each piece from its four source pixels, ~7 packed vector instructions per output piece instead of 27 and
    // three LDS round trips per chunk instead of two per piece (the f32 MFMAs share the vector issue port: interpolation was
    // 0.20-0.25 of these kernels' lifetime, profiles/r03_wino4_stamps.md).  PW x PPP lanes of IWPG waves work per row group, ING
    // groups split the rows.
    constexpr int IPAIRS = PW * PPP, IWPG = (IPAIRS + 63) / 64, ING = (NT_ / 64) / IWPG;
    constexpr int IRG = (((PH + ING - 1) / ING) + 1) & ~1;                // rows per group, even: (y >> 1) splits into group and row part
    constexpr int IK = IRG / 2 + 1;                                        // low-res lines a row group touches
    static_assert(!UP2 || (ING >= 1 && (PPP & (PPP - 1)) == 0 && TH % 2 == 0), "row groups of whole waves; even tile origin");
    static_assert(!UP2 || ((PH - 1) / 2 + 1 < LH), "the parked region holds every candidate line");
    auto interpolate = [&]() {
        const int ig = wid / IWPG;                         // this wave's row group (wave-uniform)
        const int it = tid - ig * (IWPG * 64);
        if (ig >= ING || it >= IPAIRS) return;
        const int x = it / PPP, part = it % PPP;
        const int yb = ig * IRG;
        const float4 ct = *reinterpret_cast<const float4*>(&colT[4 * x]);
        const float* const l0 = &lowres[__float_as_int(ct.x) + part * 4 + (yb / 2) * (LW * LP)];
        const float* const l1 = &lowres[__float_as_int(ct.y) + part * 4 + (yb / 2) * (LW * LP)];
        const v2f wx0 = v2f{ct.z, ct.z}, wx1 = v2f{ct.w, ct.w};
        v2f tl[IK], th_[IK];
#pragma unroll
        for (int k = 0; k < IK; ++k) {                     // wx0 * p[line][c0] + wx1 * p[line][c1], 4 channels
            if ((yb / 2 + k) < LH) {                       // (wave-uniform; lines past the region belong to rows past the patch)
                const float4 pa = *reinterpret_cast<const float4*>(l0 + k * (LW * LP)), pb = *reinterpret_cast<const float4*>(l1 + k * (LW * LP));
                tl[k] = __builtin_elementwise_fma(wx0, v2f{pa.x, pa.y}, wx1 * v2f{pb.x, pb.y});
                th_[k] = __builtin_elementwise_fma(wx0, v2f{pa.z, pa.w}, wx1 * v2f{pb.z, pb.w});
            } else { tl[k] = v2f{0.f, 0.f}; th_[k] = v2f{0.f, 0.f}; }
        }
        const int pdst = ((yb * PW + x) * CKQ) + part * 4;
#pragma unroll
        for (int j = 0; j < IRG; ++j) {
            if (yb + j < PH) {                             // (wave-uniform)
                const float4 rw = *reinterpret_cast<const float4*>(&rowT[4 * (yb + j)]);
                const v2f wa = v2f{rw.x, rw.x}, wb = v2f{rw.y, rw.y};
                const int k = j >> 1;
                const v2f olo = __builtin_elementwise_fma(wa, tl[k], wb * tl[k + 1]);
                const v2f ohi = __builtin_elementwise_fma(wa, th_[k], wb * th_[k + 1]);
                put_patch(pdst + j * (PW * CKQ), make_float4(olo.x, olo.y, ohi.x, ohi.y));
            }
        }
    };
    // ---- FIRST: the denoiser's first layer (sigma-plane cat + conv 2 -> 32 + LeakyReLU, noise.py:157-162,104; conv_first_kernel)
    // evaluated for the patch pixels of chunk c's 8 channels straight into the patch: its 537 MB output is neither written nor
    // read back (that layer is bound by its stores, this one by its loads).  Thread item = (patch pixel, 4 channels); taps outer,
    // items inner: one weight vector and NIT x 4 accumulators live.  Workgroups whose patch windows lie inside the image fold the
    // sigma plane (= 1 everywhere) into the bias; border workgroups add its taps under the in-image mask.  Pixels outside the
    // image are this conv's zero padding.
    bool first_interior = false;
    int fbase[FIRST ? NIT : 1];
    if constexpr (FIRST) {
        first_interior = ty0 >= 2 && ty0 + TH + 2 <= a.H && tx0 >= 2 && tx0 + TW + 2 <= a.W;
        const float sg = a.first_sigma[n];
        for (int i = tid; i < (PH + 2) * (PW + 2); i += NT_) {
            const int r = i / (PW + 2), q = i % (PW + 2);
            const int gy = ty0 - 2 + r, gx = tx0 - 2 + q;
            float d = 0.f, m = 0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const size_t qx = ((size_t)n * a.H + gy) * a.W + gx;
                d = a.last_ximg != nullptr ? a.last_ximg[qx] : (a.last_z[qx].x - a.last_u[qx].x);
                m = 1.f;
            }
            dimg[r * IW + q] = d;
            mimg[r * IW + q] = m;
        }
        for (int i = tid; i < 32 * 18; i += NT_) {
            const int ch = i / 18, t = i % 18;
            wl[t * 32 + ch] = a.first_w[i] * (t >= 9 ? sg : 1.f);
        }
        if (tid < 32) {
            float sc = 0.f;
#pragma unroll
            for (int t = 9; t < 18; ++t) sc += a.first_w[tid * 18 + t] * sg;
            bl[tid] = a.first_b[tid];
            bl[32 + tid] = a.first_b[tid] + sc;
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int pp = (tid + k * NT_) / PPP;
            fbase[k] = (tid + k * NT_ < ITEMS) ? (pp / PW) * IW + pp % PW : 0;
        }
        __syncthreads();
    }
    auto first_patch = [&](int c) {
        if constexpr (FIRST) {
            const int part = tid % PPP;                    // (NT_ is a multiple of PPP: the same part for all of a thread's items)
            const float* const wq = wl + CK * c + 4 * part;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + (first_interior ? 32 : 0) + CK * c + 4 * part);
            f32x4 acc[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) acc[k] = b4;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wq + t * 32);
#pragma unroll
                for (int k = 0; k < NIT; ++k) acc[k] = __builtin_elementwise_fma(w4, f32x4(dimg[fbase[k] + (t / 3) * IW + t % 3]), acc[k]);
            }
            if (!first_interior) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wq + (9 + t) * 32);
#pragma unroll
                    for (int k = 0; k < NIT; ++k) acc[k] = __builtin_elementwise_fma(w4, f32x4(mimg[fbase[k] + (t / 3) * IW + t % 3]), acc[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int idx = tid + k * NT_;
                if (idx < ITEMS) {
                    const float in = first_interior ? 1.f : mimg[fbase[k] + IW + 1];      // the patch pixel itself inside the image?
                    const f32x4 v = in * __builtin_elementwise_max(acc[k], kLeaky * acc[k]);
                    put_patch((idx / PPP) * CKQ + part * 4, make_float4(v[0], v[1], v[2], v[3]));
                }
            }
        }
    };
    if constexpr (UP2) {
        // the interpolation's coordinates are the same for every chunk: one table entry per patch row and column, built once
        // (first read behind the loop-top barrier).  Pixels outside the image (the conv's zero padding) get zero weights.
        // Row entry: the weights of the row's two candidate source lines s, s + 1 (see interpolate()); column entry:
        // {offsets of the two source columns in the region, their weights}.
        if (tid < PH + PW) {
            const bool isrow = tid < PH;
            const int pq = isrow ? tid : tid - PH;
            const int g = (isrow ? ty0 : tx0) + pq - 1;
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g >= 0 && g < (isrow ? a.H : a.W)) {
                const float sc = (isrow ? a.rh : a.rw) * (float)g;
                const int i0 = (int)sc;
                const int i1 = i0 + (i0 < (isrow ? Hs : Ws) - 1 ? 1 : 0);
                const float l = fminf(fmaxf(sc - (float)i0, 0.f), 1.f);
                if (isrow) {
                    const int s0 = ylo + (pq >> 1);        // the row's first candidate line
                    const float w0 = 1.f - l, w1 = l;
                    e.x = (i0 == s0 ? w0 : 0.f) + (i1 == s0 ? w1 : 0.f);
                    e.y = (i0 == s0 + 1 ? w0 : 0.f) + (i1 == s0 + 1 ? w1 : 0.f);
                    // (a line outside the two candidates can only carry the weight l = 0 of image row 0, where sc = 0:
                    // upsample_lines_regular(), checked by winograd_plan)
                } else {
                    e = make_float4(__int_as_float((i0 - xlo) * LP), __int_as_float((i1 - xlo) * LP), 1.f - l, l);
                }
            }
            *reinterpret_cast<float4*>(&(isrow ? rowT : colT)[4 * pq]) = e;
        }
    }
    if constexpr (FIRST) first_patch(0);
    else {
        issue(0); park(0);
        if constexpr (SPLIT) { issue(0, 1); park(0, 1); }
    }

    // this thread's transform item: tile tq of the workgroup's 32 tiles (TC per row), channels [2*hc, 2*hc+2), frequency
    // column group tj (first half of the waves: columns 0..2, second half: columns 3..5; all six rows).  HCN lanes cover a
    // tile's channels and a 32-lane ds_read_b64 group covers 4 tiles 80 floats apart (WN = 2) or 8 tiles 40 floats apart
    // (WN = 1): all 64 banks, conflict-free.
    // V has NO padding (pixel stride CK floats); the channel granules of a tile are permuted instead (swz) so that the MFMA
    // phase's A reads are conflict-free: WN = 2, ds_read_b128 in 16-lane groups {tiles 0-3, 12-15 | granule g} + {tiles 4-11 |
    // granule g+1}: granule position = g ^ m[tile / 4], m = {0, 3, 2, 1};  WN = 1, ds_read_b64 in 32-lane groups {16 tiles x
    // 2 granules}: position = g ^ 2 * (tile / 8).
    auto swz = [](int t16, int g) { return WN == 2 ? (g ^ ((4 - (t16 >> 2)) & 3)) : (g ^ ((t16 >> 3) << 1)); };
    const int hc = tid % HCN, tq = (tid / HCN) % MT;
    const int tj = wid / (NT_ / 128);
    const int trow = tq / TC;                                                  // tile row; stacked: rows 0..3 / 4..7 = slice 0 / 1
    const int wrow = STK ? (trow / (TR / 2)) * SUBH + 4 * (trow % (TR / 2)) : 4 * trow;
    const int win = (wrow * PW + 4 * (tq % TC)) * CKQ + 2 * hc;                // top-left of tile tq's 6x6 input window
    const int vout = WN == 2 ? tq * CKP + 4 * swz(tq & 15, hc >> 1) + 2 * (hc & 1)      // this item's slot in every frequency plane
                             : tq * CKP + 2 * swz(tq & 15, hc);

    f32x4 acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef typename std::conditional<WN == 2, float4, float2>::type frag_t;      // S channels of one lane's A / B fragment
    const size_t stream = (size_t)nchunks * 36 * 64 + kW4Tail / S;                // fragments per 16-channel block's stream
    // Upsample + concat variants: B fragments come through a buffer descriptor over this wave's stream - per-lane offset lane *
    // sizeof(frag_t) (one register, constant), fragment offset SCALAR: no 64-bit vector address arithmetic in the MFMA loop
    // (global_load costs two v_add per ~4 loads on the port the f32 MFMAs need, and an address register pair): -2...3.5 % on those
    // layers.  The plain variants measured 3-9 % SLOWER that way (the B stream is their dominant traffic and global_load streams
    // it faster than offen buffer loads) and keep global loads.
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const frag_t*>(a.wpack) + (size_t)cb * stream), 0, (int)(stream * sizeof(frag_t)), 0x00020000);
    const int boff = lane * (int)sizeof(frag_t);
    const frag_t* const bptr = reinterpret_cast<const frag_t*>(a.wpack) + (size_t)cb * stream + lane;
    // fragment i of the chunk whose first fragment is number f0 of the stream / sits at bp (both wave-uniform; the global-load
    // form keeps ONE base pointer per chunk and immediate offsets - an address computed per load costs vector instructions)
    auto bfrag = [&](const frag_t* bp, int f0, int i) -> frag_t {
        if constexpr (!UP2) return bp[i * 64];
        else if constexpr (WN == 2) return __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rB, boff, (f0 + i) * (64 * 16), 0));
        else return __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b64(rB, boff, (f0 + i) * (64 * 8), 0));
    };
    frag_t bq[PF];
#pragma unroll
    for (int p = 0; p < KEEP; ++p) bq[p] = bfrag(bptr, 0, p);
    const int aoff = (16 * th + cl) * CKP + S * swz(cl, tg);                      // this lane's fragment of V[0]

#ifdef PNP_STAMPS
    st_setup = W4T();
    st_loop0 = st_setup;
#endif
    for (int c = 0; c < nchunks; ++c) {
#ifdef PNP_STAMPS
        st_tmp = W4T();
#endif
        __syncthreads();                                   // chunk c is parked; the previous chunk's MFMA phase is done with V
        if (UP2 && c >= nskip) {
            if constexpr (NT_ == 256) __builtin_amdgcn_s_setprio(3);
            interpolate();
            if constexpr (NT_ == 256) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_commit += t - st_tmp; st_tmp = t; }
#endif

        // ---- input transform V = B^T d B: all six frequency rows of this thread's column group ------------------------------
        // WN = 1 (32 -> 32 layers: 4 short chunks): the next chunk's loads go out BEFORE the transform - a chunk's MFMA phase
        // alone (~2 us) is shorter than an HBM round trip under load
        if constexpr (WN == 1 && !FIRST) { if (c + 1 < nchunks) issue(c + 1); }
        __builtin_amdgcn_sched_barrier(0);                 // WN = 2: keep the next chunk's loads (20 registers) behind the transform
        if constexpr (NT_ == 256) __builtin_amdgcn_s_setprio(3);   // two workgroups per CU: this one's few VALU instructions go
        if (tj == 0) wino4_input_transform<0, PW, CKQ, PLANE>(patch + win, V + vout);   // ahead of the other's MFMA stream
        else wino4_input_transform<1, PW, CKQ, PLANE>(patch + win, V + vout);
        if constexpr (NT_ == 256) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        const frag_t* bp = bptr + (size_t)c * 36 * 64;     // this chunk's fragments
        const int bf0 = c * 36;
#pragma unroll
        for (int p = KEEP; p < PF; ++p) bq[p] = bfrag(bp, bf0, p);
        if constexpr (WN == 2) { if (c + 1 < nchunks) issue(c + 1); }   // next chunk's loads fly under this chunk's MFMAs
        __syncthreads();
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_trans += t - st_tmp; st_tmp = t; }
#endif

        // ---- 36 GEMMs per wave: A from V (LDS), B from the packed U stream (L2), S MFMAs per frequency ----------------------
        frag_t ar[3];                                      // A fragments two frequencies ahead (LDS latency > one group of MFMAs)
        ar[0] = *reinterpret_cast<const frag_t*>(&V[aoff]);
        ar[1] = *reinterpret_cast<const frag_t*>(&V[PLANE + aoff]);
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) {
            if (xi + 2 < 36) ar[(xi + 2) % 3] = *reinterpret_cast<const frag_t*>(&V[(xi + 2) * PLANE + aoff]);
            __builtin_amdgcn_sched_barrier(0);
            const frag_t a0 = ar[xi % 3], b0 = bq[xi % PF];
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc[xi], 0, 0, 0);
            if constexpr (WN == 2) {
                acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc[xi], 0, 0, 0);
                acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc[xi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SPLIT) {                         // half way: park the first half-batch, send the second
                if (xi == 17 && c + 1 < nchunks) { park(c + 1, 0); issue(c + 1, 1); }
            }
            // refill the slot just read: this chunk's fragment xi + PF, or one of the next chunk's first KEEP (the stream is
            // contiguous across chunks; tail zero-padded)
#ifdef PNP_DIAG
            if ((a.diag & 4) && (xi & 1)) continue;        // timing only: every second weight fragment is not fetched (stale one reused)
#endif
            if (xi + PF < 36 + KEEP) bq[xi % PF] = bfrag(bp, bf0, xi + PF);
        }
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_mfma += t - st_tmp; }
#endif
        if (c + 1 < nchunks) {
            if constexpr (FIRST) first_patch(c + 1);
            else park(c + 1, SPLIT ? 1 : 0);
        }
    }
#ifdef PNP_STAMPS
    st_loop1 = W4T();
    st_tmp = st_loop1;
#endif

    // ---- output transform, lane-local -----------------------------------------------------------------------------------------
    wino4_epilogue<TW, STK, MT, WN == 1 && MT == 32 && SRC == SRC_PLAIN>(a, acc, n, th, tg, cl, cb, tx0, ty0, STK ? (th == 0 ? live0 : live1) : true, smem);
#ifdef PNP_STAMPS
    {
        const int w = (int)blockIdx.x - (int)(gridDim.x / 2);
        if (tid == 0 && w >= 0 && w < W4_WGS && a.stamp_slot < W4_SLOTS) {
            unsigned long long* o = g_w4_stamps + ((size_t)a.stamp_slot * W4_WGS + w) * W4_N;
            const unsigned long long te = W4T();
            o[0] = 1; o[1] = st_setup - st_t0; o[2] = st_commit; o[3] = st_trans; o[4] = st_mfma; o[5] = st_loop1 - st_loop0;
            o[6] = te - st_tmp; o[7] = 0; o[8] = te - st_loop1; o[9] = te - st_t0; o[10] = (unsigned long long)nchunks;
        }
    }
#endif
}

// ---- PHASED variant (WN = 2, plain source) ----------------------------------------------------------------------------------
// Same arithmetic and the same wave roles as conv3x3_wino4_kernel<TW, SRC_PLAIN, STK, 2>, different schedule.  The input
// transform is LDS-bound (18 ds_write_b64 + 30 ds_read_b64 per thread, ~80 packed VALU instructions) and the f32 MFMAs hold the
// SIMD's vector port, so with all eight waves in the same phase the matrix pipe idles through every transform (22 % of a
// chunk, profiles/r02_wino4_stamps.md).  Here the two halves of the workgroup run HALF A CHUNK APART: waves 0-3 (tiles 0-15)
// and waves 4-7 (tiles 16-31) each transform only their own 16 tiles - a wave's MFMAs read no other rows of V - so between two
// workgroup barriers every SIMD has one wave in its MFMA phase and one in its transform:
//     barrier | G0: T(c)      G1: M(c-1) | barrier | G0: M(c)      G1: T(c) | barrier | G0: T(c+1) ...
// The halo patch is shared by both halves and now lives twice (chunk c+1 is parked while the lagging half still reads chunk c):
// channel-granule planes [4][pixels][4 floats] (+4 floats per plane: the 8 x 4 (channel pair, tile) lanes of a ds_read_b64
// group fall on 64 different banks without any pixel padding), 39 KB per copy instead of the padded 49 KB.
template <int TW, bool STK>
__global__ __launch_bounds__(512, 2) void conv3x3_wino4p_kernel(const ConvArgs a) {
    constexpr int CK = 16, CKP = 16, PPP = 4, NT_ = 512, S = 4;
    constexpr int TC = TW / 4, TR = 32 / TC;           // tiles per row / rows of tiles in the 32-tile M-block
    constexpr int TH = 4 * TR;
    constexpr int SUBH = STK ? TH / 2 + 2 : 0;         // rows of one slice's halo patch in the stacked layout
    constexpr int PH = STK ? 2 * SUBH : TH + 2, PW = TW + 2;
    constexpr int NPIX = PH * PW, ITEMS = NPIX * PPP;
    constexpr int NIT = (ITEMS + NT_ - 1) / NT_;
    constexpr int PLSZ = ((NPIX * 4 + 15) / 16) * 16 + 4;  // floats per channel-granule plane, = 4 (mod 16)
    constexpr int PATCH = 4 * PLSZ;                    // floats per copy of the patch
    constexpr int PF = 12, KEEP = 3;                   // B fragment ring (conv3x3_wino4_kernel), deeper: one MFMA wave per SIMD at a time
    constexpr int PLANE = 32 * CKP;                    // floats per frequency plane of V
    constexpr unsigned OOB = 0x80000000u;
    static_assert(!STK || TW == 16, "stacked slices: 16-wide tiles");
    static_assert((2 * PATCH + 36 * PLANE) * 4 <= 160 * 1024, "two patch copies + V in one CU's LDS");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const patch = smem;                             // [2][4 granules][PLSZ]
    float* const V = smem + 2 * PATCH;                     // [36][32 tiles][16], granules swizzled (see swz)
#ifdef PNP_STAMPS
    unsigned long long st_t0 = W4T(), st_setup = 0, st_loop0 = 0, st_commit = 0, st_trans = 0, st_mfma = 0, st_loop1 = 0, st_tmp = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, cq = wid & 3;                // half of the 32 tiles (the phase group), 16-channel group of this wave
    const int th = grp;
    const int tg = lane >> 4, cl = lane & 15;              // MFMA lane = (k / row group tg, column / row cl)

    const int ny = a.Cout / 64;
    const int bid = blockIdx.x;
    const int ntl = a.tilesX * a.tilesY * (STK ? (a.N + 1) / 2 : a.N);
    int cby, bt;                                           // this workgroup's block of 64 channels, its spatial tile index (XCD-aware
    wino4_decode(bid, ny, ntl, a.order, cby, bt);          // order, see conv3x3_wino4_kernel)
    if (bt >= ntl) return;
    const int tx0 = (bt % a.tilesX) * TW;
    bt /= a.tilesX;
    const int ty0 = (bt % a.tilesY) * TH;
    const int n = STK ? 2 * (bt / a.tilesY) : bt / a.tilesY;               // first (or only) slice of this workgroup
    const int nsl = STK ? (n + 1 < a.N ? 2 : 1) : 1;                       // slices this workgroup holds
    const bool live0 = !(a.tact != nullptr && a.tact[n] > 0.5f);
    const bool live1 = STK && nsl == 2 && !(a.tact != nullptr && a.tact[n + 1] > 0.5f);
    if (!live0 && !live1) return;
    const int cb = cby * 4 + cq;                           // this wave's 16-channel block
    const int nchunks = a.Cin / CK;

    // staging addresses, once (see conv3x3_wino4_kernel): item idx = (pixel, channel granule), granule fastest
    const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.src0 + (size_t)n * a.H * a.W * a.Cin), 0, nsl * a.H * a.W * a.Cin * 4, 0x00020000);
    unsigned voff[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int idx = tid + k * NT_;
        const int part = idx % PPP, pp = idx / PPP;
        const int py = pp / PW, px = pp % PW;
        const int sub = STK ? py / SUBH : 0;                               // stacked: which of the two slices
        const int gy = STK ? py % SUBH - 1 : ty0 + py - 1, gx = tx0 + px - 1;
        voff[k] = (idx < ITEMS && sub < nsl && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                      ? (unsigned)((((sub * a.H + gy) * a.W + gx) * a.Cin + part * 4) * 4) : OOB;
    }
    const int ldst = (tid & 3) * PLSZ + (tid >> 2) * 4;    // LDS slot of item 0; item k is 128 pixels (512 floats) further
    float4 raw[NIT];
    auto issue = [&](int c) {
        const int soff = c * CK * 4;
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            raw[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc0, voff[k], soff, 0));
    };
    auto park = [&](int c) {                               // registers -> copy (c & 1) of the patch
        float* const dst = patch + (c & 1) * PATCH + ldst;
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (tid + k * NT_ < ITEMS) *reinterpret_cast<float4*>(&dst[k * (NT_ / PPP) * 4]) = raw[k];
    };

    // this thread's transform item: tile t16 of its half's 16 tiles, channels [2*hc, 2*hc+2), frequency column group tj
    // (waves 0-1 of the half: columns 0..2, waves 2-3: columns 3..5; all six rows)
    auto swz = [](int t16, int g) { return g ^ ((4 - (t16 >> 2)) & 3); };
    const int tgi = tid & 255;
    const int hc = tgi & 7, t16 = (tgi >> 3) & 15;
    const int tj = (wid >> 1) & 1;
    const int tq = 16 * grp + t16;
    const int trow = tq / TC;                                                  // tile row; stacked: rows 0..3 / 4..7 = slice 0 / 1
    const int wrow = STK ? (trow / (TR / 2)) * SUBH + 4 * (trow % (TR / 2)) : 4 * trow;
    const int win = (hc >> 1) * PLSZ + (wrow * PW + 4 * (tq % TC)) * 4 + 2 * (hc & 1);   // top-left of the 6x6 window, copy 0
    const int vout = tq * CKP + 4 * swz(t16, hc >> 1) + 2 * (hc & 1);          // this item's slot in every frequency plane

    f32x4 acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    const size_t stream = (size_t)nchunks * 36 * 64 + kW4Tail / S;                // fragments per 16-channel block's stream
    const float4* const bptr = reinterpret_cast<const float4*>(a.wpack) + (size_t)cb * stream + lane;
    float4 bq[PF];
#pragma unroll
    for (int p = 0; p < KEEP; ++p) bq[p] = bptr[p * 64];
    const int aoff = (16 * th + cl) * CKP + 4 * swz(cl, tg);                      // this lane's fragment of V[0]

    // PARK_T: a half fetches AND parks its share of the next chunk inside its own transform phase (see the loop); the stacked 16 x 16
    // level measured 1.3-2.1 % faster on the round-2 schedule (staging one phase earlier, parked after the MFMA phase) and keeps it
    constexpr bool PARK_T = !STK;
    // prologue: both halves park their share of chunk 0; on the round-2 schedule the lagging half also of chunk 1 (its in-loop
    // parks run one chunk further ahead)
    issue(0);
    park(0);
    if constexpr (!PARK_T) { if (grp == 1 && nchunks > 1) { issue(1); park(1); } }
#ifdef PNP_STAMPS
    st_setup = W4T();
    st_loop0 = st_setup;
#endif
    if (grp == 1) __syncthreads();                         // the lagging half starts one phase late
    for (int c = 0; c < nchunks; ++c) {
#ifdef PNP_STAMPS
        st_tmp = W4T();
#endif
        __syncthreads();                                   // chunk c is parked by everyone; this half's MFMAs of chunk c-1 are done
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_commit += t - st_tmp; st_tmp = t; }
#endif
        // This half's share of the NEXT chunk is fetched and parked inside its own transform phase: the loads go out here, land
        // under the transform and the wait for the partner half's MFMAs, and are parked before the phase's closing barrier (copy
        // (c + 1) & 1: its last reader, the lagging half's transform of chunk c - 1, is at least one barrier back; its first
        // reader, the leading half's transform of chunk c + 1, comes after the closing barrier of the lagging half's phase).  The
        // MFMA phase - the critical one - then carries neither the staging registers nor the park (round 2 parked after it).
        if constexpr (PARK_T) { if (c + 1 < nchunks) issue(c + 1); }
        // ---- input transform of this half's 16 tiles (its partner waves on the SIMDs are in their MFMA phase: the few VALU
        // instructions of the transform go first)
        __builtin_amdgcn_s_setprio(3);
        {
            const float* const w = patch + (c & 1) * PATCH + win;
            if (tj == 0) wino4_input_transform<0, PW, 4, PLANE>(w, V + vout);
            else wino4_input_transform<1, PW, 4, PLANE>(w, V + vout);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        const float4* bp = bptr + (size_t)c * 36 * 64;
#pragma unroll
        for (int p = KEEP; p < PF; ++p) bq[p] = bp[p * 64];
        // round-2 schedule (!PARK_T): staging runs ahead of the TRANSFORMS, and the lagging half transforms chunk c a phase after the
        // leading half: the leading half parks chunk c+1 after its MFMAs of chunk c, the lagging half chunk c+2 (both land in a copy
        // whose last reader - the lagging half's transform - is at least one barrier back, and a full phase before their first reader)
        const int nx = c + 1 + grp;
        if constexpr (PARK_T) { if (c + 1 < nchunks) park(c + 1); }
        else { if (nx < nchunks) issue(nx); }
        __syncthreads();
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_trans += t - st_tmp; st_tmp = t; }
#endif

        // ---- 36 GEMMs per wave: A from V (LDS), B from the packed U stream (L2), 4 MFMAs per frequency -----------------------
        float4 ar[3];
        ar[0] = *reinterpret_cast<const float4*>(&V[aoff]);
        ar[1] = *reinterpret_cast<const float4*>(&V[PLANE + aoff]);
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) {
            if (xi + 2 < 36) ar[(xi + 2) % 3] = *reinterpret_cast<const float4*>(&V[(xi + 2) * PLANE + aoff]);
            __builtin_amdgcn_sched_barrier(0);
            const float4 a0 = ar[xi % 3], b0 = bq[xi % PF];
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc[xi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifdef PNP_DIAG
            if ((a.diag & 4) && (xi & 1)) continue;        // timing only: every second weight fragment is not fetched (stale one reused)
#endif
            if (xi + PF < 36 + KEEP) bq[xi % PF] = bp[(xi + PF) * 64];
        }
#ifdef PNP_STAMPS
        { const unsigned long long t = W4T(); st_mfma += t - st_tmp; }
#endif
        if constexpr (!PARK_T) { if (nx < nchunks) park(nx); }
    }
    if (grp == 0) __syncthreads();                         // pairs with the lagging half's last barrier
#ifdef PNP_STAMPS
    st_loop1 = W4T();
    st_tmp = st_loop1;
#endif

    // ---- output transform, lane-local (as conv3x3_wino4_kernel) ---------------------------------------------------------------
    wino4_epilogue<TW, STK>(a, acc, n, th, tg, cl, cb, tx0, ty0, STK ? (th == 0 ? live0 : live1) : true);
#ifdef PNP_STAMPS
    {
        const int w = (int)blockIdx.x - (int)(gridDim.x / 2);
        if (tid == 0 && w >= 0 && w < W4_WGS && a.stamp_slot < W4_SLOTS) {
            unsigned long long* o = g_w4_stamps + ((size_t)a.stamp_slot * W4_WGS + w) * W4_N;
            const unsigned long long te = W4T();
            o[0] = 1; o[1] = st_setup - st_t0; o[2] = st_commit; o[3] = st_trans; o[4] = st_mfma; o[5] = st_loop1 - st_loop0;
            o[6] = te - st_tmp; o[7] = 0; o[8] = te - st_loop1; o[9] = te - st_t0; o[10] = (unsigned long long)nchunks;
        }
    }
#endif
}

template <int TW, bool STK>
static hipError_t launch_wino4p_inst(const ConvArgs& a, const WinoPlan& p, hipStream_t s) {
    constexpr int TC = TW / 4, TR = 32 / TC, TH = 4 * TR;
    constexpr int NPIX = (STK ? TH + 4 : TH + 2) * (TW + 2);
    constexpr int PLSZ = ((NPIX * 4 + 15) / 16) * 16 + 4;
    constexpr size_t lds = ((size_t)2 * 4 * PLSZ + (size_t)36 * 32 * 16) * sizeof(float);
    auto kern = conv3x3_wino4p_kernel<TW, STK>;
    static DeviceOnce cap;
    if (hipError_t e = raise_lds_cap((const void*)kern, (int)lds, cap); e != hipSuccess) return e;
    const int ntiles = p.tiles_x * p.tiles_y * (STK ? (a.N + 1) / 2 : a.N), ny = a.Cout / 64;
    dim3 grid((unsigned)(((ntiles + 7) / 8) * 8 * ny));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
    return hipGetLastError();
}

template <int TW, int SRC, bool STK = false, int WN = 2, int MT = 32>
static hipError_t launch_wino4_inst(const ConvArgs& a, const WinoPlan& p, hipStream_t s) {
    constexpr int CK = 8 * WN, CKP = CK, CKQ = WN == 2 ? CK + 4 : CK + 2, TC = TW / 4, TR = MT / TC, TH = 4 * TR;
    constexpr size_t patch_f = (((size_t)(STK ? TH + 4 : TH + 2) * (TW + 2) * CKQ + 3) / 4) * 4;
    constexpr size_t lowres_f = SRC == SRC_UPCAT ? (size_t)(TH / 2 + 3) * (TW / 2 + 3) * CK + 4 * (TH + 2 + TW + 2) : 0;   // + tables
    constexpr size_t first_f = SRC == SRC_FIRST ? (size_t)2 * (TH + 4) * (TW + 5) + 18 * 32 + 64 : 0;   // image tile + mask, weights, bias
    constexpr size_t lds = (patch_f + (size_t)36 * MT * CKP + lowres_f + first_f) * sizeof(float);
    static_assert(lds <= (WN == 2 && MT == 32 ? 160 : 80) * 1024, "one (8 waves) / two (4 waves) workgroups per CU");
    if (p.th != TH || p.tw != TW) return hipErrorInvalidValue;
    auto kern = conv3x3_wino4_kernel<TW, SRC, STK, WN, MT>;
    static DeviceOnce cap;
    if (hipError_t e = raise_lds_cap((const void*)kern, (int)lds, cap); e != hipSuccess) return e;
    const int ntiles = p.tiles_x * p.tiles_y * (STK ? (a.N + 1) / 2 : a.N), ny = a.Cout / (32 * WN);
    dim3 grid((unsigned)(((ntiles + 7) / 8) * 8 * ny));
    hipLaunchKernelGGL(kern, grid, dim3(8 * MT * WN), lds, s, a);
    return hipGetLastError();
}

// `a.wpack` must be the F(4x4) pack (pack_winograd4_weights).
hipError_t launch_conv3x3_winograd4(const ConvArgs& a0, const WinoPlan& p, int src_mode, hipStream_t s) {
    if (!p.use || p.algo != 4 || a0.Cout % p.bn || a0.Cin % p.ck || (src_mode == SRC_UPCAT && a0.Cskip % p.ck) || a0.W % 4) return hipErrorInvalidValue;
    if (a0.last_w != nullptr && !(p.bn == 32 && p.mt == 32 && src_mode == SRC_PLAIN)) return hipErrorInvalidValue;   // fused last layer: 32-channel variant
    ConvArgs a = a0;
    a.tilesX = p.tiles_x;
    a.tilesY = p.tiles_y;
    a.order = p.order;
#ifdef PNP_STAMPS
    a.stamp_slot = g_w4_slot++;
#endif
    if (p.bn == 32) {                                      // Cout = 32: 4-wave workgroups, 8-channel chunks, two per CU
        if (p.ck != 8 || p.stack) return hipErrorInvalidValue;
        if (p.tw == 32 && src_mode == SRC_PLAIN) return launch_wino4_inst<32, SRC_PLAIN, false, 1>(a, p, s);
        if (p.tw == 32 && src_mode == SRC_UPCAT) return launch_wino4_inst<32, SRC_UPCAT, false, 1>(a, p, s);
        if (p.tw == 32 && src_mode == SRC_FIRST) return launch_wino4_inst<32, SRC_FIRST, false, 1>(a, p, s);
        if (p.tw == 16 && src_mode == SRC_FIRST) return launch_wino4_inst<16, SRC_FIRST, false, 1>(a, p, s);
        if (p.tw == 16 && src_mode == SRC_PLAIN) return launch_wino4_inst<16, SRC_PLAIN, false, 1>(a, p, s);
        if (p.tw == 16 && src_mode == SRC_UPCAT) return launch_wino4_inst<16, SRC_UPCAT, false, 1>(a, p, s);
        return hipErrorInvalidValue;
    }
    if (p.ck != 16) return hipErrorInvalidValue;
    if (p.mt == 16) {                                      // 16-tile M-blocks: two independent 4-wave workgroups per CU
        if (p.stack) return hipErrorInvalidValue;
        if (p.tw == 32 && src_mode == SRC_PLAIN) return launch_wino4_inst<32, SRC_PLAIN, false, 2, 16>(a, p, s);
        if (p.tw == 32 && src_mode == SRC_UPCAT) return launch_wino4_inst<32, SRC_UPCAT, false, 2, 16>(a, p, s);
        if (p.tw == 16 && src_mode == SRC_PLAIN) return launch_wino4_inst<16, SRC_PLAIN, false, 2, 16>(a, p, s);
        if (p.tw == 16 && src_mode == SRC_UPCAT) return launch_wino4_inst<16, SRC_UPCAT, false, 2, 16>(a, p, s);
        return hipErrorInvalidValue;
    }
    if (p.phased && src_mode == SRC_PLAIN) {               // the two tile halves half a chunk apart (conv3x3_wino4p_kernel)
        if (p.tw == 32 && !p.stack) return launch_wino4p_inst<32, false>(a, p, s);
        if (p.tw == 16) {
            if (p.stack) return a.H == 16 && a.W == 16 ? launch_wino4p_inst<16, true>(a, p, s) : hipErrorInvalidValue;
            return launch_wino4p_inst<16, false>(a, p, s);
        }
        return hipErrorInvalidValue;
    }
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
