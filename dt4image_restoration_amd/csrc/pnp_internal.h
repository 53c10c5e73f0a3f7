// Internal declarations shared by the HIP translation units of libpnpadmm.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include <vector>

namespace pnp {

// Experiment / test overrides, read from the environment ONCE per handle (pnp_create) and stored with it, so the plan a
// layer's weights were packed for is the plan every later launch uses.
struct Tuning {
    int wino_min_cin = 32;        // PNP_WINO_MIN_CIN
    long wino_min_blocks = 192;   // PNP_WINO_MIN_BLOCKS (tests force 1: Winograd on small problems too)
    bool wino_big = false;        // PNP_WINO_BIG_GROUPS: the 8-wave plans everywhere
    bool wino_small = false;      // PNP_WINO_SMALL_GROUPS
    bool no_wino = false;         // PNP_NO_WINOGRAD
    int f4_min_cin = 32;          // PNP_WINO_F4_MIN_CIN: F(4x4,3x3) for layers with at least this many input channels
    bool no_f4 = false;           // PNP_NO_WINO_F4
    bool no_f4_fused_first = false;// PNP_NO_F4_FUSED_FIRST: the first layer as its own kernel (conv_first_kernel)
    bool no_f4_fused_last = false;// PNP_NO_F4_FUSED_LAST: up4.conv-2 (+ fused last layer) on the F(2x2) kernel
    bool no_f4_phased = false;    // PNP_NO_WINO_F4_PHASED: every F(4x4) layer on the all-waves-in-step schedule
    int f4_mt16 = 0;              // PNP_WINO_F4_MT16 (experiments): 0 = default rule, 1 = never, 2 = upsample+concat layers only, 3 = every
                                  // 64-channel-block layer on 16-tile M-blocks
    bool bf16_no_ws = false;      // PNP_BF16_NO_WS (ablation): bf16 mode without the producer / consumer kernel (the round-2 kernel everywhere)
    bool bf16_f32_acts = false;   // PNP_BF16_F32_ACTS (ablation): bf16 mode keeps every activation in f32, as rounds 1-2 did
    bool bf16_no_holdhi = false;  // PNP_BF16_NO_HOLDHI (ablation): the 32 -> 32 layers of the bf16 mode stream their weights instead of holding them
    bool bf16_w1 = false;         // PNP_BF16_W1 (ablation): bf16 mode with ONE bf16 term per weight (the round-3 arithmetic: 0.015 dB of
                                  // PSNR drift against the f32 reference over configs[4]'s 50 iterations) instead of hi + lo
    int f4_order = 1;             // PNP_WINO_F4_ORDER (experiments): 0 = spatial tiles dealt round-robin over the XCDs (rounds 1-2)
    int splitk_inlaunch = 0;      // PNP_SPLITK_INLAUNCH (experiments): 1 = split-K planes combined inside the conv launch (agent-scope accesses,
                                  // no fence) instead of by splitk_reduce_kernel
    bool fft_xcd = false;         // PNP_FFT_XCD=1 (experiment, off by default: it moves half the bytes and is slower, profiles/r05_ablation.md): the
                                  // data-fidelity stage of square 256 / 512 slices as ONE persistent launch with per-XCD work queues
    int slice128_min_n = 192;     // PNP_SLICE128_MIN_N: 128 x 128 slices take the one-workgroup-per-slice data-fidelity kernel from
                                  // this batch size on (measured: one workgroup per slice is LDS-bound on its CU - 46 us a slice - so it
                                  // needs a chip-filling batch to beat the three-launch path: 64.1 vs 78.8 us at 256 slices, 48.9 vs 34.9 at 64)
};
Tuning tuning_from_env();

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) must be applied once per DEVICE (the attribute lives with the device's
// code object), not once per process: one bit per device ordinal.
struct DeviceOnce { std::atomic<uint64_t> mask{0}; };
hipError_t raise_lds_cap(const void* fn, int bytes, DeviceOnce& once);

// ---- denoiser conv layer description (mirrors dt4image_restoration_amd/unet_spec.py) ----------
enum SrcMode : int { SRC_PLAIN = 0, SRC_SIGMA = 1, SRC_POOL = 2, SRC_UPCAT = 3, SRC_FIRST = 4 };   // FIRST: F(4x4) 32-channel variant only

struct LayerSpec {
    int cin, cout, ksize, level, src, cskip;
};
static constexpr int N_LAYERS = 28;
extern const LayerSpec kLayers[N_LAYERS];

// Arguments of one conv3x3 MFMA launch.  Activations are NHWC float32.
struct ConvArgs {
    const float* src0;   // PLAIN: [N,H,W,Cin]; POOL: [N,2H,2W,Cin]; UPCAT: skip [N,H,W,Cskip]
    const float* src1;   // UPCAT: low-res [N,H/2,W/2,Cin-Cskip]; else unused
    const float* wpack;  // packed weights, see pack_conv3x3_weights()
    const float* bias;   // [Cout]
    float* dst;          // [N,H,W,Cout]
    float* pooled;       // optional: also write MaxPool2d(2) of the output, [N,H/2,W/2,Cout] (H, W even)
    // optional fused last layer (1x1 conv 32 -> 1 + image residual + clamp), Cout = 32 LDS-epilogue plan only:
    const float* last_w; const float* last_b; const float* last_ximg; const float2* last_z; const float2* last_u; float* last_out;
    // optional fused FIRST layer (SRC_FIRST: sigma-plane cat + conv 2 -> 32 + LeakyReLU computed into the patch; the image channel comes
    // from last_ximg or Re(last_z - last_u)): raw [32][18] weights, [32] bias, [N] sigma
    const float* first_w; const float* first_b; const float* first_sigma;
    float* partial;      // split-K workspace, conv3x3_partial_floats() floats (small problems only)
    unsigned* arrive;    // split-K: per (tile, channel block) arrival counters, zero between launches - the planes are combined in
                         // the conv launch by the last workgroup to arrive; nullptr: by splitk_reduce_kernel
    const float* tact;   // [N] stop actions or nullptr; slice skipped when tact[n] > 0.5
    int N, H, W;         // OUTPUT spatial size
    int Cin, Cskip, Cout;
    int tilesX, tilesY;  // filled by launch_conv3x3 from the plan
    float rh, rw;        // UPCAT: (H/2-1)/(H-1), (W/2-1)/(W-1)  (bilinear align_corners=True scale)
    int bf16;            // bf16 MFMA operands (wpack = pack_conv3x3_weights_bf16), f32 accumulate: 0 = off, 2 = weights as two bf16 terms
                         // hi + lo (the mode's default), 1 = one term (PNP_BF16_W1)
    int act16;           // bf16 mode, 32-channel plan: bit 0 = src0 (PLAIN source / UPCAT skip) holds bf16, 2 B per channel; bit 1 = dst too;
                         // bit 2 = the pooled copy is written as bf16 (producer / consumer kernel's OFFLOAD store only)
    int order;           // F(4x4): blockIdx -> tile order (wino4_decode), from the plan
#ifdef PNP_STAMPS
    int stamp_slot;      // diagnostic build: launch index into the stamp buffer (winograd_kernels.hip)
#endif
#ifdef PNP_DIAG
    int diag;            // diagnostic build (`make diag`, timing only, results wrong): bit 0 = the F(4x4) epilogue's global stores are dropped
                         // (descriptor of zero records: same instruction stream), bit 1 = every workgroup stages the patch of tile 0 of
                         // slice 0 (L2 hits instead of HBM reads).  Set per launch from PNP_DIAG_L0 for the level-0 layers: the compute-only
                         // time of a layer = what a fused conv-1 -> conv-2 pair could at best pay per layer (profiles/r04_level0_bound.md)
#endif
};

// Launch the conv3x3 (+bias +LeakyReLU 0.2) implicit-GEMM kernel matching `a` (picks the tile shape
// from W and Cout).  Returns hipSuccess or the launch error.
// Tile plan of a conv3x3 launch (picked from the problem size and Cout).
struct ConvPlan {
    int tw, th;        // pixel tile
    int bm, bn;        // pixels / output channels per workgroup tile
    int mt, nt, wm, wn; // M-/N-blocks (32x32) per wave; wave grid
    int ck;            // input channels per staged chunk (16 or 32)
    int splitk;        // K ranges (of whole chunks), one workgroup each; > 1 only on small problems
    int ws;            // bf16 mode: the producer / consumer kernel (conv_bf16_kernels.hip) runs this layer; ck = 32
    int holdhi = 1;    // ws, 32 -> 32 layers under two-term weights: both weight fragments of every k-step held in registers (Tuning.bf16_no_holdhi)
    int tiles_x, tiles_y;
};
ConvPlan conv3x3_plan(int N, int H, int W, int Cin, int Cout, bool bf16 = false, int src_mode = SRC_PLAIN, bool allow_ws = true);
hipError_t launch_conv3x3_bf16ws(const ConvArgs& a, const ConvPlan& p, int src_mode, hipStream_t s);
size_t conv3x3_partial_floats(const ConvPlan& p, int N, int H, int W, int Cout);
bool conv3x3_pooled_output_ok(const ConvPlan& p);
bool conv3x3_tensor_fits(int N, int H, int W, int Cin, int Cout);   // whole-tensor buffer descriptors: < 2 GiB per activation tensor
hipError_t launch_conv3x3(const ConvArgs& a, const ConvPlan& p, int src_mode, hipStream_t s);

// Host-side repack of OIHW conv3x3 weights into the per-lane MFMA B-fragment stream (chunk size ck from the
// layer's plan).  dst must hold conv3x3_pack_floats(cin, cout) floats.
size_t conv3x3_pack_floats(int cin, int cout);
void pack_conv3x3_weights(const float* oihw, int cin, int cout, int ck, float* dst);
size_t conv3x3_pack_floats_bf16(int cin, int cout, int terms);
void pack_conv3x3_weights_bf16(const float* oihw, int cin, int cout, int ck, int terms, float* dst);

// Winograd F(2x2,3x3) path for the K-heavy layers (winograd_kernels.hip).
struct WinoPlan {
    bool use;          // layer is eligible (long K, enough workgroups)
    int algo;          // 1: F(2x2,3x3) (winograd_kernels.hip), 4: F(4x4,3x3) (winograd4_kernels.hip)
    int tw, th, bn, wm, wn, ck, tiles_x, tiles_y;
    int stack;         // F(4x4) on 16 x 16 images: two slices stacked into one 32-tile workgroup
    int mt;            // F(4x4): tiles per workgroup (32, or 16 = two independent 4-wave workgroups per CU)
    int phased;        // F(4x4), 64-channel blocks, plain source: the two tile halves run half a chunk apart (conv3x3_wino4p_kernel)
    int order;         // F(4x4): 1 = every XCD walks a contiguous range of spatial tiles (halo pixels shared through its L2), 0 = tiles dealt round-robin
};
// `src_mode` = the source mode the layer will be LAUNCHED with (a POOL layer whose producer writes the pooled copy runs PLAIN)
WinoPlan winograd_plan(int N, int H, int W, int Cin, int Cout, int src_mode, const Tuning& t);
bool upsample_lines_regular(int H);   // winograd4_kernels.hip: the x2 upsample to H rows reads lines floor((g - 1) / 2), + 1 in float32 too
size_t winograd4_pack_floats(int cin, int cout);
void pack_winograd4_weights(const float* oihw, int cin, int cout, int ck, float* dst);
hipError_t launch_conv3x3_winograd4(const ConvArgs& a, const WinoPlan& p, int src_mode, hipStream_t s);
size_t winograd_pack_floats(int cin, int cout);
void pack_winograd_weights(const float* oihw, int cin, int cout, int ck, float* dst);
hipError_t launch_conv3x3_winograd(const ConvArgs& a, const WinoPlan& p, int src_mode, hipStream_t s);

// First layer (2 -> 32, K = 18: too thin for MFMA, direct VALU) and last layer (1x1 32 -> 1 fused
// with the residual add and clamp).  `ximg` (f32 [N,H,W]) or, when null, Re(z-u) of complex64 z,u
// is the image channel.
hipError_t launch_conv_first(const float* ximg, const float2* z, const float2* u, const float* sigma,
                             const float* tact, const float* w, const float* bias, float* dst,
                             int N, int H, int W, hipStream_t s, bool dst_bf16 = false);
hipError_t launch_conv_last(const float* act, const float* ximg, const float2* z, const float2* u,
                            const float* tact, const float* w, const float* bias, float* out,
                            int N, int H, int W, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s);

// ---- FFT / data-fidelity stage -----------------------------------------------------------------
struct FftPlan {
    int h, w;
    float2* tw_h;   // device: exp(-2 pi i m / h), m < h
    float2* tw_w;
};

// generic centred-or-plain passes (pnp_fft2c)
hipError_t launch_fft_rows_generic(const float2* in, float2* out, const float2* tw, int batch, int H, int W,
                                   int inverse, int shift_in, int shift_out, hipStream_t s);
hipError_t launch_fft_cols_generic(float2* data, const float2* tw, int batch, int H, int W, int inverse,
                                   int shift_in, int shift_out, hipStream_t s);
// ADMM passes
hipError_t launch_fft_rows_fwd_admm(const float* x, const float2* u, float2* work, const float2* tw,
                                    const float* tact, int N, int H, int W, hipStream_t s);
hipError_t launch_fft_cols_prox(float2* work, const float2* tw, const float2* y0s, const uint8_t* masks,
                                int mask_n, const float* mu, const float* tact, int N, int H, int W, hipStream_t s);
hipError_t launch_fft_rows_inv_admm(const float2* work, const float* x, float2* z, float2* u, const float2* tw,
                                    const float* tact, int N, int H, int W, hipStream_t s);

// x0 / x / z / u may all be null: only the episode constants (y0s, masks) are rebuilt
// 128 x 128 only: the whole stage (both transforms each way, the solve, the dual update) in one workgroup per slice
hipError_t launch_admm_slice128(const float* x, float2* z, float2* u, const float2* tw, const float2* y0s,
                                const uint8_t* masks, int mask_n, const float* mu, const float* tact, int N, hipStream_t s);

// x0 / x / z / u may all be null: only the episode constants (y0s, masks) are rebuilt
hipError_t launch_reset(const float2* x0, const float2* y0, const uint8_t* mask, int mask_n, float* x, float2* z,
                        float2* u, float2* y0s, uint8_t* masks, int N, int H, int W, hipStream_t s);
hipError_t launch_finish(const float* tact, float* tstate, uint8_t* done, int N, hipStream_t s);
// the whole stage as ONE persistent launch with per-XCD work queues (fft_kernels.hip, admm_xcd_kernel); ctr: admm_xcd_counter_bytes() of device memory
size_t admm_xcd_counter_bytes();
bool admm_xcd_usable(int N, int H, int W);
hipError_t launch_admm_xcd(const float* x, float2* z, float2* u, float2* work, const float2* tw, const float2* y0s,
                           const uint8_t* masks, int mask_n, const float* mu, const float* tact, unsigned* ctr, unsigned epoch, int N, int H, hipStream_t s);
hipError_t launch_psnr(const float* x, const float* gt, float* out, int N, int HW, hipStream_t s);

}  // namespace pnp
