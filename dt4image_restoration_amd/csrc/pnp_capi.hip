// C ABI of libpnpadmm.so (declared in include/pnpadmm.h): engine object, weight ingest, launch
// sequencing of one PnP-ADMM iteration, kernel-level event timing.
#include "../../include/pnpadmm.h"
#include "pnp_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

using namespace pnp;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(PNP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// No C++ exception may cross the C ABI: every entry point body runs inside this guard.
#define PNP_API_BEGIN try {
#define PNP_API_END(name)                                                                          \
    } catch (const std::bad_alloc&) { return fail(PNP_ERR_NOMEM, name ": out of host memory");     \
    } catch (const std::exception& ex_) { return fail(PNP_ERR_INTERNAL, name ": %s", ex_.what());  \
    } catch (...) { return fail(PNP_ERR_INTERNAL, name ": unknown C++ exception"); }

// Entry points run on the handle's device whatever the caller's current device is, and leave the caller's current
// device as they found it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define PNP_ON_DEVICE(e)                                                                                  \
    DeviceGuard dev_guard_((e)->cfg.device);                                                              \
    if (dev_guard_.err != hipSuccess) return fail(PNP_ERR_HIP, "hipSetDevice(%d): %s", (e)->cfg.device, hipGetErrorString(dev_guard_.err))

constexpr size_t kNParams = 11773857;

struct EventPair {
    hipEvent_t a, b;
    int cls, layer;
    int count;      // kernel launches between the two events (a run of same-class kernels shares one pair)
};

struct LevelBufs {
    float* p;  // ping
    float* q;  // pong
    float* s;  // skip / stage output kept for the up path
    float* pool;  // MaxPool2d(2) of s, written by the kernel that writes s (null if that kernel cannot)
    int c, h, w;
};

}  // namespace

struct pnp_engine {
    pnp_config cfg;
    bool weights_loaded = false;
    bool reset_done = false;
    // denoiser
    float* d_wpack[N_LAYERS] = {};   // packed conv3x3 weights (layers 1..26), raw for 0 and 27
    float* d_bias[N_LAYERS] = {};
    LevelBufs lv[5] = {};
    float* d_partial = nullptr;      // split-K workspace (small problems)
    unsigned* d_arrive = nullptr;    // split-K arrival counters (PNP_SPLITK_INLAUNCH), zero between launches
    Tuning tune;                      // environment overrides, read once in pnp_create
    WinoPlan wplan[N_LAYERS] = {};    // per-layer launch plans, fixed at pnp_create: the weight pack and every launch use
    ConvPlan cplan[N_LAYERS] = {};    // the same plan
    bool wino[N_LAYERS] = {};         // layer runs on the Winograd kernel (weights packed for it)
    bool fuse_last = false;           // last 1x1 layer rides in the epilogue of up4.conv-2
    bool fuse_first = false;          // first layer (2 -> 32) is computed in the staging of inc.conv-1 (F(4x4) 32-channel variant)
    bool pool_ok[4] = {};             // level k's stage output also gets a pooled copy (its producing kernel supports it)
    int bf16_terms = 0;               // bf16 mode: bf16 terms per conv weight (2: hi + lo, the default; 1: PNP_BF16_W1); 0 = f32 mode
    bool act16 = false;               // bf16 mode: the 32-channel level-0 activations (lv[0].p/q/s) are stored as bf16 (ConvArgs.act16)
    uint8_t abits[N_LAYERS] = {};     // bf16 mode, per conv layer: ConvArgs.act16 (bit 0: src0 holds bf16, bit 1: dst holds bf16)
    // data-fidelity stage
    FftPlan plan = {};
    float2* d_work = nullptr;   // [N,H,W] complex scratch
    unsigned* d_fftq = nullptr; // per-XCD ticket / completion counters of the persistent data-fidelity kernel (nullptr: three launches)
    unsigned fftq_epoch = 0;    // launches of that kernel since the counters were zeroed (they are never reset: the kernel subtracts epoch x per-launch advance)
    float2* d_y0s = nullptr;    // [N,H,W] sgn * S y0
    uint8_t* d_masks = nullptr; // [mask_n,H,W] S mask
    int mask_n = 1;
    size_t ws_bytes = 0;
    // profiling
    std::vector<EventPair> events;
    size_t ev_used = 0;
    double cls_ms[PNP_PROFILE_CLASSES] = {};
    int64_t cls_n[PNP_PROFILE_CLASSES] = {};
    double layer_ms[N_LAYERS] = {};
    int64_t layer_n[N_LAYERS] = {};
};

namespace {

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// One HIP event pair around a kernel launch - or, with `defer_end`, around a RUN of same-class launches that follow each
// other on the stream (the 26 conv3x3 launches of a denoiser forward): `count` launches, closed by end().  Event records
// cost ~3 us of stream time each, so the default profile mode brackets the conv run once; PNP_FLAG_PROFILE_LAYERS keeps
// a pair per launch for the per-layer table.
struct Prof {
    pnp_engine* e;
    hipStream_t s;
    EventPair* ep = nullptr;
    bool deferred = false;
    Prof(pnp_engine* e_, hipStream_t s_, int cls, int layer, bool active = true, bool defer_end = false) : e(e_), s(s_), deferred(defer_end) {
        if (!active || !(e->cfg.flags & (PNP_FLAG_PROFILE | PNP_FLAG_PROFILE_LAYERS))) return;
        if (e->ev_used == e->events.size()) {
            EventPair p{};
            if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
            e->events.push_back(p);
        }
        ep = &e->events[e->ev_used++];
        ep->cls = cls;
        ep->layer = layer;
        ep->count = 1;
        (void)hipEventRecord(ep->a, s);
    }
    void end(int count) {
        if (ep) { ep->count = count; (void)hipEventRecord(ep->b, s); ep = nullptr; }
    }
    ~Prof() {
        if (ep && !deferred) (void)hipEventRecord(ep->b, s);
        else if (ep) { ep->count = 0; (void)hipEventRecord(ep->b, s); }     // a run left early (error path)
    }
};

int make_twiddles(int L, float2** out) {
    std::vector<float2> t(L);
    for (int m = 0; m < L; ++m) {
        const double a = -2.0 * M_PI * (double)m / (double)L;
        t[m] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    HIP_TRY(hipMalloc((void**)out, sizeof(float2) * L));
    HIP_TRY(hipMemcpy(*out, t.data(), sizeof(float2) * L, hipMemcpyHostToDevice));
    return PNP_OK;
}

// The 27 MFMA convs + first/last layer of one denoiser forward.  Image channel = ximg, or Re(z-u).
int run_unet(pnp_engine* e, const float* ximg, const float2* z, const float2* u, const float* sigma,
             const float* tact, float* out, hipStream_t s) {
    const int N = e->cfg.n, H = e->cfg.h, W = e->cfg.w;
    if (e->cfg.flags & PNP_FLAG_NO_DENOISER)
        return fail(PNP_ERR_STATE, "this handle was created with PNP_FLAG_NO_DENOISER");
    if (!e->fuse_first) {
        Prof p(e, s, 1, 0);
        HIP_TRY(launch_conv_first(ximg, z, u, sigma, tact, e->d_wpack[0], e->d_bias[0], e->lv[0].p, N, H, W, s, e->act16));
    }
    const bool per_layer = (e->cfg.flags & PNP_FLAG_PROFILE_LAYERS) != 0;
#ifdef PNP_DIAG
    static hipStream_t diag_pool[8];
    static hipEvent_t diag_fork, diag_join[8];
    static int diag_made = 0;
    const int diag_k = getenv("PNP_DIAG_STREAMS") ? std::min(8, std::max(1, atoi(getenv("PNP_DIAG_STREAMS")))) : 1;
    if (diag_k > 1) {
        if (!diag_made) {
            for (int i = 0; i < 8; ++i) { HIP_TRY(hipStreamCreateWithFlags(&diag_pool[i], hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&diag_join[i], hipEventDisableTiming)); }
            HIP_TRY(hipEventCreateWithFlags(&diag_fork, hipEventDisableTiming));
            diag_made = 1;
        }
        HIP_TRY(hipEventRecord(diag_fork, s));
        for (int i = 0; i < diag_k; ++i) HIP_TRY(hipStreamWaitEvent(diag_pool[i], diag_fork, 0));
    }
    struct DiagJoin {
        hipStream_t s; int k; hipStream_t* pool; hipEvent_t* ev;
        ~DiagJoin() { if (k > 1) for (int i = 0; i < k; ++i) { (void)hipEventRecord(ev[i], pool[i]); (void)hipStreamWaitEvent(s, ev[i], 0); } }
    } diag_joiner{s, diag_k, diag_pool, diag_join};
#endif
    Prof run(e, s, 0, -1, !per_layer, true);              // one event pair around the whole conv3x3 run
    int run_launches = 0;
    auto conv = [&](int li, const float* src0, const float* src1, float* dst, int lvl, float* pooled = nullptr,
                    bool src_is_pooled = false) -> int {
        const LayerSpec& L = kLayers[li];
        ConvArgs a{};
        a.pooled = pooled;
        a.src0 = src0; a.src1 = src1; a.wpack = e->d_wpack[li]; a.bias = e->d_bias[li]; a.dst = dst; a.partial = e->d_partial; a.arrive = e->d_arrive; a.tact = tact;
        a.bf16 = e->bf16_terms;
        a.act16 = e->abits[li];
        a.N = N; a.H = H >> lvl; a.W = W >> lvl; a.Cin = L.cin; a.Cskip = L.cskip; a.Cout = L.cout;
        if (L.src == SRC_UPCAT) {
            const int hs = a.H / 2, ws = a.W / 2;
            a.rh = a.H > 1 ? (float)(hs - 1) / (float)(a.H - 1) : 0.f;
            a.rw = a.W > 1 ? (float)(ws - 1) / (float)(a.W - 1) : 0.f;
        }
        int src_mode = (L.src == SRC_POOL && src_is_pooled) ? (int)SRC_PLAIN : L.src;   // pooled copy already exists
        if (li == 1 && e->fuse_first) {                    // inc.conv-1 evaluates the first layer while staging its patch
            src_mode = SRC_FIRST;
            a.first_w = e->d_wpack[0]; a.first_b = e->d_bias[0]; a.first_sigma = sigma;
            a.last_ximg = ximg; a.last_z = z; a.last_u = u;
        }
#ifdef PNP_DIAG
        if (const char* dv = getenv("PNP_DIAG_L0")) a.diag = L.level == 0 ? atoi(dv) : 0;
        if (const char* dv = getenv("PNP_DIAG_ALL")) a.diag = atoi(dv);
#endif
        Prof p(e, s, 0, li, per_layer);
        ++run_launches;
        hipStream_t ls = s;
#ifdef PNP_DIAG
        // PNP_DIAG_STREAMS=K (timing only, results wrong): conv launch i goes to side stream i % K, so consecutive layers do NOT wait for
        // each other - the ceiling of any scheme that overlaps a layer's tail with its successor's head (profiles/r05_ablation.md)
        if (diag_k > 1) ls = diag_pool[li % diag_k];
#endif
        if (e->wino[li] && e->wplan[li].algo == 4) HIP_TRY(launch_conv3x3_winograd4(a, e->wplan[li], src_mode, ls));
        else if (e->wino[li]) HIP_TRY(launch_conv3x3_winograd(a, e->wplan[li], src_mode, ls));
        else HIP_TRY(launch_conv3x3(a, e->cplan[li], src_mode, ls));
        return PNP_OK;
    };
    int rc;
    // inc; a stage's last conv also writes the 2x2 max-pooled copy the next stage starts from, when its kernel can
    if ((rc = conv(1, e->lv[0].p, nullptr, e->lv[0].q, 0))) return rc;
    if ((rc = conv(2, e->lv[0].q, nullptr, e->lv[0].s, 0, e->pool_ok[0] ? e->lv[0].pool : nullptr))) return rc;
    // down1..4: conv-0 reads the pooled copy (or pools the previous stage output while staging)
    for (int k = 1; k <= 4; ++k) {
        const int b = 3 * k;
        const bool pooled_in = e->pool_ok[k - 1];
        if ((rc = conv(b, pooled_in ? e->lv[k - 1].pool : e->lv[k - 1].s, nullptr, e->lv[k].p, k, nullptr, pooled_in))) return rc;
        if ((rc = conv(b + 1, e->lv[k].p, nullptr, e->lv[k].q, k))) return rc;
        if ((rc = conv(b + 2, e->lv[k].q, nullptr, e->lv[k].s, k, (k < 4 && e->pool_ok[k]) ? e->lv[k].pool : nullptr))) return rc;
    }
    // up1..4: conv-0 reads cat([skip, bilinear_up(low)]) while staging
    const float* low = e->lv[4].s;
    for (int k = 3; k >= 1; --k) {
        const int b = 15 + 3 * (3 - k);
        if ((rc = conv(b, e->lv[k].s, low, e->lv[k].p, k))) return rc;
        if ((rc = conv(b + 1, e->lv[k].p, nullptr, e->lv[k].q, k))) return rc;
        if ((rc = conv(b + 2, e->lv[k].q, nullptr, e->lv[k].p, k))) return rc;
        low = e->lv[k].p;
    }
    if ((rc = conv(24, e->lv[0].s, low, e->lv[0].p, 0))) return rc;
    if ((rc = conv(25, e->lv[0].p, nullptr, e->lv[0].q, 0))) return rc;
    if (e->fuse_last) {
        // up4.conv-2 with the last layer (1x1 + residual + clamp) fused into its epilogue: writes `out` directly
        const LayerSpec& L = kLayers[26];
        ConvArgs a{};
        a.src0 = e->lv[0].q; a.wpack = e->d_wpack[26]; a.bias = e->d_bias[26]; a.dst = e->lv[0].p; a.partial = e->d_partial; a.arrive = e->d_arrive;
        a.bf16 = e->bf16_terms;
        a.act16 = e->abits[26] & 1;
        a.tact = tact; a.N = N; a.H = H; a.W = W; a.Cin = L.cin; a.Cskip = 0; a.Cout = L.cout;
        a.last_w = e->d_wpack[27]; a.last_b = e->d_bias[27]; a.last_ximg = ximg; a.last_z = z; a.last_u = u; a.last_out = out;
#ifdef PNP_DIAG
        if (const char* dv = getenv("PNP_DIAG_L0")) a.diag = atoi(dv);
#endif
        {
            Prof p(e, s, 0, 26, per_layer);
            ++run_launches;
            if (e->wino[26] && e->wplan[26].algo == 4) HIP_TRY(launch_conv3x3_winograd4(a, e->wplan[26], SRC_PLAIN, s));
            else if (e->wino[26]) HIP_TRY(launch_conv3x3_winograd(a, e->wplan[26], SRC_PLAIN, s));
            else HIP_TRY(launch_conv3x3(a, e->cplan[26], SRC_PLAIN, s));
        }
        run.end(run_launches);
    } else {
        if ((rc = conv(26, e->lv[0].q, nullptr, e->lv[0].p, 0))) return rc;
        run.end(run_launches);
        Prof p(e, s, 2, 27);
        HIP_TRY(launch_conv_last(e->lv[0].p, ximg, z, u, tact, e->d_wpack[27], e->d_bias[27], out, N, H, W, s));
    }
    return PNP_OK;
}

int run_prox_dual(pnp_engine* e, const float* mu, const float* tact, const float* x, float2* z, float2* u,
                  hipStream_t s) {
    const int N = e->cfg.n, H = e->cfg.h, W = e->cfg.w;
    if (H == 128 && W == 128 && N >= e->tune.slice128_min_n) {   // chip-filling batches of the reference's slice size: 37 B/px
        Prof p(e, s, 4, -1);
        HIP_TRY(launch_admm_slice128(x, z, u, e->plan.tw_w, e->d_y0s, e->d_masks, e->mask_n, mu, tact, N, s));
        return PNP_OK;
    }
    if (e->d_fftq != nullptr) {                           // square 256 / 512 slices, >= 8 of them: one persistent launch, scratch stays in L2
        Prof p(e, s, 4, -1);
        HIP_TRY(launch_admm_xcd(x, z, u, e->d_work, e->plan.tw_w, e->d_y0s, e->d_masks, e->mask_n, mu, tact, e->d_fftq, e->fftq_epoch, N, H, s));
        ++e->fftq_epoch;
        return PNP_OK;
    }
    {
        Prof p(e, s, 3, -1);
        HIP_TRY(launch_fft_rows_fwd_admm(x, u, e->d_work, e->plan.tw_w, tact, N, H, W, s));
    }
    {
        Prof p(e, s, 4, -1);
        HIP_TRY(launch_fft_cols_prox(e->d_work, e->plan.tw_h, e->d_y0s, e->d_masks, e->mask_n, mu, tact, N, H, W, s));
    }
    {
        Prof p(e, s, 3, -1);
        HIP_TRY(launch_fft_rows_inv_admm(e->d_work, x, z, u, e->plan.tw_w, tact, N, H, W, s));
    }
    return PNP_OK;
}

}  // namespace

extern "C" {

const char* pnp_last_error(void) { return g_err.c_str(); }
const char* pnp_version(void) { return "pnpadmm 0.4 (gfx950, f32 MFMA; optional bf16-operand convs with two-term weights)"; }

// Does layer `li` (a stage's last conv, planned already) also write the 2x2 max-pooled copy of its output?  Its output must have
// even sides and its kernel must support it (a Winograd kernel, or the direct kernel's LDS-epilogue plan).  ONE predicate for the
// consumer's planned source mode and for the launch-time pool_ok[] flags.
static bool pooled_copy_ok(const pnp_engine* e, int li) {
    const LayerSpec& L = kLayers[li];
    const int lh = e->cfg.h >> L.level, lw = e->cfg.w >> L.level;
    return lh % 2 == 0 && lw % 2 == 0 && (e->wino[li] || conv3x3_pooled_output_ok(e->cplan[li]));
}

static int create_impl(const pnp_config* cfg, pnp_engine* e) {
    e->cfg = *cfg;
    e->tune = tuning_from_env();
    const size_t N = cfg->n, H = cfg->h, W = cfg->w;
    const bool bf16 = (cfg->flags & PNP_FLAG_BF16_CONVS) != 0;
    e->bf16_terms = bf16 ? (e->tune.bf16_w1 ? 1 : 2) : 0;
    static const int chan[5] = {32, 64, 128, 256, 512};
    for (int k = 0; k < 5; ++k) {
        LevelBufs& L = e->lv[k];
        L.c = chan[k]; L.h = (int)(H >> k); L.w = (int)(W >> k);
        if (cfg->flags & PNP_FLAG_NO_DENOISER) continue;   // k-space-only handle: no activation planes
        const size_t bytes_full = N * L.h * L.w * L.c * sizeof(float);
        float** bufs[4] = {&L.p, &L.q, &L.s, &L.pool};
        for (auto b : bufs) {
            const size_t bytes = (b == &L.pool) ? (k < 4 ? bytes_full / 4 : 0) : bytes_full;
            if (bytes == 0) continue;
            hipError_t er = hipMalloc((void**)b, bytes);
            if (er != hipSuccess) return fail(PNP_ERR_NOMEM, "activation planes: %s", hipGetErrorString(er));
            e->ws_bytes += bytes;
        }
    }
    if (!(cfg->flags & PNP_FLAG_NO_DENOISER)) {
        // launch plans of the 26 conv3x3 layers: fixed here, used by the weight pack and by every launch
        size_t pf = 0;
        for (int li = 1; li < N_LAYERS - 1; ++li) {
            const LayerSpec& L = kLayers[li];
            const int lh = cfg->h >> L.level, lw = cfg->w >> L.level;
            // the source mode the launch will use: a pooled stage input is read PLAIN from the producer's pooled copy, which
            // exists iff the producing layer (li - 1, planned just before) runs a Winograd kernel or the direct LDS-epilogue plan
            // (the same predicate as pool_ok[] below: the producer's output size - twice this layer's - must be even, which it
            // always is, and its kernel must be one that writes the pooled copy)
            int src_mode = L.src;
            if (L.src == SRC_POOL && pooled_copy_ok(e, li - 1)) src_mode = SRC_PLAIN;
            e->wplan[li] = winograd_plan(cfg->n, lh, lw, L.cin, L.cout, src_mode, e->tune);
            if (li == 26 && e->wplan[li].algo == 4 && !(cfg->flags & PNP_FLAG_KEEP_STAGES) &&
                (e->tune.no_f4_fused_last || e->wplan[li].bn != 32 || e->wplan[li].mt != 32)) {
                // up4.conv-2 carries the fused last layer (1x1 conv + residual + clamp) in its epilogue: the F(4x4) kernel's
                // 32-channel variant has it (DPP reduce-scatter over a pixel's channels); PNP_NO_F4_FUSED_LAST puts the layer
                // back on the F(2x2) kernel, which walks whole pixels there
                Tuning t2 = e->tune;
                t2.no_f4 = true;
                e->wplan[li] = winograd_plan(cfg->n, lh, lw, L.cin, L.cout, src_mode, t2);
            }
            // (the producer / consumer kernel's upsample is the separable form: only on heights with the regular line structure)
            const bool ws_ok = !e->tune.bf16_no_ws && (src_mode != SRC_UPCAT || upsample_lines_regular(lh));
            e->cplan[li] = conv3x3_plan(cfg->n, lh, lw, L.cin, L.cout, bf16, src_mode, ws_ok);
            // up4.conv-2 (fused last layer): on the producer / consumer kernel only in the two-term mode, where its producers evaluate the last
            // layer (OFFLOAD); the one-term form of that tile measured slower than conv_kernels.hip (conv3x3_plan)
            e->cplan[li].holdhi = e->tune.bf16_no_holdhi ? 0 : 1;
            if (li == 26 && e->cplan[li].nt == 1 && (e->bf16_terms != 2 || e->tune.bf16_no_holdhi)) e->cplan[li].ws = 0;
            e->wino[li] = e->wplan[li].use && !bf16;
            if (e->wino[li] && e->wplan[li].algo == 4) continue;          // (per-slice descriptors)
            if (!conv3x3_tensor_fits(cfg->n, lh, lw, L.cin, L.cout))
                return fail(PNP_ERR_INVALID, "pnp_create: layer %d's output tensor (%d x %d x %d x %d floats) reaches 2 GiB, past this "
                            "kernel's buffer descriptor: use a smaller batch per handle", li, cfg->n, lh, lw, L.cout);
            if (e->wino[li]) continue;
            const size_t f = conv3x3_partial_floats(e->cplan[li], cfg->n, lh, lw, L.cout);
            if (f > pf) pf = f;
        }
        if (pf > 0) {
            if (hipMalloc((void**)&e->d_partial, pf * sizeof(float)) != hipSuccess) return fail(PNP_ERR_NOMEM, "split-K workspace");
            e->ws_bytes += pf * sizeof(float);
            if (e->tune.splitk_inlaunch) {                 // one counter per output tile of a split-K launch: never more than 4096 tiles
                if (hipMalloc((void**)&e->d_arrive, 4096 * sizeof(unsigned)) != hipSuccess ||
                    hipMemset(e->d_arrive, 0, 4096 * sizeof(unsigned)) != hipSuccess) return fail(PNP_ERR_NOMEM, "split-K counters");
            }
        }
        // which stage outputs get a pooled copy: the producing conv (layers 2, 5, 8, 11) must run a kernel whose epilogue
        // goes through LDS - the Winograd kernel, or the direct kernel's Cout = 32 configuration on a large problem
        for (int k = 0; k < 4; ++k) {
            const int li = 3 * k + 2;
            const LayerSpec& L = kLayers[li];
            const int lh = cfg->h >> L.level, lw = cfg->w >> L.level;
            (void)lh; (void)lw;
            e->pool_ok[k] = pooled_copy_ok(e, li);
        }
        e->fuse_last = !(cfg->flags & PNP_FLAG_KEEP_STAGES) && (e->wino[26] || conv3x3_pooled_output_ok(e->cplan[26]));
        e->fuse_first = e->wino[1] && e->wplan[1].algo == 4 && e->wplan[1].bn == 32 && e->wplan[1].mt == 32 && !e->tune.no_f4_fused_first;
        // bf16 mode: the five 32-channel level-0 layers exchange bf16 tensors (same bits the staging would round to; half the
        // bytes of the HBM-bound level).  Needs the plan that has the variant on all five and the pooled copy for down1
        // (always so today); a KEEP_STAGES handle keeps f32 stages for pnp_unet_read_stage.
        e->act16 = bf16 && !(cfg->flags & PNP_FLAG_KEEP_STAGES) && !e->tune.bf16_f32_acts && e->pool_ok[0];
        for (int li : {1, 2, 24, 25, 26}) e->act16 = e->act16 && conv3x3_pooled_output_ok(e->cplan[li]);
        if (e->act16) {
            // ... and so do the layers of the producer / consumer kernel among themselves: a tensor is bf16 when the launch that
            // writes it and every launch that reads it as src0 (next layer, PLAIN or POOL; the decoder layer taking it as its
            // skip tensor) can; the low-res input of an upsample (outputs of layers 14, 17, 20, 23) stays f32 - its consumer
            // rounds after interpolating - and so do the pooled copy of level 0 and the input of the unfused 1x1 conv
            auto can = [&](int li) { return li >= 1 && li <= 26 && ((li < 3 || li > 23) || e->cplan[li].ws != 0); };
            bool out16[N_LAYERS] = {};
            out16[0] = true;                                       // conv_first -> inc.conv-1
            for (int li = 1; li <= 25; ++li) {
                if (li == 14 || li == 17 || li == 20 || li == 23) continue;
                const int skip_reader = li == 2 ? 24 : (li == 5 ? 21 : (li == 8 ? 18 : (li == 11 ? 15 : 0)));
                // (a stage's last layer: the next stage reads the f32 pooled copy instead when there is one)
                const bool next_reads = !(skip_reader != 0 && e->pool_ok[kLayers[li].level]);
                out16[li] = can(li) && (!next_reads || can(li + 1)) && (skip_reader == 0 || can(skip_reader));
            }
            for (int li = 1; li <= 26; ++li) {
                bool in16;
                if (kLayers[li].src == SRC_POOL && e->pool_ok[kLayers[li].level - 1]) in16 = false;   // reads the producer's f32 pooled copy (today: li = 3)
                else if (li == 15 || li == 18 || li == 21 || li == 24) in16 = out16[li == 15 ? 11 : (li == 18 ? 8 : (li == 21 ? 5 : 2))];
                else in16 = out16[li - 1];
                // the level-0 kernel writes bf16 only from its bf16-source variant
                if ((li < 3 || li > 23) && !in16) out16[li] = false;
                e->abits[li] = (uint8_t)((in16 ? 1 : 0) | (out16[li] ? 2 : 0));
            }
            // round 5: level 0's pooled copy (inc.conv-2 -> down1.conv-0) as bf16 too when both layers run the producer / consumer kernel:
            // rounding to nearest even is monotonic, so bf16(max(a, b, c, d)) == max(bf16(a), ...) - the bits down1.conv-0 stages are the
            // ones it rounded the f32 copy to, at half the bytes written and read (bit 2 of the writer's act16, bit 0 of the reader's)
            if ((e->abits[2] & 2) && e->cplan[2].ws && e->cplan[3].ws && e->cplan[2].holdhi && e->bf16_terms == 2 && e->pool_ok[0] &&
                kLayers[3].src == SRC_POOL) {
                e->abits[2] |= 4;
                e->abits[3] |= 1;
            }
        }
    }
    const size_t cbytes = N * H * W * sizeof(float2);
    if (hipMalloc((void**)&e->d_work, cbytes) != hipSuccess || hipMalloc((void**)&e->d_y0s, cbytes) != hipSuccess ||
        hipMalloc((void**)&e->d_masks, N * H * W) != hipSuccess)
        return fail(PNP_ERR_NOMEM, "k-space scratch");
    e->ws_bytes += 2 * cbytes + N * H * W;
    if (e->tune.fft_xcd && admm_xcd_usable(cfg->n, cfg->h, cfg->w)) {
        if (hipMalloc((void**)&e->d_fftq, admm_xcd_counter_bytes()) != hipSuccess || hipMemset(e->d_fftq, 0, admm_xcd_counter_bytes()) != hipSuccess)
            return fail(PNP_ERR_NOMEM, "data-fidelity work queues");
        e->ws_bytes += admm_xcd_counter_bytes();
    }
    e->plan.h = cfg->h; e->plan.w = cfg->w;
    int rc;
    if ((rc = make_twiddles(cfg->h, &e->plan.tw_h)) || (rc = make_twiddles(cfg->w, &e->plan.tw_w))) return rc;
    return PNP_OK;
}

int pnp_create(const pnp_config* cfg, pnp_handle* out) {
    PNP_API_BEGIN
    if (!cfg || !out) return fail(PNP_ERR_INVALID, "pnp_create: null argument");
    *out = nullptr;
    if (cfg->n < 1 || cfg->h < 16 || cfg->w < 16 || cfg->h % 16 || cfg->w % 16)
        return fail(PNP_ERR_INVALID, "pnp_create: need n >= 1 and h, w multiples of 16 (got n=%d h=%d w=%d)", cfg->n,
                    cfg->h, cfg->w);
    if (cfg->h > 1024 || cfg->w > 1024) return fail(PNP_ERR_INVALID, "pnp_create: h, w <= 1024");
    // the conv kernels' x2 upsample reads two compile-time source lines per output row (commit_lo / interpolate): holds in float32
    // for every even height up to 1024 - checked, not assumed
    for (int k = 0; k < 4; ++k)
        if (!(cfg->flags & PNP_FLAG_NO_DENOISER) && !upsample_lines_regular(cfg->h >> k))
            return fail(PNP_ERR_INVALID, "pnp_create: upsample to %d rows is not line-regular in float32", cfg->h >> k);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(PNP_ERR_INVALID, "pnp_create: device %d of %d", cfg->device, ndev);
    pnp_engine* e = new (std::nothrow) pnp_engine();
    if (!e) return fail(PNP_ERR_NOMEM, "pnp_create: out of host memory");
    e->cfg = *cfg;
    int rc;
    {
        DeviceGuard g(cfg->device);
        if (g.err != hipSuccess) { delete e; return fail(PNP_ERR_HIP, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(g.err)); }
        try { rc = create_impl(cfg, e); }
        catch (...) { const std::string keep = g_err; pnp_destroy(e); g_err = keep; throw; }
        if (rc != PNP_OK) { const std::string keep = g_err; pnp_destroy(e); g_err = keep; return rc; }
    }
    *out = e;
    return PNP_OK;
    PNP_API_END("pnp_create")
}

int pnp_destroy(pnp_handle e) {
    PNP_API_BEGIN
    if (!e) return PNP_OK;
    DeviceGuard g(e->cfg.device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < N_LAYERS; ++i) { (void)hipFree(e->d_wpack[i]); (void)hipFree(e->d_bias[i]); }
    for (auto& L : e->lv) { (void)hipFree(L.p); (void)hipFree(L.q); (void)hipFree(L.s); (void)hipFree(L.pool); }
    (void)hipFree(e->d_work); (void)hipFree(e->d_fftq); (void)hipFree(e->d_y0s); (void)hipFree(e->d_masks); (void)hipFree(e->d_partial); (void)hipFree(e->d_arrive);
    (void)hipFree(e->plan.tw_h); (void)hipFree(e->plan.tw_w);
    for (auto& p : e->events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    delete e;
    return PNP_OK;
    PNP_API_END("pnp_destroy")
}

size_t pnp_workspace_bytes(pnp_handle e) { return e ? e->ws_bytes : 0; }
int pnp_bf16_weight_terms(pnp_handle e) { return e ? e->bf16_terms : 0; }

// All-or-nothing: every layer is packed on the host and uploaded into NEW device buffers first; the handle's buffers are
// replaced only when all 56 uploads succeeded, so a failure leaves the handle exactly as it was (an earlier successful
// load stays usable, a never-loaded handle stays "weights not loaded").
int pnp_load_unet_weights(pnp_handle e, const float* blob, size_t n_floats) {
    PNP_API_BEGIN
    if (!e || !blob) return fail(PNP_ERR_INVALID, "pnp_load_unet_weights: null argument");
    if (n_floats != kNParams)
        return fail(PNP_ERR_INVALID, "pnp_load_unet_weights: expected %zu floats (56 tensors of UNet(2,1)), got %zu",
                    kNParams, n_floats);
    if (e->cfg.flags & PNP_FLAG_NO_DENOISER) return fail(PNP_ERR_STATE, "pnp_load_unet_weights: handle created with PNP_FLAG_NO_DENOISER");
    PNP_ON_DEVICE(e);
    float* nw_pack[N_LAYERS] = {};
    float* nw_bias[N_LAYERS] = {};
    auto drop_new = [&]() { for (int i = 0; i < N_LAYERS; ++i) { (void)hipFree(nw_pack[i]); (void)hipFree(nw_bias[i]); } };
    const bool bf16 = (e->cfg.flags & PNP_FLAG_BF16_CONVS) != 0;
    size_t off = 0;
    std::vector<float> tmp;
    hipError_t er = hipSuccess;
    try {
        for (int li = 0; li < N_LAYERS && er == hipSuccess; ++li) {
            const LayerSpec& L = kLayers[li];
            const size_t nw = (size_t)L.cout * L.cin * L.ksize * L.ksize;
            const float* w = blob + off;
            const float* b = blob + off + nw;
            off += nw + L.cout;
            size_t pf;
            const float* src;
            if (li == 0 || li == N_LAYERS - 1) {   // first (2->32, OIHW as is) and last (1x1) layers
                pf = nw; src = w;
            } else if (e->wino[li] && e->wplan[li].algo == 4) {
                pf = winograd4_pack_floats(L.cin, L.cout);
                tmp.assign(pf, 0.f);
                pack_winograd4_weights(w, L.cin, L.cout, e->wplan[li].ck, tmp.data());
                src = tmp.data();
            } else if (e->wino[li]) {
                pf = winograd_pack_floats(L.cin, L.cout);
                tmp.assign(pf, 0.f);
                pack_winograd_weights(w, L.cin, L.cout, e->wplan[li].ck, tmp.data());
                src = tmp.data();
            } else {
                pf = bf16 ? conv3x3_pack_floats_bf16(L.cin, L.cout, e->bf16_terms) : conv3x3_pack_floats(L.cin, L.cout);
                tmp.assign(pf, 0.f);
                if (bf16) pack_conv3x3_weights_bf16(w, L.cin, L.cout, e->cplan[li].ck, e->bf16_terms, tmp.data());
                else pack_conv3x3_weights(w, L.cin, L.cout, e->cplan[li].ck, tmp.data());
                src = tmp.data();
            }
            if ((er = hipMalloc((void**)&nw_pack[li], pf * sizeof(float))) != hipSuccess) break;
            if ((er = hipMemcpy(nw_pack[li], src, pf * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
            if ((er = hipMalloc((void**)&nw_bias[li], L.cout * sizeof(float))) != hipSuccess) break;
            er = hipMemcpy(nw_bias[li], b, L.cout * sizeof(float), hipMemcpyHostToDevice);
        }
    } catch (...) { drop_new(); throw; }
    if (er != hipSuccess) { drop_new(); return fail(PNP_ERR_HIP, "pnp_load_unet_weights: upload failed: %s (handle unchanged)", hipGetErrorString(er)); }
    (void)hipDeviceSynchronize();                        // no launch still reads the buffers being replaced
    for (int li = 0; li < N_LAYERS; ++li) {
        (void)hipFree(e->d_wpack[li]); (void)hipFree(e->d_bias[li]);
        e->d_wpack[li] = nw_pack[li]; e->d_bias[li] = nw_bias[li];
    }
    e->weights_loaded = true;
    return PNP_OK;
    PNP_API_END("pnp_load_unet_weights")
}

int pnp_reset(pnp_handle e, const float* x0, const float* y0, const uint8_t* mask, int mask_n, float* x, float* z,
              float* u, void* stream) {
    PNP_API_BEGIN
    if (!e || !x0 || !y0 || !mask || !x || !z || !u) return fail(PNP_ERR_INVALID, "pnp_reset: null argument");
    if (mask_n != 1 && mask_n != e->cfg.n) return fail(PNP_ERR_INVALID, "pnp_reset: mask_n must be 1 or n=%d", e->cfg.n);
    if (!is_pow2(e->cfg.h) || !is_pow2(e->cfg.w))
        return fail(PNP_ERR_INVALID, "pnp_reset: the k-space stage needs power-of-two h, w (got %dx%d)", e->cfg.h, e->cfg.w);
    PNP_ON_DEVICE(e);
    e->mask_n = mask_n;
    // the two experimental in-launch hand-over schemes keep counters between launches (PNP_SPLITK_INLAUNCH: arrival counters that return to zero;
    // PNP_FFT_XCD: ticket / completion counters read against a launch epoch): a launch that faulted half way would leave them out of step for good,
    // so an episode starts from zero
    if (e->d_arrive) HIP_TRY(hipMemsetAsync(e->d_arrive, 0, 4096 * sizeof(unsigned), (hipStream_t)stream));
    if (e->d_fftq) { HIP_TRY(hipMemsetAsync(e->d_fftq, 0, admm_xcd_counter_bytes(), (hipStream_t)stream)); e->fftq_epoch = 0; }
    HIP_TRY(launch_reset((const float2*)x0, (const float2*)y0, mask, mask_n, x, (float2*)z, (float2*)u, e->d_y0s,
                         e->d_masks, e->cfg.n, e->cfg.h, e->cfg.w, (hipStream_t)stream));
    e->reset_done = true;
    return PNP_OK;
    PNP_API_END("pnp_reset")
}

int pnp_set_kspace(pnp_handle e, const float* y0, const uint8_t* mask, int mask_n, void* stream) {
    PNP_API_BEGIN
    if (!e || !y0 || !mask) return fail(PNP_ERR_INVALID, "pnp_set_kspace: null argument");
    if (mask_n != 1 && mask_n != e->cfg.n) return fail(PNP_ERR_INVALID, "pnp_set_kspace: mask_n must be 1 or n=%d", e->cfg.n);
    if (!is_pow2(e->cfg.h) || !is_pow2(e->cfg.w))
        return fail(PNP_ERR_INVALID, "pnp_set_kspace: the k-space stage needs power-of-two h, w (got %dx%d)", e->cfg.h, e->cfg.w);
    PNP_ON_DEVICE(e);
    e->mask_n = mask_n;
    HIP_TRY(launch_reset(nullptr, (const float2*)y0, mask, mask_n, nullptr, nullptr, nullptr, e->d_y0s, e->d_masks,
                         e->cfg.n, e->cfg.h, e->cfg.w, (hipStream_t)stream));
    e->reset_done = true;
    return PNP_OK;
    PNP_API_END("pnp_set_kspace")
}

int pnp_step(pnp_handle e, const float* mu, const float* sigma_d, const float* t_action, float* x, float* z, float* u,
             float* t_state, uint8_t* done, void* stream) {
    PNP_API_BEGIN
    if (!e || !mu || !sigma_d || !x || !z || !u) return fail(PNP_ERR_INVALID, "pnp_step: null argument");
    if (!e->weights_loaded) return fail(PNP_ERR_STATE, "pnp_step: denoiser weights not loaded (pnp_load_unet_weights)");
    if (!e->reset_done) return fail(PNP_ERR_STATE, "pnp_step: pnp_reset has not been called");
    PNP_ON_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = run_unet(e, nullptr, (const float2*)z, (const float2*)u, sigma_d, t_action, x, s))) return rc;
    if ((rc = run_prox_dual(e, mu, t_action, x, (float2*)z, (float2*)u, s))) return rc;
    if (t_state || done) {
        Prof p(e, s, 5, -1);
        HIP_TRY(launch_finish(t_action, t_state, done, e->cfg.n, s));
    }
    return PNP_OK;
    PNP_API_END("pnp_step")
}

int pnp_denoise(pnp_handle e, const float* x_in, const float* sigma, float* out, void* stream) {
    PNP_API_BEGIN
    if (!e || !x_in || !sigma || !out) return fail(PNP_ERR_INVALID, "pnp_denoise: null argument");
    if (!e->weights_loaded) return fail(PNP_ERR_STATE, "pnp_denoise: denoiser weights not loaded");
    PNP_ON_DEVICE(e);
    return run_unet(e, x_in, nullptr, nullptr, sigma, nullptr, out, (hipStream_t)stream);
    PNP_API_END("pnp_denoise")
}

int pnp_fft2c(pnp_handle e, const float* in, float* out, int batch, int hh, int ww, int inverse, void* stream) {
    PNP_API_BEGIN
    if (!e || !in || !out) return fail(PNP_ERR_INVALID, "pnp_fft2c: null argument");
    if (hh != e->cfg.h || ww != e->cfg.w || batch < 1 || batch > e->cfg.n)
        return fail(PNP_ERR_INVALID, "pnp_fft2c: shape [%d,%d,%d] does not fit the engine [%d,%d,%d]", batch, hh, ww,
                    e->cfg.n, e->cfg.h, e->cfg.w);
    if (!is_pow2(hh) || !is_pow2(ww)) return fail(PNP_ERR_INVALID, "pnp_fft2c: power-of-two sizes only");
    PNP_ON_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    // fft_c = S . FFT . S : fold both shifts into the load/store indices of the two passes
    {
        Prof p(e, s, 3, -1);
        HIP_TRY(launch_fft_rows_generic((const float2*)in, (float2*)out, e->plan.tw_w, batch, hh, ww, inverse, ww / 2,
                                        ww / 2, s));
    }
    {
        Prof p(e, s, 4, -1);
        HIP_TRY(launch_fft_cols_generic((float2*)out, e->plan.tw_h, batch, hh, ww, inverse, hh / 2, hh / 2, s));
    }
    return PNP_OK;
    PNP_API_END("pnp_fft2c")
}

int pnp_prox_dual(pnp_handle e, const float* mu, const float* t_action, const float* x, float* z, float* u,
                  void* stream) {
    PNP_API_BEGIN
    if (!e || !mu || !x || !z || !u) return fail(PNP_ERR_INVALID, "pnp_prox_dual: null argument");
    if (!e->reset_done) return fail(PNP_ERR_STATE, "pnp_prox_dual: pnp_reset has not been called");
    PNP_ON_DEVICE(e);
    return run_prox_dual(e, mu, t_action, x, (float2*)z, (float2*)u, (hipStream_t)stream);
    PNP_API_END("pnp_prox_dual")
}

int pnp_psnr(pnp_handle e, const float* x, const float* gt, float* out, void* stream) {
    PNP_API_BEGIN
    if (!e || !x || !gt || !out) return fail(PNP_ERR_INVALID, "pnp_psnr: null argument");
    PNP_ON_DEVICE(e);
    Prof p(e, (hipStream_t)stream, 5, -1);
    HIP_TRY(launch_psnr(x, gt, out, e->cfg.n, e->cfg.h * e->cfg.w, (hipStream_t)stream));
    return PNP_OK;
    PNP_API_END("pnp_psnr")
}

size_t pnp_snapshot_bytes(pnp_handle e) {
    if (!e) return 0;
    const size_t px = (size_t)e->cfg.n * e->cfg.h * e->cfg.w;
    return px * (4 + 8 + 8) + (size_t)e->cfg.n * 4;
}

int pnp_snapshot(pnp_handle e, const float* x, const float* z, const float* u, const float* t_state, void* dst,
                 void* stream) {
    PNP_API_BEGIN
    if (!e || !x || !z || !u || !dst) return fail(PNP_ERR_INVALID, "pnp_snapshot: null argument");
    PNP_ON_DEVICE(e);
    const size_t px = (size_t)e->cfg.n * e->cfg.h * e->cfg.w;
    char* d = static_cast<char*>(dst);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(d, x, px * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + px * 4, z, px * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + px * 12, u, px * 8, hipMemcpyDeviceToDevice, s));
    if (t_state) HIP_TRY(hipMemcpyAsync(d + px * 20, t_state, (size_t)e->cfg.n * 4, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(hipMemsetAsync(d + px * 20, 0, (size_t)e->cfg.n * 4, s));
    return PNP_OK;
    PNP_API_END("pnp_snapshot")
}

int pnp_restore(pnp_handle e, const void* src, float* x, float* z, float* u, float* t_state, void* stream) {
    PNP_API_BEGIN
    if (!e || !x || !z || !u || !src) return fail(PNP_ERR_INVALID, "pnp_restore: null argument");
    PNP_ON_DEVICE(e);
    const size_t px = (size_t)e->cfg.n * e->cfg.h * e->cfg.w;
    const char* d = static_cast<const char*>(src);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(x, d, px * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(z, d + px * 4, px * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(u, d + px * 12, px * 8, hipMemcpyDeviceToDevice, s));
    if (t_state) HIP_TRY(hipMemcpyAsync(t_state, d + px * 20, (size_t)e->cfg.n * 4, hipMemcpyDeviceToDevice, s));
    return PNP_OK;
    PNP_API_END("pnp_restore")
}

int pnp_unet_read_stage(pnp_handle e, int which, float* dst, int* c, int* hh, int* ww, void* stream) {
    PNP_API_BEGIN
    if (!e || which < 0 || which > 8) return fail(PNP_ERR_INVALID, "pnp_unet_read_stage: which must be 0..8");
    if (which == 8 && e->fuse_last) return fail(PNP_ERR_STATE, "pnp_unet_read_stage: stage 8 is fused away; create the handle with PNP_FLAG_KEEP_STAGES");
    if (which <= 3 && (e->abits[3 * which + 2] & 2))
        return fail(PNP_ERR_STATE, "pnp_unet_read_stage: this stage is held as bf16 on this handle; create it with PNP_FLAG_KEEP_STAGES");
    PNP_ON_DEVICE(e);
    // stage outputs: inc, down1..4 live in lv[k].s; up1..4 in lv[3..0].p
    const int lvl = which <= 4 ? which : 8 - which;
    const LevelBufs& L = e->lv[lvl];
    const float* src = which <= 4 ? L.s : L.p;
    if (c) *c = L.c;
    if (hh) *hh = L.h;
    if (ww) *ww = L.w;
    if (dst) HIP_TRY(launch_nhwc_to_nchw(src, dst, e->cfg.n, L.c, L.h, L.w, (hipStream_t)stream));
    return PNP_OK;
    PNP_API_END("pnp_unet_read_stage")
}

int pnp_conv_algorithms(pnp_handle e, int32_t* algo28) {
    PNP_API_BEGIN
    if (!e || !algo28) return fail(PNP_ERR_INVALID, "pnp_conv_algorithms: null argument");
    if (e->cfg.flags & PNP_FLAG_NO_DENOISER) return fail(PNP_ERR_STATE, "pnp_conv_algorithms: handle has no denoiser");
    for (int i = 0; i < N_LAYERS; ++i) algo28[i] = i == 0 ? 2 : (i == N_LAYERS - 1 ? 3 : (e->wino[i] ? e->wplan[i].algo : (e->cplan[i].ws ? 5 : 0)));
    return PNP_OK;
    PNP_API_END("pnp_conv_algorithms")
}

int pnp_profile_reset(pnp_handle e) {
    PNP_API_BEGIN
    if (!e) return fail(PNP_ERR_INVALID, "null handle");
    e->ev_used = 0;
    memset(e->cls_ms, 0, sizeof e->cls_ms); memset(e->cls_n, 0, sizeof e->cls_n);
    memset(e->layer_ms, 0, sizeof e->layer_ms); memset(e->layer_n, 0, sizeof e->layer_n);
    return PNP_OK;
    PNP_API_END("pnp_profile_reset")
}

int pnp_profile_collect(pnp_handle e, double* total_ms, int64_t* launches) {
    PNP_API_BEGIN
    if (!e) return fail(PNP_ERR_INVALID, "null handle");
    PNP_ON_DEVICE(e);
    for (size_t i = 0; i < e->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->events[i].a, e->events[i].b));
        const EventPair& p = e->events[i];
        e->cls_ms[p.cls] += ms; e->cls_n[p.cls] += p.count;
        if (p.layer >= 0) { e->layer_ms[p.layer] += ms; e->layer_n[p.layer] += 1; }
    }
    e->ev_used = 0;
    for (int i = 0; i < PNP_PROFILE_CLASSES; ++i) {
        if (total_ms) total_ms[i] = e->cls_ms[i];
        if (launches) launches[i] = e->cls_n[i];
    }
    return PNP_OK;
    PNP_API_END("pnp_profile_collect")
}

int pnp_profile_layers(pnp_handle e, double* layer_ms, int64_t* layer_launches) {
    PNP_API_BEGIN
    if (!e) return fail(PNP_ERR_INVALID, "null handle");
    for (int i = 0; i < N_LAYERS; ++i) {
        if (layer_ms) layer_ms[i] = e->layer_ms[i];
        if (layer_launches) layer_launches[i] = e->layer_n[i];
    }
    return PNP_OK;
    PNP_API_END("pnp_profile_layers")
}

}  // extern "C"
