// Host-side sanitizer check of libpnpadmm (tests/test_host_logic.py::test_host_side_under_asan_ubsan builds and runs it;
// `make -C dt4image_restoration_amd/csrc asan` builds the instrumented library, host code only).
// AddressSanitizer + UBSan see: every tile plan, every weight repack into buffers of exactly the size the library asks for
// (heap redzones catch an overrun by one float), and the argument validation of every C-ABI entry point.
// No GPU is needed: nothing here launches a kernel.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/pnpadmm.h"
#include "../dt4image_restoration_amd/csrc/pnp_internal.h"

using namespace pnp;

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)

int main() {
    // ---- plans + packs over the sizes the tests and the bench use -----------------------------------------------------
    const int shapes[][3] = {{1, 128, 128}, {64, 256, 256}, {4, 256, 256}, {2, 48, 64}, {16, 512, 512}, {1, 16, 16}, {3, 64, 16}};
    Tuning forced = tuning_from_env();
    forced.wino_min_blocks = 1;
    Tuning variants[2] = {tuning_from_env(), forced};
    std::vector<float> w;
    size_t packed_layers = 0;
    for (const auto& t : variants)
        for (const auto& sh : shapes)
            for (int li = 1; li < N_LAYERS - 1; ++li) {
                const LayerSpec& L = kLayers[li];
                const int n = sh[0], h = sh[1] >> L.level, wd = sh[2] >> L.level;
                const WinoPlan wp = winograd_plan(n, h, wd, L.cin, L.cout, L.src == SRC_POOL ? (int)SRC_PLAIN : L.src, t);
                const ConvPlan cp = conv3x3_plan(n, h, wd, L.cin, L.cout, false);
                const ConvPlan cb = conv3x3_plan(n, h, wd, L.cin, L.cout, true);
                CHECK(cp.tiles_x * cp.tw >= wd && cp.tiles_y * cp.th >= h && cp.splitk >= 1);
                CHECK(L.cin % cp.ck == 0 && L.cin % cb.ck == 0);
                // bf16 mode, the producer / consumer plan: only with bf16 operands, always 32-channel chunks and whole tiles, one of its
                // five tile shapes, never split-K; the f32 plan and the ablation switch never carry it
                CHECK(cp.ws == 0 && conv3x3_plan(n, h, wd, L.cin, L.cout, true, L.src, false).ws == 0);
                for (int src : {(int)SRC_PLAIN, (int)L.src}) {
                    const ConvPlan cw = conv3x3_plan(n, h, wd, L.cin, L.cout, true, src);
                    if (!cw.ws) continue;
                    CHECK(cw.ck == 32 && cw.splitk == 1 && cw.tw >= 16 && cw.bm == cw.th * cw.tw && L.cout % cw.bn == 0);
                    CHECK((cw.mt == 4 && cw.nt == 2 && cw.wm * cw.wn == 4) || (cw.mt == 2 && cw.nt == 1 && cw.wm == 4 && cw.tw == 32 && src == SRC_PLAIN));
                    CHECK((long)cw.tiles_x * cw.tiles_y * n * (L.cout / cw.bn) >= 192);
                    CHECK(src != SRC_POOL || cw.wn == 2);
                }
                (void)conv3x3_partial_floats(cp, n, h, wd, L.cout);
                (void)conv3x3_pooled_output_ok(cp);
                if (&sh != &shapes[0] && &sh != &shapes[3]) continue;     // repack (slow under ASan) for two shapes only
                w.assign((size_t)L.cout * L.cin * 9, 0.f);
                for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
                if (wp.use) {
                    CHECK(wp.tiles_x * wp.tw >= wd && wp.tiles_y * wp.th >= h && L.cin % wp.ck == 0);
                    const size_t pf = wp.algo == 4 ? winograd4_pack_floats(L.cin, L.cout) : winograd_pack_floats(L.cin, L.cout);
                    float* dst = (float*)std::malloc(pf * sizeof(float));          // exact size: redzones right behind it
                    if (wp.algo == 4) pack_winograd4_weights(w.data(), L.cin, L.cout, wp.ck, dst);
                    else pack_winograd_weights(w.data(), L.cin, L.cout, wp.ck, dst);
                    std::free(dst);
                    ++packed_layers;
                }
                const size_t pf = conv3x3_pack_floats(L.cin, L.cout);
                float* dst = (float*)std::malloc(pf * sizeof(float));
                pack_conv3x3_weights(w.data(), L.cin, L.cout, cp.ck, dst);
                std::free(dst);
                for (int terms = 1; terms <= 2; ++terms) {                          // bf16 mode: one- and two-term weight packs
                    float* d16 = (float*)std::malloc(conv3x3_pack_floats_bf16(L.cin, L.cout, terms) * sizeof(float));
                    pack_conv3x3_weights_bf16(w.data(), L.cin, L.cout, cb.ck, terms, d16);
                    std::free(d16);
                }
                ++packed_layers;
            }
    CHECK(packed_layers > 100);

    // ---- C ABI argument validation (every entry point, no GPU behind it) ------------------------------------------------
    pnp_handle h = nullptr;
    pnp_config bad = {0, 128, 128, 0, 0};
    CHECK(pnp_create(&bad, &h) == PNP_ERR_INVALID && h == nullptr && std::strlen(pnp_last_error()) > 0);
    bad = {1, 100, 128, 0, 0};
    CHECK(pnp_create(&bad, &h) == PNP_ERR_INVALID);
    bad = {1, 2048, 128, 0, 0};
    CHECK(pnp_create(&bad, &h) == PNP_ERR_INVALID);
    CHECK(pnp_create(nullptr, &h) == PNP_ERR_INVALID && pnp_create(&bad, nullptr) == PNP_ERR_INVALID);
    pnp_config ok = {1, 128, 128, 0, 0};
    const int rc = pnp_create(&ok, &h);                 // no GPU here: must fail cleanly, with a message, leaking nothing
    if (rc != PNP_OK) CHECK(h == nullptr && std::strlen(pnp_last_error()) > 0);
    else CHECK(pnp_destroy(h) == PNP_OK);
    float f = 0.f; uint8_t b = 0; int c;
    CHECK(pnp_destroy(nullptr) == PNP_OK);
    CHECK(pnp_load_unet_weights(nullptr, &f, 1) == PNP_ERR_INVALID);
    CHECK(pnp_reset(nullptr, &f, &f, &b, 1, &f, &f, &f, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_set_kspace(nullptr, &f, &b, 1, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_step(nullptr, &f, &f, &f, &f, &f, &f, &f, &b, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_denoise(nullptr, &f, &f, &f, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_fft2c(nullptr, &f, &f, 1, 128, 128, 0, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_prox_dual(nullptr, &f, &f, &f, &f, &f, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_psnr(nullptr, &f, &f, &f, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_snapshot(nullptr, &f, &f, &f, &f, &f, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_restore(nullptr, &f, &f, &f, &f, &f, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_unet_read_stage(nullptr, 0, &f, &c, &c, &c, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_conv_algorithms(nullptr, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_profile_reset(nullptr) == PNP_ERR_INVALID && pnp_profile_collect(nullptr, nullptr, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_profile_layers(nullptr, nullptr, nullptr) == PNP_ERR_INVALID);
    CHECK(pnp_snapshot_bytes(nullptr) == 0 && pnp_workspace_bytes(nullptr) == 0);
    CHECK(std::strstr(pnp_version(), "gfx950") != nullptr);
    std::printf("asan_host: %zu layer repacks, %d failures\n", packed_layers, fails);
    return fails ? 1 : 0;
}
