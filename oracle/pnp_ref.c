/*
 * pnp_ref.c - plain-C restatement of the operators on the PnP-ADMM hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference (/root/reference, pure Python) reaches conv2d / max_pool2d / upsample_bilinear2d / fftn through
 * PyTorch ATen, a third-party dependency that is not part of the reference's tree (no version pinned; fixtures were
 * made with torch 2.10.0).  oracle/pnp_oracle.py calls the same ATen entry points; THIS file restates their published
 * semantics with no ATen at all - direct loops, double accumulation - so the oracle can be cross-checked
 * independently (tests/test_oracle_c.py).  Only tests/ load it; the product never does.
 *
 * Layouts are the reference's: NCHW float32, weights OIHW, complex as interleaved (re, im).
 * Each function cites the reference call site whose operator it restates.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LEAKY 0.2

/* nn.Conv2d(cin, cout, k, stride 1, padding k/2, bias) [+ LeakyReLU(0.2)]  - evaluation/noise.py:75-85, :88-98 */
static void conv2d(const float* x, const float* w, const float* b, float* y, int n, int cin, int cout, int h, int wd,
                   int k, int act) {
    const int pad = k / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int in = 0; in < n; ++in)
        for (int co = 0; co < cout; ++co)
            for (int oy = 0; oy < h; ++oy)
                for (int ox = 0; ox < wd; ++ox) {
                    double acc = b[co];
                    for (int ci = 0; ci < cin; ++ci)
                        for (int ky = 0; ky < k; ++ky) {
                            const int iy = oy + ky - pad;
                            if (iy < 0 || iy >= h) continue;      /* zero padding */
                            for (int kx = 0; kx < k; ++kx) {
                                const int ix = ox + kx - pad;
                                if (ix < 0 || ix >= wd) continue;
                                acc += (double)x[(((size_t)in * cin + ci) * h + iy) * wd + ix] *
                                       (double)w[(((size_t)co * cin + ci) * k + ky) * k + kx];
                            }
                        }
                    if (act && acc < 0) acc *= LEAKY;
                    y[(((size_t)in * cout + co) * h + oy) * wd + ox] = (float)acc;
                }
}

/* nn.MaxPool2d(2) - evaluation/noise.py:22-25 */
static void maxpool2(const float* x, float* y, int nc, int h, int w) {
    const int ho = h / 2, wo = w / 2;
    for (int c = 0; c < nc; ++c)
        for (int oy = 0; oy < ho; ++oy)
            for (int ox = 0; ox < wo; ++ox) {
                const float* p = x + ((size_t)c * h + 2 * oy) * w + 2 * ox;
                float m = p[0];
                if (p[1] > m) m = p[1];
                if (p[w] > m) m = p[w];
                if (p[w + 1] > m) m = p[w + 1];
                y[((size_t)c * ho + oy) * wo + ox] = m;
            }
}

/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) - evaluation/noise.py:39,46
 * src = dst * (in-1)/(out-1); neighbours i0, i0+1 (clamped); weights (1-l, l). */
static void upsample2(const float* x, float* y, int nc, int h, int w) {
    const int ho = 2 * h, wo = 2 * w;
    const double rh = ho > 1 ? (double)(h - 1) / (ho - 1) : 0.0, rw = wo > 1 ? (double)(w - 1) / (wo - 1) : 0.0;
    for (int c = 0; c < nc; ++c)
        for (int oy = 0; oy < ho; ++oy) {
            const double sy = rh * oy;
            const int y0 = (int)sy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const double ly = sy - y0;
            for (int ox = 0; ox < wo; ++ox) {
                const double sx = rw * ox;
                const int x0 = (int)sx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const double lx = sx - x0;
                const float* p = x + (size_t)c * h * w;
                const double v = (1 - ly) * ((1 - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1]) +
                                 ly * ((1 - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1]);
                y[((size_t)c * ho + oy) * wo + ox] = (float)v;
            }
        }
}

static const float* take(const float** blob, size_t count) {
    const float* p = *blob;
    *blob += count;
    return p;
}

/* ConvBlock: 3 x (conv3x3 + LeakyReLU)  - evaluation/noise.py:88-98; weights consumed from the state_dict-ordered blob */
static float* conv_block(const float** blob, const float* x, int n, int cin, int cout, int h, int w) {
    float* a = (float*)malloc(sizeof(float) * (size_t)n * cout * h * w);
    float* b = (float*)malloc(sizeof(float) * (size_t)n * cout * h * w);
    const float* wt = take(blob, (size_t)cout * cin * 9);
    const float* bs = take(blob, cout);
    conv2d(x, wt, bs, a, n, cin, cout, h, w, 3, 1);
    wt = take(blob, (size_t)cout * cout * 9); bs = take(blob, cout);
    conv2d(a, wt, bs, b, n, cout, cout, h, w, 3, 1);
    wt = take(blob, (size_t)cout * cout * 9); bs = take(blob, cout);
    conv2d(b, wt, bs, a, n, cout, cout, h, w, 3, 1);
    free(b);
    return a;
}

static float* down(const float** blob, const float* x, int n, int cin, int cout, int h, int w) {
    float* p = (float*)malloc(sizeof(float) * (size_t)n * cin * (h / 2) * (w / 2));
    maxpool2(x, p, n * cin, h, w);
    float* y = conv_block(blob, p, n, cin, cout, h / 2, w / 2);
    free(p);
    return y;
}

/* up.forward: upsample x1, cat([x2 (skip), x1], dim=1), ConvBlock  - evaluation/noise.py:44-61 */
static float* up(const float** blob, const float* x1, int c1, const float* x2, int c2, int n, int cout, int h, int w) {
    /* x1: [n,c1,h/2,w/2]; x2: [n,c2,h,w] */
    float* u = (float*)malloc(sizeof(float) * (size_t)n * c1 * h * w);
    upsample2(x1, u, n * c1, h / 2, w / 2);
    float* cat = (float*)malloc(sizeof(float) * (size_t)n * (c1 + c2) * h * w);
    const size_t hw = (size_t)h * w;
    for (int in = 0; in < n; ++in) {
        memcpy(cat + (size_t)in * (c1 + c2) * hw, x2 + (size_t)in * c2 * hw, sizeof(float) * c2 * hw);
        memcpy(cat + ((size_t)in * (c1 + c2) + c2) * hw, u + (size_t)in * c1 * hw, sizeof(float) * c1 * hw);
    }
    free(u);
    float* y = conv_block(blob, cat, n, c1 + c2, cout, h, w);
    free(cat);
    return y;
}

/* UNetDenoiser2D.forward: cat[x, sigma plane] -> UNet(2,1) -> + x -> clamp(0,1)
 * - evaluation/noise.py:155-164 and :119-133.  blob = 56 tensors in state_dict order. */
int ref_denoise(const float* blob, const float* x, const float* sigma, float* out, int n, int h, int w, int clamp01) {
    const size_t hw = (size_t)h * w;
    float* in = (float*)malloc(sizeof(float) * (size_t)n * 2 * hw);
    for (int i = 0; i < n; ++i) {
        memcpy(in + (size_t)i * 2 * hw, x + (size_t)i * hw, sizeof(float) * hw);
        for (size_t p = 0; p < hw; ++p) in[((size_t)i * 2 + 1) * hw + p] = sigma[i];
    }
    const float* bp = blob;
    float* x1 = conv_block(&bp, in, n, 2, 32, h, w);
    float* x2 = down(&bp, x1, n, 32, 64, h, w);
    float* x3 = down(&bp, x2, n, 64, 128, h / 2, w / 2);
    float* x4 = down(&bp, x3, n, 128, 256, h / 4, w / 4);
    float* x5 = down(&bp, x4, n, 256, 512, h / 8, w / 8);
    float* y1 = up(&bp, x5, 512, x4, 256, n, 256, h / 8, w / 8);
    float* y2 = up(&bp, y1, 256, x3, 128, n, 128, h / 4, w / 4);
    float* y3 = up(&bp, y2, 128, x2, 64, n, 64, h / 2, w / 2);
    float* y4 = up(&bp, y3, 64, x1, 32, n, 32, h, w);
    const float* wt = take(&bp, 32);
    const float* bs = take(&bp, 1);
    conv2d(y4, wt, bs, out, n, 32, 1, h, w, 1, 0);
    for (size_t i = 0; i < (size_t)n * hw; ++i) {
        float v = x[i] + out[i];
        if (clamp01) v = v < 0 ? 0 : (v > 1 ? 1 : v);
        out[i] = v;
    }
    free(in); free(x1); free(x2); free(x3); free(x4); free(x5); free(y1); free(y2); free(y3); free(y4);
    return 0;
}

/* fft / ifft: fftshift(fftn(ifftshift(img), norm='ortho'))  - evaluation/utils/transformations.py:6-19.
 * Separable naive DFT in double (O(HW(H+W))): slow and obviously right. */
static void dft_centred(const float* in, float* out, int h, int w, int inverse) {
    const double sgn = inverse ? 2.0 * M_PI : -2.0 * M_PI;
    double* t = (double*)malloc(sizeof(double) * 2 * (size_t)h * w);
    double* s = (double*)malloc(sizeof(double) * 2 * (size_t)h * w);
    /* ifftshift: out[i] = in[(i + (n - n/2)) % n]... for the sizes used (even) shift by n/2 */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int sy = (y + h / 2) % h, sx = (x + w / 2) % w;   /* ifftshift for even n */
            s[2 * ((size_t)y * w + x)] = in[2 * ((size_t)sy * w + sx)];
            s[2 * ((size_t)y * w + x) + 1] = in[2 * ((size_t)sy * w + sx) + 1];
        }
    for (int y = 0; y < h; ++y)                       /* rows */
        for (int k = 0; k < w; ++k) {
            double re = 0, im = 0;
            for (int x = 0; x < w; ++x) {
                const double a = sgn * (double)((long)k * x % w) / w;
                const double c = cos(a), sn = sin(a);
                re += s[2 * ((size_t)y * w + x)] * c - s[2 * ((size_t)y * w + x) + 1] * sn;
                im += s[2 * ((size_t)y * w + x)] * sn + s[2 * ((size_t)y * w + x) + 1] * c;
            }
            t[2 * ((size_t)y * w + k)] = re; t[2 * ((size_t)y * w + k) + 1] = im;
        }
    const double sc = 1.0 / sqrt((double)h * w);
    for (int x = 0; x < w; ++x)                       /* columns */
        for (int k = 0; k < h; ++k) {
            double re = 0, im = 0;
            for (int y = 0; y < h; ++y) {
                const double a = sgn * (double)((long)k * y % h) / h;
                const double c = cos(a), sn = sin(a);
                re += t[2 * ((size_t)y * w + x)] * c - t[2 * ((size_t)y * w + x) + 1] * sn;
                im += t[2 * ((size_t)y * w + x)] * sn + t[2 * ((size_t)y * w + x) + 1] * c;
            }
            const int dy = (k + h / 2) % h, dx = (x + w / 2) % w;   /* fftshift */
            out[2 * ((size_t)dy * w + dx)] = (float)(re * sc);
            out[2 * ((size_t)dy * w + dx) + 1] = (float)(im * sc);
        }
    free(t); free(s);
}

int ref_fft2c(const float* in, float* out, int batch, int h, int w, int inverse) {
    if (h % 2 || w % 2) return -1;
#pragma omp parallel for
    for (int b = 0; b < batch; ++b) dft_centred(in + 2 * (size_t)b * h * w, out + 2 * (size_t)b * h * w, h, w, inverse);
    return 0;
}

/* One PnPEnv.step for n independent slices  - evaluation/env.py:74-100.
 * x [n,h,w] out; z, u complex [n,h,w] in/out; y0 complex; mask u8 [h,w]; mu, sigma [n]. */
int ref_admm_step(const float* blob, float* x, float* z, float* u, const float* y0, const unsigned char* mask,
                  const float* mu, const float* sigma, int n, int h, int w) {
    const size_t hw = (size_t)h * w;
    float* d = (float*)malloc(sizeof(float) * n * hw);
    float* v = (float*)malloc(sizeof(float) * 2 * n * hw);
    float* f = (float*)malloc(sizeof(float) * 2 * n * hw);
    for (size_t i = 0; i < n * hw; ++i) d[i] = z[2 * i] - u[2 * i];                      /* env.py:85-86 */
    ref_denoise(blob, d, sigma, x, n, h, w, 1);
    for (size_t i = 0; i < n * hw; ++i) { v[2 * i] = x[i] + u[2 * i]; v[2 * i + 1] = u[2 * i + 1]; }
    ref_fft2c(v, f, n, h, w, 0);                                                        /* env.py:87 */
    for (int s = 0; s < n; ++s)
        for (size_t p = 0; p < hw; ++p)
            if (mask[p]) {                                                              /* env.py:88-90 */
                const size_t i = s * hw + p;
                f[2 * i] = (mu[s] * f[2 * i] + y0[2 * i]) / (1 + mu[s]);
                f[2 * i + 1] = (mu[s] * f[2 * i + 1] + y0[2 * i + 1]) / (1 + mu[s]);
            }
    ref_fft2c(f, z, n, h, w, 1);                                                        /* env.py:91 */
    for (size_t i = 0; i < n * hw; ++i) {                                               /* env.py:93 */
        u[2 * i] = u[2 * i] + x[i] - z[2 * i];
        u[2 * i + 1] = u[2 * i + 1] - z[2 * i + 1];
    }
    free(d); free(v); free(f);
    return 0;
}

/* torch_psnr - evaluation/env.py:120-125 */
int ref_psnr(const float* x, const float* gt, float* out, int n, int hw) {
    for (int s = 0; s < n; ++s) {
        double acc = 0;
        for (int p = 0; p < hw; ++p) {
            float v = x[(size_t)s * hw + p];
            v = v < 0 ? 0 : (v > 1 ? 1 : v);
            const double e = (double)v - gt[(size_t)s * hw + p];
            acc += e * e;
        }
        out[s] = (float)(10.0 * log10(1.0 / (acc / hw)));
    }
    return 0;
}
