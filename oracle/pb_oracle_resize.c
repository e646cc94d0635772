/*
 * pb_oracle_resize.c -- CPU ORACLE (test infrastructure; see pb_oracle.h) for the pre-processing step
 * `img.resize_to_fill(W, H, FilterType::Triangle).to_rgb8()` of efficientnet.rs:20.
 *
 * PARITY UNPINNED.  The arithmetic lives in the `image` crate (Cargo.toml:22, image = "0.25.9"), a third-party
 * dependency that is not under /root/reference, and nothing in the reference pins a resized pixel (its only
 * embedding test checks determinism, efficientnet.rs:54-67).  What follows restates the crate's published
 * algorithm for an RGB8 source:
 *   DynamicImage::resize_to_fill   (src/dynimage.rs)      : scale-to-cover dimensions, resize_exact, centre crop
 *   resize_dimensions(.., fill)    (src/math/utils.rs)    : f64 ratios, round(), max(.., 1)
 *   imageops::resize               (src/imageops/sample.rs): same size -> copy; else vertical_sample into an
 *                                                            f32 image, then horizontal_sample back to u8
 *   triangle_kernel                                        : 1 - |x| for |x| < 1, support 1.0
 * All sample arithmetic is f32 with separate multiply and add (rustc does not contract), weights normalised by
 * their f32 sum, the final value clamped to [0, 255] and rounded half away from zero (FloatNearest -> f32::round).
 */
#include "pb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/math/utils.rs resize_dimensions with fill = true (the > u32::MAX branches cannot occur for a 128..224 target) */
void pbo_resize_dimensions_fill(uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint32_t *ow, uint32_t *oh) {
    const double wratio = (double)nw / (double)w;
    const double hratio = (double)nh / (double)h;
    const double ratio = wratio > hratio ? wratio : hratio; /* f64::max */
    double a = round((double)w * ratio), b = round((double)h * ratio);
    uint64_t x = a < 1.0 ? 1 : (uint64_t)a, y = b < 1.0 ? 1 : (uint64_t)b;
    *ow = (uint32_t)x;
    *oh = (uint32_t)y;
}

static float triangle_kernel(float x) {
    const float a = fabsf(x);
    return a < 1.0f ? 1.0f - a : 0.0f;
}

/* the tap window and f32 weights of output index `o` when `in_size` samples become `out_size` (identical code in
 * vertical_sample and horizontal_sample).  ws must hold in_size floats.  Returns left; *count = right - left. */
static uint32_t sample_weights(uint32_t o, uint32_t in_size, uint32_t out_size, float *ws, uint32_t *count) {
    const float ratio = (float)in_size / (float)out_size;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 1.0f * sratio;
    float input = ((float)o + 0.5f) * ratio;
    int64_t left = (int64_t)floorf(input - src_support);
    if (left < 0) left = 0;
    if (left > (int64_t)in_size - 1) left = (int64_t)in_size - 1;
    int64_t right = (int64_t)ceilf(input + src_support);
    if (right < left + 1) right = left + 1;
    if (right > (int64_t)in_size) right = (int64_t)in_size;
    input = input - 0.5f;
    float sum = 0.0f;
    uint32_t n = 0;
    for (int64_t i = left; i < right; ++i) {
        const float w = triangle_kernel(((float)i - input) / sratio);
        ws[n++] = w;
        sum = sum + w;
    }
    for (uint32_t i = 0; i < n; ++i) ws[i] = ws[i] / sum;
    *count = n;
    return (uint32_t)left;
}

static uint8_t to_u8_nearest(float t) {
    if (t < 0.0f) t = 0.0f; /* clamp(t, 0, 255); NaN cannot occur */
    if (t > 255.0f) t = 255.0f;
    return (uint8_t)roundf(t); /* f32::round: half away from zero */
}

/* imageops::resize(img, nw, nh, Triangle) for RGB8: out[nh][nw][3] */
static void resize_exact_rgb8(const uint8_t *src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint8_t *out) {
    if (nw == w && nh == h) {
        memcpy(out, src, (size_t)w * h * 3);
        return;
    }
    float *tmp = (float *)malloc((size_t)w * nh * 3 * sizeof(float));
    float *ws = (float *)malloc(((size_t)(w > h ? w : h) + 1) * sizeof(float));
    for (uint32_t oy = 0; oy < nh; ++oy) { /* vertical_sample */
        uint32_t cnt;
        const uint32_t left = sample_weights(oy, h, nh, ws, &cnt);
        for (uint32_t x = 0; x < w; ++x) {
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint8_t *p = src + ((size_t)(left + i) * w + x) * 3;
                const float wgt = ws[i];
                float m;
                m = (float)p[0] * wgt; t0 = t0 + m;
                m = (float)p[1] * wgt; t1 = t1 + m;
                m = (float)p[2] * wgt; t2 = t2 + m;
            }
            float *o = tmp + ((size_t)oy * w + x) * 3;
            o[0] = t0; o[1] = t1; o[2] = t2;
        }
    }
    for (uint32_t ox = 0; ox < nw; ++ox) { /* horizontal_sample */
        uint32_t cnt;
        const uint32_t left = sample_weights(ox, w, nw, ws, &cnt);
        for (uint32_t y = 0; y < nh; ++y) {
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
            for (uint32_t i = 0; i < cnt; ++i) {
                const float *p = tmp + ((size_t)y * w + left + i) * 3;
                const float wgt = ws[i];
                float m;
                m = p[0] * wgt; t0 = t0 + m;
                m = p[1] * wgt; t1 = t1 + m;
                m = p[2] * wgt; t2 = t2 + m;
            }
            uint8_t *o = out + ((size_t)y * nw + ox) * 3;
            o[0] = to_u8_nearest(t0); o[1] = to_u8_nearest(t1); o[2] = to_u8_nearest(t2);
        }
    }
    free(ws);
    free(tmp);
}

/* DynamicImage::resize_to_fill(nw, nh, Triangle) on RGB8: out[nh][nw][3].  Returns 0, or -1 for empty input. */
int pbo_resize_to_fill_rgb8(const uint8_t *src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint8_t *out) {
    if (!w || !h || !nw || !nh) return -1;
    uint32_t w2, h2;
    pbo_resize_dimensions_fill(w, h, nw, nh, &w2, &h2);
    uint8_t *mid = (uint8_t *)malloc((size_t)w2 * h2 * 3);
    resize_exact_rgb8(src, w, h, w2, h2, mid);
    /* centre crop (dynimage.rs): compare iwidth*nheight with nwidth*iheight */
    const uint64_t ratio = (uint64_t)w2 * nh, nratio = (uint64_t)nw * h2;
    uint32_t cx = 0, cy = 0;
    if (nratio > ratio) cy = (h2 - nh) / 2;
    else cx = (w2 - nw) / 2;
    /* crop clamps the window to the image (GenericImageView::view semantics of crop_imm) */
    for (uint32_t y = 0; y < nh; ++y)
        for (uint32_t x = 0; x < nw; ++x) {
            const uint32_t sx = cx + x < w2 ? cx + x : w2 - 1, sy = cy + y < h2 ? cy + y : h2 - 1;
            memcpy(out + ((size_t)y * nw + x) * 3, mid + ((size_t)sy * w2 + sx) * 3, 3);
        }
    free(mid);
    return 0;
}
