/*
 * pb_oracle_phash.c -- CPU ORACLE (test infrastructure; see pb_oracle.h) for `phash` (src/image_hashes/phash.rs:3-22):
 *
 *     let small = img.resize(16, 16, image::imageops::Gaussian);
 *     let grey  = imageops::grayscale(&small).to_vec();
 *     let mean  = (sum(grey) / (16 * 16)) as u8;                    // divides by 256 whatever the size of `small`
 *     byte b    = OR over i in 0..8 of (grey[8 b + i] > mean) << i   // grey.len() / 8 bytes, LSB first
 *
 * PARITY: pinned by the one known answer the reference holds that does not need its (absent) test images --
 * phash.rs:36-41, a flat white image hashes to 32 zero bytes (any flat image does: no pixel exceeds the mean) --
 * and otherwise UNPINNED: the arithmetic of `resize` and `grayscale` lives in the `image` crate (Cargo.toml:22,
 * image = "0.25.9"), which is not under /root/reference, and phash.rs:43-78 compare hashes of files that are absent
 * (test_resources/).  What follows restates the crate's published algorithm for an RGB8 source:
 *   DynamicImage::resize (src/dynimage.rs)            : KEEPS the aspect ratio -- resize_dimensions(w, h, 16, 16, fill = false),
 *                                                       so `small` is 16 x n or n x 16 and the hash has (w2 * h2) / 8 bytes
 *   imageops::resize (src/imageops/sample.rs)         : same size -> copy; else vertical_sample into an f32 image, then
 *                                                       horizontal_sample back to u8 (the code pb_oracle_resize.c restates
 *                                                       for Triangle), with the Gaussian filter: support 3.0,
 *                                                       kernel gaussian(x, 0.5) = 1 / (sqrt(2 pi) 0.5) * exp(-x^2 / (2 * 0.25))
 *   imageops::grayscale -> Pixel::to_luma (src/color.rs): luma = (2126 R + 7152 G + 722 B) / 10000 in u32, truncating
 * All sample arithmetic is f32 with separate multiply and add; expf is libm's (the one thing here that is not
 * arithmetic: the product computes the same weights on the HOST with the same libm and only multiplies on the GPU).
 */
#include "pb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/math/utils.rs resize_dimensions with fill = false */
void pbo_resize_dimensions_fit(uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint32_t *ow, uint32_t *oh) {
    const double wratio = (double)nw / (double)w;
    const double hratio = (double)nh / (double)h;
    const double ratio = wratio < hratio ? wratio : hratio; /* f64::min */
    double a = round((double)w * ratio), b = round((double)h * ratio);
    uint64_t x = a < 1.0 ? 1 : (uint64_t)a, y = b < 1.0 ? 1 : (uint64_t)b;
    *ow = (uint32_t)x;
    *oh = (uint32_t)y;
}

/* sample.rs: gaussian(x, r) = ((2 pi).sqrt() * r).recip() * (-x.powi(2) / (2.0 * r.powi(2))).exp(), r = 0.5 */
float pbo_gaussian_kernel(float x) {
    const float pi = 3.14159274101257324f; /* f32::consts::PI */
    const float r = 0.5f;
    const float norm = 1.0f / (sqrtf(2.0f * pi) * r);
    const float x2 = x * x;
    const float den = 2.0f * (r * r);
    const float arg = -x2 / den;
    return norm * expf(arg);
}

/* tap window + normalised f32 weights of output index o when in_size samples become out_size (sample.rs, the code shared
 * by vertical_sample and horizontal_sample), Gaussian filter.  ws must hold in_size floats.  Returns left; *count taps. */
uint32_t pbo_gaussian_weights(uint32_t o, uint32_t in_size, uint32_t out_size, float *ws, uint32_t *count) {
    const float ratio = (float)in_size / (float)out_size;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 3.0f * sratio;
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
        const float w = pbo_gaussian_kernel(((float)i - input) / sratio);
        ws[n++] = w;
        sum = sum + w;
    }
    for (uint32_t i = 0; i < n; ++i) ws[i] = ws[i] / sum;
    *count = n;
    return (uint32_t)left;
}

static uint8_t to_u8_nearest(float t) {
    if (t < 0.0f) t = 0.0f;
    if (t > 255.0f) t = 255.0f;
    return (uint8_t)roundf(t);
}

/* imageops::resize(img, nw, nh, Gaussian) for RGB8: out[nh][nw][3] */
static void resize_gaussian_rgb8(const uint8_t *src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint8_t *out) {
    if (nw == w && nh == h) {
        memcpy(out, src, (size_t)w * h * 3);
        return;
    }
    float *tmp = (float *)malloc((size_t)w * nh * 3 * sizeof(float));
    float *ws = (float *)malloc(((size_t)(w > h ? w : h) + 1) * sizeof(float));
    for (uint32_t oy = 0; oy < nh; ++oy) { /* vertical_sample */
        uint32_t cnt;
        const uint32_t left = pbo_gaussian_weights(oy, h, nh, ws, &cnt);
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
        const uint32_t left = pbo_gaussian_weights(ox, w, nw, ws, &cnt);
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

/* phash.rs:3-22 for an RGB8 image [h][w][3].  out must hold 32 bytes; *n_bytes receives (w2 * h2) / 8 <= 32.
 * small_rgb (optional, 16*16*3 bytes) receives the resized image, *sw / *sh its size.  Returns 0, -1 for empty input. */
int pbo_phash_rgb8(const uint8_t *src, uint32_t w, uint32_t h, uint8_t *out, uint32_t *n_bytes, uint8_t *small_rgb, uint32_t *sw,
                   uint32_t *sh) {
    if (!w || !h) return -1;
    uint32_t w2, h2;
    pbo_resize_dimensions_fit(w, h, 16, 16, &w2, &h2); /* DynamicImage::resize keeps the aspect ratio */
    uint8_t small[16 * 16 * 3];
    resize_gaussian_rgb8(src, w, h, w2, h2, small);
    uint8_t grey[256];
    const uint32_t n = w2 * h2;
    uint64_t sum = 0;
    for (uint32_t i = 0; i < n; ++i) { /* color.rs rgb_to_luma: u32 arithmetic, truncating divide */
        const uint32_t l = 2126u * small[3 * i] + 7152u * small[3 * i + 1] + 722u * small[3 * i + 2];
        grey[i] = (uint8_t)(l / 10000u);
        sum += grey[i];
    }
    const uint8_t mean = (uint8_t)(sum / 256u); /* phash.rs:10: / (img_width * img_height) with the CONSTANTS 16, 16 */
    const uint32_t nb = n / 8;
    for (uint32_t b = 0; b < nb; ++b) {
        uint8_t acc = 0;
        for (int i = 0; i < 8; ++i)
            if (grey[8 * b + i] > mean) acc |= (uint8_t)(1u << i);
        out[b] = acc;
    }
    *n_bytes = nb;
    if (small_rgb) memcpy(small_rgb, small, (size_t)n * 3);
    if (sw) *sw = w2;
    if (sh) *sh = h2;
    return 0;
}
