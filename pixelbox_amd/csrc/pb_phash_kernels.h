// pb_phash_kernels.h -- gfx950 device code for `phash` (src/image_hashes/phash.rs:3-22): Gaussian resize to fit
// 16 x 16 (aspect ratio kept), luma, mean threshold, LSB-first bytes.
//
// The resize is the image crate's two-pass resampler (imageops::resize: vertical_sample into an f32 image, then
// horizontal_sample back to u8); its filter weights -- the only non-arithmetic step (expf) -- are computed on the HOST
// by pb_phash.hip with libm, exactly as the CPU restatement does, normalised there, and only multiplied and added here, in the
// crate's order (f32, separate multiply and add: this TU is compiled with -ffp-contract=off).  So the kernels reproduce the
// restatement bit for bit (tests/test_phash.py); against the crate itself the step is unpinned (absent from the reference
// tree) except for the flat-image known answer of phash.rs:36-41.
//
// Sizes: the source is any RGB8 image; the vertical pass reads a window of ~6 h / 16 source rows per output row (Gaussian
// support 3 stretched by the down-scaling ratio), i.e. ~6x the image in total -- a few hundred MB/s of L2 / HBM reads for
// a 12 MP photo, 50-100 us; the rest is a single workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace pbp {

// vertical_sample: tmp[oy][x][c] = sum_i w_v[oy][i] * src[left_v[oy] + i][x][c]   (f32, taps in order)
// grid (ceil(w * 3 / 256), h2); wts: [h2][stride] normalised weights, left / cnt per output row
__global__ __launch_bounds__(256) void k_phash_v(const uint8_t *__restrict__ src, uint32_t w, const float *__restrict__ wts,
                                                 const uint32_t *__restrict__ left, const uint32_t *__restrict__ cnt, uint32_t stride,
                                                 float *__restrict__ tmp) {
    const uint32_t xc = blockIdx.x * blockDim.x + threadIdx.x;  // column * 3 + channel
    const uint32_t oy = blockIdx.y;
    if (xc >= w * 3) return;
    const uint32_t l = left[oy], n = cnt[oy];
    const float *ws = wts + (size_t)oy * stride;
    float t = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        const float m = (float)src[(size_t)(l + i) * w * 3 + xc] * ws[i];
        t = t + m;
    }
    tmp[(size_t)oy * w * 3 + xc] = t;
}

// horizontal_sample + grayscale + threshold, ONE workgroup of 256 threads (thread = output pixel, <= 16 x 16).
// in_f32: the vertical pass ran (tmp is f32 [h2][w][3]); else `src8` is the u8 source used as it is (same-size copy path
// of imageops::resize, w == w2 and h == h2).
__global__ __launch_bounds__(256) void k_phash_h(const float *__restrict__ tmp, const uint8_t *__restrict__ src8, int in_f32, uint32_t w,
                                                 uint32_t w2, uint32_t h2, const float *__restrict__ wts, const uint32_t *__restrict__ left,
                                                 const uint32_t *__restrict__ cnt, uint32_t stride, uint8_t *__restrict__ out_hash,
                                                 uint32_t *__restrict__ out_nbytes, uint8_t *__restrict__ out_small) {
    __shared__ uint32_t s_sum;
    __shared__ uint8_t s_grey[256];
    const uint32_t i = threadIdx.x;
    const uint32_t n = w2 * h2;
    if (i == 0) s_sum = 0;
    __syncthreads();
    uint32_t grey = 0;
    if (i < n) {
        const uint32_t y = i / w2, ox = i % w2;
        uint8_t px[3];
        if (in_f32) {
            const uint32_t l = left[ox], c = cnt[ox];
            const float *ws = wts + (size_t)ox * stride;
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
            for (uint32_t k = 0; k < c; ++k) {
                const float *p = tmp + ((size_t)y * w + l + k) * 3;
                const float wgt = ws[k];
                const float m0 = p[0] * wgt, m1 = p[1] * wgt, m2 = p[2] * wgt;
                t0 = t0 + m0; t1 = t1 + m1; t2 = t2 + m2;
            }
            auto to_u8 = [](float t) -> uint8_t {
                t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
                return (uint8_t)roundf(t);  // f32::round: half away from zero
            };
            px[0] = to_u8(t0); px[1] = to_u8(t1); px[2] = to_u8(t2);
        } else {
            const uint8_t *p = src8 + (size_t)i * 3;
            px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
        }
        if (out_small) {
            out_small[3 * i] = px[0]; out_small[3 * i + 1] = px[1]; out_small[3 * i + 2] = px[2];
        }
        grey = (2126u * px[0] + 7152u * px[1] + 722u * px[2]) / 10000u;  // color.rs rgb_to_luma (u32, truncating)
        s_grey[i] = (uint8_t)grey;
        atomicAdd(&s_sum, grey);
    }
    __syncthreads();
    const uint32_t mean = (s_sum / 256u) & 0xFFu;  // phash.rs:10: sum / (16 * 16) as u8
    const uint32_t nb = n / 8;
    if (i < nb) {
        uint32_t acc = 0;
        for (int b = 0; b < 8; ++b) acc |= (s_grey[8 * i + b] > mean) ? (1u << b) : 0u;
        out_hash[i] = (uint8_t)acc;
    }
    if (i == 0) *out_nbytes = nb;
}

// ---- a whole batch of images in two launches (pb_phash_batch_images): the same arithmetic per image, driven by a descriptor
struct PhashDesc {
    unsigned long long src_off;  // bytes into the staged source block: u8 [h][w][3]
    unsigned long long tmp_off;  // floats into the scratch block: f32 [h2][w][3]
    uint32_t w, h, w2, h2;
    uint32_t wv_off, wh_off;     // floats into the weights block: [h2][sv] and [w2][sh]
    uint32_t sv, sh;             // longest window of either pass (row pitch of its weights)
    uint32_t resample;           // 0: the source already has the fitted size (imageops::resize copies)
    uint32_t pad;
};
// per image 64 words of meta: left_v[16] cnt_v[16] left_h[16] cnt_h[16].  grid (ceil(max w * 3 / 256), 16, images)
__global__ __launch_bounds__(256) void k_phash_v_batch(const uint8_t *__restrict__ src_base, const PhashDesc *__restrict__ desc,
                                                       const float *__restrict__ wts_base, const uint32_t *__restrict__ meta_base,
                                                       float *__restrict__ tmp_base) {
    const PhashDesc d = desc[blockIdx.z];
    const uint32_t xc = blockIdx.x * blockDim.x + threadIdx.x;  // column * 3 + channel
    const uint32_t oy = blockIdx.y;
    if (!d.resample || oy >= d.h2 || xc >= d.w * 3) return;
    const uint32_t *meta = meta_base + (size_t)blockIdx.z * 64;
    const uint8_t *src = src_base + d.src_off;
    const uint32_t l = meta[oy], n = meta[16 + oy];
    const float *ws = wts_base + d.wv_off + (size_t)oy * d.sv;
    float t = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        const float m = (float)src[(size_t)(l + i) * d.w * 3 + xc] * ws[i];
        t = t + m;
    }
    tmp_base[d.tmp_off + (size_t)oy * d.w * 3 + xc] = t;
}
// one workgroup per image; out_hash [images][32] (unused bytes zero), out_nbytes [images]
__global__ __launch_bounds__(256) void k_phash_h_batch(const float *__restrict__ tmp_base, const uint8_t *__restrict__ src_base,
                                                       const PhashDesc *__restrict__ desc, const float *__restrict__ wts_base,
                                                       const uint32_t *__restrict__ meta_base, uint8_t *__restrict__ out_hash,
                                                       uint32_t *__restrict__ out_nbytes) {
    __shared__ uint32_t s_sum;
    __shared__ uint8_t s_grey[256];
    const PhashDesc d = desc[blockIdx.x];
    const uint32_t *meta = meta_base + (size_t)blockIdx.x * 64;
    const uint32_t i = threadIdx.x;
    const uint32_t n = d.w2 * d.h2;
    if (i == 0) s_sum = 0;
    __syncthreads();
    if (i < n) {
        const uint32_t y = i / d.w2, ox = i % d.w2;
        uint8_t px[3];
        if (d.resample) {
            const float *tmp = tmp_base + d.tmp_off;
            const uint32_t l = meta[32 + ox], c = meta[48 + ox];
            const float *ws = wts_base + d.wh_off + (size_t)ox * d.sh;
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
            for (uint32_t k = 0; k < c; ++k) {
                const float *pp = tmp + ((size_t)y * d.w + l + k) * 3;
                const float wgt = ws[k];
                const float m0 = pp[0] * wgt, m1 = pp[1] * wgt, m2 = pp[2] * wgt;
                t0 = t0 + m0; t1 = t1 + m1; t2 = t2 + m2;
            }
            auto to_u8 = [](float t) -> uint8_t {
                t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
                return (uint8_t)roundf(t);  // f32::round: half away from zero
            };
            px[0] = to_u8(t0); px[1] = to_u8(t1); px[2] = to_u8(t2);
        } else {
            const uint8_t *pp = src_base + d.src_off + (size_t)i * 3;
            px[0] = pp[0]; px[1] = pp[1]; px[2] = pp[2];
        }
        const uint32_t grey = (2126u * px[0] + 7152u * px[1] + 722u * px[2]) / 10000u;  // color.rs rgb_to_luma (u32, truncating)
        s_grey[i] = (uint8_t)grey;
        atomicAdd(&s_sum, grey);
    }
    __syncthreads();
    const uint32_t mean = (s_sum / 256u) & 0xFFu;  // phash.rs:10: sum / (16 * 16) as u8
    const uint32_t nb = n / 8;
    if (i < 32) {
        uint32_t acc = 0;
        if (i < nb)
            for (int b = 0; b < 8; ++b) acc |= (s_grey[8 * i + b] > mean) ? (1u << b) : 0u;
        out_hash[(size_t)blockIdx.x * 32 + i] = (uint8_t)acc;
    }
    if (i == 0) out_nbytes[blockIdx.x] = nb;
}

}  // namespace pbp
