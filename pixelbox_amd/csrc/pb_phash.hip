// pb_phash.hip -- host side of `image_hashes::phash` (src/image_hashes/phash.rs:3-22) behind the C ABI.  gfx950 only; no
// CPU fallback: every entry point fails with PB_ERR_HIP when there is no GPU.
//
// The 32-byte average hash is stored in the `phashes` table (engine.rs:106-109,248-250) and compared with
// hamming_distance (engine.rs:594-604); pb_index_create_metric(PB_METRIC_HAMMING) scans such a table.
#include <cmath>
#include <mutex>
#include <new>
#include <vector>

#include "pb_common.h"
#include "pb_phash_kernels.h"

#pragma clang fp contract(off)

struct pb_phasher {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_src = nullptr;   // source image bytes (grow-only)
    float *d_tmp = nullptr;     // vertical-pass output f32 [h2][w][3] (grow-only)
    float *d_wts = nullptr;     // filter weights of both passes (grow-only)
    uint32_t *d_meta = nullptr; // left / count of both passes: 4 x 16
    uint8_t *d_out = nullptr;   // 32 hash bytes + 768 bytes of the resized image
    uint32_t *d_nb = nullptr;
    size_t src_cap = 0, tmp_cap = 0, wts_cap = 0;
    // batch form (pb_phash_batch_images): pinned staging of a sub-batch's sources, weights, windows and descriptors; results
    uint8_t *h_bsrc = nullptr, *d_bsrc = nullptr;
    float *h_bwts = nullptr, *d_bwts = nullptr, *d_btmp = nullptr;
    uint32_t *h_bmeta = nullptr, *d_bmeta = nullptr, *h_bnb = nullptr, *d_bnb = nullptr;
    pbp::PhashDesc *h_bdesc = nullptr, *d_bdesc = nullptr;
    uint8_t *h_bout = nullptr, *d_bout = nullptr;
    size_t bsrc_cap = 0, bwts_cap = 0, btmp_cap = 0, bimg_cap = 0;
    std::mutex mu;
};

namespace {

// src/math/utils.rs resize_dimensions(w, h, 16, 16, fill = false)
void fit16(uint32_t w, uint32_t h, uint32_t *w2, uint32_t *h2) {
    const double wratio = 16.0 / (double)w, hratio = 16.0 / (double)h;
    const double ratio = wratio < hratio ? wratio : hratio;
    const double a = std::round((double)w * ratio), b = std::round((double)h * ratio);
    *w2 = a < 1.0 ? 1u : (uint32_t)a;
    *h2 = b < 1.0 ? 1u : (uint32_t)b;
}

// sample.rs gaussian(x, 0.5): ((2 pi).sqrt() * r).recip() * (-x.powi(2) / (2.0 * r.powi(2))).exp()
float gaussian_kernel(float x) {
    const float pi = 3.14159274101257324f, r = 0.5f;
    const float norm = 1.0f / (std::sqrt(2.0f * pi) * r);
    const float x2 = x * x;
    const float den = 2.0f * (r * r);
    const float arg = -x2 / den;
    return norm * std::exp(arg);
}

// windows and normalised weights of all out_size outputs (sample.rs, shared by both passes); returns the longest window
uint32_t make_weights(uint32_t in_size, uint32_t out_size, std::vector<std::vector<float>> &ws, uint32_t *left, uint32_t *cnt) {
    const float ratio = (float)in_size / (float)out_size;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 3.0f * sratio;
    uint32_t longest = 0;
    ws.assign(out_size, {});
    for (uint32_t o = 0; o < out_size; ++o) {
        float input = ((float)o + 0.5f) * ratio;
        int64_t l = (int64_t)std::floor(input - src_support);
        if (l < 0) l = 0;
        if (l > (int64_t)in_size - 1) l = (int64_t)in_size - 1;
        int64_t r = (int64_t)std::ceil(input + src_support);
        if (r < l + 1) r = l + 1;
        if (r > (int64_t)in_size) r = (int64_t)in_size;
        input = input - 0.5f;
        float sum = 0.0f;
        for (int64_t i = l; i < r; ++i) {
            const float wv = gaussian_kernel(((float)i - input) / sratio);
            ws[o].push_back(wv);
            sum = sum + wv;
        }
        for (float &wv : ws[o]) wv = wv / sum;
        left[o] = (uint32_t)l;
        cnt[o] = (uint32_t)ws[o].size();
        longest = std::max<uint32_t>(longest, cnt[o]);
    }
    return longest;
}

template <typename T>
int grow(T **p, size_t *cap, size_t want) {
    if (want <= *cap) return PB_OK;
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    PB_HIP(hipMalloc(p, want * sizeof(T)));
    *cap = want;
    return PB_OK;
}

int phash_one(pb_phasher *p, const uint8_t *rgb, uint32_t w, uint32_t h, uint8_t *out, uint32_t *n_bytes, uint8_t *small_rgb) {
    PB_CHECK(rgb, PB_ERR_INVALID, "pb_phash: null image");
    PB_CHECK(w >= 1 && h >= 1 && w <= 65535 && h <= 65535, PB_ERR_INVALID, "pb_phash: image size %ux%u outside 1..65535", w, h);
    uint32_t w2, h2;
    fit16(w, h, &w2, &h2);
    const size_t src_bytes = (size_t)w * h * 3;
    int rc = grow(&p->d_src, &p->src_cap, src_bytes);
    if (rc) return rc;
    PB_HIP(hipMemcpyAsync(p->d_src, rgb, src_bytes, hipMemcpyHostToDevice, p->stream));
    uint8_t *d_small = p->d_out + 32;
    if (w2 == w && h2 == h) {  // imageops::resize: same dimensions -> copy
        hipLaunchKernelGGL(pbp::k_phash_h, dim3(1), dim3(256), 0, p->stream, (const float *)nullptr, p->d_src, 0, w, w2, h2,
                           (const float *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, p->d_out, p->d_nb, d_small);
        PB_HIP(hipGetLastError());
    } else {
        std::vector<std::vector<float>> wv, wh;
        uint32_t meta[64];  // left_v[16] cnt_v[16] left_h[16] cnt_h[16]
        const uint32_t sv = make_weights(h, h2, wv, meta, meta + 16);
        const uint32_t sh = make_weights(w, w2, wh, meta + 32, meta + 48);
        std::vector<float> flat((size_t)16 * sv + (size_t)16 * sh, 0.0f);
        for (uint32_t o = 0; o < h2; ++o) std::copy(wv[o].begin(), wv[o].end(), flat.begin() + (size_t)o * sv);
        for (uint32_t o = 0; o < w2; ++o) std::copy(wh[o].begin(), wh[o].end(), flat.begin() + (size_t)16 * sv + (size_t)o * sh);
        if ((rc = grow(&p->d_wts, &p->wts_cap, flat.size()))) return rc;
        if ((rc = grow(&p->d_tmp, &p->tmp_cap, (size_t)h2 * w * 3))) return rc;
        // pageable sources of queued copies must outlive them: both copies are waited for below, before flat / meta die
        PB_HIP(hipMemcpyAsync(p->d_wts, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice, p->stream));
        PB_HIP(hipMemcpyAsync(p->d_meta, meta, sizeof(meta), hipMemcpyHostToDevice, p->stream));
        hipLaunchKernelGGL(pbp::k_phash_v, dim3((w * 3 + 255) / 256, h2), dim3(256), 0, p->stream, p->d_src, w, p->d_wts, p->d_meta,
                           p->d_meta + 16, sv, p->d_tmp);
        PB_HIP(hipGetLastError());
        hipLaunchKernelGGL(pbp::k_phash_h, dim3(1), dim3(256), 0, p->stream, p->d_tmp, (const uint8_t *)nullptr, 1, w, w2, h2,
                           p->d_wts + (size_t)16 * sv, p->d_meta + 32, p->d_meta + 48, sh, p->d_out, p->d_nb, d_small);
        PB_HIP(hipGetLastError());
        PB_HIP(hipStreamSynchronize(p->stream));
    }
    uint8_t host[32 + 768];
    uint32_t nb = 0;
    PB_HIP(hipMemcpyAsync(host, p->d_out, sizeof(host), hipMemcpyDeviceToHost, p->stream));
    PB_HIP(hipMemcpyAsync(&nb, p->d_nb, sizeof(nb), hipMemcpyDeviceToHost, p->stream));
    PB_HIP(hipStreamSynchronize(p->stream));
    memcpy(out, host, nb);
    *n_bytes = nb;
    if (small_rgb) memcpy(small_rgb, host + 32, (size_t)w2 * h2 * 3);
    return PB_OK;
}

// ---- batch form: n images of individual sizes -> out[n][32] (+ n_bytes[n]) with two launches per sub-batch of <= 64 MB of
// source pixels: sources packed into one pinned block (one transfer), the Gaussian weights and windows of every image computed
// on the host (libm expf, as for one image) into one block, a descriptor per image; ONE wait per sub-batch.
constexpr size_t PH_STAGE_BYTES = 64u << 20;
constexpr size_t PH_STAGE_TMP_FLOATS = 64u << 20;  // 256 MB of vertical-pass scratch per sub-batch
constexpr uint32_t PH_STAGE_IMAGES = 1024;

template <typename T>
int grow_pair(T **h, T **d, size_t *cap, size_t want) {
    if (want <= *cap) return PB_OK;
    if (*h) (void)hipHostFree(*h);
    (void)hipFree(*d);
    *h = nullptr; *d = nullptr; *cap = 0;
    PB_HIP(hipHostMalloc(reinterpret_cast<void **>(h), want * sizeof(T), hipHostMallocDefault));
    PB_HIP(hipMalloc(reinterpret_cast<void **>(d), want * sizeof(T)));
    *cap = want;
    return PB_OK;
}

int phash_batch(pb_phasher *p, const uint8_t *const *rgb, const uint32_t *widths, const uint32_t *heights, uint32_t n, uint8_t *out,
                uint32_t *n_bytes) {
    if (p->bimg_cap < PH_STAGE_IMAGES) {
        size_t c1 = p->bimg_cap, c2 = p->bimg_cap, c3 = p->bimg_cap, c4 = p->bimg_cap;
        int rc = grow_pair(&p->h_bdesc, &p->d_bdesc, &c1, (size_t)PH_STAGE_IMAGES);
        if (!rc) rc = grow_pair(&p->h_bmeta, &p->d_bmeta, &c2, (size_t)PH_STAGE_IMAGES * 64);
        if (!rc) rc = grow_pair(&p->h_bout, &p->d_bout, &c3, (size_t)PH_STAGE_IMAGES * 32);
        if (!rc) rc = grow_pair(&p->h_bnb, &p->d_bnb, &c4, (size_t)PH_STAGE_IMAGES);
        if (rc) return rc;
        p->bimg_cap = PH_STAGE_IMAGES;
    }
    std::vector<float> wts;
    std::vector<std::vector<float>> wv, wh;
    for (uint32_t i0 = 0; i0 < n;) {
        size_t src_total = 0, tmp_total = 0;
        uint32_t i1 = i0, max_w = 1;
        wts.clear();
        for (; i1 < n && i1 - i0 < PH_STAGE_IMAGES; ++i1) {
            const uint32_t w = widths[i1], h = heights[i1];
            PB_CHECK(rgb[i1], PB_ERR_INVALID, "pb_phash_batch_images: image %u: null pointer", i1);
            PB_CHECK(w >= 1 && h >= 1 && w <= 65535 && h <= 65535, PB_ERR_INVALID, "pb_phash_batch_images: image %u: size %ux%u outside 1..65535", i1, w, h);
            const size_t sb = (size_t)w * h * 3;
            if (i1 > i0 && src_total + sb > PH_STAGE_BYTES) break;
            if (i1 > i0 && tmp_total + (size_t)16 * w * 3 > PH_STAGE_TMP_FLOATS) break;  // scratch: <= 16 rows x w x 3 floats per image (wide, short sources)
            pbp::PhashDesc d{};
            d.w = w; d.h = h;
            fit16(w, h, &d.w2, &d.h2);
            d.resample = (d.w2 == w && d.h2 == h) ? 0u : 1u;
            d.src_off = src_total;
            d.tmp_off = tmp_total;
            uint32_t *meta = p->h_bmeta + (size_t)(i1 - i0) * 64;
            memset(meta, 0, 64 * sizeof(uint32_t));
            if (d.resample) {
                d.sv = make_weights(h, d.h2, wv, meta, meta + 16);
                d.sh = make_weights(w, d.w2, wh, meta + 32, meta + 48);
                d.wv_off = (uint32_t)wts.size();
                wts.resize(wts.size() + (size_t)16 * d.sv, 0.0f);
                for (uint32_t o = 0; o < d.h2; ++o) std::copy(wv[o].begin(), wv[o].end(), wts.begin() + d.wv_off + (size_t)o * d.sv);
                d.wh_off = (uint32_t)wts.size();
                wts.resize(wts.size() + (size_t)16 * d.sh, 0.0f);
                for (uint32_t o = 0; o < d.w2; ++o) std::copy(wh[o].begin(), wh[o].end(), wts.begin() + d.wh_off + (size_t)o * d.sh);
                tmp_total += ((size_t)d.h2 * w * 3 + 3) & ~(size_t)3;
            }
            src_total += (sb + 15) & ~(size_t)15;
            max_w = std::max(max_w, w);
            p->h_bdesc[i1 - i0] = d;
        }
        const uint32_t m = i1 - i0;
        int rc = grow_pair(&p->h_bsrc, &p->d_bsrc, &p->bsrc_cap, std::max(src_total, PH_STAGE_BYTES));
        if (!rc) rc = grow_pair(&p->h_bwts, &p->d_bwts, &p->bwts_cap, std::max<size_t>(wts.size(), 1u << 16));
        if (!rc) rc = grow(&p->d_btmp, &p->btmp_cap, std::max<size_t>(tmp_total, 1));
        if (rc) return rc;
        for (uint32_t i = 0; i < m; ++i) memcpy(p->h_bsrc + p->h_bdesc[i].src_off, rgb[i0 + i], (size_t)p->h_bdesc[i].w * p->h_bdesc[i].h * 3);
        if (!wts.empty()) memcpy(p->h_bwts, wts.data(), wts.size() * sizeof(float));
        PB_HIP(hipMemcpyAsync(p->d_bsrc, p->h_bsrc, src_total, hipMemcpyHostToDevice, p->stream));
        if (!wts.empty()) PB_HIP(hipMemcpyAsync(p->d_bwts, p->h_bwts, wts.size() * sizeof(float), hipMemcpyHostToDevice, p->stream));
        PB_HIP(hipMemcpyAsync(p->d_bmeta, p->h_bmeta, (size_t)m * 64 * sizeof(uint32_t), hipMemcpyHostToDevice, p->stream));
        PB_HIP(hipMemcpyAsync(p->d_bdesc, p->h_bdesc, (size_t)m * sizeof(pbp::PhashDesc), hipMemcpyHostToDevice, p->stream));
        if (tmp_total) {
            hipLaunchKernelGGL(pbp::k_phash_v_batch, dim3((max_w * 3 + 255) / 256, 16, m), dim3(256), 0, p->stream, p->d_bsrc, p->d_bdesc, p->d_bwts,
                               p->d_bmeta, p->d_btmp);
            PB_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(pbp::k_phash_h_batch, dim3(m), dim3(256), 0, p->stream, p->d_btmp, p->d_bsrc, p->d_bdesc, p->d_bwts, p->d_bmeta, p->d_bout,
                           p->d_bnb);
        PB_HIP(hipGetLastError());
        PB_HIP(hipMemcpyAsync(p->h_bout, p->d_bout, (size_t)m * 32, hipMemcpyDeviceToHost, p->stream));
        PB_HIP(hipMemcpyAsync(p->h_bnb, p->d_bnb, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToHost, p->stream));
        PB_HIP(hipStreamSynchronize(p->stream));  // the pinned blocks are reused by the next sub-batch
        memcpy(out + (size_t)i0 * 32, p->h_bout, (size_t)m * 32);
        memcpy(n_bytes + i0, p->h_bnb, (size_t)m * sizeof(uint32_t));
        i0 = i1;
    }
    return PB_OK;
}

}  // namespace

extern "C" {

int pb_phash_create(pb_phasher **out, int device) {
    PB_CHECK(out, PB_ERR_INVALID, "pb_phash_create: null out pointer");
    *out = nullptr;
    int n_dev = 0;
    PB_HIP(hipGetDeviceCount(&n_dev));
    PB_CHECK(device >= 0 && device < n_dev, PB_ERR_INVALID, "pb_phash_create: device %d of %d", device, n_dev);
    pb::DeviceGuard guard(device);
    PB_CHECK(guard.ok, PB_ERR_HIP, "hipSetDevice(%d) failed", device);
    pb_phasher *p = new (std::nothrow) pb_phasher();
    PB_CHECK(p, PB_ERR_NOMEM, "out of host memory");
    p->device = device;
    auto body = [&]() -> int {
        PB_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        PB_HIP(hipMalloc(&p->d_meta, 64 * sizeof(uint32_t)));
        PB_HIP(hipMalloc(&p->d_out, 32 + 768));
        PB_HIP(hipMalloc(&p->d_nb, sizeof(uint32_t)));
        return PB_OK;
    };
    int rc = body();
    if (rc) {
        pb_phash_destroy(p);
        return rc;
    }
    *out = p;
    return PB_OK;
}

int pb_phash_destroy(pb_phasher *p) {
    if (!p) return PB_OK;
    {
        pb::DeviceGuard guard(p->device);
        if (p->stream) (void)hipStreamSynchronize(p->stream);
        (void)hipFree(p->d_src);
        (void)hipFree(p->d_tmp);
        (void)hipFree(p->d_wts);
        (void)hipFree(p->d_meta);
        (void)hipFree(p->d_out);
        (void)hipFree(p->d_nb);
        (void)hipFree(p->d_bsrc); (void)hipFree(p->d_bwts); (void)hipFree(p->d_btmp); (void)hipFree(p->d_bmeta); (void)hipFree(p->d_bnb);
        (void)hipFree(p->d_bdesc); (void)hipFree(p->d_bout);
        if (p->h_bsrc) (void)hipHostFree(p->h_bsrc);
        if (p->h_bwts) (void)hipHostFree(p->h_bwts);
        if (p->h_bmeta) (void)hipHostFree(p->h_bmeta);
        if (p->h_bnb) (void)hipHostFree(p->h_bnb);
        if (p->h_bdesc) (void)hipHostFree(p->h_bdesc);
        if (p->h_bout) (void)hipHostFree(p->h_bout);
        if (p->stream) (void)hipStreamDestroy(p->stream);
    }
    delete p;
    return PB_OK;
}

int pb_phash_image(pb_phasher *p, const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out, size_t out_len, uint32_t *n_bytes) {
    PB_CHECK(p, PB_ERR_INVALID, "pb_phash_image: null handle");
    PB_CHECK(out && n_bytes && out_len >= 32, PB_ERR_INVALID, "pb_phash_image: out must hold 32 bytes");
    std::lock_guard<std::mutex> lock(p->mu);
    pb::DeviceGuard guard(p->device);
    return phash_one(p, rgb, width, height, out, n_bytes, nullptr);
}

int pb_phash_batch_images(pb_phasher *p, const uint8_t *const *rgb, const uint32_t *widths, const uint32_t *heights, uint32_t n, uint8_t *out,
                          uint32_t *n_bytes) {
    PB_CHECK(p, PB_ERR_INVALID, "pb_phash_batch_images: null handle");
    PB_CHECK(n == 0 || (rgb && widths && heights && out && n_bytes), PB_ERR_INVALID, "pb_phash_batch_images: null buffer");
    if (n == 0) return PB_OK;
    std::lock_guard<std::mutex> lock(p->mu);
    pb::DeviceGuard guard(p->device);
    const int rc = phash_batch(p, rgb, widths, heights, n, out, n_bytes);
    if (rc) (void)hipStreamSynchronize(p->stream);  // nothing of this call may still be reading its staging blocks
    return rc;
}

int pb_phash_small_image(pb_phasher *p, const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out_rgb, uint32_t *out_w, uint32_t *out_h) {
    PB_CHECK(p && out_rgb && out_w && out_h, PB_ERR_INVALID, "pb_phash_small_image: null pointer");
    std::lock_guard<std::mutex> lock(p->mu);
    pb::DeviceGuard guard(p->device);
    uint8_t hash[32];
    uint32_t nb = 0;
    int rc = phash_one(p, rgb, width, height, hash, &nb, out_rgb);
    if (rc) return rc;
    fit16(width, height, out_w, out_h);
    return PB_OK;
}

}  // extern "C"
