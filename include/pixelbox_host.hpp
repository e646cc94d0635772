// pixelbox_host.hpp -- header-only C++ host layer above the C ABI (include/pixelbox_hip.h), mirroring the
// reference's Rust interface for the hot path name for name, so that the parity tests read like the
// reference's own code (the Rust toolchain is absent here; the Rust binding is in INTEGRATION.md):
//
//   pixelbox::image_hashes::mlhash(embedder, img) -> Vec<u8>          src/image_hashes/efficientnet.rs:31-42
//   pixelbox::IndexedImage {id, filename, path, visual_hash, distance_from_query, ...}
//                                                                      src/indexed_image.rs:16-32
//   pixelbox::Engine::insert_image_from_memory(IndexedImage)           src/engine.rs:224-259
//   pixelbox::Engine::query_by_image_hash_from_image(&IndexedImage)    src/engine.rs:363-396
//   pixelbox::Engine::query_by_image_hash_from_file(&Path)             src/engine.rs:352-361
//   pixelbox::Engine::get_query_results() -> Option<Vec<IndexedImage>> src/engine.rs:398-400
//   pixelbox::image_hashes::phash(hasher, img) -> Vec<u8>             src/image_hashes/phash.rs:3-22
//   pixelbox::IndexedImage::from_file_path / from_memory              src/indexed_image.rs:35-91 (decode is a callback)
//   pixelbox::Engine::max_distance_from_query (default 1e3)            src/engine.rs:23,92
//
// What stays in SQLite in a real integration (images/tags tables, persistence) is modelled here by an
// in-memory `images` map: enough to reproduce the INNER JOIN of engine.rs:377 (results whose image row
// is missing are dropped) and the UNIQUE(path) + INSERT OR IGNORE behaviour of engine.rs:40,230-233.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "pixelbox_hip.h"

namespace pixelbox {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc != PB_OK) throw Error(rc, pb_last_error());
}

// A decoded RGB8 image (what `DynamicImage::resize_to_fill(W, H, Triangle).to_rgb8()` yields).
struct RgbImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> pixels;  // HWC
};

class Embedder {
  public:
    Embedder(const void *weights_blob, size_t len, uint32_t max_batch = 512, int device = 0) {
        pb_embedder *h = nullptr;
        check(pb_embed_create(&h, device, weights_blob, len, max_batch));
        h_.reset(h);
        check(pb_embed_info(h, &height_, &width_, &dim_, &max_batch_));
    }
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    uint32_t dim() const { return dim_; }
    uint32_t max_batch() const { return max_batch_; }
    pb_embedder *raw() const { return h_.get(); }
    // Whoever needs the embedder's DEVICE output to stay put beyond one call holds this lock across the call and its use of the
    // pointer (pb_embed_batch_images_device: "valid until the next call on this embedder") -- the crawler's embed thread from
    // the forward pass to the end of the device-to-device insert.  Every hashing helper below takes it for its own call, so a
    // query hashed on the same embedder while indexing runs (the reference allows it: engine.rs:352-361 beside :177-205) waits
    // for the insert instead of overwriting the batch it reads.  Recursive: the holder calls those helpers.
    std::recursive_mutex &exclusive() const { return use_mu_; }

  private:
    struct Del {
        void operator()(pb_embedder *p) const { pb_embed_destroy(p); }
    };
    std::unique_ptr<pb_embedder, Del> h_;
    uint32_t width_ = 0, height_ = 0, dim_ = 0, max_batch_ = 0;
    mutable std::recursive_mutex use_mu_;
};

class PHasher {
  public:
    explicit PHasher(int device = 0) {
        pb_phasher *h = nullptr;
        check(pb_phash_create(&h, device));
        h_.reset(h);
    }
    pb_phasher *raw() const { return h_.get(); }

  private:
    struct Del {
        void operator()(pb_phasher *p) const { pb_phash_destroy(p); }
    };
    std::unique_ptr<pb_phasher, Del> h_;
};

namespace image_hashes {
// pub fn phash(img:&DynamicImage) -> Vec<u8>   (phash.rs:3-22; 32 bytes for a square image, (w2 * h2) / 8 otherwise)
inline std::vector<uint8_t> phash(const PHasher &hasher, const RgbImage &img) {
    if (img.width == 0 || img.height == 0 || img.pixels.size() != (size_t)img.width * img.height * 3)
        throw Error(PB_ERR_INVALID, "phash: empty image or pixel buffer of the wrong size");
    std::vector<uint8_t> out(32);
    uint32_t n = 0;
    check(pb_phash_image(hasher.raw(), img.pixels.data(), img.width, img.height, out.data(), out.size(), &n));
    out.resize(n);
    return out;
}
// the same for a batch of images in one call (pb_phash_batch_images: two launches per <= 64 MB of source pixels)
inline std::vector<std::vector<uint8_t>> phash_batch(const PHasher &hasher, const std::vector<RgbImage> &imgs) {
    std::vector<const uint8_t *> ptrs(imgs.size());
    std::vector<uint32_t> ws(imgs.size()), hs(imgs.size()), nb(imgs.size());
    for (size_t i = 0; i < imgs.size(); ++i) {
        if (imgs[i].width == 0 || imgs[i].height == 0 || imgs[i].pixels.size() != (size_t)imgs[i].width * imgs[i].height * 3)
            throw Error(PB_ERR_INVALID, "phash_batch: empty image or pixel buffer of the wrong size");
        ptrs[i] = imgs[i].pixels.data();
        ws[i] = imgs[i].width;
        hs[i] = imgs[i].height;
    }
    std::vector<uint8_t> out(imgs.size() * 32);
    check(pb_phash_batch_images(hasher.raw(), ptrs.data(), ws.data(), hs.data(), (uint32_t)imgs.size(), out.data(), nb.data()));
    std::vector<std::vector<uint8_t>> res(imgs.size());
    for (size_t i = 0; i < imgs.size(); ++i) res[i].assign(out.begin() + i * 32, out.begin() + i * 32 + nb[i]);
    return res;
}
// ... of pixel blocks that are not RgbImages (the staging slots of pb_embed_stage_close)
inline std::vector<std::vector<uint8_t>> phash_batch(const PHasher &hasher, const std::vector<const uint8_t *> &ptrs, const std::vector<uint32_t> &ws,
                                                     const std::vector<uint32_t> &hs) {
    std::vector<uint32_t> nb(ptrs.size());
    std::vector<uint8_t> out(ptrs.size() * 32);
    check(pb_phash_batch_images(hasher.raw(), ptrs.data(), ws.data(), hs.data(), (uint32_t)ptrs.size(), out.data(), nb.data()));
    std::vector<std::vector<uint8_t>> res(ptrs.size());
    for (size_t i = 0; i < ptrs.size(); ++i) res[i].assign(out.begin() + i * 32, out.begin() + i * 32 + nb[i]);
    return res;
}
// pub fn mlhash(img:&DynamicImage) -> Vec<u8>   (the model is an explicit handle instead of a lazy static)
// Any image size: `resize_to_fill(W, H, Triangle)` of efficientnet.rs:20 runs on the GPU (pb_mlhash_image); an
// image that already has the model's input size goes straight in, as in the image crate.
inline std::vector<uint8_t> mlhash(const Embedder &model, const RgbImage &img) {
    if (img.width == 0 || img.height == 0 || img.pixels.size() != (size_t)img.width * img.height * 3)
        throw Error(PB_ERR_INVALID, "mlhash: empty image or pixel buffer of the wrong size");
    std::vector<uint8_t> out(model.dim());
    std::lock_guard<std::recursive_mutex> use(model.exclusive());
    check(pb_mlhash_image(model.raw(), img.pixels.data(), img.width, img.height, out.data(), out.size()));
    return out;
}
// batched form used by a re-built crawler stage
// d_hashes (optional): the batch (at most the embedder's max_batch images) is run by pb_embed_batch_images_device and
// *d_hashes receives the hashes' address in the GPU's memory (valid until the next call on this embedder)
inline std::vector<std::vector<uint8_t>> mlhash_batch(const Embedder &model, const std::vector<RgbImage> &imgs,
                                                      const uint8_t **d_hashes = nullptr) {
    std::vector<uint8_t> out(imgs.size() * model.dim());
    std::vector<const uint8_t *> ptrs(imgs.size());
    std::vector<uint32_t> ws(imgs.size()), hs(imgs.size());
    for (size_t i = 0; i < imgs.size(); ++i) {
        if (imgs[i].pixels.size() != (size_t)imgs[i].width * imgs[i].height * 3) throw Error(PB_ERR_INVALID, "mlhash_batch: wrong pixel buffer size");
        ptrs[i] = imgs[i].pixels.data();
        ws[i] = imgs[i].width;
        hs[i] = imgs[i].height;
    }
    std::lock_guard<std::recursive_mutex> use(model.exclusive());
    if (d_hashes) check(pb_embed_batch_images_device(model.raw(), ptrs.data(), ws.data(), hs.data(), (uint32_t)imgs.size(), out.data(), d_hashes));
    else check(pb_embed_batch_images(model.raw(), ptrs.data(), ws.data(), hs.data(), (uint32_t)imgs.size(), out.data(), nullptr));
    std::vector<std::vector<uint8_t>> res(imgs.size());
    for (size_t i = 0; i < imgs.size(); ++i) res[i].assign(out.begin() + i * model.dim(), out.begin() + (i + 1) * model.dim());
    return res;
}
}  // namespace image_hashes

// src/indexed_image.rs:16-32 (the fields the hot path touches)
struct IndexedImage {
    int64_t id = 0;
    std::string filename;
    std::string path;
    std::pair<uint32_t, uint32_t> resolution{0, 0};
    std::optional<std::vector<uint8_t>> phash;
    std::optional<std::vector<uint8_t>> visual_hash;
    std::optional<double> distance_from_query;
};

// ---- decode: the image crate's job in the reference (indexed_image.rs:47-56 `ImageReader::with_guessed_format().decode()`).
// Codecs stay on the CPU and outside this library; a host plugs its decoder in as a callback.  Built in: binary PNM (P6 RGB,
// P5 grey, maxval <= 255) -- "pnm" is one of the twelve extensions the crawler accepts (crawler.rs:7) and needs no codec.
using Decoder = std::function<std::optional<RgbImage>(const std::vector<uint8_t> &bytes)>;

// A decoder that writes where it is told (round 5: pb_embed_stage_*): it parses the header, asks `alloc(w, h)` for room -- w * h * 3
// bytes, rows top to bottom; null: give up -- and writes the RGB8 rows there.  false: not an image of this decoder's kind, or alloc
// said no.  (The image crate's `ImageDecoder::read_image(buf)` has this shape.)
using PixelAlloc = std::function<uint8_t *(uint32_t w, uint32_t h)>;
using StagedDecoder = std::function<bool(const std::vector<uint8_t> &bytes, const PixelAlloc &alloc)>;

inline bool decode_pnm_into(const std::vector<uint8_t> &b, const PixelAlloc &alloc) {
    size_t pos = 0;
    auto token = [&]() -> std::string {
        for (;;) {  // whitespace and # comments
            while (pos < b.size() && (b[pos] == ' ' || b[pos] == '\n' || b[pos] == '\r' || b[pos] == '\t')) ++pos;
            if (pos < b.size() && b[pos] == '#') {
                while (pos < b.size() && b[pos] != '\n') ++pos;
                continue;
            }
            break;
        }
        std::string t;
        while (pos < b.size() && b[pos] > ' ') t.push_back((char)b[pos++]);
        return t;
    };
    const std::string magic = token();
    if (magic != "P6" && magic != "P5") return false;
    const std::string ws = token(), hs = token(), ms = token();
    if (ws.empty() || hs.empty() || ms.empty()) return false;
    const long w = std::strtol(ws.c_str(), nullptr, 10), h = std::strtol(hs.c_str(), nullptr, 10), mx = std::strtol(ms.c_str(), nullptr, 10);
    if (w < 1 || h < 1 || w > 65535 || h > 65535 || mx < 1 || mx > 255) return false;
    ++pos;  // the single whitespace byte after maxval
    const size_t ch = magic == "P6" ? 3 : 1, need = (size_t)w * h * ch;
    if (pos + need > b.size()) return false;
    uint8_t *px = alloc((uint32_t)w, (uint32_t)h);
    if (!px) return false;
    if (ch == 3 && mx == 255) {
        std::memcpy(px, b.data() + pos, need);  // the common case: the file's samples are the pixels
        return true;
    }
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (size_t c = 0; c < 3; ++c) {
            const uint32_t v = b[pos + i * ch + (ch == 3 ? c : 0)];
            px[3 * i + c] = (uint8_t)(mx == 255 ? v : (v * 255 + mx / 2) / mx);
        }
    return true;
}

inline std::optional<RgbImage> decode_pnm(const std::vector<uint8_t> &b) {
    RgbImage img;
    const bool ok = decode_pnm_into(b, [&](uint32_t w, uint32_t h) {
        img.width = w;
        img.height = h;
        img.pixels.resize((size_t)w * h * 3);
        return img.pixels.data();
    });
    if (!ok) return std::nullopt;
    return img;
}

inline std::optional<std::vector<uint8_t>> read_file(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return std::nullopt;
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return bytes;
}

// IndexedImage::from_memory (indexed_image.rs:47-91): decode, hashes (phash, mlhash).  The thumbnail / QOI / EXIF steps of
// the reference are CPU work outside the hot path and are not modelled.  std::nullopt = undecodable (the crawler skips it,
// crawler.rs:78).
inline std::optional<IndexedImage> indexed_image_from_memory(const std::vector<uint8_t> &bytes, const std::string &filename,
                                                             const std::string &path, const Embedder &model, const PHasher *hasher,
                                                             const Decoder &decode = decode_pnm) {
    std::optional<RgbImage> img = decode(bytes);
    if (!img) return std::nullopt;
    IndexedImage r;
    r.filename = filename;
    r.path = path;
    r.resolution = {img->width, img->height};
    if (hasher) r.phash = image_hashes::phash(*hasher, *img);   // indexed_image.rs:70
    r.visual_hash = image_hashes::mlhash(model, *img);          // indexed_image.rs:71
    return r;
}
// IndexedImage::from_file_path (indexed_image.rs:35-45)
inline std::optional<IndexedImage> indexed_image_from_file_path(const std::string &path, const Embedder &model, const PHasher *hasher,
                                                                const Decoder &decode = decode_pnm) {
    std::optional<std::vector<uint8_t>> bytes = read_file(path);
    if (!bytes) return std::nullopt;
    const size_t slash = path.find_last_of('/');
    return indexed_image_from_memory(*bytes, slash == std::string::npos ? path : path.substr(slash + 1), path, model, hasher, decode);
}

class Engine {
  public:
    static constexpr uint32_t RESULT_LIMIT = 100;  // `LIMIT 100`, engine.rs:314,381
    double max_distance_from_query = 1e3;          // engine.rs:23,92

    Engine(uint32_t hash_dim, uint64_t capacity_rows, int device = 0) : dim_(hash_dim) {
        pb_index *h = nullptr;
        check(pb_index_create(&h, device, hash_dim, capacity_rows));
        idx_.reset(h);
    }

    // engine.rs:224-259: INSERT OR IGNORE INTO images (...) keyed by UNIQUE(path); then
    // INSERT OR IGNORE INTO semantic_hashes (image_id, hash).
    void insert_image_from_memory(IndexedImage img) {
        auto known = by_path_.find(img.path);
        if (known == by_path_.end()) {
            img.id = ++last_rowid_;
            by_path_[img.path] = img.id;
            images_[img.id] = img;
        } else {
            img.id = known->second;  // row exists: the image insert is ignored, the hash insert below too
        }
        if (img.visual_hash) {
            if (img.visual_hash->size() != dim_) throw Error(PB_ERR_INVALID, "visual_hash length != index dim");
            uint64_t stored = 0;
            check(pb_index_append(idx_.get(), &img.id, img.visual_hash->data(), 1, &stored));
        }
    }

    // engine.rs:352-361: hash the file (decode + resize_to_fill + network, all but the decode on the GPU), then query.
    // Returns false where the reference would panic (`from_file_path(img).unwrap()`): unreadable or undecodable file.
    bool query_by_image_hash_from_file(const std::string &path, const Embedder &model, const PHasher *hasher = nullptr,
                                       const Decoder &decode = decode_pnm) {
        cached_search_results_.reset();
        const auto t0 = std::chrono::steady_clock::now();
        std::optional<IndexedImage> indexed_image = indexed_image_from_file_path(path, model, hasher, decode);
        last_hash_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();  // engine.rs:355-358
        if (!indexed_image) return false;
        query_by_image_hash_from_image(*indexed_image);
        return true;
    }

    // engine.rs:363-396
    void query_by_image_hash_from_image(const IndexedImage &indexed_image) {
        if (!indexed_image.visual_hash) return;  // engine.rs:364-368: logs and returns
        cached_search_results_.reset();
        const auto t0 = std::chrono::steady_clock::now();
        // The reference's INNER JOIN runs BEFORE `LIMIT 100`: a hash whose image row is missing does not use up a result
        // slot.  Ask the index for more than 100 when orphans turn up (PB_MAX_K at most) and cut after the join.
        std::vector<IndexedImage> out;
        for (uint32_t k = RESULT_LIMIT;; k = (2 * k < PB_MAX_K ? 2 * k : PB_MAX_K)) {
            std::vector<int64_t> ids(k);
            std::vector<float> dist(k);
            uint32_t count = 0;
            check(pb_index_search(idx_.get(), indexed_image.visual_hash->data(), 1, k, max_distance_from_query, ids.data(), dist.data(),
                                  &count));
            out.clear();
            for (uint32_t i = 0; i < count && out.size() < RESULT_LIMIT; ++i) {
                auto it = images_.find(ids[i]);
                if (it == images_.end()) continue;  // INNER JOIN images ON images.id = semantic_hashes.image_id
                IndexedImage r = it->second;
                r.distance_from_query = (double)dist[i];  // engine.rs:619 `Ok(dist as f64)`
                out.push_back(std::move(r));
            }
            if (out.size() == RESULT_LIMIT || count < k || k == PB_MAX_K) break;
        }
        cached_search_results_ = std::move(out);
        last_search_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();  // engine.rs:391-395
    }
    // the two timings the reference prints (engine.rs:355-358, 391-395)
    double last_hash_ms = 0.0, last_search_ms = 0.0;

    // engine.rs:398-400
    std::optional<std::vector<IndexedImage>> get_query_results() const { return cached_search_results_; }

    uint64_t get_num_indexed_images() const {
        uint64_t n = 0;
        check(pb_index_size(idx_.get(), &n));
        return n;
    }
    pb_index *raw() const { return idx_.get(); }

  private:
    struct Del {
        void operator()(pb_index *p) const { pb_index_destroy(p); }
    };
    std::unique_ptr<pb_index, Del> idx_;
    uint32_t dim_;
    int64_t last_rowid_ = 0;
    std::map<int64_t, IndexedImage> images_;
    std::unordered_map<std::string, int64_t> by_path_;
    std::optional<std::vector<IndexedImage>> cached_search_results_;
};

}  // namespace pixelbox
