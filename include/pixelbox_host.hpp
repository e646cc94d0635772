// pixelbox_host.hpp -- header-only C++ host layer above the C ABI (include/pixelbox_hip.h), mirroring the
// reference's Rust interface for the hot path name for name, so that the parity tests read like the
// reference's own code (the Rust toolchain is absent here; the Rust binding is in INTEGRATION.md):
//
//   pixelbox::image_hashes::mlhash(embedder, img) -> Vec<u8>          src/image_hashes/efficientnet.rs:31-42
//   pixelbox::IndexedImage {id, filename, path, visual_hash, distance_from_query, ...}
//                                                                      src/indexed_image.rs:16-32
//   pixelbox::Engine::insert_image_from_memory(IndexedImage)           src/engine.rs:224-259
//   pixelbox::Engine::query_by_image_hash_from_image(&IndexedImage)    src/engine.rs:363-396
//   pixelbox::Engine::get_query_results() -> Option<Vec<IndexedImage>> src/engine.rs:398-400
//   pixelbox::Engine::max_distance_from_query (default 1e3)            src/engine.rs:23,92
//
// What stays in SQLite in a real integration (images/tags tables, persistence) is modelled here by an
// in-memory `images` map: enough to reproduce the INNER JOIN of engine.rs:377 (results whose image row
// is missing are dropped) and the UNIQUE(path) + INSERT OR IGNORE behaviour of engine.rs:40,230-233.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
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
    pb_embedder *raw() const { return h_.get(); }

  private:
    struct Del {
        void operator()(pb_embedder *p) const { pb_embed_destroy(p); }
    };
    std::unique_ptr<pb_embedder, Del> h_;
    uint32_t width_ = 0, height_ = 0, dim_ = 0, max_batch_ = 0;
};

namespace image_hashes {
// pub fn mlhash(img:&DynamicImage) -> Vec<u8>   (the model is an explicit handle instead of a lazy static)
// Any image size: `resize_to_fill(W, H, Triangle)` of efficientnet.rs:20 runs on the GPU (pb_mlhash_image); an
// image that already has the model's input size goes straight in, as in the image crate.
inline std::vector<uint8_t> mlhash(const Embedder &model, const RgbImage &img) {
    if (img.width == 0 || img.height == 0 || img.pixels.size() != (size_t)img.width * img.height * 3)
        throw Error(PB_ERR_INVALID, "mlhash: empty image or pixel buffer of the wrong size");
    std::vector<uint8_t> out(model.dim());
    check(pb_mlhash_image(model.raw(), img.pixels.data(), img.width, img.height, out.data(), out.size()));
    return out;
}
// batched form used by a re-built crawler stage
inline std::vector<std::vector<uint8_t>> mlhash_batch(const Embedder &model, const std::vector<RgbImage> &imgs) {
    std::vector<uint8_t> out(imgs.size() * model.dim());
    std::vector<const uint8_t *> ptrs(imgs.size());
    std::vector<uint32_t> ws(imgs.size()), hs(imgs.size());
    for (size_t i = 0; i < imgs.size(); ++i) {
        if (imgs[i].pixels.size() != (size_t)imgs[i].width * imgs[i].height * 3) throw Error(PB_ERR_INVALID, "mlhash_batch: wrong pixel buffer size");
        ptrs[i] = imgs[i].pixels.data();
        ws[i] = imgs[i].width;
        hs[i] = imgs[i].height;
    }
    check(pb_embed_batch_images(model.raw(), ptrs.data(), ws.data(), hs.data(), (uint32_t)imgs.size(), out.data(), nullptr));
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
    std::optional<std::vector<uint8_t>> visual_hash;
    std::optional<double> distance_from_query;
};

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

    // engine.rs:363-396
    void query_by_image_hash_from_image(const IndexedImage &indexed_image) {
        if (!indexed_image.visual_hash) return;  // engine.rs:364-368: logs and returns
        cached_search_results_.reset();
        std::vector<int64_t> ids(RESULT_LIMIT);
        std::vector<float> dist(RESULT_LIMIT);
        uint32_t count = 0;
        check(pb_index_search(idx_.get(), indexed_image.visual_hash->data(), 1, RESULT_LIMIT, max_distance_from_query,
                              ids.data(), dist.data(), &count));
        std::vector<IndexedImage> out;
        for (uint32_t i = 0; i < count; ++i) {
            auto it = images_.find(ids[i]);
            if (it == images_.end()) continue;  // INNER JOIN images ON images.id = semantic_hashes.image_id
            IndexedImage r = it->second;
            r.distance_from_query = (double)dist[i];  // engine.rs:619 `Ok(dist as f64)`
            out.push_back(std::move(r));
        }
        cached_search_results_ = std::move(out);
    }

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
