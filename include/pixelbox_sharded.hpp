// pixelbox_sharded.hpp -- the reference's Engine for a node with several GPUs, in ONE host process (the reference is one Rust
// process: src/engine.rs:79-145), above the row-sharded table of the C ABI (pb_sharded_*, include/pixelbox_hip.h):
//
//   ShardedEngine::start_indexing(folders)           src/engine.rs:177-205 + src/crawler.rs:21-122
//       the Crawler's decode workers feed ONE EMBED THREAD PER GPU (a pb_embedder beside every shard); each finished batch is
//       given image ids (the `last_insert_rowid()` of engine.rs:233,249), recorded in the `images` map and its hashes are
//       stored on that GPU's shard straight from the embedder's device buffer (pb_sharded_append_device): N GPUs embed and
//       insert concurrently, no hash crosses PCIe on its way into the index (BASELINE configs[4]);
//   ShardedEngine::insert_image_from_memory          src/engine.rs:224-259 (INSERT OR IGNORE over all shards)
//   ShardedEngine::query_by_image_hash_from_image    src/engine.rs:363-396: every shard answers on its own GPU, one RCCL
//       all-gather of the per-shard top-k over xGMI, device merge, then the INNER JOIN with `images` BEFORE the limit;
//   ShardedEngine::query_by_image_hash_from_file     src/engine.rs:352-361.
// What stays in SQLite in a real integration is the in-memory `images` map here, as in pixelbox_host.hpp.
#pragma once
#include <mutex>

#include "pixelbox_crawler.hpp"

namespace pixelbox {

class ShardedEngine {
  public:
    static constexpr uint32_t RESULT_LIMIT = 100;  // `LIMIT 100`, engine.rs:314,381
    double max_distance_from_query = 1e3;          // engine.rs:23,92

    // one shard per entry of device_ids (a device may repeat: the one-GPU test topology) and, when a weight blob is given,
    // one embedder beside every shard
    ShardedEngine(uint32_t hash_dim, uint64_t capacity_rows, const std::vector<int> &device_ids, const void *weights_blob = nullptr,
                  size_t blob_len = 0, uint32_t max_batch = 512)
        : dim_(hash_dim) {
        pb_sharded *h = nullptr;
        check(pb_sharded_create(&h, device_ids.data(), (int)device_ids.size(), hash_dim, capacity_rows));
        idx_.reset(h);
        if (weights_blob)
            for (size_t g = 0; g < device_ids.size(); ++g) {
                int dev = -1;
                check(pb_sharded_shard_device(h, (int)g, &dev));
                models_.push_back(std::make_unique<Embedder>(weights_blob, blob_len, max_batch, dev));
                if (models_.back()->dim() != hash_dim) throw Error(PB_ERR_INVALID, "ShardedEngine: the model's hash length != index dim");
            }
    }
    ~ShardedEngine() { crawler_.reset(); }  // the stage's threads use the table: they go first

    size_t n_shards() const { return models_.empty() ? shard_count() : models_.size(); }
    const Embedder &model(size_t g = 0) const { return *models_.at(g); }

    // engine.rs:177-205.  Returns at once; drain with recv_indexed() (the reference's insert thread, engine.rs:186-203) or
    // wait_for_indexing().  The records that arrive are already stored.
    void start_indexing(const std::vector<std::string> &folders, size_t num_workers = 4, const PHasher *hasher = nullptr,
                        Decoder decode = decode_pnm, uint32_t max_batch = 512) {
        if (models_.empty()) throw Error(PB_ERR_INVALID, "ShardedEngine::start_indexing: created without a model");
        std::vector<const Embedder *> ms;
        for (auto &m : models_) ms.push_back(m.get());
        crawler_ = std::make_unique<Crawler>(ms, hasher, std::move(decode), max_batch,
                                             [this](size_t g, std::vector<IndexedImage> &batch, const uint8_t *d_hashes) { store_batch(g, batch, d_hashes); });
        crawler_->start_indexing(folders, num_workers);
    }
    bool recv_indexed(IndexedImage &out) { return crawler_ && crawler_->recv(out); }
    // drains the channel; returns the number of images indexed by this run; throws what an embed thread failed with
    uint64_t wait_for_indexing() {
        uint64_t n = 0;
        IndexedImage r;
        while (recv_indexed(r)) ++n;
        if (crawler_ && !crawler_->error().empty()) throw Error(PB_ERR_HIP, crawler_->error());
        return n;
    }
    Crawler::Stats indexing_stats() const { return crawler_ ? crawler_->stats() : Crawler::Stats{}; }

    // engine.rs:224-259 for one record with a host-side hash
    void insert_image_from_memory(IndexedImage img) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto known = by_path_.find(img.path);
            if (known == by_path_.end()) {
                img.id = ++last_rowid_;
                by_path_[img.path] = img.id;
                images_[img.id] = img;
            } else {
                img.id = known->second;
            }
        }
        if (img.visual_hash) {
            if (img.visual_hash->size() != dim_) throw Error(PB_ERR_INVALID, "visual_hash length != index dim");
            uint64_t stored = 0;
            check(pb_sharded_append(idx_.get(), &img.id, img.visual_hash->data(), 1, &stored));
        }
    }

    // engine.rs:363-396
    void query_by_image_hash_from_image(const IndexedImage &indexed_image) {
        if (!indexed_image.visual_hash) return;
        cached_search_results_.reset();
        std::vector<IndexedImage> out;
        for (uint32_t k = RESULT_LIMIT;; k = std::min<uint32_t>(2 * k, PB_MAX_K)) {  // the JOIN runs before the LIMIT: over-fetch while orphans use up slots
            std::vector<int64_t> ids(k);
            std::vector<float> dist(k);
            uint32_t count = 0;
            check(pb_sharded_search(idx_.get(), indexed_image.visual_hash->data(), 1, k, max_distance_from_query, ids.data(), dist.data(), &count));
            out.clear();
            std::lock_guard<std::mutex> lk(mu_);
            for (uint32_t i = 0; i < count && out.size() < RESULT_LIMIT; ++i) {
                auto it = images_.find(ids[i]);
                if (it == images_.end()) continue;
                IndexedImage r = it->second;
                r.distance_from_query = (double)dist[i];
                out.push_back(std::move(r));
            }
            if (out.size() == RESULT_LIMIT || count < k || k == PB_MAX_K) break;
        }
        cached_search_results_ = std::move(out);
    }
    // engine.rs:352-361 (the query image is hashed on shard 0's GPU; while indexing runs it waits, inside the hashing helper, for
    // that GPU's embed thread to finish the batch it is storing: Embedder::exclusive)
    bool query_by_image_hash_from_file(const std::string &path, const PHasher *hasher = nullptr, const Decoder &decode = decode_pnm) {
        cached_search_results_.reset();
        std::optional<IndexedImage> img = indexed_image_from_file_path(path, model(0), hasher, decode);
        if (!img) return false;
        query_by_image_hash_from_image(*img);
        return true;
    }
    std::optional<std::vector<IndexedImage>> get_query_results() const { return cached_search_results_; }

    uint64_t get_num_indexed_images(std::vector<uint64_t> *per_shard = nullptr) const {
        uint64_t n = 0;
        std::vector<uint64_t> per(shard_count());
        check(pb_sharded_size(idx_.get(), &n, per.data()));
        if (per_shard) *per_shard = per;
        return n;
    }
    pb_sharded *raw() const { return idx_.get(); }

  private:
    size_t shard_count() const {
        int n = 0, r = 0;
        uint64_t x = 0;
        check(pb_sharded_info(idx_.get(), &n, &r, &x));
        return (size_t)n;
    }
    // a finished batch of GPU g: ids under the lock (ascending per shard because a GPU's batches are numbered in the order it
    // finishes them), then the device-to-device insert outside it.  Two things can refuse that insert without anything being
    // wrong: the shard is full while others have room (shards hold ceil(capacity / n) rows each and the GPUs take batches from
    // one queue, so a fast GPU fills its shard first), or an id smaller than the batch's reached this shard first
    // (insert_image_from_memory during indexing places its row on the least-full shard).  Those rows then go through
    // pb_sharded_append with the host copies of their hashes -- any order, spilling to the least-full shard.  If the rows cannot
    // be stored at all, the records are taken back out of the maps: a later start_indexing must not skip their paths.
    void store_batch(size_t g, std::vector<IndexedImage> &batch, const uint8_t *d_hashes) {
        std::vector<int64_t> ids;
        std::vector<uint32_t> rows;  // positions of the batch's NEW images (UNIQUE(path): a known path is ignored, engine.rs:40,231)
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < batch.size(); ++i) {
                auto known = by_path_.find(batch[i].path);
                if (known != by_path_.end()) {
                    batch[i].id = known->second;
                    continue;
                }
                batch[i].id = ++last_rowid_;
                by_path_[batch[i].path] = batch[i].id;
                images_[batch[i].id] = batch[i];
                ids.push_back(batch[i].id);
                rows.push_back((uint32_t)i);
            }
        }
        if (ids.empty()) return;
        try {
            // the new rows are runs of the device buffer (one run when the whole batch is new: the common case)
            size_t a = 0;
            while (a < rows.size()) {
                size_t b = a + 1;
                while (b < rows.size() && rows[b] == rows[b - 1] + 1) ++b;
                // (d_hashes null: the stage could not line the records up with the device buffer -- a staged batch that held an
                // undecodable file -- and the host copies go in)
                const int rc = d_hashes ? pb_sharded_append_device(idx_.get(), (int)g, ids.data() + a, d_hashes + (size_t)rows[a] * dim_, b - a) : PB_ERR_INVALID;
                if (rc == PB_ERR_CAPACITY || rc == PB_ERR_INVALID) {  // this shard cannot take the run: the host path places it
                    std::vector<uint8_t> host((b - a) * (size_t)dim_);
                    for (size_t i = a; i < b; ++i) {
                        const std::vector<uint8_t> &vh = *batch[rows[i]].visual_hash;
                        if (vh.size() != dim_) throw Error(PB_ERR_INVALID, "visual_hash length != index dim");
                        std::copy(vh.begin(), vh.end(), host.begin() + (i - a) * (size_t)dim_);
                    }
                    uint64_t stored = 0;
                    check(pb_sharded_append(idx_.get(), ids.data() + a, host.data(), b - a, &stored));
                } else {
                    check(rc);
                }
                a = b;
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < rows.size(); ++i) {  // rows that did reach a shard keep their records
                int found = 0;
                if (pb_sharded_contains(idx_.get(), ids[i], &found) == PB_OK && found) continue;
                images_.erase(ids[i]);
                by_path_.erase(batch[rows[i]].path);
            }
            throw;
        }
    }

    struct Del {
        void operator()(pb_sharded *p) const { pb_sharded_destroy(p); }
    };
    std::unique_ptr<pb_sharded, Del> idx_;
    std::vector<std::unique_ptr<Embedder>> models_;
    std::unique_ptr<Crawler> crawler_;
    uint32_t dim_;
    mutable std::mutex mu_;
    int64_t last_rowid_ = 0;
    std::map<int64_t, IndexedImage> images_;
    std::unordered_map<std::string, int64_t> by_path_;
    std::optional<std::vector<IndexedImage>> cached_search_results_;
};

}  // namespace pixelbox
