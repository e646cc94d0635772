// pixelbox_batching.hpp -- micro-batching front end for concurrent `mlhash` callers (SURVEY.md section 8b
// "Embed threading", section 8f rank 2).  The reference hashes one image per call from 4 crawler threads plus the
// UI thread (src/engine.rs:22,180,356; src/crawler.rs:68-78); a GPU wants batches.  BatchingEmbedder keeps the
// blocking `mlhash(img) -> Vec<u8>` signature for every caller and lets one worker thread collect whatever
// requests are pending (up to max_batch, waiting at most max_wait_us after the first) into one pb_embed_batch.
#pragma once
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "pixelbox_host.hpp"

namespace pixelbox {

class BatchingEmbedder {
  public:
    BatchingEmbedder(const void *weights_blob, size_t len, uint32_t max_batch = 512, int device = 0, uint32_t max_wait_us = 300)
        : model_(weights_blob, len, max_batch, device), max_batch_(max_batch), max_wait_us_(max_wait_us),
          per_((size_t)model_.width() * model_.height() * 3), worker_([this] { run(); }) {}
    ~BatchingEmbedder() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        worker_.join();
    }
    BatchingEmbedder(const BatchingEmbedder &) = delete;

    // image_hashes::mlhash for any number of concurrent callers (blocks until this image's batch has run)
    std::vector<uint8_t> mlhash(const RgbImage &img) {
        if (img.pixels.size() != per_) throw Error(PB_ERR_INVALID, "mlhash: image must be resized to the model input size first");
        Req r;
        r.pixels = img.pixels.data();
        r.out.resize(model_.dim());
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.push_back(&r);
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return r.done; });
        if (r.rc != PB_OK) throw Error(r.rc, r.err);
        return std::move(r.out);
    }
    uint64_t batches_run() const { return batches_; }
    uint64_t images_run() const { return images_; }
    uint32_t dim() const { return model_.dim(); }
    uint32_t width() const { return model_.width(); }
    uint32_t height() const { return model_.height(); }

  private:
    struct Req {
        const uint8_t *pixels = nullptr;
        std::vector<uint8_t> out;
        bool done = false;
        int rc = PB_OK;
        std::string err;
    };
    void run() {
        std::vector<Req *> batch;
        std::vector<uint8_t> in, out;
        for (;;) {
            batch.clear();
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) return;
                // give concurrent callers a moment to pile up, bounded by max_wait_us after the first request
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(max_wait_us_);
                while (q_.size() < max_batch_ && !stop_)
                    if (cv_.wait_until(lk, deadline) == std::cv_status::timeout) break;
                while (!q_.empty() && batch.size() < max_batch_) {
                    batch.push_back(q_.front());
                    q_.pop_front();
                }
            }
            const size_t n = batch.size(), d = model_.dim();
            in.resize(n * per_);
            out.resize(n * d);
            for (size_t i = 0; i < n; ++i) std::copy(batch[i]->pixels, batch[i]->pixels + per_, in.begin() + i * per_);
            const int rc = pb_embed_batch(model_.raw(), in.data(), (uint32_t)n, out.data(), nullptr);
            const std::string err = rc == PB_OK ? "" : pb_last_error();
            {
                std::lock_guard<std::mutex> lk(mu_);
                for (size_t i = 0; i < n; ++i) {
                    if (rc == PB_OK) std::copy(out.begin() + i * d, out.begin() + (i + 1) * d, batch[i]->out.begin());
                    batch[i]->rc = rc;
                    batch[i]->err = err;
                    batch[i]->done = true;
                }
                batches_ += 1;
                images_ += n;
            }
            done_cv_.notify_all();
        }
    }
    Embedder model_;
    const uint32_t max_batch_, max_wait_us_;
    const size_t per_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<Req *> q_;
    bool stop_ = false;
    uint64_t batches_ = 0, images_ = 0;
    std::thread worker_;  // last member: starts after everything else is initialised
};

}  // namespace pixelbox
