// pixelbox_crawler.hpp -- the crawler -> embed stage re-built for GPUs, mirroring
// src/crawler.rs:10-123 + the hashing half of IndexedImage::from_memory (src/indexed_image.rs:47-91).
//
// Reference: one glob thread walks "<folder>/**/*.*" and keeps the files whose extension is one of twelve (crawler.rs:7,
// :35-65); N worker threads (PARALLEL_FILE_PROCESSORS = 4, engine.rs:22) each read + decode + thumbnail + hash ONE image at
// a time -- `mlhash` is a batch-1 `MODEL.run` per image (efficientnet.rs:34) -- and push IndexedImages into a bounded(128)
// channel that the insert thread drains (crawler.rs:27-28, engine.rs:186-203).
//
// Here the workers only READ and DECODE (CPU codecs; the decoder is the host's callback, PNM built in) and hand the
// decoded pixels, at whatever size, to the EMBED THREADS -- one per pb_embedder, i.e. one per GPU -- each of which gathers
// up to `max_batch` (512) of them and runs `resize_to_fill(W, H, Triangle)` + the network for the whole batch on its GPU and
// `phash` per image (pb_phash_image); the finished IndexedImages go into the same bounded channel, which the caller drains
// with recv() exactly as engine.rs:189-200 drains its Receiver.  Same 12-extension allow-list, same "undecodable files are
// skipped" rule (crawler.rs:78), same back-pressure (a full channel stalls the producers).
//
// Multi-GPU ingest (BASELINE configs[4]): with a BatchSink the embed thread of GPU g hands every finished batch -- records
// AND a pointer to the batch's hashes in that GPU's memory -- to the sink before the records enter the channel; ShardedEngine
// (pixelbox_sharded.hpp) uses it to assign image ids and store the hashes on shard g device-to-device
// (pb_sharded_append_device), so that N GPUs embed and insert concurrently with no host hop for the hashes.
//
// Staged decoding (round 5): given a StagedDecoder (a decoder that writes where it is told: pixelbox_host.hpp) the workers write their
// pixels ONCE, into the pinned staging block of the embedder the image goes to (pb_embed_stage_acquire / _release), and the embed
// thread closes and commits the batch (pb_embed_stage_close / _commit) -- no packing pass, the transfer is one command per batch;
// phash runs on the same pinned pixels between close and commit.  Same records as the Decoder path (tests/cpp/crawler_demo.cpp).
//
// Shutdown: like the reference's detached workers, which stop when the Receiver is dropped (`TrySendError::Disconnected`,
// crawler.rs:97-101), every thread here leaves as soon as the stage is cancelled -- by the destructor, by cancel(), or by an
// embed thread that failed.  A failure (a pb_* call returning an error throws pixelbox::Error on the embed thread) is
// caught there, recorded, closes the channel and is reported by error(); nothing terminates the process.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <filesystem>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include "pixelbox_host.hpp"

namespace pixelbox {

class Crawler {
  public:
    static constexpr size_t MAX_PENDING_TX = 128;  // crawler.rs:8
    static inline const char *const SUPPORTED_IMAGE_EXTENSIONS[12] = {"png", "bmp", "jpg",  "jpeg", "jfif", "gif",
                                                                      "tiff", "pnm", "webp", "ico",  "tga",  "exr"};  // crawler.rs:7

    // records of one finished batch + its hashes in the memory of embedder `model_index`'s GPU (uint8[batch][dim], valid for
    // the duration of the call; row i = record i; NULL when the records do not line up with the device buffer -- a staged batch that
    // held an undecodable file -- and the sink must use the records' own visual_hash); may set IndexedImage::id.  Runs on that embedder's thread: sinks of different embedders
    // run concurrently.  Throwing pixelbox::Error stops the stage.
    using BatchSink = std::function<void(size_t model_index, std::vector<IndexedImage> &batch, const uint8_t *d_hashes)>;

    Crawler(const Embedder &model, const PHasher *hasher, Decoder decode = decode_pnm, uint32_t max_batch = 512)
        : models_{&model}, hasher_(hasher), decode_(std::move(decode)), max_batch_(std::min(max_batch, model.max_batch())) {}
    // one embed thread per model (one model per GPU); `hasher` (optional) must be usable from all of them
    Crawler(std::vector<const Embedder *> models, const PHasher *hasher, Decoder decode = decode_pnm, uint32_t max_batch = 512,
            BatchSink sink = nullptr)
        : models_(std::move(models)), hasher_(hasher), decode_(std::move(decode)), max_batch_(max_batch), sink_(std::move(sink)) {
        for (const Embedder *m : models_) max_batch_ = std::min(max_batch_, m->max_batch());  // a batch must fit every embedder's workspace
        if (models_.empty()) throw Error(PB_ERR_INVALID, "Crawler: no embedder");
    }
    // the same stage with decoders that write into the embedders' staging slots (see the header comment)
    Crawler(std::vector<const Embedder *> models, const PHasher *hasher, StagedDecoder decode, uint32_t max_batch = 512, BatchSink sink = nullptr)
        : models_(std::move(models)), hasher_(hasher), decode_(nullptr), max_batch_(max_batch), sink_(std::move(sink)), staged_decode_(std::move(decode)) {
        for (const Embedder *m : models_) max_batch_ = std::min(max_batch_, m->max_batch());
        if (models_.empty()) throw Error(PB_ERR_INVALID, "Crawler: no embedder");
        if (!staged_decode_) throw Error(PB_ERR_INVALID, "Crawler: null staged decoder");
    }
    ~Crawler() {
        cancel();
        join();
    }
    Crawler(const Crawler &) = delete;

    // stop everything: waiting threads wake up and leave, recv() returns false once the channel is empty
    void cancel() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            cancelled_ = true;
            cv_files_.notify_all();
            cv_decoded_.notify_all();
            cv_out_.notify_all();
            cv_space_.notify_all();
        }
        // an embed thread parked in pb_embed_stage_close (a decoder that holds a ticket and is slow, or stuck) gets its call back
        if (staged_decode_)
            for (const auto &mdl : models_) (void)pb_embed_stage_abort(mdl->raw());
    }
    // message of the first failure on an embed thread ("" = none); the stage is cancelled when one occurs
    std::string error() const {
        std::lock_guard<std::mutex> lk(mu_);
        return error_;
    }

    // crawler.rs:21-122.  Returns at once; the IndexedImages arrive through recv().
    void start_indexing(std::vector<std::string> folders, size_t num_workers) {
        cancel();
        join();
        files_.clear();
        decoded_.clear();
        out_.clear();
        cancelled_ = false;
        error_.clear();
        files_done_ = false;
        decoders_left_ = num_workers;
        embedders_left_ = models_.size();
        out_closed_ = false;
        stats_ = Stats{};
        staged_.assign(models_.size(), StagedState{});
        if (staged_decode_)  // nothing of an earlier run (of this Crawler or of another user of the embedder) joins this run's first batch
            for (const auto &mdl : models_) check(pb_embed_stage_abort(mdl->raw()));
        next_model_ = 0;
        threads_.emplace_back([this, folders] { glob_thread(folders); });
        for (size_t i = 0; i < num_workers; ++i) threads_.emplace_back([this] { staged_decode_ ? staged_worker() : decode_worker(); });
        for (size_t m = 0; m < models_.size(); ++m) threads_.emplace_back([this, m] { embed_thread(m); });
    }

    // the Receiver<IndexedImage>: blocks for the next image; false once every file has been processed
    bool recv(IndexedImage &out) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_out_.wait(lk, [&] { return !out_.empty() || out_closed_ || cancelled_; });
        if (out_.empty()) return false;
        out = std::move(out_.front());
        out_.pop_front();
        cv_space_.notify_all();
        return true;
    }

    struct Stats {
        uint64_t files_seen = 0, files_matched = 0, decoded = 0, skipped = 0, batches = 0, largest_batch = 0;
        uint64_t out_of_range = 0;  // images of batches the embedder refused with PB_ERR_RANGE (an activation outside the fixed-point domain of
                                    // the squeeze-excite sums: pixelbox_hip.h) -- dropped like undecodable files, the crawl goes on
    };
    Stats stats() const {
        std::lock_guard<std::mutex> lk(mu_);
        return stats_;
    }

    static bool is_image_file(const std::filesystem::path &p) {
        std::string ext = p.extension().string();  // ".png"
        if (ext.size() < 2) return false;          // "*.*": files without an extension are skipped (crawler.rs:57)
        ext = ext.substr(1);
        std::transform(ext.begin(), ext.end(), ext.begin(), [](unsigned char c) { return (char)std::tolower(c); });  // eq_ignore_ascii_case
        for (const char *e : SUPPORTED_IMAGE_EXTENSIONS)
            if (ext == e) return true;
        return false;
    }

  private:
    struct Decoded {
        std::string filename, path;
        RgbImage img;
    };

    void join() {
        for (std::thread &t : threads_)
            if (t.joinable()) t.join();
        threads_.clear();
    }

    void glob_thread(const std::vector<std::string> &folders) {  // crawler.rs:35-65
        namespace fs = std::filesystem;
        for (const std::string &dir : folders) {
            std::error_code ec;
            for (fs::recursive_directory_iterator it(dir, fs::directory_options::skip_permission_denied, ec), end; !ec && it != end; it.increment(ec)) {
                if (!it->is_regular_file(ec)) continue;
                std::lock_guard<std::mutex> lk(mu_);
                if (cancelled_) break;
                ++stats_.files_seen;
                if (!is_image_file(it->path())) continue;
                ++stats_.files_matched;
                files_.push_back(it->path().string());
                cv_files_.notify_one();
            }
        }
        std::lock_guard<std::mutex> lk(mu_);
        files_done_ = true;
        cv_files_.notify_all();
    }

    void decode_worker() {  // the CPU half of crawler.rs:68-119 / indexed_image.rs:35-56
        for (;;) {
            std::string path;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_files_.wait(lk, [&] { return !files_.empty() || files_done_ || cancelled_; });
                if (files_.empty() || cancelled_) break;
                path = std::move(files_.front());
                files_.pop_front();
            }
            std::optional<std::vector<uint8_t>> bytes = read_file(path);
            std::optional<RgbImage> img = bytes ? decode_(*bytes) : std::nullopt;
            std::unique_lock<std::mutex> lk(mu_);
            if (!img) {  // crawler.rs:78: `if let Ok(img) = ...` -- anything else is dropped silently
                ++stats_.skipped;
                continue;
            }
            ++stats_.decoded;
            // back-pressure: decoded pixels are the big items; hold at most two batches of them
            cv_space_.wait(lk, [&] { return decoded_.size() < 2 * (size_t)max_batch_ * models_.size() || cancelled_; });
            if (cancelled_) break;
            Decoded d;
            d.path = path;
            const size_t slash = path.find_last_of('/');
            d.filename = slash == std::string::npos ? path : path.substr(slash + 1);
            d.img = std::move(*img);
            decoded_.push_back(std::move(d));
            cv_decoded_.notify_one();
        }
        std::lock_guard<std::mutex> lk(mu_);
        if (--decoders_left_ == 0) cv_decoded_.notify_all();
    }

    // ---- staged decoding: a worker writes into the staging slot of embedder m (images go round robin over the embedders)
    struct StagedMeta {
        std::string filename, path;
        uint32_t w = 0, h = 0;
        bool valid = false;  // acquired AND decoded (an image whose decoder failed after it got room stays in the batch and is dropped afterwards)
    };
    struct StagedState {
        std::map<uint32_t, std::vector<StagedMeta>> meta;  // generation -> records by position in the batch
        // images acquired into batches that are not closed yet, PER GENERATION: a worker counts its image after the acquire has
        // returned, which may be after the embed thread closed that very batch -- a single counter clipped at zero then stays one too
        // high for ever and the loop never sees `nothing pending`
        std::map<uint32_t, size_t> pending_gen;
        uint32_t closed_through = 0;                        // generations <= this one are closed: late counts for them are dropped
        bool want_commit = false;                           // a worker found the open batch full
        size_t pending() const {
            size_t t = 0;
            for (const auto &kv : pending_gen) t += kv.second;
            return t;
        }
        void count(uint32_t gen) {
            if (gen > closed_through) ++pending_gen[gen];
        }
        void closed(uint32_t gen) {
            closed_through = std::max(closed_through, gen);
            pending_gen.erase(pending_gen.begin(), pending_gen.upper_bound(gen));
        }
    };
    // whatever happens between a successful acquire and the release (a throwing StagedDecoder -- it is a std::function --, bad_alloc
    // in the bookkeeping), the ticket goes back: an unreleased ticket parks the embed thread inside pb_embed_stage_close for ever
    struct TicketGuard {
        Crawler *c;
        size_t m = 0;
        uint64_t ticket = 0;
        bool armed = false;
        ~TicketGuard() {
            if (!armed) return;
            (void)pb_embed_stage_release(c->models_[m]->raw(), ticket);  // the record of this position stays `not valid`
            std::lock_guard<std::mutex> lk(c->mu_);
            c->cv_decoded_.notify_all();
        }
    };

    void staged_worker() {
        try {
            for (;;) {
                std::string path;
                size_t m;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_files_.wait(lk, [&] { return !files_.empty() || files_done_ || cancelled_; });
                    if (files_.empty() || cancelled_) break;
                    path = std::move(files_.front());
                    files_.pop_front();
                    m = next_model_++ % models_.size();
                }
                std::optional<std::vector<uint8_t>> bytes = read_file(path);
                uint64_t ticket = 0;
                bool acquired = false;
                uint32_t iw = 0, ih = 0;
                TicketGuard guard{this, m};
                const PixelAlloc alloc = [&](uint32_t w, uint32_t h) -> uint8_t * {
                    for (;;) {
                        uint8_t *px = nullptr;
                        if (acquired) return nullptr;  // one image per file: a decoder that asks twice gets no second room
                        const int rc = pb_embed_stage_acquire(models_[m]->raw(), w, h, &px, &ticket);
                        if (rc == PB_OK) {
                            acquired = true;
                            guard.ticket = ticket;
                            guard.armed = true;
                            iw = w;
                            ih = h;
                            std::lock_guard<std::mutex> lk(mu_);
                            staged_[m].count((uint32_t)(ticket >> 32));
                            return px;
                        }
                        if (rc != PB_STAGE_FULL) check(rc);
                        std::unique_lock<std::mutex> lk(mu_);  // the open batch is full: the embed thread closes it
                        if (cancelled_) return nullptr;
                        staged_[m].want_commit = true;
                        cv_decoded_.notify_all();
                        cv_space_.wait_for(lk, std::chrono::milliseconds(2));
                        if (cancelled_) return nullptr;
                    }
                };
                const bool ok = bytes && staged_decode_(*bytes, alloc);
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    if (ok && acquired) ++stats_.decoded;
                    else ++stats_.skipped;  // crawler.rs:78: anything that does not decode is dropped silently
                    if (acquired) {
                        std::vector<StagedMeta> &v = staged_[m].meta[(uint32_t)(ticket >> 32)];
                        const size_t pos = (size_t)(ticket & 0xFFFFu);
                        if (v.size() <= pos) v.resize(pos + 1);
                        StagedMeta &mt = v[pos];
                        mt.path = path;
                        const size_t slash = path.find_last_of('/');
                        mt.filename = slash == std::string::npos ? path : path.substr(slash + 1);
                        mt.w = iw;
                        mt.h = ih;
                        mt.valid = ok;
                    }
                }
                if (acquired) {
                    guard.armed = false;
                    check(pb_embed_stage_release(models_[m]->raw(), ticket));
                    std::lock_guard<std::mutex> lk(mu_);
                    cv_decoded_.notify_all();
                }
            }
        } catch (const std::exception &ex) {
            std::lock_guard<std::mutex> lk(mu_);
            if (error_.empty()) error_ = ex.what();
            cancelled_ = true;
            cv_files_.notify_all();
            cv_space_.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu_);
        if (--decoders_left_ == 0) {}
        cv_decoded_.notify_all();
    }

    void staged_embed_loop(size_t m) {
        const Embedder &model = *models_[m];
        std::vector<uint32_t> ws(model.max_batch()), hs(model.max_batch());
        std::vector<const uint8_t *> ptrs(model.max_batch());
        // Whatever THIS run leaves in the staging (a cancel with an open batch, a failure between close and commit) is discarded on
        // every way out -- break, return, exception; start_indexing does the same before the workers of the next run exist (not
        // here: this thread starts beside them, and a worker may already have its room in the first batch).
        struct StageReset {
            pb_embedder *e;
            ~StageReset() { (void)pb_embed_stage_abort(e); }
        } stage_reset{model.raw()};
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                // a batch goes when it is full, when a worker found it full, when the workers are done, or when something has waited 5 ms
                cv_decoded_.wait_for(lk, std::chrono::milliseconds(5), [&] {
                    return staged_[m].pending() >= max_batch_ || staged_[m].want_commit || decoders_left_ == 0 || cancelled_;
                });
                if (cancelled_) break;
                if (staged_[m].pending() == 0 && !staged_[m].want_commit) {
                    if (decoders_left_ == 0) break;
                    continue;
                }
            }
            std::unique_lock<std::recursive_mutex> own(model.exclusive());
            uint32_t n = 0, gen = 0;
            const int rc_close = pb_embed_stage_close(model.raw(), &n, &gen, ws.data(), hs.data(), ptrs.data());
            if (rc_close == PB_STAGE_ABORTED) continue;  // cancel() discarded the staging while this thread waited for the writers
            check(rc_close);
            std::vector<StagedMeta> metas;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (gen) staged_[m].closed(gen);
                staged_[m].want_commit = false;
                auto it = staged_[m].meta.find(gen);
                if (it != staged_[m].meta.end()) {
                    metas = std::move(it->second);
                    staged_[m].meta.erase(it);
                }
                if (n) {
                    ++stats_.batches;
                    stats_.largest_batch = std::max<uint64_t>(stats_.largest_batch, n);
                }
            }
            if (n == 0) continue;
            metas.resize(n);
            std::vector<std::vector<uint8_t>> phashes;
            if (hasher_) {  // on the pinned pixels, before the commit gives the block back
                std::vector<const uint8_t *> pp(ptrs.begin(), ptrs.begin() + n);
                std::vector<uint32_t> pw(ws.begin(), ws.begin() + n), ph(hs.begin(), hs.begin() + n);
                phashes = image_hashes::phash_batch(*hasher_, pp, pw, ph);
            }
            std::vector<uint8_t> hashes((size_t)n * model.dim());
            const uint8_t *d_hashes = nullptr;
            {
                const int rc_commit = pb_embed_stage_commit(model.raw(), hashes.data(), &d_hashes);
                if (rc_commit == PB_ERR_RANGE) {  // as in embed_loop: the batch is dropped and counted, the crawl goes on
                    std::lock_guard<std::mutex> lk(mu_);
                    stats_.out_of_range += n;
                    stats_.skipped += n;
                    cv_space_.notify_all();
                    continue;
                }
                check(rc_commit);
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                cv_space_.notify_all();  // workers that found the batch full: the other slot is open for them (or this one is free again)
            }
            std::vector<IndexedImage> recs;
            recs.reserve(n);
            // the sink stores the batch's hashes device-to-device by POSITION: with a sink every position must be a record, so a
            // batch that holds an undecodable image goes to the sink from the host copy of its valid rows' positions only if all are valid
            bool all_valid = true;
            for (uint32_t i = 0; i < n; ++i) all_valid = all_valid && metas[i].valid;
            for (uint32_t i = 0; i < n; ++i) {
                if (!metas[i].valid) continue;
                IndexedImage r;
                r.filename = std::move(metas[i].filename);
                r.path = std::move(metas[i].path);
                r.resolution = {ws[i], hs[i]};
                if (hasher_) r.phash = std::move(phashes[i]);
                r.visual_hash = std::vector<uint8_t>(hashes.begin() + (size_t)i * model.dim(), hashes.begin() + (size_t)(i + 1) * model.dim());
                recs.push_back(std::move(r));
            }
            if (sink_) sink_(m, recs, all_valid ? d_hashes : nullptr);
            own.unlock();
            for (IndexedImage &r : recs) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_space_.wait(lk, [&] { return out_.size() < MAX_PENDING_TX || cancelled_; });
                if (cancelled_) return;
                out_.push_back(std::move(r));
                cv_out_.notify_one();
            }
        }
    }

    void embed_thread(size_t m) {  // the GPU half: whatever is pending, up to max_batch images per forward pass
        try {
            if (staged_decode_) staged_embed_loop(m);
            else embed_loop(m);
        } catch (const std::exception &ex) {  // pixelbox::Error from a pb_* call, bad_alloc, ...: never out of a std::thread
            std::lock_guard<std::mutex> lk(mu_);
            if (error_.empty()) error_ = ex.what();
            cancelled_ = true;
            cv_files_.notify_all();
            cv_decoded_.notify_all();
            cv_space_.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu_);
        if (--embedders_left_ == 0 || cancelled_) {
            out_closed_ = true;
            cv_out_.notify_all();
        }
    }

    void embed_loop(size_t m) {
        const Embedder &model = *models_[m];
        for (;;) {
            std::vector<Decoded> batch;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_decoded_.wait(lk, [&] { return !decoded_.empty() || decoders_left_ == 0 || cancelled_; });
                if (decoded_.empty() || cancelled_) break;
                while (!decoded_.empty() && batch.size() < max_batch_) {
                    batch.push_back(std::move(decoded_.front()));
                    decoded_.pop_front();
                }
                ++stats_.batches;
                stats_.largest_batch = std::max<uint64_t>(stats_.largest_batch, batch.size());
                cv_space_.notify_all();
            }
            std::vector<RgbImage> imgs;
            imgs.reserve(batch.size());
            for (Decoded &d : batch) imgs.push_back(std::move(d.img));
            // resize_to_fill + network, one batch; the hashes stay on the GPU as well (for the sink).  The embedder is this
            // thread's from here to the end of the sink's device-to-device insert: d_hashes is the embedder's own output buffer,
            // and a query hashed on the same embedder meanwhile (ShardedEngine::query_by_image_hash_from_file) would overwrite it
            std::unique_lock<std::recursive_mutex> own(model.exclusive());
            const uint8_t *d_hashes = nullptr;
            std::vector<std::vector<uint8_t>> hashes;
            try {
                hashes = image_hashes::mlhash_batch(model, imgs, sink_ ? &d_hashes : nullptr);
            } catch (const Error &ex) {
                if (ex.code != PB_ERR_RANGE) throw;
                std::lock_guard<std::mutex> lk(mu_);  // one out-of-domain image fails its batch, not the crawl: the batch is dropped and counted
                stats_.out_of_range += batch.size();
                stats_.skipped += batch.size();
                continue;
            }
            std::vector<std::vector<uint8_t>> phashes;
            if (hasher_) phashes = image_hashes::phash_batch(*hasher_, imgs);  // the batch's phashes in one call too
            std::vector<IndexedImage> recs(batch.size());
            for (size_t i = 0; i < batch.size(); ++i) {
                IndexedImage &r = recs[i];
                r.filename = batch[i].filename;
                r.path = batch[i].path;
                r.resolution = {imgs[i].width, imgs[i].height};
                if (hasher_) r.phash = std::move(phashes[i]);
                r.visual_hash = hashes[i];
            }
            if (sink_) sink_(m, recs, d_hashes);
            own.unlock();
            for (IndexedImage &r : recs) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_space_.wait(lk, [&] { return out_.size() < MAX_PENDING_TX || cancelled_; });  // bounded(128): a slow consumer stalls the stage
                if (cancelled_) return;
                out_.push_back(std::move(r));
                cv_out_.notify_one();
            }
        }
    }

    std::vector<const Embedder *> models_;
    const PHasher *hasher_;
    Decoder decode_;
    uint32_t max_batch_;
    BatchSink sink_;
    StagedDecoder staged_decode_;
    std::vector<StagedState> staged_;
    size_t next_model_ = 0;
    mutable std::mutex mu_;
    std::condition_variable cv_files_, cv_decoded_, cv_out_, cv_space_;
    std::deque<std::string> files_;
    std::deque<Decoded> decoded_;
    std::deque<IndexedImage> out_;
    bool files_done_ = true, out_closed_ = true, cancelled_ = false;
    size_t decoders_left_ = 0, embedders_left_ = 0;
    std::string error_;
    Stats stats_;
    std::vector<std::thread> threads_;
};

}  // namespace pixelbox
