// ingest_staged.cpp -- the crawler's hot loop with DECODERS in it (crawler.rs:68-119 -> indexed_image.rs:47-91 -> efficientnet.rs:19-29 ->
// engine.rs:251-256), driven natively so that the host side is threads, not an interpreter: E embedders (an embed thread each) x D
// "decoder" threads per embedder.  A decoder's output is a w x h RGB8 image that it WRITES (here: a copy of one of 64 pool images, the
// bytes a real decoder's last pass would store) either
//   mode 0  into a buffer of its own, handed to the embed thread, which calls pb_embed_batch_images_device on batches of 512 (the
//           library packs them into pinned staging: one more pass through host memory), or
//   mode 1  straight into the embedder's staging slot (pb_embed_stage_acquire / _release), the embed thread closing and committing
//           batches (pb_embed_stage_close / _commit): the decoder's store is the only host pass.
// Every batch's hashes are appended device-to-device to one index (ids in insertion order).  Prints images/s.
// build: g++ -O2 -std=c++17 -pthread -I include profiles/micro/ingest_staged.cpp -L pixelbox_amd -lpixelbox_hip -Wl,-rpath,...
// argv: weights.pbxw n_images w h embedders decoders_per_embedder mode
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <mutex>
#include <thread>
#include <vector>

#include "pixelbox_hip.h"

#define CK(x)                                                                  \
    do {                                                                       \
        const int rc_ = (x);                                                   \
        if (rc_ < 0) {                                                         \
            std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, pb_last_error()); \
            std::exit(1);                                                      \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 8) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const long n_images = std::atol(argv[2]);
    const uint32_t w = (uint32_t)std::atoi(argv[3]), h = (uint32_t)std::atoi(argv[4]);
    const int E = std::atoi(argv[5]), D = std::atoi(argv[6]), mode = std::atoi(argv[7]);
    const uint32_t NB = 512;
    const size_t per = (size_t)w * h * 3;
    std::vector<std::vector<uint8_t>> pool(64, std::vector<uint8_t>(per));
    uint64_t z = 0x9E3779B97F4A7C15ull;
    for (auto &p : pool)
        for (size_t i = 0; i < per; i += 8) {
            z += 0x9E3779B97F4A7C15ull;
            uint64_t x = z;
            x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
            x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
            x ^= x >> 31;
            std::memcpy(&p[i], &x, std::min<size_t>(8, per - i));
        }
    std::vector<pb_embedder *> emb(E);
    for (int e = 0; e < E; ++e) CK(pb_embed_create(&emb[e], 0, blob.data(), blob.size(), NB));
    pb_index *index = nullptr;
    CK(pb_index_create(&index, 0, 256, (uint64_t)n_images + 16));
    std::mutex id_mu;
    int64_t next_id = 1;
    auto append = [&](const uint8_t *d_hashes, uint32_t n) {
        std::lock_guard<std::mutex> lk(id_mu);
        std::vector<int64_t> ids(n);
        for (uint32_t i = 0; i < n; ++i) ids[i] = next_id++;
        CK(pb_index_append_device(index, ids.data(), d_hashes, n));
    };
    // warm-up: the kernel forms of this batch size, the staging blocks
    for (int e = 0; e < E; ++e) {
        std::vector<const uint8_t *> ptrs(NB);
        std::vector<uint32_t> ws(NB, w), hs(NB, h);
        for (uint32_t i = 0; i < NB; ++i) ptrs[i] = pool[i % 64].data();
        const uint8_t *d = nullptr;
        CK(pb_embed_batch_images_device(emb[e], ptrs.data(), ws.data(), hs.data(), NB, nullptr, &d));
    }
    struct Q {
        std::mutex mu;
        std::condition_variable cv, cv_space;
        std::deque<std::vector<uint8_t>> items;
        int producers = 0;
    };
    struct S {
        std::mutex mu;
        std::condition_variable cv, cv_room;
        long pending = 0;
        bool full = false;
        int producers = 0;
    };
    std::vector<Q> qs(E);  // (outlive the threads below)
    std::vector<S> ss(E);
    if (mode == 1) {  // slots that hold a whole batch of this image size; one staged batch per embedder so that its size's kernel forms are chosen
        for (int e = 0; e < E; ++e) {
            CK(pb_embed_set_option(emb[e], PB_OPT_EMBED_STAGE_BYTES, (int64_t)std::max<size_t>((per + 15) / 16 * 16 * NB, 48u << 20)));
            for (int rep = 0; rep < 2; ++rep) {
                for (uint32_t i = 0; i < NB; ++i) {
                    uint8_t *px = nullptr;
                    uint64_t ticket = 0;
                    if (pb_embed_stage_acquire(emb[e], w, h, &px, &ticket) != PB_OK) break;
                    std::memcpy(px, pool[i % 64].data(), per);
                    CK(pb_embed_stage_release(emb[e], ticket));
                }
                uint32_t n = 0, gen = 0;
                const uint8_t *d = nullptr;
                CK(pb_embed_stage_close(emb[e], &n, &gen, nullptr, nullptr, nullptr));
                CK(pb_embed_stage_commit(emb[e], nullptr, &d));
            }
        }
    }
    std::atomic<long> next_image{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> threads;
    if (mode == 0) {
        for (int e = 0; e < E; ++e) qs[e].producers = D;
        for (int e = 0; e < E; ++e) {
            for (int dth = 0; dth < D; ++dth)
                threads.emplace_back([&, e] {
                    for (;;) {
                        const long i = next_image.fetch_add(1);
                        if (i >= n_images) break;
                        std::vector<uint8_t> img(per);
                        std::memcpy(img.data(), pool[i % 64].data(), per);  // the decoder's output
                        std::unique_lock<std::mutex> lk(qs[e].mu);
                        qs[e].cv_space.wait(lk, [&] { return qs[e].items.size() < 2 * NB; });
                        qs[e].items.push_back(std::move(img));
                        qs[e].cv.notify_one();
                    }
                    std::lock_guard<std::mutex> lk(qs[e].mu);
                    --qs[e].producers;
                    qs[e].cv.notify_all();
                });
            threads.emplace_back([&, e] {
                std::vector<std::vector<uint8_t>> batch;
                std::vector<const uint8_t *> ptrs(NB);
                std::vector<uint32_t> ws(NB, w), hs(NB, h);
                for (;;) {
                    batch.clear();
                    {
                        std::unique_lock<std::mutex> lk(qs[e].mu);
                        qs[e].cv.wait(lk, [&] { return qs[e].items.size() >= NB || qs[e].producers == 0; });
                        while (!qs[e].items.empty() && batch.size() < NB) {
                            batch.push_back(std::move(qs[e].items.front()));
                            qs[e].items.pop_front();
                        }
                        qs[e].cv_space.notify_all();
                        if (batch.empty()) break;
                    }
                    for (size_t i = 0; i < batch.size(); ++i) ptrs[i] = batch[i].data();
                    const uint8_t *d = nullptr;
                    CK(pb_embed_batch_images_device(emb[e], ptrs.data(), ws.data(), hs.data(), (uint32_t)batch.size(), nullptr, &d));
                    append(d, (uint32_t)batch.size());
                }
            });
        }
    } else {
        for (int e = 0; e < E; ++e) ss[e].producers = D;
        for (int e = 0; e < E; ++e) {
            for (int dth = 0; dth < D; ++dth)
                threads.emplace_back([&, e] {
                    for (;;) {
                        const long i = next_image.fetch_add(1);
                        if (i >= n_images) break;
                        uint8_t *px = nullptr;
                        uint64_t ticket = 0;
                        for (;;) {
                            const int rc = pb_embed_stage_acquire(emb[e], w, h, &px, &ticket);
                            if (rc == PB_OK) break;
                            CK(rc);
                            std::unique_lock<std::mutex> lk(ss[e].mu);  // PB_STAGE_FULL: the embed thread closes the batch
                            ss[e].full = true;
                            ss[e].cv.notify_all();
                            ss[e].cv_room.wait_for(lk, std::chrono::microseconds(200));
                        }
                        std::memcpy(px, pool[i % 64].data(), per);  // the decoder's output, written where the transfer starts from
                        CK(pb_embed_stage_release(emb[e], ticket));
                        std::lock_guard<std::mutex> lk(ss[e].mu);
                        if (++ss[e].pending >= (long)NB) ss[e].cv.notify_all();
                    }
                    std::lock_guard<std::mutex> lk(ss[e].mu);
                    --ss[e].producers;
                    ss[e].cv.notify_all();
                });
            threads.emplace_back([&, e] {
                for (;;) {
                    {
                        std::unique_lock<std::mutex> lk(ss[e].mu);
                        ss[e].cv.wait(lk, [&] { return ss[e].pending >= (long)NB || ss[e].full || ss[e].producers == 0; });
                        if (ss[e].pending == 0 && ss[e].producers == 0) break;
                        if (ss[e].pending == 0) {
                            ss[e].full = false;
                            continue;
                        }
                    }
                    uint32_t n = 0, gen = 0;
                    CK(pb_embed_stage_close(emb[e], &n, &gen, nullptr, nullptr, nullptr));
                    {
                        std::lock_guard<std::mutex> lk(ss[e].mu);
                        ss[e].pending -= n;
                        ss[e].full = false;
                        ss[e].cv_room.notify_all();
                    }
                    if (n == 0) continue;
                    const uint8_t *d = nullptr;
                    CK(pb_embed_stage_commit(emb[e], nullptr, &d));
                    append(d, n);
                    std::lock_guard<std::mutex> lk(ss[e].mu);
                    ss[e].cv_room.notify_all();
                }
            });
        }
    }
    for (auto &t : threads) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t rows = 0;
    CK(pb_index_size(index, &rows));
    std::printf("{\"mode\": %d, \"images\": %ld, \"stored\": %llu, \"seconds\": %.4f, \"images_per_s\": %.1f, \"embedders\": %d, \"decoders_per_embedder\": %d}\n", mode,
                n_images, (unsigned long long)rows, dt, n_images / dt, E, D);
    for (auto *e : emb) pb_embed_destroy(e);
    pb_index_destroy(index);
    return rows == (uint64_t)n_images ? 0 : 3;
}
