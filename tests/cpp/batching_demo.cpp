// batching_demo.cpp -- N caller threads use the blocking mlhash() of pixelbox::BatchingEmbedder concurrently, the way
// PixelBox's crawler workers call image_hashes::mlhash; checks every hash against a plain batched run.
//   usage: batching_demo <weights.pbxw> <images.u8> <n_images> <n_threads>
#include <atomic>
#include <cstdio>
#include <fstream>
#include <iterator>

#include "pixelbox_batching.hpp"

static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}

int main(int argc, char **argv) {
    if (argc != 5) return 2;
    try {
        const auto blob = slurp(argv[1]);
        const auto raw = slurp(argv[2]);
        const int n = atoi(argv[3]), n_threads = atoi(argv[4]);
        pixelbox::BatchingEmbedder be(blob.data(), blob.size(), 64, 0, 500);
        const size_t per = (size_t)be.width() * be.height() * 3;
        std::vector<pixelbox::RgbImage> imgs(n);
        for (int i = 0; i < n; ++i) imgs[i] = {be.width(), be.height(), {raw.begin() + i * per, raw.begin() + (i + 1) * per}};
        // reference: one explicit batch through a second handle
        pixelbox::Embedder plain(blob.data(), blob.size(), 64);
        const auto want = pixelbox::image_hashes::mlhash_batch(plain, imgs);
        std::vector<std::vector<uint8_t>> got(n);
        std::atomic<int> next{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&] {
                for (int i; (i = next.fetch_add(1)) < n;) got[i] = be.mlhash(imgs[i]);
            });
        for (auto &x : th) x.join();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += got[i] != want[i];
        printf("images %d threads %d mismatches %d batches %llu mean_batch %.2f ms %.2f\n", n, n_threads, bad,
               (unsigned long long)be.batches_run(), (double)be.images_run() / (double)be.batches_run(), ms);
        return bad ? 1 : 0;
    } catch (const pixelbox::Error &e) {
        fprintf(stderr, "pixelbox error %d: %s\n", e.code, e.what());
        return 1;
    }
}
