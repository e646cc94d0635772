// crawler_faults_demo.cpp -- the staged crawler under the failures ADVICE r5 named: a StagedDecoder that throws AFTER it got room in
// the batch (the ticket must go back, or the embed thread sits in pb_embed_stage_close for ever), a cancel() with a batch open, and
// a restart of the same Crawler on the same Embedder afterwards (the staging must come back clean: no `closed and never committed`
// state, no stale pixels in the next run's first batch).  Reference loop: crawler.rs:68-119.
// argv: weights.pbxw good_folder poisoned_folder out.txt
#include <chrono>
#include <cstdio>
#include <fstream>
#include <map>
#include <stdexcept>

#include "pixelbox_crawler.hpp"

static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

using Hashes = std::map<std::string, std::vector<uint8_t>>;

static Hashes drain(pixelbox::Crawler &c) {
    Hashes h;
    pixelbox::IndexedImage img;
    while (c.recv(img)) h[img.filename] = *img.visual_hash;
    return h;
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    try {
        const std::vector<uint8_t> blob = slurp(argv[1]);
        pixelbox::Embedder model(blob.data(), blob.size(), 16);
        FILE *out = std::fopen(argv[4], "w");
        // the answer: the plain decoder path on the good folder
        Hashes want;
        {
            pixelbox::Crawler plain(model, nullptr, pixelbox::decode_pnm, 16);
            plain.start_indexing({argv[2]}, 3);
            want = drain(plain);
        }
        std::fprintf(out, "plain %zu\n", want.size());
        // a decoder that throws once it has written into the room it was given (every 5th call)
        std::atomic<int> calls{0};
        std::atomic<bool> poison{true};
        pixelbox::StagedDecoder poisoned = [&](const std::vector<uint8_t> &b, const pixelbox::PixelAlloc &alloc) -> bool {
            const int c = ++calls;
            return pixelbox::decode_pnm_into(b, [&](uint32_t w, uint32_t h) -> uint8_t * {
                uint8_t *px = alloc(w, h);
                if (px && poison && c % 5 == 0) throw std::runtime_error("decoder blew up with a ticket in hand");
                return px;
            });
        };
        pixelbox::Crawler crawler(std::vector<const pixelbox::Embedder *>{&model}, nullptr, poisoned, 16);
        crawler.start_indexing({argv[3]}, 4);
        const Hashes partial = drain(crawler);  // must RETURN: the stage cancels itself and every thread leaves
        std::fprintf(out, "poisoned error=%d got=%zu\n", (int)!crawler.error().empty(), partial.size());
        // same Crawler object, same Embedder, decoder behaving now
        for (int round = 0; round < 2; ++round) {
            poison = false;
            crawler.start_indexing({argv[2]}, 4);
            const Hashes again = drain(crawler);
            std::fprintf(out, "restart%d error=%d same=%d n=%zu\n", round, (int)!crawler.error().empty(), (int)(again == want), again.size());
            // a cancel with a batch open: start, let some images in, cancel, drain
            crawler.start_indexing({argv[2]}, 2);
            std::this_thread::sleep_for(std::chrono::milliseconds(3));
            crawler.cancel();
            (void)drain(crawler);
            std::fprintf(out, "cancelled%d error=%d\n", round, (int)!crawler.error().empty());
        }
        // and the C ABI by itself: a closed batch that is never committed, then abort, then a normal cycle
        uint8_t *px = nullptr;
        uint64_t ticket = 0;
        pixelbox::check(pb_embed_stage_acquire(model.raw(), 8, 8, &px, &ticket));
        pixelbox::check(pb_embed_stage_release(model.raw(), ticket));
        uint32_t n = 0, gen = 0;
        pixelbox::check(pb_embed_stage_close(model.raw(), &n, &gen, nullptr, nullptr, nullptr));
        const int rc_second_close = pb_embed_stage_close(model.raw(), &n, &gen, nullptr, nullptr, nullptr);  // not committed: refused
        pixelbox::check(pb_embed_stage_abort(model.raw()));
        const int rc_after_abort = pb_embed_stage_close(model.raw(), &n, &gen, nullptr, nullptr, nullptr);   // nothing staged: fine, n = 0
        // an abort with a writer still inside: the slot is not handed out again before that writer releases
        pixelbox::check(pb_embed_stage_acquire(model.raw(), 8, 8, &px, &ticket));
        pixelbox::check(pb_embed_stage_abort(model.raw()));
        const int rc_release_late = pb_embed_stage_release(model.raw(), ticket);
        std::fprintf(out, "abi second_close=%d after_abort=%d n=%u release_late=%d\n", rc_second_close, rc_after_abort, n, rc_release_late);
        std::fclose(out);
    } catch (const pixelbox::Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
