// crawler_demo.cpp -- the reference's indexing flow written against the C++ host mirror: Crawler::start_indexing ->
// drain the channel into Engine::insert_image_from_memory (engine.rs:177-205), then Engine::query_by_image_hash_from_file
// (engine.rs:352-361).  argv: weights.pbxw folder query.pnm out.txt workers [staged: 1 = the decoders write into the embedder's
// staging slots (pb_embed_stage_*), 0 / absent = decoders with buffers of their own]
#include <cstdio>
#include <fstream>

#include "pixelbox_crawler.hpp"

static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static std::string hex(const std::vector<uint8_t> &v) {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (uint8_t b : v) {
        s.push_back(d[b >> 4]);
        s.push_back(d[b & 15]);
    }
    return s;
}

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    try {
        const std::vector<uint8_t> blob = slurp(argv[1]);
        pixelbox::Embedder model(blob.data(), blob.size(), 64);
        pixelbox::PHasher hasher;
        pixelbox::Engine engine(model.dim(), 4096);
        const bool staged = argc > 6 && std::atoi(argv[6]) != 0;
        pixelbox::Crawler crawler = staged ? pixelbox::Crawler(std::vector<const pixelbox::Embedder *>{&model}, &hasher,
                                                               pixelbox::StagedDecoder(pixelbox::decode_pnm_into), 64)
                                           : pixelbox::Crawler(model, &hasher, pixelbox::decode_pnm, 64);
        crawler.start_indexing({argv[2]}, (size_t)std::atoi(argv[5]));
        std::vector<pixelbox::IndexedImage> got;
        pixelbox::IndexedImage img;
        while (crawler.recv(img)) {  // engine.rs:189-200
            got.push_back(img);
            engine.insert_image_from_memory(img);
        }
        const pixelbox::Crawler::Stats st = crawler.stats();
        FILE *out = std::fopen(argv[4], "w");
        std::fprintf(out, "seen %llu matched %llu decoded %llu skipped %llu indexed %llu batches %llu largest %llu\n",
                     (unsigned long long)st.files_seen, (unsigned long long)st.files_matched, (unsigned long long)st.decoded,
                     (unsigned long long)st.skipped, (unsigned long long)engine.get_num_indexed_images(), (unsigned long long)st.batches,
                     (unsigned long long)st.largest_batch);
        for (const pixelbox::IndexedImage &g : got)
            std::fprintf(out, "img %s %u %u %s %s\n", g.filename.c_str(), g.resolution.first, g.resolution.second, hex(*g.visual_hash).c_str(),
                         hex(*g.phash).c_str());
        const bool ok = engine.query_by_image_hash_from_file(argv[3], model, &hasher);
        std::fprintf(out, "query %d hash_ms %.3f search_ms %.3f\n", (int)ok, engine.last_hash_ms, engine.last_search_ms);
        if (ok) {
            const auto res = engine.get_query_results();  // Option<Vec<IndexedImage>>, cloned like engine.rs:398-400
            for (const pixelbox::IndexedImage &r : *res) std::fprintf(out, "res %s %.9g\n", r.filename.c_str(), *r.distance_from_query);
        }
        const bool bad = engine.query_by_image_hash_from_file("/nonexistent/file.pnm", model, &hasher);
        std::fprintf(out, "missing %d\n", (int)bad);
        std::fclose(out);
    } catch (const pixelbox::Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
