// host_mirror_demo.cpp -- exercises the C++ host mirror (include/pixelbox_host.hpp) the way the reference's
// own code would: mlhash every image, Engine::insert_image_from_memory, then
// Engine::query_by_image_hash_from_image + get_query_results.  Driven by tests/test_host_mirror_gpu.py.
//   usage: host_mirror_demo <weights.pbxw> <images.u8> <n_images> <out.txt>
#include <cstdio>
#include <fstream>
#include <iterator>

#include "pixelbox_host.hpp"

static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}

int main(int argc, char **argv) {
    if (argc != 5) return 2;
    try {
        const auto blob = slurp(argv[1]);
        const auto raw = slurp(argv[2]);
        const int n = atoi(argv[3]);
        pixelbox::Embedder model(blob.data(), blob.size(), 16);
        const size_t per = (size_t)model.width() * model.height() * 3;
        pixelbox::Engine engine(model.dim(), 1000);
        std::vector<pixelbox::IndexedImage> all;
        for (int i = 0; i < n; ++i) {
            pixelbox::RgbImage img{model.width(), model.height(), {raw.begin() + i * per, raw.begin() + (i + 1) * per}};
            pixelbox::IndexedImage rec;
            rec.filename = "img" + std::to_string(i) + ".png";
            rec.path = "/synthetic/" + rec.filename;
            rec.resolution = {model.width(), model.height()};
            rec.visual_hash = pixelbox::image_hashes::mlhash(model, img);
            all.push_back(rec);
            engine.insert_image_from_memory(rec);
        }
        engine.insert_image_from_memory(all[0]);  // re-index of a known path: ignored (UNIQUE(path), OR IGNORE)
        FILE *out = fopen(argv[4], "w");
        fprintf(out, "indexed %llu\n", (unsigned long long)engine.get_num_indexed_images());
        for (int qi : {0, n / 2}) {
            engine.query_by_image_hash_from_image(all[qi]);
            const auto res = engine.get_query_results();
            fprintf(out, "query %d results %zu\n", qi, res ? res->size() : 0);
            for (const auto &r : *res) fprintf(out, "%lld %.9g %s\n", (long long)r.id, *r.distance_from_query, r.path.c_str());
        }
        for (const auto &rec : all) {
            for (uint8_t b : *rec.visual_hash) fprintf(out, "%02x", b);
            fprintf(out, "\n");
        }
        fclose(out);
    } catch (const pixelbox::Error &e) {
        fprintf(stderr, "pixelbox error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
