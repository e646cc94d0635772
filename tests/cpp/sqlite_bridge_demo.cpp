// sqlite_bridge_demo.cpp -- drives include/pixelbox_sqlite.hpp the way PixelBox's Engine is used:
// open an existing database (bulk load of semantic_hashes), insert new images (write-through), query.
//   usage: sqlite_bridge_demo <db> <dim> <new_hashes.u8> <n_new> <out.txt>
#include <cstdio>
#include <fstream>
#include <iterator>

#include "pixelbox_sqlite.hpp"

int main(int argc, char **argv) {
    if (argc != 6) return 2;
    try {
        const uint32_t dim = (uint32_t)atoi(argv[2]);
        std::ifstream f(argv[3], std::ios::binary);
        const std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), {});
        const int n_new = atoi(argv[4]);
        FILE *out = fopen(argv[5], "w");
        pixelbox::PersistentEngine engine(argv[1], dim, 100000);
        fprintf(out, "loaded %llu skipped %llu orphans %llu\n", (unsigned long long)engine.get_num_indexed_images(),
                (unsigned long long)engine.num_skipped_hashes(), (unsigned long long)engine.num_orphan_hashes());
        std::vector<pixelbox::IndexedImage> fresh;
        for (int i = 0; i < n_new; ++i) {
            pixelbox::IndexedImage rec;
            rec.filename = "new" + std::to_string(i) + ".png";
            rec.path = "/new/" + rec.filename;
            rec.resolution = {128, 128};
            rec.visual_hash = std::vector<uint8_t>(raw.begin() + (size_t)i * dim, raw.begin() + (size_t)(i + 1) * dim);
            const long long id = engine.insert_image_from_memory(rec);
            fprintf(out, "inserted %lld\n", id);
            fresh.push_back(rec);
        }
        engine.insert_image_from_memory(fresh[0]);  // known path: both INSERT OR IGNOREs are no-ops
        fprintf(out, "indexed %llu\n", (unsigned long long)engine.get_num_indexed_images());
        for (int qi = 0; qi < 2; ++qi) {
            engine.query_by_image_hash_from_image(fresh[qi]);
            const auto res = engine.get_query_results();
            fprintf(out, "query %d results %zu\n", qi, res->size());
            for (const auto &r : *res)
                fprintf(out, "%lld %.9g %s %zu\n", (long long)r.id, *r.distance_from_query, r.path.c_str(),
                        r.visual_hash ? r.visual_hash->size() : 0);
        }
        fclose(out);
    } catch (const pixelbox::Error &e) {
        fprintf(stderr, "pixelbox error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
