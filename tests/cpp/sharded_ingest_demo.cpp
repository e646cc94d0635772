// sharded_ingest_demo.cpp -- BASELINE configs[4] in the product form, on the C++ host mirror: ShardedEngine (one process, a
// shard + an embedder per entry of the device list) runs the reference's indexing flow (engine.rs:177-205: crawler ->
// embed -> insert) with ONE EMBED THREAD PER SHARD storing its batches device-to-device (pb_sharded_append_device), then
// answers Engine::query_by_image_hash_from_file (engine.rs:352-361) over all shards.  Also: a consumer that stops
// receiving and drops the stage must not hang (the reference's workers leave on a disconnected channel).
// argv: weights.pbxw folder query.pnm out.txt workers device_list(e.g. 0,0)
#include <atomic>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <thread>

#include "pixelbox_sharded.hpp"

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
    if (argc < 7) return 2;
    try {
        const std::vector<uint8_t> blob = slurp(argv[1]);
        std::vector<int> devices;
        {
            std::stringstream ss(argv[6]);
            std::string tok;
            while (std::getline(ss, tok, ',')) devices.push_back(std::atoi(tok.c_str()));
        }
        FILE *out = std::fopen(argv[4], "w");
        {
            // a consumer that takes ONE record and walks away: the destructor must cancel and join (no hang)
            pixelbox::ShardedEngine tmp(32, 4096, devices, blob.data(), blob.size(), 16);
            tmp.start_indexing({argv[2]}, (size_t)std::atoi(argv[5]), nullptr, pixelbox::decode_pnm, 16);
            pixelbox::IndexedImage one;
            const bool got_one = tmp.recv_indexed(one);
            std::fprintf(out, "dropped_early %d\n", (int)got_one);
        }
        pixelbox::ShardedEngine engine(32, 4096, devices, blob.data(), blob.size(), 16);
        pixelbox::PHasher hasher(devices[0]);
        engine.start_indexing({argv[2]}, (size_t)std::atoi(argv[5]), &hasher, pixelbox::decode_pnm, 16);  // batches of <= 16: several per shard
        // the UI thread of the reference queries while the indexing thread runs (engine.rs:352-361 beside :177-205): hash the
        // query image on shard 0's embedder over and over while that embedder's thread embeds and stores batches (the
        // embedder's device buffer must not be overwritten between a batch's forward pass and its insert)
        std::atomic<bool> indexing_done{false};
        std::atomic<int> queries_meanwhile{0};
        std::thread ui([&] {
            while (!indexing_done.load()) {
                if (engine.query_by_image_hash_from_file(argv[3], nullptr)) queries_meanwhile.fetch_add(1);
            }
        });
        std::vector<pixelbox::IndexedImage> got;
        pixelbox::IndexedImage img;
        while (engine.recv_indexed(img)) got.push_back(img);  // already stored: the records only pass through
        indexing_done.store(true);
        ui.join();
        std::fprintf(out, "queries_while_indexing %d\n", queries_meanwhile.load() > 0 ? 1 : 0);
        {   // every record's hash is what the table stores under its id: the row answers its own hash at the self-distance
            int ok = 0;
            for (const pixelbox::IndexedImage &g : got) {
                int64_t ids[64];
                float dist[64];
                uint32_t cnt = 0;
                pixelbox::check(pb_sharded_search(engine.raw(), g.visual_hash->data(), 1, 64, 1e3, ids, dist, &cnt));
                for (uint32_t i = 0; i < cnt; ++i)
                    if (ids[i] == g.id && dist[i] <= 1e-6f) {
                        ++ok;
                        break;
                    }
            }
            std::fprintf(out, "stored_rows_match_records %d of %zu\n", ok, got.size());
        }
        const pixelbox::Crawler::Stats st = engine.indexing_stats();
        std::vector<uint64_t> per;
        const uint64_t total = engine.get_num_indexed_images(&per);
        std::fprintf(out, "seen %llu matched %llu decoded %llu skipped %llu indexed %llu batches %llu largest %llu shards %zu", (unsigned long long)st.files_seen,
                     (unsigned long long)st.files_matched, (unsigned long long)st.decoded, (unsigned long long)st.skipped, (unsigned long long)total,
                     (unsigned long long)st.batches, (unsigned long long)st.largest_batch, per.size());
        for (uint64_t p : per) std::fprintf(out, " %llu", (unsigned long long)p);
        std::fprintf(out, "\n");
        for (const pixelbox::IndexedImage &g : got)
            std::fprintf(out, "img %s %lld %s\n", g.filename.c_str(), (long long)g.id, hex(*g.visual_hash).c_str());
        // indexing the same folder again stores nothing new (UNIQUE(path) + INSERT OR IGNORE, engine.rs:40,231)
        engine.start_indexing({argv[2]}, 2, nullptr, pixelbox::decode_pnm, 16);
        const uint64_t again = engine.wait_for_indexing();
        std::fprintf(out, "reindexed %llu total %llu\n", (unsigned long long)again, (unsigned long long)engine.get_num_indexed_images());
        const bool ok = engine.query_by_image_hash_from_file(argv[3], &hasher);
        std::fprintf(out, "query %d\n", (int)ok);
        if (ok) {
            const auto res = engine.get_query_results();  // Option<Vec<IndexedImage>>, cloned like engine.rs:398-400 (kept alive for the loop)
            for (const pixelbox::IndexedImage &r : *res) std::fprintf(out, "res %s %lld %.9g\n", r.filename.c_str(), (long long)r.id, *r.distance_from_query);
        }
        {   // a table with hardly any slack: shards hold ceil(80 / n) rows and the GPUs take batches from one queue, so a shard
            // can run full while the table has room -- those batches spill through the host path, nothing is lost
            pixelbox::ShardedEngine tight(32, 80, devices, blob.data(), blob.size(), 16);
            tight.start_indexing({argv[2]}, (size_t)std::atoi(argv[5]), nullptr, pixelbox::decode_pnm, 16);
            const uint64_t n_tight = tight.wait_for_indexing();
            std::fprintf(out, "tight_capacity indexed %llu stored %llu\n", (unsigned long long)n_tight, (unsigned long long)tight.get_num_indexed_images());
        }
        std::fclose(out);
    } catch (const pixelbox::Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
