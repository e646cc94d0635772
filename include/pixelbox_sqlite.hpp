// pixelbox_sqlite.hpp -- SQLite persistence bridge for the device-resident hash table (SURVEY.md section 8f,
// rank 1): `semantic_hashes` <-> the GPU matrix behind `pb_index`.
//
//   PersistentEngine::open(db)          Engine::open (src/engine.rs:117-145): SELECT image_id, hash FROM
//                                       semantic_hashes ORDER BY image_id  ->  pb_index_load
//   insert_image_from_memory(img)       engine.rs:224-259: INSERT OR IGNORE INTO images / semantic_hashes (SQLite
//                                       stays the system of record) + write-through pb_index_append
//   query_by_image_hash_from_image      engine.rs:363-396: pb_index_search, then the INNER JOIN against `images`
//                                       by id (SELECT {SELECT_FIELDS} FROM images WHERE id = ?, engine.rs:52-58)
//
// The reference links SQLite through rusqlite's bundled copy; here the system libsqlite3.so.0 is loaded with
// dlopen and the few entry points are declared by hand (the image ships the library without headers).
// Schema strings are the reference's (engine.rs:30-48), with IF NOT EXISTS.
#pragma once
#include <algorithm>
#include <dlfcn.h>

#include <cstring>

#include "pixelbox_host.hpp"

namespace pixelbox {

struct SqliteApi {
    void *lib = nullptr;
    int (*open_v2)(const char *, void **, int, const char *) = nullptr;
    int (*close_v2)(void *) = nullptr;
    int (*exec)(void *, const char *, int (*)(void *, int, char **, char **), void *, char **) = nullptr;
    int (*prepare_v2)(void *, const char *, int, void **, const char **) = nullptr;
    int (*step)(void *) = nullptr;
    int (*reset)(void *) = nullptr;
    int (*finalize)(void *) = nullptr;
    int (*bind_int64)(void *, int, long long) = nullptr;
    int (*bind_blob)(void *, int, const void *, int, void (*)(void *)) = nullptr;
    int (*bind_text)(void *, int, const char *, int, void (*)(void *)) = nullptr;
    long long (*column_int64)(void *, int) = nullptr;
    const void *(*column_blob)(void *, int) = nullptr;
    int (*column_bytes)(void *, int) = nullptr;
    const unsigned char *(*column_text)(void *, int) = nullptr;
    long long (*last_insert_rowid)(void *) = nullptr;
    int (*changes)(void *) = nullptr;
    const char *(*errmsg)(void *) = nullptr;
    void (*free_)(void *) = nullptr;

    static const SqliteApi &get() {
        static SqliteApi api = [] {
            SqliteApi a;
            for (const char *name : {"libsqlite3.so.0", "libsqlite3.so"}) {
                a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (a.lib) break;
            }
            if (!a.lib) throw Error(PB_ERR_INTERNAL, "libsqlite3.so.0 not found");
#define PB_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, sym)); if (!a.field) throw Error(PB_ERR_INTERNAL, "missing " sym)
            PB_SYM(open_v2, "sqlite3_open_v2"); PB_SYM(close_v2, "sqlite3_close_v2"); PB_SYM(exec, "sqlite3_exec");
            PB_SYM(prepare_v2, "sqlite3_prepare_v2"); PB_SYM(step, "sqlite3_step"); PB_SYM(reset, "sqlite3_reset");
            PB_SYM(finalize, "sqlite3_finalize"); PB_SYM(bind_int64, "sqlite3_bind_int64"); PB_SYM(bind_blob, "sqlite3_bind_blob");
            PB_SYM(bind_text, "sqlite3_bind_text"); PB_SYM(column_int64, "sqlite3_column_int64");
            PB_SYM(column_blob, "sqlite3_column_blob"); PB_SYM(column_bytes, "sqlite3_column_bytes");
            PB_SYM(column_text, "sqlite3_column_text"); PB_SYM(last_insert_rowid, "sqlite3_last_insert_rowid");
            PB_SYM(changes, "sqlite3_changes"); PB_SYM(errmsg, "sqlite3_errmsg"); PB_SYM(free_, "sqlite3_free");
#undef PB_SYM
            return a;
        }();
        return api;
    }
};

class PersistentEngine {
  public:
    static constexpr uint32_t RESULT_LIMIT = 100;  // LIMIT 100 (engine.rs:314,381)
    double max_distance_from_query = 1e3;          // engine.rs:23

    // Engine::new + Engine::open: create the tables if absent, then mirror semantic_hashes onto the GPU.
    PersistentEngine(const std::string &db_path, uint32_t hash_dim, uint64_t capacity_rows, int device = 0)
        : S(SqliteApi::get()), dim_(hash_dim) {
        try {
            open_and_load(db_path, hash_dim, capacity_rows, device);
        } catch (...) {
            close_all();  // a throwing constructor runs no destructor: release the statements and the connection here
            throw;
        }
    }
    // hashes in `semantic_hashes` whose length differs from the index dimension (left out of the device table)
    uint64_t num_skipped_hashes() const { return skipped_hashes_; }
    uint64_t num_orphan_hashes() const { return orphan_hashes_; }  // hashes without an images row: left out of the device index at open

  private:
    void close_all() {
        for (void **st : {&ins_img_, &ins_hash_, &sel_img_, &sel_hash_})
            if (*st) {
                S.finalize(*st);
                *st = nullptr;
            }
        if (db_) S.close_v2(db_);
        db_ = nullptr;
    }
    void open_and_load(const std::string &db_path, uint32_t hash_dim, uint64_t capacity_rows, int device) {
        constexpr int RW_CREATE = 0x2 | 0x4;  // SQLITE_OPEN_READWRITE | SQLITE_OPEN_CREATE
        if (S.open_v2(db_path.c_str(), &db_, RW_CREATE, nullptr) != 0) fail("open");
        run("CREATE TABLE IF NOT EXISTS images (id INTEGER PRIMARY KEY, filename TEXT NOT NULL, path TEXT NOT NULL, "
            "image_width INTEGER, image_height INTEGER, thumbnail BLOB, created DATETIME, indexed DATETIME, UNIQUE(path))");
        run("CREATE TABLE IF NOT EXISTS semantic_hashes (image_id INTEGER PRIMARY KEY, hash BLOB)");
        run("PRAGMA journal_mode = WAL;");  // engine.rs:119-122
        pb_index *h = nullptr;
        check(pb_index_create(&h, device, hash_dim, capacity_rows));
        idx_.reset(h);
        // bulk load in image_id order, chunked (hashes of another length are skipped: fixed dim per index)
        // the reference's query JOINs semantic_hashes with images BEFORE its LIMIT (engine.rs:375-381): a hash whose image
        // row is missing can never be a result.  Loading only the joined rows keeps such orphans out of the device index
        // altogether, so a query's first 100 device results ARE the first 100 joined rows however many orphans the file
        // holds (the over-fetch in query_by_image_hash_from_image only has to cover rows deleted AFTER the load)
        void *st = prepare("SELECT semantic_hashes.image_id, semantic_hashes.hash FROM semantic_hashes INNER JOIN images ON images.id = "
                           "semantic_hashes.image_id ORDER BY semantic_hashes.image_id");
        {
            void *cnt = prepare("SELECT COUNT(*) FROM semantic_hashes WHERE image_id NOT IN (SELECT id FROM images)");
            if (S.step(cnt) == 100) orphan_hashes_ = (uint64_t)S.column_int64(cnt, 0);
            S.finalize(cnt);
        }
        struct Fin {  // the statement is finalised however this scope is left
            const SqliteApi &S;
            void *st;
            ~Fin() { S.finalize(st); }
        } fin{S, st};
        std::vector<int64_t> ids;
        std::vector<uint8_t> rows;
        while (S.step(st) == 100 /*SQLITE_ROW*/) {
            if (S.column_bytes(st, 1) != (int)dim_) {  // the reference keeps such rows (engine.rs:585 zip-truncates); the device
                ++skipped_hashes_;                     // table has ONE row length, so they are left out and COUNTED
                continue;
            }
            ids.push_back(S.column_int64(st, 0));
            const uint8_t *p = static_cast<const uint8_t *>(S.column_blob(st, 1));
            rows.insert(rows.end(), p, p + dim_);
        }
        if (!ids.empty()) check(pb_index_load(idx_.get(), ids.data(), rows.data(), ids.size()));
        ins_img_ = prepare("INSERT OR IGNORE INTO images (filename, path, image_width, image_height, thumbnail) VALUES (?, ?, ?, ?, ?)");
        ins_hash_ = prepare("INSERT OR IGNORE INTO semantic_hashes (image_id, hash) VALUES (?, ?)");
        sel_img_ = prepare("SELECT images.id, images.filename, images.path, images.image_width, images.image_height FROM images WHERE id = ?");
        sel_hash_ = prepare("SELECT hash FROM semantic_hashes WHERE image_id = ?");
    }

  public:
    ~PersistentEngine() { close_all(); }
    PersistentEngine(const PersistentEngine &) = delete;

    // engine.rs:228-259.  Returns the image id SQLite assigned (or the stale last_insert_rowid when the path is
    // already known, exactly like the reference, in which case the hash insert is ignored as well).
    int64_t insert_image_from_memory(IndexedImage img) {
        S.reset(ins_img_);
        S.bind_text(ins_img_, 1, img.filename.c_str(), -1, nullptr);
        S.bind_text(ins_img_, 2, img.path.c_str(), -1, nullptr);
        S.bind_int64(ins_img_, 3, img.resolution.first);
        S.bind_int64(ins_img_, 4, img.resolution.second);
        S.bind_blob(ins_img_, 5, nullptr, 0, nullptr);
        if (S.step(ins_img_) != 101 /*SQLITE_DONE*/) fail("insert image");
        img.id = S.last_insert_rowid(db_);
        if (img.visual_hash) {
            if (img.visual_hash->size() != dim_) throw Error(PB_ERR_INVALID, "visual_hash length != index dim");
            S.reset(ins_hash_);
            S.bind_int64(ins_hash_, 1, img.id);
            S.bind_blob(ins_hash_, 2, img.visual_hash->data(), (int)dim_, nullptr);
            if (S.step(ins_hash_) != 101) fail("insert hash");
            if (S.changes(db_) > 0) {  // a new row: write through to the device mirror
                uint64_t stored = 0;
                check(pb_index_append(idx_.get(), &img.id, img.visual_hash->data(), 1, &stored));
            }
        }
        return img.id;
    }

    // engine.rs:352-361 (see Engine::query_by_image_hash_from_file)
    bool query_by_image_hash_from_file(const std::string &path, const Embedder &model, const PHasher *hasher = nullptr,
                                       const Decoder &decode = decode_pnm) {
        cached_.reset();
        std::optional<IndexedImage> indexed_image = indexed_image_from_file_path(path, model, hasher, decode);
        if (!indexed_image) return false;
        query_by_image_hash_from_image(*indexed_image);
        return true;
    }

    void query_by_image_hash_from_image(const IndexedImage &indexed_image) {
        if (!indexed_image.visual_hash) return;  // engine.rs:364-368
        cached_.reset();
        // INNER JOIN before LIMIT 100 (engine.rs:377-381): hashes without an `images` row must not use up result slots, so
        // the index is asked for more (PB_MAX_K at most) when orphans turn up
        std::vector<IndexedImage> out;
        for (uint32_t k = RESULT_LIMIT;; k = std::min<uint32_t>(2 * k, PB_MAX_K)) {
            std::vector<int64_t> ids(k);
            std::vector<float> dist(k);
            uint32_t count = 0;
            check(pb_index_search(idx_.get(), indexed_image.visual_hash->data(), 1, k, max_distance_from_query, ids.data(), dist.data(),
                                  &count));
            out.clear();
            for (uint32_t i = 0; i < count && out.size() < RESULT_LIMIT; ++i) {
                S.reset(sel_img_);
                S.bind_int64(sel_img_, 1, ids[i]);
                if (S.step(sel_img_) != 100) continue;  // INNER JOIN: no images row -> dropped
                IndexedImage r;
                r.id = S.column_int64(sel_img_, 0);
                r.filename = reinterpret_cast<const char *>(S.column_text(sel_img_, 1));
                r.path = reinterpret_cast<const char *>(S.column_text(sel_img_, 2));
                r.resolution = {(uint32_t)S.column_int64(sel_img_, 3), (uint32_t)S.column_int64(sel_img_, 4)};
                S.reset(sel_hash_);
                S.bind_int64(sel_hash_, 1, ids[i]);
                if (S.step(sel_hash_) == 100) {
                    const uint8_t *p = static_cast<const uint8_t *>(S.column_blob(sel_hash_, 0));
                    r.visual_hash = std::vector<uint8_t>(p, p + S.column_bytes(sel_hash_, 0));  // row.get(6)
                }
                r.distance_from_query = (double)dist[i];  // row.get(7)
                out.push_back(std::move(r));
            }
            if (out.size() == RESULT_LIMIT || count < k || k == PB_MAX_K) break;
        }
        cached_ = std::move(out);
    }
    std::optional<std::vector<IndexedImage>> get_query_results() const { return cached_; }
    uint64_t get_num_indexed_images() const {
        uint64_t n = 0;
        check(pb_index_size(idx_.get(), &n));
        return n;
    }

  private:
    void fail(const char *what) const { throw Error(PB_ERR_INTERNAL, std::string("sqlite ") + what + ": " + S.errmsg(db_)); }
    void run(const char *sql) {
        char *err = nullptr;
        const int rc = S.exec(db_, sql, nullptr, nullptr, &err);
        if (err) S.free_(err);  // the message is also available through sqlite3_errmsg (fail)
        if (rc != 0) fail(sql);
    }
    void *prepare(const char *sql) {
        void *st = nullptr;
        if (S.prepare_v2(db_, sql, -1, &st, nullptr) != 0) fail(sql);
        return st;
    }
    struct Del {
        void operator()(pb_index *p) const { pb_index_destroy(p); }
    };
    const SqliteApi &S;
    void *db_ = nullptr;
    uint32_t dim_;
    std::unique_ptr<pb_index, Del> idx_;
    void *ins_img_ = nullptr, *ins_hash_ = nullptr, *sel_img_ = nullptr, *sel_hash_ = nullptr;
    std::optional<std::vector<IndexedImage>> cached_;
    uint64_t skipped_hashes_ = 0, orphan_hashes_ = 0;
};

}  // namespace pixelbox
