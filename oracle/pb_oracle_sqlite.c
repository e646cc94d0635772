/*
 * pb_oracle_sqlite.c -- CPU ORACLE / BASELINE (test infrastructure; see pb_oracle.h): the scan the way the reference pays
 * for it.  A real SQLite database with the reference's schema (engine.rs:30-48), the restated cosine_distance registered
 * as a scalar function exactly like engine.rs:608-622 (two blob arguments, both COPIED -- the reference's `.to_vec()` --
 * result widened to f64, SQLITE_UTF8 | SQLITE_DETERMINISTIC), and the reference's literal query text (engine.rs:375-381)
 * with its SELECT_FIELDS (engine.rs:50-57) prepared, bound and stepped.  On top of the arithmetic the bare C scan
 * (pbo_scan_topk) measures, this pays the B-tree cell fetches, the INNER JOIN lookup into `images`, the repeated
 * evaluation of `dist` (it appears in SELECT, WHERE and ORDER BY) and the temp-B-tree sort -- BASELINE.md section 3,
 * "scan-cpu-sqlite".
 *
 * The system libsqlite3.so.0 is loaded with dlopen and the few entry points declared by hand (the image ships the library
 * without headers); rusqlite 0.38 bundles a newer SQLite, so this is "a SQLite", not the reference's exact build.
 */
#include "pb_oracle.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct sqlite3 sqlite3;
typedef struct sqlite3_stmt sqlite3_stmt;
typedef struct sqlite3_context sqlite3_context;
typedef struct sqlite3_value sqlite3_value;

static struct {
    void *lib;
    int (*open_v2)(const char *, sqlite3 **, int, const char *);
    int (*close_v2)(sqlite3 *);
    int (*exec)(sqlite3 *, const char *, int (*)(void *, int, char **, char **), void *, char **);
    int (*prepare_v2)(sqlite3 *, const char *, int, sqlite3_stmt **, const char **);
    int (*step)(sqlite3_stmt *);
    int (*reset)(sqlite3_stmt *);
    int (*finalize)(sqlite3_stmt *);
    int (*bind_int64)(sqlite3_stmt *, int, long long);
    int (*bind_double)(sqlite3_stmt *, int, double);
    int (*bind_blob)(sqlite3_stmt *, int, const void *, int, void (*)(void *));
    int (*bind_text)(sqlite3_stmt *, int, const char *, int, void (*)(void *));
    long long (*column_int64)(sqlite3_stmt *, int);
    double (*column_double)(sqlite3_stmt *, int);
    int (*create_function)(sqlite3 *, const char *, int, int, void *, void (*)(sqlite3_context *, int, sqlite3_value **),
                           void (*)(sqlite3_context *, int, sqlite3_value **), void (*)(sqlite3_context *));
    const void *(*value_blob)(sqlite3_value *);
    int (*value_bytes)(sqlite3_value *);
    int (*value_type)(sqlite3_value *);
    void (*result_double)(sqlite3_context *, double);
    void (*result_error)(sqlite3_context *, const char *, int);
    const char *(*errmsg)(sqlite3 *);
} S;

static int load_sqlite(void) {
    if (S.lib) return 0;
    const char *names[] = {"libsqlite3.so.0", "libsqlite3.so"};
    for (int i = 0; i < 2 && !S.lib; ++i) S.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!S.lib) return -1;
#define SYM(f, n) *(void **)(&S.f) = dlsym(S.lib, n); if (!S.f) return -1
    SYM(open_v2, "sqlite3_open_v2"); SYM(close_v2, "sqlite3_close_v2"); SYM(exec, "sqlite3_exec"); SYM(prepare_v2, "sqlite3_prepare_v2");
    SYM(step, "sqlite3_step"); SYM(reset, "sqlite3_reset"); SYM(finalize, "sqlite3_finalize"); SYM(bind_int64, "sqlite3_bind_int64");
    SYM(bind_double, "sqlite3_bind_double"); SYM(bind_blob, "sqlite3_bind_blob"); SYM(bind_text, "sqlite3_bind_text");
    SYM(column_int64, "sqlite3_column_int64"); SYM(column_double, "sqlite3_column_double"); SYM(create_function, "sqlite3_create_function");
    SYM(value_blob, "sqlite3_value_blob"); SYM(value_bytes, "sqlite3_value_bytes"); SYM(value_type, "sqlite3_value_type");
    SYM(result_double, "sqlite3_result_double"); SYM(result_error, "sqlite3_result_error"); SYM(errmsg, "sqlite3_errmsg");
#undef SYM
    return 0;
}

/* engine.rs:608-622: as_blob() on both arguments (a non-blob is a UserFunctionError), `.to_vec()` copies, f32 -> f64 */
static void udf_cosine_distance(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    (void)argc;
    if (S.value_type(argv[0]) != 4 /*SQLITE_BLOB*/ || S.value_type(argv[1]) != 4) {
        S.result_error(ctx, "cosine_distance: arguments must be blobs", -1);
        return;
    }
    const int na = S.value_bytes(argv[0]), nb = S.value_bytes(argv[1]);
    uint8_t *a = (uint8_t *)malloc((size_t)na + 1), *b = (uint8_t *)malloc((size_t)nb + 1);
    memcpy(a, S.value_blob(argv[0]), (size_t)na); /* lhs.to_vec() */
    memcpy(b, S.value_blob(argv[1]), (size_t)nb); /* rhs.to_vec() */
    const float dist = pbo_cosine_distance(a, (size_t)na, b, (size_t)nb);
    free(a);
    free(b);
    S.result_double(ctx, (double)dist);
}

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* Build the database (in memory), then run nq queries with the reference's SQL.  out_ids / out_dist: [nq][k] (dist as the
 * f64 SQLite returned, narrowed back to f32 exactly), out_count[nq].  *secs_per_query = mean wall time of one query
 * (prepare + bind + step to completion, like engine.rs:375-390).  Returns 0; -1 no sqlite; -2 SQL error (msg on stderr). */
int pbo_sqlite_scan(const uint8_t *rows, const int64_t *ids, size_t n, size_t d, const uint8_t *queries, size_t nq, size_t k,
                    double max_dist, int64_t *out_ids, float *out_dist, uint32_t *out_count, double *secs_per_query) {
    if (load_sqlite()) return -1;
    if (k != 100) return -2; /* the reference's text says LIMIT 100 */
    sqlite3 *db = NULL;
    if (S.open_v2(":memory:", &db, 0x2 | 0x4, NULL)) return -2;
    int rc = -2;
    sqlite3_stmt *ins_img = NULL, *ins_hash = NULL, *q = NULL;
    char *err = NULL;
    /* engine.rs:30-48 */
    if (S.exec(db, "CREATE TABLE images (id INTEGER PRIMARY KEY, filename TEXT NOT NULL, path TEXT NOT NULL, image_width INTEGER, "
                   "image_height INTEGER, thumbnail BLOB, created DATETIME, indexed DATETIME, UNIQUE(path))", NULL, NULL, &err)) goto done;
    if (S.exec(db, "CREATE TABLE semantic_hashes (image_id INTEGER PRIMARY KEY, hash BLOB)", NULL, NULL, &err)) goto done;
    if (S.create_function(db, "cosine_distance", 2, 1 /*SQLITE_UTF8*/ | 0x800 /*SQLITE_DETERMINISTIC*/, NULL, udf_cosine_distance, NULL, NULL)) goto done;
    if (S.exec(db, "BEGIN", NULL, NULL, &err)) goto done;
    if (S.prepare_v2(db, "INSERT INTO images (id, filename, path, image_width, image_height) VALUES (?, ?, ?, 128, 128)", -1, &ins_img, NULL)) goto done;
    if (S.prepare_v2(db, "INSERT OR IGNORE INTO semantic_hashes (image_id, hash) VALUES (?, ?)", -1, &ins_hash, NULL)) goto done;
    for (size_t i = 0; i < n; ++i) {
        char name[48];
        snprintf(name, sizeof(name), "/s/%lld.png", (long long)ids[i]);
        S.reset(ins_img);
        S.bind_int64(ins_img, 1, ids[i]);
        S.bind_text(ins_img, 2, name + 3, -1, (void (*)(void *)) - 1 /*SQLITE_TRANSIENT*/);
        S.bind_text(ins_img, 3, name, -1, (void (*)(void *)) - 1);
        if (S.step(ins_img) != 101) goto done;
        S.reset(ins_hash);
        S.bind_int64(ins_hash, 1, ids[i]);
        S.bind_blob(ins_hash, 2, rows + i * d, (int)d, NULL /*SQLITE_STATIC: rows outlive the statement*/);
        if (S.step(ins_hash) != 101) goto done;
    }
    if (S.exec(db, "COMMIT", NULL, NULL, &err)) goto done;
    /* engine.rs:375-381 with SELECT_FIELDS (engine.rs:50-57), verbatim */
    static const char *SQL =
        "\n\t\t\tSELECT \n\timages.id,\n\timages.filename,\n\timages.path,\n\timages.image_width,\n\timages.image_height,\n\timages.thumbnail\n, "
        "semantic_hashes.hash, cosine_distance(?, semantic_hashes.hash) AS dist\n"
        "\t\t\tFROM semantic_hashes\n"
        "\t\t\tINNER JOIN images images ON images.id = semantic_hashes.image_id\n"
        "\t\t\tWHERE dist < ?\n"
        "\t\t\tORDER BY dist ASC\n"
        "\t\t\tLIMIT 100";
    double total = 0.0;
    for (size_t qi = 0; qi < nq; ++qi) {
        const double t0 = now_s();
        if (S.prepare_v2(db, SQL, -1, &q, NULL)) goto done; /* conn.prepare(...) per call, engine.rs:375 */
        S.bind_blob(q, 1, queries + qi * d, (int)d, NULL);
        S.bind_double(q, 2, max_dist);
        uint32_t c = 0;
        int st;
        while ((st = S.step(q)) == 100 /*SQLITE_ROW*/) {
            if (c < k) {
                out_ids[qi * k + c] = S.column_int64(q, 0);
                out_dist[qi * k + c] = (float)S.column_double(q, 7);
            }
            ++c;
        }
        S.finalize(q);
        q = NULL;
        if (st != 101) goto done;
        total += now_s() - t0;
        out_count[qi] = c < k ? c : (uint32_t)k;
    }
    *secs_per_query = nq ? total / (double)nq : 0.0;
    rc = 0;
done:
    if (rc) fprintf(stderr, "pbo_sqlite_scan: %s\n", db ? S.errmsg(db) : "open failed");
    if (ins_img) S.finalize(ins_img);
    if (ins_hash) S.finalize(ins_hash);
    if (q) S.finalize(q);
    if (db) S.close_v2(db);
    return rc;
}
