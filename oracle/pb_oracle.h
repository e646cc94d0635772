/*
 * pb_oracle.h -- CPU ORACLE for the PixelBox visual-similarity hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the
 * __graft_entry__.smoke() check and bench.py's cpu_baseline leg may call it.
 * The shipped library (pixelbox_amd/csrc, include/pixelbox_hip.h) never links
 * or falls back to anything in this directory.
 *
 * It is a plain-C restatement of the reference's algorithm (the reference is
 * Rust and cannot be compiled here: no cargo/rustc, crates not vendored).
 * Every function cites the reference file:line it follows.
 *
 * Parity pinning: checked against every known-answer the reference holds for
 * this path (tests/test_oracle.py): engine.rs:703-708 (3 cosine KATs),
 * README.md:54 (quantiser KAT), engine.rs:693-701 (6 hamming KATs), and against
 * the numpy restatement driven through Python's sqlite3 with the reference's
 * literal SQL (engine.rs:375-381).  The EMBEDDING floats are "parity unpinned":
 * the reference pins no embedding value anywhere (efficientnet.rs:54-67 checks
 * determinism only) and tract-onnx 0.22.0 / the ONNX weights are absent.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction: rustc
 * never fuses a*b+c, and the top-k order depends on it -- SURVEY.md F10).
 */
#ifndef PB_ORACLE_H
#define PB_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* engine.rs:576  x = ((v as f32 / 255.0) * 2.0) - 1.0, for v = 0..255 */
void pbo_dequant_lut(float lut[256]);

/* engine.rs:572-588  cosine_distance(&Vec<u8>, &Vec<u8>) -> f32 */
float pbo_cosine_distance(const uint8_t *a, size_t na, const uint8_t *b, size_t nb);
/* the intermediate `cosine_similarity` of engine.rs:586 (for error-bound tests) */
float pbo_cosine_similarity(const uint8_t *a, size_t na, const uint8_t *b, size_t nb);

/* engine.rs:590-592 byte_distance; engine.rs:594-604 hamming_distance
 * (hamming keeps the reference's u8-wrapping sum: `.sum::<u8>()`; in a release
 * build that wraps mod 256, which is what is restated here). */
float pbo_byte_distance(const uint8_t *a, size_t na, const uint8_t *b, size_t nb);
float pbo_hamming_distance(const uint8_t *a, size_t na, const uint8_t *b, size_t nb);

/* efficientnet.rs:39  128u8.saturating_add_signed((f*128).max(-128).min(128) as i8) */
uint8_t pbo_quantize1(float f);
void pbo_quantize(const float *f, size_t n, uint8_t *out);

/* engine.rs:375-390 -- SELECT ... cosine_distance(?, hash) AS dist ... WHERE dist < ?
 * ORDER BY dist ASC LIMIT k, over a contiguous table rows[n][d] with explicit ids.
 * Order: (dist asc, image_id asc) -- SQLite's observed stable rowid order for ties.
 * The comparison `dist < max_dist` is done in f64 on the exactly-widened f32
 * (engine.rs:619 `Ok(dist as f64)`).  Returns the number of results (<= k). */
size_t pbo_scan_topk(const uint8_t *query, const uint8_t *rows, const int64_t *ids,
                     size_t n, size_t d, size_t k, double max_dist,
                     int64_t *out_ids, float *out_dist);

/* metric: 0 cosine, 1 byte_distance, 2 hamming_distance -- the same query with another UDF */
size_t pbo_scan_topk_metric(int metric, const uint8_t *query, const uint8_t *rows, const int64_t *ids, size_t n, size_t d,
                            size_t k, double max_dist, int64_t *out_ids, float *out_dist);

/* All n distances (for tests that need the full distance vector). */
void pbo_scan_all(const uint8_t *query, const uint8_t *rows, size_t n, size_t d, float *out_dist);

/* Synthetic data: counter-based splitmix64 (SURVEY.md 8d "fixed-seed integer PRNG").
 * Byte j of the stream is byte (j & 7), little-endian, of mix(seed + (j/8 + 1)*GOLDEN).
 * Same definition as the device generator pb_fill_synthetic (include/pixelbox_hip.h). */
uint64_t pbo_splitmix64_at(uint64_t seed, uint64_t word_index);
void pbo_fill_synthetic(uint64_t seed, uint64_t byte_offset, size_t nbytes, uint8_t *out);

/* ---- embed half (pb_oracle_effnet.c) ---------------------------------------
 * efficientnet.rs:19-42 with the network of resources/train.py:30-46 (EfficientNet-B0
 * features -> avgpool -> Linear(1280,D) -> tanh; BN folded).  PARITY UNPINNED (see the
 * header of pb_oracle_effnet.c).  blob = PBXW0001 weight blob (pixelbox_amd/weights.py);
 * img = H*W*3 u8 RGB, HWC (what `to_rgb8()` yields, efficientnet.rs:20). */
int pbo_effnet_forward(const uint8_t *blob, size_t blob_len, const uint8_t *img, float *out_f32);
/* mlhash (efficientnet.rs:31-42) over n images, batch-1 per call on nthreads threads. */
int pbo_mlhash_batch(const uint8_t *blob, size_t blob_len, const uint8_t *imgs, size_t n,
                     int nthreads, uint8_t *out_u8, float *out_f32);
/* The same network evaluated in f64 (pb_oracle_effnet_f64.c): the third point under the embed bar -- both f32 evaluations
 * (the oracle's, the HIP path's) are measured against it.  out[D] / out_f64[n][D] = tanh outputs before the quantiser. */
int pbo_effnet_forward_f64(const uint8_t *blob, size_t blob_len, const uint8_t *img, double *out);
int pbo_effnet_batch_f64(const uint8_t *blob, size_t blob_len, const uint8_t *imgs, size_t n, int nthreads, double *out_f64);

/* ---- pre-processing (pb_oracle_resize.c): efficientnet.rs:20 `resize_to_fill(W, H, Triangle).to_rgb8()` --------
 * image 0.25.x's published algorithm restated for an RGB8 source.  PARITY UNPINNED (third-party crate, absent). */
void pbo_resize_dimensions_fill(uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint32_t *ow, uint32_t *oh);
int pbo_resize_to_fill_rgb8(const uint8_t *src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint8_t *out);

/* ---- phash (pb_oracle_phash.c): src/image_hashes/phash.rs:3-22 -- Gaussian resize to fit 16x16 (aspect kept), luma, mean
 * threshold, LSB-first bytes.  Pinned by phash.rs:36-41 (flat white -> 32 zero bytes); the image crate's arithmetic is unpinned. */
void pbo_resize_dimensions_fit(uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint32_t *ow, uint32_t *oh);
float pbo_gaussian_kernel(float x);
uint32_t pbo_gaussian_weights(uint32_t o, uint32_t in_size, uint32_t out_size, float *ws, uint32_t *count);
int pbo_phash_rgb8(const uint8_t *src, uint32_t w, uint32_t h, uint8_t *out, uint32_t *n_bytes, uint8_t *small_rgb, uint32_t *sw,
                   uint32_t *sh);

/* ---- the scan through SQLite (pb_oracle_sqlite.c): reference schema, cosine_distance registered as a scalar function
 * (engine.rs:608-622), the reference's literal query text (engine.rs:375-381).  Checker and the "scan-cpu-sqlite" baseline
 * of BASELINE.md section 3.  k must be 100 (the text says LIMIT 100).  Returns 0, -1 (no libsqlite3), -2 (SQL error). */
int pbo_sqlite_scan(const uint8_t *rows, const int64_t *ids, size_t n, size_t d, const uint8_t *queries, size_t nq, size_t k,
                    double max_dist, int64_t *out_ids, float *out_dist, uint32_t *out_count, double *secs_per_query);

#ifdef __cplusplus
}
#endif
#endif
