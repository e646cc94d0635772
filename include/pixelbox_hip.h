/*
 * pixelbox_hip.h -- C ABI of the MI355X-native PixelBox visual-similarity hot path.
 *
 * This is the drop-in boundary: what a Rust `extern "C"` block in PixelBox would bind in
 * place of (a) tract-onnx inside `image_hashes::mlhash` and (b) the SQLite `cosine_distance`
 * UDF + `ORDER BY dist LIMIT 100` scan inside `Engine`.  Plain pointers and sizes only.
 * Every entry point cites the reference interface it replaces (paths relative to the
 * PixelBox repository).  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - every function returns PB_OK (0) or a negative pb_status; it never throws, aborts or
 *     unwinds across the boundary (the reference panics via unwrap/expect: efficientnet.rs:12,34,
 *     engine.rs:99-109).  pb_last_error() returns a thread-local message for the last failure.
 *   - the caller owns every buffer it passes; the library never frees caller memory.
 *   - handles are thread-safe: calls on one handle serialise internally (the reference calls
 *     mlhash from 4 crawler threads + the UI thread against one model, engine.rs:22,180,356, and
 *     searches while another thread inserts, engine.rs:184-203,374).
 *   - "host" pointers are ordinary CPU memory; "device" pointers are HIP device memory on the
 *     handle's GPU (e.g. a torch tensor's data_ptr()).
 */
#ifndef PIXELBOX_HIP_H
#define PIXELBOX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pb_status {
    PB_OK = 0,
    PB_ERR_INVALID = -1,   /* bad argument (null pointer, wrong length, k out of range, ...) */
    PB_ERR_HIP = -2,       /* a HIP runtime call failed; pb_last_error() has the HIP error string */
    PB_ERR_NOMEM = -3,     /* host or device allocation failed */
    PB_ERR_CAPACITY = -4,  /* index is full (capacity_rows reached) */
    PB_ERR_FORMAT = -5,    /* malformed weight blob */
    PB_ERR_INTERNAL = -6,
    PB_ERR_RANGE = -7      /* embed: an activation left the domain of the fixed-point squeeze-excite sums (|x| >= 128 after a depthwise
                              SiLU); the call's outputs are not valid.  No such value occurs in a batch-normalised EfficientNet. */
} pb_status;

#define PB_MAX_K 256u /* reference: LIMIT 100 (engine.rs:314,381) */

const char *pb_last_error(void);
int pb_version(void);
/* number of visible HIP devices (0 when there is no GPU: every other call then fails loudly) */
int pb_device_count(int *n);

/* ======================================================================================
 *  scan half:  semantic_hashes table  +  cosine_distance top-k query
 * ====================================================================================== */
typedef struct pb_index pb_index;

/* Device-resident mirror of `CREATE TABLE semantic_hashes (image_id INTEGER PRIMARY KEY, hash BLOB)`
 * (engine.rs:48,109): a contiguous uint8[capacity_rows][dim] matrix in HBM, rows kept in ascending
 * image_id order (the order SQLite scans the rowid B-tree in).  dim is fixed per index; the reference
 * does not enforce blob length (engine.rs:585 zip-truncates) -- documented deviation: other lengths
 * are rejected.  `device` is the HIP device ordinal.
 * (A table streams 2.5-6.5 % slower WHILE the driver scrubs memory that some process has just released -- tens of GB written
 * and freed cost the next second or two -- and at its full rate afterwards: a property of the moment, not of the allocation.) */
int pb_index_create(pb_index **out, int device, uint32_t dim, uint64_t capacity_rows);
/* Same table for the reference's other two blob distances: the `phashes` table has
 * the semantic_hashes schema (engine.rs:106-109) and the UDFs byte_distance / hamming_distance are registered
 * beside cosine_distance (engine.rs:124-128, 590-604, 624-663).  Queries on such an index return
 * `<udf>(?, hash) AS dist ... WHERE dist < ? ORDER BY dist ASC LIMIT k` with the reference's f32 value
 * (hamming keeps its u8 wrap-around, engine.rs:603).  These metrics use the exhaustive exact pass. */
#define PB_METRIC_COSINE 0
#define PB_METRIC_BYTE 1
#define PB_METRIC_HAMMING 2
int pb_index_create_metric(pb_index **out, int device, uint32_t dim, uint64_t capacity_rows, int metric);
int pb_index_destroy(pb_index *idx);
int pb_index_size(const pb_index *idx, uint64_t *n_rows);
int pb_index_dim(const pb_index *idx, uint32_t *dim);
/* *found = 1 iff a row with this image_id is stored (the `image_id INTEGER PRIMARY KEY` lookup, engine.rs:48,109) */
int pb_index_contains(const pb_index *idx, int64_t image_id, int *found);

/* Replaces `INSERT OR IGNORE INTO semantic_hashes (image_id, hash) VALUES (?, ?)` (engine.rs:251-256),
 * batched: n (image_id, hash) pairs from HOST memory.  OR IGNORE semantics: a pair whose image_id is
 * already present, or appeared earlier in the same call, is skipped (first write wins).  ids greater than every
 * stored id append at the end (the common case: ids come from last_insert_rowid(), engine.rs:233); any other order
 * is accepted too: the new pairs of a call are appended and merged into image_id order in ONE permutation pass over
 * the affected suffix.  *n_inserted (optional) receives the number stored -- also when the call fails: an index that
 * runs full stores what fits and returns PB_ERR_CAPACITY. */
int pb_index_append(pb_index *idx, const int64_t *image_ids, const uint8_t *rows, uint64_t n,
                    uint64_t *n_inserted);

/* The same for rows that are already in DEVICE memory (the hashes pb_embed_batch_device just wrote): the crawler ->
 * embed -> insert pipeline without a host round trip for the rows.  image_ids stay a HOST array; they must be
 * strictly ascending and greater than every stored id (fresh `last_insert_rowid()` values, engine.rs:249) -- updates
 * and out-of-order ids go through pb_index_append. */
int pb_index_append_device(pb_index *idx, const int64_t *image_ids, const uint8_t *d_rows, uint64_t n);

/* Bulk (re)load at Engine::open (engine.rs:117-145) from
 * `SELECT image_id, hash FROM semantic_hashes ORDER BY image_id`: replaces the index content.
 * image_ids must be strictly increasing. */
int pb_index_load(pb_index *idx, const int64_t *image_ids, const uint8_t *rows, uint64_t n);

/* Replaces Engine::query_by_image_hash_from_image (engine.rs:363-396), i.e.
 *   SELECT ..., cosine_distance(?, semantic_hashes.hash) AS dist FROM semantic_hashes ...
 *   WHERE dist < ? ORDER BY dist ASC LIMIT 100
 * for nq query hashes at once (queries: HOST uint8[nq][dim]).  For query q the results are written to
 * out_ids[q*k .. ], out_dist[q*k .. ] (HOST), sorted by (dist ascending, image_id ascending), and
 * out_count[q] <= k receives their number.  dist is the reference's f32 value bit for bit
 * (engine.rs:572-588: 1/max(cos,1e-6) - 1 on (v/255)*2-1 de-quantised bytes, sequential unfused f32);
 * the filter is `(f64)dist < max_dist` (engine.rs:379,619).  1 <= k <= PB_MAX_K.
 * The caller joins the ids back to the `images` table (engine.rs:377,384-388). */
int pb_index_search(pb_index *idx, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist,
                    int64_t *out_ids, float *out_dist, uint32_t *out_count);

/* Same query, results left in DEVICE memory of the index's GPU (int64[nq*k], float[nq*k],
 * uint32[nq]; unused slots hold id = INT64_MAX, dist = +inf): the per-shard top-k that the multi-GPU
 * path all-gathers over RCCL before pb_topk_merge. */
int pb_index_search_device(pb_index *idx, const uint8_t *queries, uint32_t nq, uint32_t k,
                           double max_dist, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count);

/* Same query, results packed for ONE all-gather and left in DEVICE memory: d_packed is int64[nq][2k+1] with
 * [0..k) image_ids, [k..2k) the f32 distance bits (zero-extended), [2k] the count.  This is the per-rank
 * message of the row-sharded multi-GPU query (pixelbox_amd/sharded.py). */
int pb_index_search_packed(pb_index *idx, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist,
                           int64_t *d_packed);

/* Merge of the all-gathered messages (HOST memory): gathered is int64[n_lists][nq][2k+1] as produced by
 * pb_index_search_packed on every rank; writes the global top-k per query, (dist, image_id) order. */
int pb_topk_merge_packed(const int64_t *gathered, uint32_t n_lists, uint32_t nq, uint32_t k, int64_t *out_ids,
                         float *out_dist, uint32_t *out_count);

/* G-way merge of per-shard results (HOST buffers): list g holds counts[g] entries at
 * ids[g*stride ..], dist[g*stride ..], each sorted by (dist, id).  Writes the first k of the merged
 * order.  This is the step after the all-gather in the row-sharded multi-GPU query. */
int pb_topk_merge(const int64_t *ids, const float *dist, const uint32_t *counts, uint32_t n_lists,
                  uint32_t stride, uint32_t k, int64_t *out_ids, float *out_dist, uint32_t *out_count);

/* The same merge on the GPU: d_gathered is the all-gathered message block in DEVICE memory of `device` (one rank's receive
 * buffer); results to HOST buffers.  One workgroup per query ranks every listed entry by binary searches over the other
 * lists (pb_merge_kernels.h).  This is the merge pb_sharded_search runs internally; it is exported for hosts that run
 * their own exchange (bench.py under torchrun: one process per GPU).  STREAM CONTRACT: the kernel runs on the device's NULL
 * stream; the caller must have synchronised whatever stream produced d_gathered (e.g. the collective's) before the call.
 * Calls are serialised by a process-wide lock; the scratch buffers are cached per device for the life of the process. */
int pb_topk_merge_packed_device(int device, const int64_t *d_gathered, uint32_t n_lists, uint32_t nq, uint32_t k,
                                int64_t *out_ids, float *out_dist, uint32_t *out_count);

/* Read back stored rows [first, first+n) in image_id order to HOST buffers (either may be NULL):
 * the checkpoint path -- the SQLite file stays the system of record. */
int pb_index_read(const pb_index *idx, uint64_t first, uint64_t n, int64_t *image_ids, uint8_t *rows);

/* Synthetic table for benchmarks: rows [first_row, first_row+n) of splitmix64 stream `seed`
 * (bytes first_row*dim ..), generated on the device, appended with image_id = first_id + i. */
int pb_index_fill_synthetic(pb_index *idx, uint64_t seed, uint64_t first_row, uint64_t n,
                            int64_t first_id);

/* Options (pb_index_set_option) */
#define PB_OPT_SEARCH_PATH 1 /* 0 = auto; 1 = exhaustive exact scan only; 2 = one filter pass per query (never the shared pass); \
                                3 = always the concurrent-query pass (dim 256).  auto: calls with >= PB_OPT_MQ_MIN_QUERIES (default 8) \
                                queries share ONE pass over the table (i8 MFMA), fewer get one HBM pass each */
#define PB_OPT_PROFILE 2     /* 1 = bracket the scan kernel with HIP events (pb_index_get_stats) */
#define PB_OPT_STREAM 3      /* value = hipStream_t to launch on (0 = the index's own stream) */
#define PB_OPT_MQ_MIN_QUERIES 9 /* auto path: minimum queries per call for the concurrent-query pass */
/* tuning knobs of the filter pass (dim 256), used by profiles/scan_sweep.py; defaults are the measured best */
#define PB_OPT_SCAN_VARIANT 4   /* bit 0: plain instead of non-temporal loads; bits 1-2: loads in flight per lane 8/16/4; bit 3: wave-fastest tiles */
#define PB_OPT_SCAN_WG_PER_CU 5 /* workgroups per CU (default 1) */
#define PB_OPT_SCAN_WAVES 6     /* waves per workgroup: 16, 8 (default) or 4 */
#define PB_OPT_SCAN_GRID 7      /* explicit grid size (0 = workgroups per CU x CUs) */
#define PB_OPT_SCAN_LAUNCH 8    /* 0: one launch per query = one HBM pass each; 1: one launch for the chunk, queries side by side (they share \
                                   reads through the caches); 2 (default): one launch for the chunk in which every workgroup answers the \
                                   queries one after the other, streaming its rows once per query (one HBM pass per query as with 0, \
                                   without the launch gap / ramp / tail between them; dim 256, else as 0) */
#define PB_OPT_MQ_WG_PER_CU 10  /* concurrent-query pass: workgroups per CU (default 2) */
#define PB_OPT_APPEND_ASYNC 12 /* 1: pb_index_append_device returns with its copies and the per-row norms queued on the index's stream \
                                 instead of waiting for them (a producer on the same stream -- PB_OPT_STREAM, PB_OPT_EMBED_STREAM -- \
                                 can then run ahead of the GPU); searches wait as before */
#define PB_OPT_EXACT_QN 13 /* exhaustive pass over 256-byte cosine rows: queries answered per table sweep (0 = auto, 1, 2, 4) */
#define PB_OPT_SECOND_CHANCE 14 /* uncertified queries that have k results: 0 = a second filter pass at the error margin of
                                   their k-th cosine when a cost model (table size, queries, its success rate so far on
                                   this index) puts it below the exhaustive pass, 1 = always, 2 = never */
#define PB_OPT_MQ_PER_CHUNK 11  /* 1: bursts of > 64 queries run one 64-query pass at a time instead of sharing row tiles
                                  among 512 queries per workgroup (default 0; for measurement) */
int pb_index_set_option(pb_index *idx, int option, int64_t value);

typedef struct pb_scan_stats {
    uint64_t queries;         /* queries answered */
    uint64_t fast_path;       /* answered by the int-dot filter pass with a passing certificate */
    uint64_t fallback;        /* re-run through the exhaustive exact scan */
    uint64_t profiled_launches; /* scan-kernel launches timed by events (PB_OPT_PROFILE): attached to the dispatch where a call
                                   has one filter launch, recorded before and after the launches otherwise */
    double profiled_ms;       /* their summed duration */
    uint64_t profiled_bytes;  /* algorithmic bytes those launches streamed (rows * dim per query) */
    uint64_t second_chance;   /* no certificate at first, answered exactly by the second-chance pass: every row within
                                 the error margin of the first attempt's k-th cosine listed and re-scored */
    uint64_t stamp_timeouts;  /* one-query calls whose results (tagged granules in pinned memory, polled by the host) did not
                                 arrive within 20 ms and that fell back to the stream wait (a GPU shared with ingest or another
                                 process); after 3 in a row the index waits on the stream for its next 256 one-query calls,
                                 then polls again */
} pb_scan_stats;                /* queries = fast_path + second_chance + fallback */
int pb_index_get_stats(pb_index *idx, pb_scan_stats *out, int reset);

/* ======================================================================================
 *  scan half, row-sharded over the GPUs of one node -- ONE host process (the reference is one Rust process,
 *  engine.rs:79-145), N devices.  SURVEY.md section 8b ("device_ids, n") / 8e.
 * ====================================================================================== */
typedef struct pb_sharded pb_sharded;

/* `semantic_hashes` split by rows over n_devices device-resident shards (device_ids[g] = HIP ordinal of shard g;
 * capacity_rows is the total, ceil(capacity_rows / n) per shard).  With more than one distinct device the library opens
 * RCCL (dlopen librccl.so.1) and builds one communicator per device with ncclCommInitAll; a query batch is answered
 * by every shard concurrently, the per-shard top-k messages (int64[nq][2k+1], pb_index_search_packed) are exchanged
 * with ONE ncclAllGather per batch over xGMI and merged by a device kernel in (dist, image_id) order.  device_ids may
 * repeat a device (several shards on one GPU: the 1-GPU test topology); such a set exchanges by device-to-device
 * copies, since RCCL refuses duplicate devices. */
int pb_sharded_create(pb_sharded **out, const int *device_ids, int n_devices, uint32_t dim, uint64_t capacity_rows);
int pb_sharded_destroy(pb_sharded *s);
/* n_shards; uses_rccl = 1 when the exchange is the RCCL all-gather; n_exchanges = exchanges issued so far */
int pb_sharded_info(const pb_sharded *s, int *n_shards, int *uses_rccl, uint64_t *n_exchanges);
/* total rows; per_shard (optional) receives n_shards counts */
int pb_sharded_size(pb_sharded *s, uint64_t *n_rows, uint64_t *per_shard);
/* Engine::open (engine.rs:117-145): bulk load, contiguous row ranges of ceil(n / n_shards) rows per shard */
int pb_sharded_load(pb_sharded *s, const int64_t *image_ids, const uint8_t *rows, uint64_t n);
/* INSERT OR IGNORE (engine.rs:251-256) over all shards: a pair whose image_id is stored on ANY shard is skipped; the new
 * pairs of a call go to the least-full shard (spilling to the next when it fills up). */
int pb_sharded_append(pb_sharded *s, const int64_t *image_ids, const uint8_t *rows, uint64_t n, uint64_t *n_inserted);
/* The multi-GPU form of the crawler -> embed -> insert pipeline (engine.rs:177-205,228-259; crawler.rs:68-119): rows that are
 * already in the DEVICE memory of shard `shard`'s GPU -- the hashes a pb_embedder on that device has just written -- are
 * stored on that shard with no host round trip (pb_index_append_device).  image_ids: HOST array of fresh ids (strictly
 * ascending, greater than every id of that shard, stored on no other shard: last_insert_rowid() values) -- an id that is
 * stored already, or that a concurrent call is storing, REJECTS the call (PB_ERR_INVALID, nothing stored): INSERT OR IGNORE
 * and out-of-order ids are pb_sharded_append's.  PB_ERR_CAPACITY when the shard (ceil(capacity / n) rows) or the table is
 * full -- rows that other shards still have room for can go through pb_sharded_append, which spills.  Calls for DIFFERENT
 * shards may run concurrently from different host threads (one embed thread per device): what a call is about to store is
 * reserved under the table's lock first, so concurrent calls cannot overrun the capacity or store one id twice.
 * pb_sharded_shard_device tells which device a shard lives on, so that an embedder can be created beside it. */
int pb_sharded_append_device(pb_sharded *s, int shard, const int64_t *image_ids, const uint8_t *d_rows, uint64_t n);
int pb_sharded_shard_device(const pb_sharded *s, int shard, int *device);
/* *found = 1 iff a row with this image_id is stored on some shard (pb_index_contains over the shards; rows a concurrent
 * pb_sharded_append_device is still copying do not count yet) */
int pb_sharded_contains(pb_sharded *s, int64_t image_id, int *found);
/* Engine::query_by_image_hash_from_image (engine.rs:363-396) over all shards; arguments as pb_index_search */
int pb_sharded_search(pb_sharded *s, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist, int64_t *out_ids,
                      float *out_dist, uint32_t *out_count);
/* rows [0, n) of synthetic stream `seed` (pb_index_fill_synthetic), contiguous ranges per shard, ids first_id + row */
int pb_sharded_fill_synthetic(pb_sharded *s, uint64_t seed, uint64_t n, int64_t first_id);
/* pb_index_set_option on every shard (PB_OPT_STREAM excepted: shards keep their own streams); stats summed over shards */
int pb_sharded_set_option(pb_sharded *s, int option, int64_t value);
int pb_sharded_get_stats(pb_sharded *s, pb_scan_stats *out, int reset);

/* ======================================================================================
 *  embed half:  image_hashes::mlhash
 * ====================================================================================== */
typedef struct pb_embedder pb_embedder;

/* Replaces the lazy-static tract model (efficientnet.rs:10-14).  weights_blob: PBXW0001 blob
 * (pixelbox_amd/weights.py; BN-folded EfficientNet-B0 + Linear(1280, D), resources/train.py:30-46)
 * in HOST memory; it fixes H, W and D.  max_batch is the number of images one forward pass takes (the workspace is sized
 * for it); pb_embed_batch accepts any n and runs larger calls as chunks of max_batch. */
int pb_embed_create(pb_embedder **out, int device, const void *weights_blob, size_t blob_len,
                    uint32_t max_batch);
int pb_embed_destroy(pb_embedder *e);
int pb_embed_info(const pb_embedder *e, uint32_t *h, uint32_t *w, uint32_t *d, uint32_t *max_batch);

/* Replaces mlhash (efficientnet.rs:31-42) for n images at once.  rgb: HOST uint8[n][H][W][3]
 * (what `resize_to_fill(W,H,Triangle).to_rgb8()` yields, efficientnet.rs:20); the px/255 NCHW
 * conversion of efficientnet.rs:21-28 is fused into the first kernel.  out_u8: HOST uint8[n][D], the
 * quantised hash of efficientnet.rs:39 (bit-exact quantiser).  out_f32 (optional): the D tanh outputs.
 * n may exceed max_batch: the call then runs as chunks of max_batch through a two-slot pipeline (the input copy of the next
 * chunk and the output copy of the previous one beside the current forward pass; pageable input is staged through pinned
 * buffers, pinned / registered caller memory is transferred directly).  An image's hash does not depend on the chunking.
 * The call returns with all of its device work finished and nothing reading or writing the caller's buffers. */
int pb_embed_batch(pb_embedder *e, const uint8_t *rgb, uint32_t n, uint8_t *out_u8, float *out_f32);

/* Same with DEVICE input/output pointers (no PCIe in the timed region; bench.py uses this).
 * STREAM CONTRACT: the forward pass runs on the embedder's stream (its own non-blocking stream unless
 * PB_OPT_EMBED_STREAM names another).  By default the call returns after that stream has been waited for, so the
 * outputs are complete for any consumer (pb_index_append_device on the index's own stream, a hipMemcpy on the null
 * stream, ...).  With PB_OPT_EMBED_ASYNC = 1 it returns with the work queued: d_out_* may then only be read by work
 * queued on the SAME stream (give the index that stream with PB_OPT_STREAM) or after the caller has synchronised it. */
int pb_embed_batch_device(pb_embedder *e, const uint8_t *d_rgb, uint32_t n, uint8_t *d_out_u8,
                          float *d_out_f32);

/* For PB_OPT_EMBED_ASYNC callers: PB_ERR_RANGE if a forward pass that has completed since the last check left the domain of the
 * fixed-point squeeze-excite sums (its outputs are not valid), PB_OK otherwise; clears the flag.  Call it after waiting for the
 * embedder's stream and before using the outputs of the batches queued since the last check: a queued pb_embed_batch_device call
 * itself never reports (or consumes) the flag.  Every synchronous entry point checks by itself (efficientnet.rs:34: tract evaluates
 * any range; this path refuses instead of hashing a saturated sum). */
int pb_embed_check_range(pb_embedder *e);

/* mlhash(img) -> Vec<u8> for one image: writes D bytes to out (out_len must be >= D).
 * One-image calls (this, pb_embed_batch with n = 1, pb_mlhash_image) replay the forward pass as one hipGraph from the third call on
 * (input copy + ~50 kernels + output copies captured once on the embedder's stream; same results; PB_NO_GRAPH=1 in the environment at
 * pb_embed_create launches them one by one). */
int pb_mlhash(pb_embedder *e, const uint8_t *rgb, uint8_t *out, size_t out_len);

/* pb_embed_batch_images for n <= max_batch images whose hashes are wanted IN DEVICE MEMORY as well: *d_out_u8 receives a
 * pointer to the embedder's own output buffer on its GPU (uint8[n][D], valid until the next call on this embedder) -- what
 * pb_index_append_device / pb_sharded_append_device take, so the crawler -> embed -> insert pipeline (crawler.rs:68-119,
 * engine.rs:186-203) stores a batch without uploading its hashes again.  out_u8 (HOST, optional) receives a copy for the
 * record (IndexedImage.visual_hash, the write-through to SQLite).  The call returns with the forward pass complete. */
int pb_embed_batch_images_device(pb_embedder *e, const uint8_t *const *rgb, const uint32_t *widths, const uint32_t *heights, uint32_t n,
                                 uint8_t *out_u8, const uint8_t **d_out_u8);

/* ---- staging slots for decoders (round 5).  The reference's workers decode a file into a buffer of the decoder's
 * own and hash it at once (crawler.rs:68-119, indexed_image.rs:47-91); pb_embed_batch_images* take such buffers and PACK them into
 * the embedder's pinned staging before the transfer -- one more pass of every pixel through host memory.  Here a decode worker asks
 * the embedder where to put the pixels and writes them once:
 *   pb_embed_stage_acquire(e, w, h, &pixels, &ticket)   room for a w x h RGB8 image (rows top to bottom, 3 bytes per pixel, w * 3 per
 *        row) in the batch being filled; thread-safe, any number of workers; PB_STAGE_FULL (> 0, not an error) when the batch cannot
 *        take the image (max_batch images, the slot's bytes -- 48 MB unless PB_OPT_EMBED_STAGE_BYTES says otherwise -- or 256 MB of resize
 *        scratch reached): close + commit it, acquire again (the reference crawler's workers poll).  The call only BLOCKS when no batch is
 *        open and the slot whose turn it is has not come back from its commit (or from an abort with writers still in it).
 *   pb_embed_stage_release(e, ticket)                   the pixels are written.
 *   pb_embed_stage_close(e, &n, &generation, widths, heights, pixels)   one thread (the embed thread): no more images join the batch;
 *        waits until every acquired image has been released; returns the batch's images in acquisition order (ticket & 0xFFFF = the
 *        image's position, ticket >> 32 = generation) -- their pixel pointers stay valid until the commit returns, for whatever
 *        else wants them on the host (phash).  widths / heights / pixels: room for max_batch entries each, or null.
 *   pb_embed_stage_commit(e, out_u8, &d_out_u8)         the closed batch: ONE transfer of the block, resize_to_fill + network as
 *        pb_embed_batch_images_device (same bits), hashes to out_u8 (host, optional) and in the embedder's own device buffer
 *        (*d_out_u8, valid until the next call on this embedder); returns with the forward pass complete and the slot free again.
 *   pb_embed_stage_abort(e)                             discards what the staging holds: the batch being filled and a batch closed and
 *        not yet committed (a batch inside pb_embed_stage_commit is left to that call).  Tickets handed out for them stay releasable
 *        (and must still be released: a slot with writers in it is free again at its last release); a pb_embed_stage_close blocked on
 *        such writers returns PB_STAGE_ABORTED (> 0, not an error) with *n = 0.  What a cancelled or failed run must call before the
 *        embedder's staging is used again: without it a batch closed and never committed fails every later close (PB_ERR_INVALID),
 *        and an open batch's stale pixels would join the next run's first batch.  Never blocks.
 * Decoders fill one slot while the other is being committed. */
#define PB_STAGE_FULL 1
#define PB_STAGE_ABORTED 2
int pb_embed_stage_acquire(pb_embedder *e, uint32_t w, uint32_t h, uint8_t **pixels, uint64_t *ticket);
int pb_embed_stage_release(pb_embedder *e, uint64_t ticket);
int pb_embed_stage_close(pb_embedder *e, uint32_t *n, uint32_t *generation, uint32_t *widths, uint32_t *heights, const uint8_t **pixels);
int pb_embed_stage_commit(pb_embedder *e, uint8_t *out_u8, const uint8_t **d_out_u8);
int pb_embed_stage_abort(pb_embedder *e);

/* The same for images of ANY size: efficientnet.rs:19-29 `image_to_tensor` in full -- the image crate's
 * `resize_to_fill(W, H, FilterType::Triangle)` (scale to cover, separable triangle filter through an f32
 * intermediate, centre crop; image 0.25.x semantics, restated -- the crate is not part of the reference tree, so
 * this step is unpinned: oracle/pb_oracle_resize.c) runs on the GPU, then the forward pass.  rgb: width*height*3
 * bytes, RGB8, row-major.  An image that already is W x H is passed through untouched, like the crate does. */
int pb_mlhash_image(pb_embedder *e, const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out, size_t out_len);
/* n images of individual sizes: rgb[i] -> widths[i]*heights[i]*3 bytes; out_u8[n][D], out_f32[n][D] or NULL */
int pb_embed_batch_images(pb_embedder *e, const uint8_t *const *rgb, const uint32_t *widths, const uint32_t *heights, uint32_t n,
                          uint8_t *out_u8, float *out_f32);
/* the pre-processed W x H RGB8 image itself (what the network sees before /255): out_rgb[H*W*3] */
int pb_resize_to_fill(pb_embedder *e, const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out_rgb);

/* Page-locked host memory (hipHostMalloc) for a caller's image batches: pb_embed_batch transfers a batch that sits in such a
 * buffer without staging it first (what the decode workers of crawler.rs:68-119 would fill).  Free with pb_pinned_free. */
int pb_pinned_alloc(void **out, size_t bytes);
int pb_pinned_free(void *p);

#define PB_OPT_EMBED_STREAM 3 /* value = hipStream_t the forward pass is launched on (0 = the embedder's own stream) */
#define PB_OPT_EMBED_ASYNC 4  /* 1: pb_embed_batch_device returns with the forward pass queued (see its stream contract); default 0 */
#define PB_OPT_EMBED_STAGE_BYTES 5 /* capacity of each of the two staging slots of pb_embed_stage_* in bytes (default 48 MB = 244 images of 256 x 256);
                                      takes effect the next time a slot is opened */
#define PB_OPT_EMBED_FRONT_SUB 6 /* images per sub-batch of the network's front (stem .. the last block with a large expanded map: blocks 0-4
                                    at 128 x 128): their maps then stay in the 256 MiB Infinity Cache between writer and reader.  0 = one
                                    pass over the whole batch (the default: every split measured slower at 128 x 128, profiles/r06_front_sub.txt); bits do not depend on it */
#define PB_OPT_EMBED_DUAL 7 /* batch size from which a forward pass runs as two half-batches side by side (the second on a stream and a workspace of the
                               embedder's own, joined back into the caller's stream before the call's outputs are touched): the halves' launches
                               fill each other's ramp-ups, drains and latency-bound stretches.  0 = never.  Bits do not depend on it */
int pb_embed_set_option(pb_embedder *e, int option, int64_t value);

/* The embedder picks a kernel form per (layer, batch-size bucket) by timing the candidates at first use (all forms give the
 * same bits; only speed depends on the pick).  The reference creates its model lazily and runs the first mlhash on whatever
 * thread asks (efficientnet.rs:10-14; the UI thread for a query, engine.rs:352-361), so that first-use cost is user-visible:
 *   pb_embed_tune_ms     host milliseconds spent in those timing loops so far on this embedder;
 *   pb_embed_get_tuning  serialises the picks made so far: *len receives the size; the bytes are written when out != NULL and
 *                        cap >= *len (call once with out = NULL to size the buffer);
 *   pb_embed_set_tuning  restores picks saved by an embedder created from a blob of the same H, W, D (any max_batch, any
 *                        process, any MI355X): calls at batch sizes they cover start without a timing loop.  Picks for
 *                        layers this embedder does not have, or of another library build, are rejected (PB_ERR_FORMAT) whole. */
int pb_embed_tune_ms(pb_embedder *e, double *ms);
int pb_embed_get_tuning(pb_embedder *e, uint8_t *out, size_t cap, size_t *len);
int pb_embed_set_tuning(pb_embedder *e, const uint8_t *data, size_t len);

/* ======================================================================================
 *  image_hashes::phash (src/image_hashes/phash.rs:3-22) -- the reference's other hash, stored in `phashes`
 *  (engine.rs:106-109,248-250) and compared with hamming_distance (engine.rs:594-604; pb_index_create_metric)
 * ====================================================================================== */
typedef struct pb_phasher pb_phasher;
int pb_phash_create(pb_phasher **out, int device);
int pb_phash_destroy(pb_phasher *p);
/* phash(img) -> Vec<u8>: `img.resize(16, 16, Gaussian)` (the aspect ratio is KEPT: 16 x n or n x 16), `grayscale`, mean
 * threshold with the reference's fixed divisor 256, LSB-first bytes.  rgb: width*height*3 bytes, RGB8, row-major, any size.
 * out must hold 32 bytes; *n_bytes receives the hash length (w2 * h2) / 8: 32 for a square image, fewer otherwise, as in
 * the reference.  The resampling (image 0.25.x semantics, restated: oracle/pb_oracle_phash.c) runs on the GPU; its filter
 * weights are computed on the host with libm's expf. */
int pb_phash_image(pb_phasher *p, const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out, size_t out_len, uint32_t *n_bytes);
/* The same for n images of individual sizes in ONE call (the crawler's batch, crawler.rs:68-119 -> indexed_image.rs:70): two
 * launches and one wait per <= 64 MB of source pixels instead of a transfer, three launches and two waits per image.
 * out: HOST uint8[n][32] (image i's hash in out[32 i ..], unused bytes zero), n_bytes: HOST uint32[n]. */
int pb_phash_batch_images(pb_phasher *p, const uint8_t *const *rgb, const uint32_t *widths, const uint32_t *heights, uint32_t n, uint8_t *out,
                          uint32_t *n_bytes);
/* the resized image itself (what `small` holds in phash.rs:7): out_rgb[<= 16*16*3], its size in *out_w x *out_h */
int pb_phash_small_image(pb_phasher *p, const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out_rgb, uint32_t *out_w, uint32_t *out_h);

/* ======================================================================================
 *  utilities
 * ====================================================================================== */
/* splitmix64 byte stream on the device: d_out[0..nbytes) = bytes [byte_offset, ..) of stream `seed`
 * (definition in pixelbox_amd/synth.py).  byte_offset must be a multiple of 8. */
int pb_fill_synthetic(int device, uint64_t seed, uint64_t byte_offset, uint64_t nbytes, uint8_t *d_out);

/* Synthetic RGB8 images [start, start + n) of image stream `seed` on the device, d_out[n][h][w][3] -- the
 * definition of pixelbox_amd/synth.py:synthetic_images (stream noise squeezed into a per-(image, channel)
 * brightness window, integer arithmetic only).  h*w*3 must be a multiple of 8.  Lets the end-to-end
 * configuration (BASELINE.json configs[4]: embed + insert 1M synthetic images) run without staging 49 GB of
 * pixels through the host. */
int pb_fill_synthetic_images(int device, uint64_t seed, uint64_t start, uint64_t n, uint32_t h, uint32_t w, uint8_t *d_out);
/* The STRUCTURED synthetic stream (pixelbox_amd/synth.py:synthetic_scenes): a brightness window per (image, cell of a
 * grid x grid partition, channel) instead of one per (image, channel), so that a million images give a million different
 * hashes (the end-to-end leg of bench.py; with pb_fill_synthetic_images ~40 % of them are exact duplicates). */
int pb_fill_synthetic_scenes(int device, uint64_t seed, uint64_t start, uint64_t n, uint32_t h, uint32_t w, uint32_t grid, uint8_t *d_out);

#ifdef __cplusplus
}
#endif
#endif /* PIXELBOX_HIP_H */
