// pb_scan.hip -- host side of the scan half of the C ABI (include/pixelbox_hip.h): the device-resident
// `semantic_hashes` table and the cosine_distance top-k query.  gfx950 only; no CPU fallback: every
// entry point fails with PB_ERR_HIP when there is no GPU.
//
// Reference: src/engine.rs:48,109 (table), :228-259 (insert), :363-396 (query), :572-588 (distance).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <mutex>
#include <new>
#include <unordered_set>
#include <vector>

#include <hip/hip_ext.h>

#include "pb_common.h"
#include "pb_scan_kernels.h"

#pragma clang fp contract(off)

using namespace pbk;
#ifdef PB_CALL_TRACE
// instrumented build: host-side time points of a pb_index_search call (us since its entry), mean of every 64 calls printed to stderr
static double g_ct_sum[8];
static uint64_t g_ct_n;
static std::chrono::steady_clock::time_point g_ct0;
#define PB_CT_BEGIN() (g_ct0 = std::chrono::steady_clock::now())
#define PB_CT(i) (g_ct_sum[i] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g_ct0).count())
#else
#define PB_CT_BEGIN() ((void)0)
#define PB_CT(i) ((void)0)
#endif

namespace {

constexpr uint32_t Q_CHUNK = 64;   // queries per launch group
constexpr uint32_t PIPE_Q = 1024;  // queries staged / answered per host round trip on the concurrent-query path

// engine.rs:576 -- the de-quantisation table, computed exactly as the reference does per element
void make_lut(float lut[256]) {
    for (int v = 0; v < 256; ++v) {
        volatile float t = (float)v / 255.0f;
        volatile float t2 = t * 2.0f;
        lut[v] = t2 - 1.0f;
    }
}

bool fast_dim(uint32_t d) { return d >= 16 && d <= 1024 && (d & (d - 1)) == 0; }

}  // namespace

struct pb_index {
    int device = 0;
    int metric = 0;  // PB_METRIC_*
    uint32_t dim = 0;
    uint64_t capacity = 0;
    uint64_t n_rows = 0;
    uint8_t *d_rows = nullptr;
    int64_t *d_ids = nullptr;
    float *d_norms = nullptr;
    int32_t *d_sumb = nullptr;  // per-row integer sum of bytes, sum (2b-255)^2: side tables of the multi-query pass
    int32_t *d_denb = nullptr;
    int32_t *d_min_den = nullptr;     // min over stored rows of sum (2b-255)^2 (device scalar, kept by k_row_norms)
    int32_t min_den_b = 0x7FFFFFFF;   // host copy
    bool min_dirty = false;           // d_min_den is newer than min_den_b (asynchronous appends): refreshed before the next search
    int opt_append_async = 0;         // PB_OPT_APPEND_ASYNC
    uint64_t ids_dirty_lo = UINT64_MAX;  // d_ids[ids_dirty_lo ..) not uploaded yet (asynchronous appends keep the ids on the host
                                         // until something reads d_ids: only the re-scoring kernels of a search do)
    float *d_lut = nullptr;
    std::vector<int64_t> h_ids;  // ascending, mirrors d_ids
    float lut[256];

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed_by_launch = false;   // this call's filter launch carries ev0 / ev1 itself (PB_LAUNCH_TIMED)
    int n_cu = 256;

    // search workspace (device)
    uint8_t *d_queries = nullptr;   // Q_CHUNK * dim
    QParams *d_qp = nullptr;        // Q_CHUNK
    uint64_t *d_lists = nullptr;    // filter lists: Q_CHUNK * F_MAX_WG * F_KWG ; exact lists reuse
    ListHdr *d_hdrs = nullptr;      // Q_CHUNK * F_MAX_WG
    uint64_t *d_dropkeys = nullptr; // Q_CHUNK * F_MAX_WG (byte / hamming pass: smallest key each workgroup dropped)
    uint64_t *d_xlists[2] = {nullptr, nullptr};  // exact pass ping-pong: Q_CHUNK * X_MAX_WG * PB_MAX_K
    uint32_t *d_xcounts[2] = {nullptr, nullptr};
    uint32_t *d_qsel = nullptr;     // PIPE_Q (a chunk uses the first Q_CHUNK; the burst path's fallback lists all its uncertified queries)
    uint32_t *d_tail = nullptr;     // DYN_REGIONS ticket counters of the one-query filter launch (zero between launches)
    float *d_qf = nullptr;          // Q_CHUNK * 256: de-quantised queries of the coalesced exhaustive pass
    float *d_tau = nullptr;         // multi-query pass: per-query candidate threshold (PIPE_Q)
    uint64_t *d_cand = nullptr;     // PIPE_Q * MQ_CAP
    uint32_t *d_cand_cnt = nullptr; // PIPE_Q
    uint32_t *d_ghist = nullptr;    // PIPE_Q * MQ_BINS
    // second chance (k_mq_rescore_big): compacted queries / params / thresholds / big candidate lists of one chunk
    uint8_t *d_queries2 = nullptr;  // Q_CHUNK * dim
    QParams *d_qp2 = nullptr;       // Q_CHUNK
    float *d_tau2 = nullptr;        // Q_CHUNK
    uint64_t *d_cand2 = nullptr;    // Q_CHUNK * MQ_CAP2
    uint32_t *d_cand_cnt2 = nullptr;  // Q_CHUNK
    uint32_t *d_qsel2 = nullptr;    // Q_CHUNK
    int64_t *d_res_ids = nullptr;   // PIPE_Q * PB_MAX_K
    float *d_res_dist = nullptr;
    ResultHdr *d_res_hdr = nullptr;
    // where the result-writing kernels of the current call put ids / distances / headers: the device arrays above, or
    // -- for calls that hand results to the host -- the pinned host arrays below directly (hipHostMalloc memory is
    // device-writable: the kernels' stores cross PCIe as posted writes and are visible once the stream has been waited
    // for, so a query costs no device-to-host copy command at all and exactly ONE wait)
    int64_t *r_ids = nullptr;
    float *r_dist = nullptr;
    ResultHdr *r_hdr = nullptr;
    QArg256 argq{};             // a single query travelling as a kernel argument of the filter launch (search_chunk -> run_fast)
    bool argq_pending = false;
    // pinned host staging
    uint8_t *h_stage = nullptr;  // queries + params + qsel (one chunk)
    uint8_t *h_pipe = nullptr;   // queries + params of up to PIPE_Q queries (concurrent-query path)
    int64_t *h_res_ids = nullptr;
    float *h_res_dist = nullptr;
    ResultHdr *h_res_hdr = nullptr;
    uint32_t *h_done = nullptr;     // pinned: the result granules of a one-query call ({payload[3], tag} x (2 + PB_MAX_K), sel_put_granule), polled by the host
    uint32_t done_seq = 0;
    bool poll_pending = false;      // the select launch of this call carries a stamp
    bool env_no_poll = false;       // PB_NO_POLL: wait for the stream instead of the completion stamp (comparison)
    int64_t poll_timeout_us = 20000;  // PB_POLL_TIMEOUT_US: how long a one-query call polls for its result granules before it waits for the stream (tests: 0)
    uint32_t stamp_timeouts_row = 0;  // consecutive stamp time-outs
    uint32_t no_poll_calls = 0;       // one-query calls left on the stream wait after three time-outs in a row
    uint64_t stats_refused_fused = 0;  // one-launch calls that handed over to k_select_rescore (more candidates than threads)
    bool last_fused = false;  // the filter launch just queued carries the selection / re-scoring itself (k_scan_filter FUSE)
    int last_n_wg = 0;
    bool env_no_fuse = true;   // one-query calls as two launches (filter, k_select_rescore) unless PB_FUSE=1: the one-launch form is built, exact and
                               // measured -- 68.1-68.3 us against 67.1-67.7 per call at 1M rows, 0.381 against 0.381 ms at 10M (profiles/
                               // r06_one_launch.txt): the second launch's host cost was never on the critical path (it is queued while the
                               // filter runs), and what the fusion saves at the boundary (~1.3 us) the arrival hand-off and a 512-thread
                               // selection give back
    uint32_t filter_tile_rows = 32;  // rows per tile of the filter launch just queued (loads in flight x rows per load): the certificate's tile sums
    bool tail_dirty = false;        // a filter launch that uses the counters in d_tail (STEAL / DYN) was queued without the k_select_rescore that clears them
    bool env_loop_static = false;   // PB_LOOP_STATIC: the looped filter launch with fixed tile strides per wave (comparison)

    bool env_static_tail = false;      // PB_STATIC_TAIL: the one-query filter launch with static shares per workgroup at any size (comparison)
    bool env_force_tickets = false;    // PB_FORCE_TAIL_TICKETS: round 2's form of the dynamic tail (one device ticket per wave), at any size (comparison, tests)
    uint32_t env_steal_lead = 0;       // PB_STEAL_LEAD: chunks a request runs ahead of its chunk (experiments)
    bool env_force_steal = false;      // PB_FORCE_STEAL: the chunked tail also under 2M rows (tests)
    bool env_exact_lane_rows = false;  // PB_EXACT_LANE_ROWS: the lane-per-row exhaustive kernel also for 256-byte cosine rows (comparison)
    int opt_second_chance = 0;         // PB_OPT_SECOND_CHANCE: 0 = cost model, 1 = always, 2 = never
    float sc_success = 1.0f;           // running success rate of the second chance on this index (optimistic start)
    uint32_t sc_skipped = 0;           // eligible chunks sent straight to the exhaustive pass since the last attempt
    int opt_exact_qn = 0;              // PB_OPT_EXACT_QN: queries per sweep of the coalesced exhaustive pass (0 = auto: 4 from three queries on, 2 for two)
    bool env_seed_always = false;  // PB_SEED=1: seeds for HBM-sized tables too (they are for cache-sized ones: run_fast)
    bool env_no_seed = false;  // PB_NO_SEED: the looped filter launch starts every query at the max_dist floor (no sample pass)
    bool env_no_second_chance = false, env_trace_cert = false;  // PB_NO_SECOND_CHANCE / PB_TRACE_CERT, read once at create
    int opt_path = 0;
    int opt_profile = 0;
    int opt_variant = 0;    // tuning knob (dim 256 only): bit 0 = plain (temporal) loads, bits 1-2 = U in {8,16,4}, bit 3 = wave-fastest tiles
    int opt_wg_per_cu = 1;  // filter-pass workgroups per CU
    int opt_waves = F_WAVES;  // filter-pass waves per workgroup (dim 256: 16, 8, 4)
    int opt_mode = 2;       // PB_OPT_SCAN_LAUNCH: 0 = one filter launch per query, 1 = one launch, queries side by side (shared reads),
                            // 2 (default) = one launch, every workgroup answers the queries one after the other (one table pass each)
    int opt_mq_min_queries = 8;  // chunks with at least this many queries take the concurrent-query pass
    int opt_mq_wg_per_cu = 2;
    int opt_mq_per_chunk = 0;  // 1: bursts run one 64-query pass at a time (the pre-workgroup-sharing form, for comparison)
    uint64_t stats_multi = 0;
    int opt_grid = 0;       // explicit filter-pass grid size (0: wg_per_cu * CUs)
    pb_scan_stats stats{};
    mutable std::mutex mu;
};

namespace {

int alloc_workspace(pb_index *ix) {
    const size_t d = ix->dim;
    PB_HIP(hipMalloc(&ix->d_queries, PIPE_Q * d));
    PB_HIP(hipMalloc(&ix->d_qp, PIPE_Q * sizeof(QParams)));
    PB_HIP(hipMalloc(&ix->d_lists, (size_t)Q_CHUNK * F_MAX_WG * F_KWG * sizeof(uint64_t)));
    PB_HIP(hipMalloc(&ix->d_hdrs, (size_t)Q_CHUNK * F_MAX_WG * sizeof(ListHdr)));
    PB_HIP(hipMalloc(&ix->d_dropkeys, (size_t)Q_CHUNK * F_MAX_WG * sizeof(uint64_t)));
    for (int i = 0; i < 2; ++i) {
        const size_t lists = i == 0 ? X_MAX_WG : (X_MAX_WG + M_FANIN - 1) / M_FANIN;
        PB_HIP(hipMalloc(&ix->d_xlists[i], (size_t)Q_CHUNK * lists * PB_MAX_K * sizeof(uint64_t)));
        PB_HIP(hipMalloc(&ix->d_xcounts[i], (size_t)Q_CHUNK * lists * sizeof(uint32_t)));
    }
    PB_HIP(hipMalloc(&ix->d_qsel, PIPE_Q * sizeof(uint32_t)));
    // (+ one stride: the arrival counter of the one-launch one-query call, k_scan_filter FUSE)
    PB_HIP(hipMalloc(&ix->d_tail, (size_t)(DYN_REGIONS + 1) * DYN_CTR_STRIDE * sizeof(uint32_t)));
    PB_HIP(hipMemset(ix->d_tail, 0, (size_t)(DYN_REGIONS + 1) * DYN_CTR_STRIDE * sizeof(uint32_t)));
    PB_HIP(hipMalloc(&ix->d_qf, ((size_t)Q_CHUNK * 256 + 16) * sizeof(float)));  // + one piece of slack: k_scan_exact_co fetches one piece ahead
    PB_HIP(hipMalloc(&ix->d_tau, PIPE_Q * sizeof(float)));
    PB_HIP(hipMalloc(&ix->d_cand, (size_t)PIPE_Q * MQ_CAP * sizeof(uint64_t)));
    PB_HIP(hipMalloc(&ix->d_cand_cnt, PIPE_Q * sizeof(uint32_t)));
    PB_HIP(hipMalloc(&ix->d_ghist, (size_t)PIPE_Q * MQ_BINS * sizeof(uint32_t)));
    PB_HIP(hipMalloc(&ix->d_queries2, (size_t)Q_CHUNK * d));
    PB_HIP(hipMalloc(&ix->d_qp2, Q_CHUNK * sizeof(QParams)));
    PB_HIP(hipMalloc(&ix->d_tau2, Q_CHUNK * sizeof(float)));
    PB_HIP(hipMalloc(&ix->d_cand2, (size_t)Q_CHUNK * MQ_CAP2 * sizeof(uint64_t)));
    PB_HIP(hipMalloc(&ix->d_cand_cnt2, Q_CHUNK * sizeof(uint32_t)));
    PB_HIP(hipMalloc(&ix->d_qsel2, Q_CHUNK * sizeof(uint32_t)));
    PB_HIP(hipMalloc(&ix->d_res_ids, (size_t)PIPE_Q * PB_MAX_K * sizeof(int64_t)));
    PB_HIP(hipMalloc(&ix->d_res_dist, (size_t)PIPE_Q * PB_MAX_K * sizeof(float)));
    PB_HIP(hipMalloc(&ix->d_res_hdr, PIPE_Q * sizeof(ResultHdr)));
    PB_HIP(hipHostMalloc(&ix->h_stage, Q_CHUNK * (d + sizeof(QParams) + sizeof(uint32_t)), hipHostMallocDefault));
    PB_HIP(hipHostMalloc(&ix->h_pipe, PIPE_Q * (d + sizeof(QParams)), hipHostMallocDefault));
    // kernels store results and the completion stamp here and the host reads them while the stream is still busy: the memory
    // must be fine-grained COHERENT whatever HIP_HOST_COHERENT says (with non-coherent memory every one-query call would spin
    // for the stamp's whole time-out before it falls back to the stream wait)
    PB_HIP(hipHostMalloc(&ix->h_res_ids, (size_t)PIPE_Q * PB_MAX_K * sizeof(int64_t), hipHostMallocCoherent));
    PB_HIP(hipHostMalloc(&ix->h_res_dist, (size_t)PIPE_Q * PB_MAX_K * sizeof(float), hipHostMallocCoherent));
    PB_HIP(hipHostMalloc(&ix->h_res_hdr, PIPE_Q * sizeof(ResultHdr), hipHostMallocCoherent));
    PB_HIP(hipHostMalloc(&ix->h_done, (size_t)(PB_MAX_K + 2) * 16, hipHostMallocCoherent));
    memset(ix->h_done, 0, (size_t)(PB_MAX_K + 2) * 16);
    return PB_OK;
}

void free_all(pb_index *ix) {
    (void)hipFree(ix->d_rows);
    (void)hipFree(ix->d_ids);
    (void)hipFree(ix->d_norms);
    (void)hipFree(ix->d_lut);
    (void)hipFree(ix->d_queries);
    (void)hipFree(ix->d_qp);
    (void)hipFree(ix->d_lists);
    (void)hipFree(ix->d_hdrs);
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(ix->d_xlists[i]);
        (void)hipFree(ix->d_xcounts[i]);
    }
    (void)hipFree(ix->d_qsel);
    (void)hipFree(ix->d_qf);
    (void)hipFree(ix->d_tail);
    (void)hipFree(ix->d_tau);
    (void)hipFree(ix->d_cand);
    (void)hipFree(ix->d_cand_cnt);
    (void)hipFree(ix->d_ghist);
    (void)hipFree(ix->d_dropkeys);
    (void)hipFree(ix->d_queries2);
    (void)hipFree(ix->d_qp2);
    (void)hipFree(ix->d_tau2);
    (void)hipFree(ix->d_cand2);
    (void)hipFree(ix->d_cand_cnt2);
    (void)hipFree(ix->d_qsel2);
    (void)hipFree(ix->d_sumb);
    (void)hipFree(ix->d_denb);
    (void)hipFree(ix->d_min_den);
    (void)hipFree(ix->d_res_ids);
    (void)hipFree(ix->d_res_dist);
    (void)hipFree(ix->d_res_hdr);
    if (ix->h_stage) (void)hipHostFree(ix->h_stage);
    if (ix->h_pipe) (void)hipHostFree(ix->h_pipe);
    if (ix->h_res_ids) (void)hipHostFree(ix->h_res_ids);
    if (ix->h_res_dist) (void)hipHostFree(ix->h_res_dist);
    if (ix->h_res_hdr) (void)hipHostFree(ix->h_res_hdr);
    if (ix->h_done) (void)hipHostFree(ix->h_done);
    if (ix->ev0) (void)hipEventDestroy(ix->ev0);
    if (ix->ev1) (void)hipEventDestroy(ix->ev1);
    if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
}

// norms of rows [first, first+n) (append-time; engine.rs:580-581 `hash_b` half, exact arithmetic)
int launch_norms(pb_index *ix, uint64_t first, uint64_t n) {
    if (n == 0) return PB_OK;
    const int block = 256;
    const uint64_t want = (n + block - 1) / block;
    const int grid = (int)std::min<uint64_t>(want, (uint64_t)ix->n_cu * 8);
    hipLaunchKernelGGL(k_row_norms, dim3(grid), dim3(block), 0, ix->stream, ix->d_rows, first, n, (int)ix->dim,
                       ix->d_lut, ix->d_norms, ix->d_sumb, ix->d_denb, ix->d_min_den);
    PB_HIP(hipGetLastError());
    // the running minimum feeds the per-query error margin (finish_qparams)
    if (ix->opt_append_async) {
        ix->min_dirty = true;  // read back by refresh_min_den before the next search
        return PB_OK;
    }
    PB_HIP(hipMemcpyAsync(&ix->min_den_b, ix->d_min_den, sizeof(int32_t), hipMemcpyDeviceToHost, ix->stream));
    PB_HIP(hipStreamSynchronize(ix->stream));
    return PB_OK;
}

// Catch up with asynchronous appends before anything reads the index: upload the image_ids they left on the host
// (from the index's own h_ids, which outlives the caller's array -- the caller's may be a temporary that is gone by
// the time a queued copy would run) and read back the running minimum that feeds the error margin.  One wait.
int refresh_min_den(pb_index *ix) {
    if (!ix->min_dirty && ix->ids_dirty_lo == UINT64_MAX) return PB_OK;
    if (ix->ids_dirty_lo < ix->n_rows)
        PB_HIP(hipMemcpyAsync(ix->d_ids + ix->ids_dirty_lo, ix->h_ids.data() + ix->ids_dirty_lo,
                              (ix->n_rows - ix->ids_dirty_lo) * sizeof(int64_t), hipMemcpyHostToDevice, ix->stream));
    if (ix->min_dirty)
        PB_HIP(hipMemcpyAsync(&ix->min_den_b, ix->d_min_den, sizeof(int32_t), hipMemcpyDeviceToHost, ix->stream));
    PB_HIP(hipStreamSynchronize(ix->stream));  // h_ids may be reallocated by the next append: the copy must have run
    ix->min_dirty = false;
    ix->ids_dirty_lo = UINT64_MAX;
    return PB_OK;
}

// Per-query constants.  The f32 fold is the reference's (engine.rs:580-581, `hash_a` half); the
// integer sums feed the filter pass.  Host arithmetic: this TU is compiled with -ffp-contract=off.
void finish_qparams(const pb_index *ix, float acc, int64_t sum_a, int64_t sum_a2, uint32_t k, double max_dist, QParams *out);

void make_qparams(const pb_index *ix, const uint8_t *q, uint32_t k, double max_dist, QParams *out) {
    const uint32_t d = ix->dim;
    float acc = 0.0f;
    int64_t sum_a = 0, sum_a2 = 0;
    for (uint32_t i = 0; i < d; ++i) {
        const float x = ix->lut[q[i]];
        const float p = x * x;
        acc = acc + p;
        sum_a += q[i];
        sum_a2 += (int64_t)q[i] * q[i];
    }
    finish_qparams(ix, acc, sum_a, sum_a2, k, max_dist, out);
}

// nq queries: the reference's left-to-right f32 fold is a dependent chain of d additions per query (~1 us at
// d = 256 when done one query at a time: 0.4 ms of host time per 1024-query burst); eight chains are advanced
// side by side, each in its own order, so the sums are bit-identical to make_qparams
void make_qparams_batch(const pb_index *ix, const uint8_t *q, uint32_t nq, uint32_t k, double max_dist, QParams *out) {
    const uint32_t d = ix->dim;
    constexpr uint32_t W = 8;
    uint32_t q0 = 0;
    for (; q0 + W <= nq; q0 += W) {
        float acc[W];
        int64_t sa[W], sa2[W];
        for (uint32_t j = 0; j < W; ++j) {
            acc[j] = 0.0f;
            sa[j] = 0;
            sa2[j] = 0;
        }
        const uint8_t *base = q + (size_t)q0 * d;
        for (uint32_t i = 0; i < d; ++i) {
            for (uint32_t j = 0; j < W; ++j) {
                const uint8_t v = base[(size_t)j * d + i];
                const float x = ix->lut[v];
                const float p = x * x;
                acc[j] = acc[j] + p;
                sa[j] += v;
                sa2[j] += (int64_t)v * v;
            }
        }
        for (uint32_t j = 0; j < W; ++j) finish_qparams(ix, acc[j], sa[j], sa2[j], k, max_dist, &out[q0 + j]);
    }
    for (; q0 < nq; ++q0) make_qparams(ix, q + (size_t)q0 * d, k, max_dist, &out[q0]);
}

void finish_qparams(const pb_index *ix, float acc, int64_t sum_a, int64_t sum_a2, uint32_t k, double max_dist, QParams *out) {
    const uint32_t d = ix->dim;
    QParams P{};
    P.max_dist = max_dist;
    P.sqrt_sa = std::sqrt(acc);
    P.den_a = (float)(4 * sum_a2 - 1020 * sum_a + 65025ll * d);
    P.sum_a = (int32_t)sum_a;
    P.k = k;
    // c_floor: every row whose reference cosine is below it has dist >= max_dist.  dist = fl(fl(1/c) - 1)
    // is non-increasing in c; 1/c > (max_dist + 1)(1 + 1e-6) survives both roundings.
    float c_floor = 0.0f;
    if (max_dist == max_dist && max_dist > -0.5 && max_dist < 9.0e5) {
        c_floor = (float)((1.0 / (max_dist + 1.0)) * (1.0 - 2e-6));
    }
    // |cos_filter - cos_ref| <= m for this query against every stored row: de-quantisation LUT
    // rounding 6u sqrt(n) (1/|X| + 1/|Y|) with |X| = sqrt(den_a)/255 and |Y| >= sqrt(min den_b)/255, the reference's
    // fold/norm/divide roundings (2n+4)u, the filter's own f32 steps 1e-6; 25 % on top.  All-127/128 rows give the
    // old fixed 4e-4-class margin; ordinary data ~3.5e-5, which is what lets clustered tables (many rows within a
    // few 1e-4 of the k-th cosine) certify
    const double u = 1.0 / 16777216.0, nn = (double)d;
    const double den_q = (double)(4 * sum_a2 - 1020 * sum_a + 65025ll * d);
    const double den_r = (double)std::max<int32_t>(1, ix->min_den_b);
    double m = 1.25 * (6.0 * u * std::sqrt(nn) * 255.0 * (1.0 / std::sqrt(den_q) + 1.0 / std::sqrt(den_r)) + (2.0 * nn + 4.0) * u + 1e-6);
    if (!(m < (double)M_GLOB)) m = (double)M_GLOB;  // never looser than the all-vectors worst case budgeted before
    P.m = (float)m * (1.0f + 1e-6f);
    const float thr_min = 2.0f * P.m + 2e-6f;  // stay clear of the cos <= 1e-6 plateau (dist = 999999)
    const float thr_filter = c_floor - 1.01f * P.m;
    P.c_floor = c_floor;
    if (thr_filter >= thr_min) {
        P.thr0 = thr_filter;
        P.floor_is_filter = 1;
    } else {
        P.thr0 = thr_min;
        P.floor_is_filter = 0;
    }
    *out = P;
}

// Filter pass launch.  Default (mode 0): ONE launch per query, one workgroup per CU -- each query is its own pass
// over HBM (queries that run concurrently would share row reads through L2 / Infinity Cache and the measured
// bandwidth would no longer be an HBM number).  The tuning knobs (options 4-7) exist for profiles/scan_sweep.py.
// A filter launch that is the ONLY one of its call (one-query calls, the looped launch): with PB_OPT_PROFILE the timing events
// are attached to the dispatch itself (hipExtLaunchKernelGGL: start / stop from the packet's own timestamps, what a kernel
// trace reports) instead of being recorded as packets of their own before and after it -- those brackets add the command
// processor's hand-over between packets, ~2 us, which is nothing beside 22 ms of 64 passes but 5 % of a 1M-row pass (1M rows,
// same run: 47.4 us by brackets, 45.2 us in the kernel trace, profiles/r04_full_kernel_stats.csv).
#define PB_LAUNCH_TIMED(KERN, GRID, BLOCK, ...)                                                                    \
    do {                                                                                                           \
        if (ix->opt_profile) {                                                                                     \
            hipExtLaunchKernelGGL(KERN, GRID, BLOCK, 0, ix->stream, ix->ev0, ix->ev1, 0, __VA_ARGS__);             \
            ix->timed_by_launch = true;                                                                            \
        } else {                                                                                                   \
            hipLaunchKernelGGL(KERN, GRID, BLOCK, 0, ix->stream, __VA_ARGS__);                                     \
        }                                                                                                          \
    } while (0)
template <int LPR, int U, bool NT, int NW, int MAPB>
void launch_filter_t(pb_index *ix, int n_wg, uint32_t q_base, uint32_t nq) {
    constexpr int HS = (LPR == 16 && U == 8) ? 4 : 0;  // k_scan_filter: load placement
    ix->filter_tile_rows = U * (64 / LPR);
    hipLaunchKernelGGL((k_scan_filter<LPR, U, NT, NW, MAPB, false, false, false, false, HS>), dim3(n_wg, nq), dim3(NW * 64), 0, ix->stream, ix->d_rows,
                       ix->n_rows, ix->d_queries, ix->d_qp, ix->d_lists, ix->d_hdrs, (int)q_base, 1, (uint8_t *)nullptr,
                       (QParams *)nullptr, QArg256{});
}
// one launch, every workgroup answers the nq queries one after the other (k_scan_filter LOOPQ)
template <int NW, int U = 8, int MAPB = 0, bool WGT = false>
void launch_filter_loop(pb_index *ix, int n_wg, uint32_t q_base, uint32_t nq) {
    ix->filter_tile_rows = U * 4;
    PB_LAUNCH_TIMED((k_scan_filter<16, U, true, NW, MAPB, true, false, false, WGT, (U == 8 ? 4 : 0)>), dim3(n_wg, 1), dim3(NW * 64),
                    (const uint8_t *)ix->d_rows, (uint64_t)ix->n_rows, (const uint8_t *)ix->d_queries, (const QParams *)ix->d_qp, ix->d_lists,
                    ix->d_hdrs, (int)q_base, (int)nq, (uint8_t *)nullptr, (QParams *)nullptr, QArg256{}, (uint32_t *)nullptr, StealGeo{}, FuseArgs{});
}

int filter_u(const pb_index *ix) {
    if (ix->dim != 256) return 8;
    const int sel = (ix->opt_variant >> 1) & 3;
    return sel == 1 ? 16 : (sel == 2 ? 4 : 8);
}

int filter_grid(const pb_index *ix) {
    const uint32_t lpr = ix->dim / 16;
    const uint64_t rows_it = (uint64_t)filter_u(ix) * (64 / lpr);
    const uint64_t n_super = (ix->n_rows + rows_it - 1) / rows_it;
    const uint64_t nw = ix->dim == 256 ? (uint64_t)ix->opt_waves : (uint64_t)F_WAVES;
    const uint64_t want = (n_super + nw - 1) / nw;
    uint64_t cap = std::min<uint64_t>(F_MAX_WG, (uint64_t)ix->n_cu * ix->opt_wg_per_cu);
    if (ix->opt_grid > 0) cap = std::min<uint64_t>(F_MAX_WG, (uint64_t)ix->opt_grid);
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(want, cap));
}

int launch_filter(pb_index *ix, int n_wg, uint32_t q_base, uint32_t nq) {
    switch (ix->dim / 16) {
        case 1: launch_filter_t<1, 8, true, F_WAVES, 0>(ix, n_wg, q_base, nq); break;
        case 2: launch_filter_t<2, 8, true, F_WAVES, 0>(ix, n_wg, q_base, nq); break;
        case 4: launch_filter_t<4, 8, true, F_WAVES, 0>(ix, n_wg, q_base, nq); break;
        case 8: launch_filter_t<8, 8, true, F_WAVES, 0>(ix, n_wg, q_base, nq); break;
        case 32: launch_filter_t<32, 8, true, F_WAVES, 0>(ix, n_wg, q_base, nq); break;
        case 64: launch_filter_t<64, 8, true, F_WAVES, 0>(ix, n_wg, q_base, nq); break;
        case 16: {
            const bool nt = !(ix->opt_variant & 1);  // bit 0 set = plain (temporal) loads
            const int u = filter_u(ix);
            const bool mapb = ix->opt_variant & 8;
#define PB_F(UV, NTV, NWV)                                                     \
    do {                                                                       \
        if (mapb) launch_filter_t<16, UV, NTV, NWV, 1>(ix, n_wg, q_base, nq);  \
        else launch_filter_t<16, UV, NTV, NWV, 0>(ix, n_wg, q_base, nq);       \
    } while (0)
#define PB_FU(NTV, NWV)                            \
    do {                                           \
        if (u == 16) PB_F(16, NTV, NWV);           \
        else if (u == 4) PB_F(4, NTV, NWV);        \
        else PB_F(8, NTV, NWV);                    \
    } while (0)
            if (ix->opt_waves == 16) { if (nt) PB_FU(true, 16); else PB_FU(false, 16); }
            else if (ix->opt_waves == 4) { if (nt) PB_FU(true, 4); else PB_FU(false, 4); }
            else { if (nt) PB_FU(true, 8); else PB_FU(false, 8); }
#undef PB_FU
#undef PB_F
            break;
        }
        default: return pb::fail(PB_ERR_INTERNAL, "filter pass: unsupported dim %u", ix->dim);
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// PB_OPT_SCAN_LAUNCH = 2: the queries of a chunk in one launch, one table pass per query (dim 256, default load variant)
bool loop_mode(const pb_index *ix, uint32_t nq) {
    const int v = ix->opt_variant & 15;
    return ix->opt_mode == 2 && nq > 1 && ix->dim == 256 && (ix->opt_waves == 8 ? (v == 0 || v == 8 || v == 2 || v == 4) : (ix->opt_waves == 4 && v == 0));
}

template <int QT>
void launch_multi(pb_index *ix, bool hist, int grid, uint32_t base, uint32_t nq);

// Looped filter launch: every query's starting threshold from ONE shared pass over a 1/32 sample of the table (k_seed_thr,
// pb_scan_kernels.h) -- without it a wave starts at the max_dist floor and prunes its way up.  PB_NO_SEED=1 for the comparison.
int seed_thresholds(pb_index *ix, uint32_t nq) {
    if (ix->env_no_seed || ix->metric != 0 || ix->dim != 256 || nq > (uint32_t)MQ_MAXQ || ix->n_rows < (1u << 16)) return PB_OK;
    const uint64_t n_tiles = (ix->n_rows + 15) / 16, tiles = (n_tiles + MQ_SAMPLE - 1) / MQ_SAMPLE;
    PB_HIP(hipMemsetAsync(ix->d_ghist, 0, (size_t)nq * MQ_BINS * sizeof(uint32_t), ix->stream));
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((tiles + MQ_WAVES - 1) / MQ_WAVES, (uint64_t)ix->n_cu));
    switch ((int)((nq + 15) / 16)) {
        case 1: launch_multi<1>(ix, true, grid, 0, nq); break;
        case 2: launch_multi<2>(ix, true, grid, 0, nq); break;
        case 3: launch_multi<3>(ix, true, grid, 0, nq); break;
        default: launch_multi<4>(ix, true, grid, 0, nq); break;
    }
    PB_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_seed_thr, dim3(nq), dim3(64), 0, ix->stream, ix->d_ghist, ix->d_qp, (int)nq, 1e-5f);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// fast path for nq staged queries; results + status land in d_res_*
int run_fast(pb_index *ix, uint32_t nq) {
    int n_wg = filter_grid(ix);
    bool dyn = false, fused = false;
    // A table that (nearly) fits the 256 MiB Infinity Cache -- a 1.25M-row shard of the 8-GPU split is 320 MB -- is not an
    // HBM stream any more: more loads in flight pay there.  Looped launch, nothing tuned by the caller: 16 loads per lane
    // and two workgroups per CU (profiles/loop_ticket_probe.py: 6.29 -> 6.79 TB/s at 1.25M rows, 5.67 -> 6.59 at 625k; from
    // 2.5M rows on the default shape wins, 6.84 vs 6.58)
    const bool default_shape = ix->opt_variant == 0 && ix->opt_waves == F_WAVES && ix->opt_wg_per_cu == 1 && ix->opt_grid == 0;
    const bool cache_sized = default_shape && !ix->argq_pending && loop_mode(ix, nq) && !ix->env_loop_static &&
                             ix->n_rows * (uint64_t)ix->dim <= (400ull << 20);
    if (cache_sized) {
        const uint64_t n_super = (ix->n_rows + 63) / 64, want = (n_super + F_WAVES - 1) / F_WAVES;
        n_wg = (int)std::max<uint64_t>(1, std::min<uint64_t>(want, std::min<uint64_t>(F_MAX_WG, 2ull * ix->n_cu)));
    }
    ix->timed_by_launch = false;
    const bool one_launch = ix->argq_pending || cache_sized || loop_mode(ix, nq);  // PB_LAUNCH_TIMED carries the events
    if (ix->opt_profile && !one_launch) PB_HIP(hipEventRecord(ix->ev0, ix->stream));
    if (ix->argq_pending) {
        // one 256-byte query, default launch shape: the query rides in the kernel arguments (k_scan_filter ARGQ)
        ix->argq_pending = false;
        // Three forms of the one-query launch (k_scan_filter ARGQ):
        //   STEAL (from 2M rows): a workgroup's own 7/8 share by LDS tickets, the rest of the table in chunks from per-region
        //     device counters, one request per workgroup and chunk (pb_scan_kernels.h).  One box, us per call, this form /
        //     the next: 10M rows 380 / 389 (round 2's tickets), 4M 172 / 175, 2M 101 / 102, 1M 73.8 / 72.8, 500k 57.2 / 55.4 --
        //     a small table does not earn the requests back (PB_FORCE_STEAL: this form at any size, for tests);
        //   static (under 2M rows, or PB_STATIC_TAIL): equal shares per workgroup, LDS tickets inside it, nothing crosses
        //     workgroups;
        //   PB_FORCE_TAIL_TICKETS: round 2's form -- fixed strides per wave, the last eighth by one device ticket per wave and
        //     4 tiles (what tables from 4M rows up used until round 4).
        const uint64_t n_super = (ix->n_rows + 31) / 32;
        ix->filter_tile_rows = 32;  // the ARGQ forms: 8 loads in flight x 4 rows per load
        StealGeo sg{};
        sg.S = (uint32_t)((n_super - n_super / 8) / (uint64_t)n_wg);
        const uint64_t pool = n_super - (uint64_t)sg.S * n_wg;
        sg.n_reg = std::min<uint32_t>((uint32_t)(n_wg + 7) / 8, (uint32_t)DYN_REGIONS);
        const uint64_t per = (pool + sg.n_reg - 1) / sg.n_reg;
        sg.shift = 3;  // 8 tiles = one per wave
        while (((per + (1ull << sg.shift) - 1) >> sg.shift) > (uint64_t)ST_MAXC - 6) ++sg.shift;
        sg.per_reg = ((per + (1ull << sg.shift) - 1) >> sg.shift) << sg.shift;
        // a request leads its chunk by one chunk of tile time; an 8-tile chunk is ONE tile time (~2.4 us), about what the atomic's
        // round trip takes under load, so the smallest chunks are asked for two ahead
        sg.lead = sg.shift == 3 ? (ix->env_steal_lead ? ix->env_steal_lead : 2u) : 1u;
        const bool can_steal = sg.S >= sg.lead * (1u << sg.shift) && n_wg >= 8 && (ix->n_rows >= (2ull << 20) || ix->env_force_steal);
        const bool old_tickets = ix->env_force_tickets;
        if (ix->tail_dirty) {  // an earlier call failed between a ticketed launch and the launch that resets the counters
            PB_HIP(hipMemsetAsync(ix->d_tail, 0, (size_t)(DYN_REGIONS + 1) * DYN_CTR_STRIDE * sizeof(uint32_t), ix->stream));
            ix->tail_dirty = false;
        }
        // (the arguments of every form; the query and its constants ride in ix->argq)
#define PB_ARGQ_ARGS(TAIL, GEO)                                                                                                    \
    (const uint8_t *)ix->d_rows, (uint64_t)ix->n_rows, (const uint8_t *)ix->d_queries, (const QParams *)ix->d_qp, ix->d_lists, ix->d_hdrs, \
        0, 1, ix->d_queries, ix->d_qp, ix->argq, TAIL, GEO, fa
        const dim3 grid(n_wg, 1), block(F_WAVES * 64);
        const bool pin4 = ix->n_rows >= (4ull << 20);  // load placement (k_scan_filter HS): left to hipcc under 4M rows, pinned above
        // ONE launch per call (k_scan_filter FUSE): the launch's last workgroup selects, re-scores and certifies.  Its 512 threads
        // bound the count of the k-th largest list head (n_lists * ceil(k / n_lists) values) -- larger k: the two-launch form.
        FuseArgs fa{};
        const uint32_t k_q = ix->argq.p.k;
        fused = !old_tickets && !ix->env_no_fuse && (uint64_t)n_wg * ((k_q + n_wg - 1) / n_wg) <= (uint64_t)F_BLOCK && n_wg <= F_BLOCK;
        if (fused) {
            fa.ids = ix->d_ids; fa.norms = ix->d_norms; fa.lut = ix->d_lut; fa.out_ids = ix->r_ids; fa.out_dist = ix->r_dist; fa.out_hdr = ix->r_hdr;
            fa.done_flag = ix->poll_pending ? ix->h_done : nullptr;
            fa.arrive = ix->d_tail + (size_t)DYN_REGIONS * DYN_CTR_STRIDE;
            fa.out_stride = (uint32_t)PB_MAX_K; fa.done_seq = ix->done_seq; fa.tile_rows = 32;
        }
        if (fused && (ix->env_static_tail || !can_steal)) {
            ix->tail_dirty = true;  // (the arrival counter lives behind the ticket counters: a failed launch leaves it for the memset above)
            if (!pin4) PB_LAUNCH_TIMED((k_scan_filter<16, 8, true, F_WAVES, 0, false, true, false, true, 0, false, true>), grid, block, PB_ARGQ_ARGS((uint32_t *)nullptr, StealGeo{}));
            else PB_LAUNCH_TIMED((k_scan_filter<16, 8, true, F_WAVES, 0, false, true, false, true, 4, false, true>), grid, block, PB_ARGQ_ARGS((uint32_t *)nullptr, StealGeo{}));
        } else if (fused) {
            ix->tail_dirty = true;
            if (!pin4) PB_LAUNCH_TIMED((k_scan_filter<16, 8, true, F_WAVES, 0, false, true, false, true, 0, true, true>), grid, block, PB_ARGQ_ARGS(ix->d_tail, sg));
            else PB_LAUNCH_TIMED((k_scan_filter<16, 8, true, F_WAVES, 0, false, true, false, true, 4, true, true>), grid, block, PB_ARGQ_ARGS(ix->d_tail, sg));
        } else if (old_tickets) {
            ix->tail_dirty = true;
            PB_LAUNCH_TIMED((k_scan_filter<16, 8, true, F_WAVES, 0, false, true, true, false, 4>), grid, block, PB_ARGQ_ARGS(ix->d_tail, StealGeo{}));
            dyn = true;
        } else if (ix->env_static_tail || !can_steal) {
            if (!pin4) PB_LAUNCH_TIMED((k_scan_filter<16, 8, true, F_WAVES, 0, false, true, false, true>), grid, block, PB_ARGQ_ARGS((uint32_t *)nullptr, StealGeo{}));
            else PB_LAUNCH_TIMED((k_scan_filter<16, 8, true, F_WAVES, 0, false, true, false, true, 4>), grid, block, PB_ARGQ_ARGS((uint32_t *)nullptr, StealGeo{}));
        } else {
            ix->tail_dirty = true;
            if (!pin4) PB_LAUNCH_TIMED((k_scan_filter<16, 8, true, F_WAVES, 0, false, true, false, true, 0, true>), grid, block, PB_ARGQ_ARGS(ix->d_tail, sg));
            else PB_LAUNCH_TIMED((k_scan_filter<16, 8, true, F_WAVES, 0, false, true, false, true, 4, true>), grid, block, PB_ARGQ_ARGS(ix->d_tail, sg));
            dyn = true;
        }
#undef PB_ARGQ_ARGS
        PB_HIP(hipGetLastError());
    } else if (cache_sized) {
        { int rc_s = seed_thresholds(ix, nq); if (rc_s) return rc_s; }
        launch_filter_loop<8, 16, 0, true>(ix, n_wg, 0, nq);
        PB_HIP(hipGetLastError());
    } else if (loop_mode(ix, nq)) {
        // (HBM-sized tables: the seeds buy nothing -- 64 passes over 10M rows 22.76 against 22.78 ms, the clustered table 22.73-22.82
        // against 22.78-22.83, and the sample pass costs 0.03 ms of the step; a 1.25M-row shard gains 2 % in the kernel, 1 % in the
        // step: profiles/r06_seed_thresholds.txt.  PB_SEED=1 seeds here too.)
        if (ix->env_seed_always) { int rc_s = seed_thresholds(ix, nq); if (rc_s) return rc_s; }
        const int v = ix->opt_variant & 15;
        if (v == 8) launch_filter_loop<8, 8, 1>(ix, n_wg, 0, nq);
        else if (v == 2) launch_filter_loop<8, 16, 0, true>(ix, n_wg, 0, nq);
        else if (v == 4) launch_filter_loop<8, 4, 0, true>(ix, n_wg, 0, nq);
        else if (ix->opt_waves == 4) launch_filter_loop<4, 8, 0, true>(ix, n_wg, 0, nq);
        else if (ix->env_loop_static) launch_filter_loop<8>(ix, n_wg, 0, nq);
        else launch_filter_loop<8, 8, 0, true>(ix, n_wg, 0, nq);
        PB_HIP(hipGetLastError());
    } else if (ix->opt_mode == 0 || ix->opt_mode == 2) {
        for (uint32_t q = 0; q < nq; ++q) {
            int rc = launch_filter(ix, n_wg, q, 1);
            if (rc) return rc;
        }
    } else {
        int rc = launch_filter(ix, n_wg, 0, nq);
        if (rc) return rc;
    }
    PB_CT(2);
    if (ix->opt_profile && !ix->timed_by_launch) PB_HIP(hipEventRecord(ix->ev1, ix->stream));
    ix->last_fused = fused;
    ix->last_n_wg = n_wg;
    if (fused) {
        ix->tail_dirty = false;  // queued: its last workgroup leaves every counter zero
        return PB_OK;
    }
    hipLaunchKernelGGL(k_select_rescore, dim3(nq), dim3(SEL_BLOCK), 0, ix->stream, ix->d_rows, ix->d_ids, ix->d_norms,
                       (int)ix->dim, ix->d_queries, ix->d_qp, ix->d_lut, ix->d_lists, ix->d_hdrs, n_wg, ix->r_ids,
                       ix->r_dist, ix->r_hdr, (uint32_t)PB_MAX_K, dyn ? ix->d_tail : nullptr,
                       ix->poll_pending ? ix->h_done : nullptr, ix->done_seq, (uint32_t)ix->n_rows, ix->filter_tile_rows);
    PB_HIP(hipGetLastError());
    if (dyn) ix->tail_dirty = false;  // k_select_rescore is queued: it leaves the ticket counters zero
    return PB_OK;
}

// byte_distance / hamming_distance: coalesced exact-key pass (one launch per query, like the cosine filter) + merge
template <int METRIC>
int launch_dist(pb_index *ix, int n_wg, uint32_t q, uint32_t nq_loop) {
#define PB_D(LPRV)                                                                                                   \
    hipLaunchKernelGGL((k_scan_dist<LPRV, METRIC>), dim3(n_wg, 1), dim3(F_WAVES * 64), 0, ix->stream, ix->d_rows, ix->n_rows, \
                       ix->d_queries, ix->d_qp, ix->d_lists, ix->d_hdrs, ix->d_dropkeys, (int)q, (int)nq_loop)
    switch (ix->dim / 16) {
        case 1: PB_D(1); break;
        case 2: PB_D(2); break;
        case 4: PB_D(4); break;
        case 8: PB_D(8); break;
        case 16: PB_D(16); break;
        case 32: PB_D(32); break;
        case 64: PB_D(64); break;
        default: return pb::fail(PB_ERR_INTERNAL, "distance pass: unsupported dim %u", ix->dim);
    }
#undef PB_D
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int run_fast_dist(pb_index *ix, uint32_t nq) {
    const int n_wg = std::min(filter_grid(ix), (int)F_MAX_WG / 2);  // k_select_keys sorts 8192 keys: <= 256 lists
    if (ix->opt_profile) PB_HIP(hipEventRecord(ix->ev0, ix->stream));
    if (ix->opt_mode == 2) {  // one launch, the queries one after the other (one table pass each)
        int rc = ix->metric == 1 ? launch_dist<1>(ix, n_wg, 0, nq) : launch_dist<2>(ix, n_wg, 0, nq);
        if (rc) return rc;
    } else {
        for (uint32_t q = 0; q < nq; ++q) {
            int rc = ix->metric == 1 ? launch_dist<1>(ix, n_wg, q, 1) : launch_dist<2>(ix, n_wg, q, 1);
            if (rc) return rc;
        }
    }
    if (ix->opt_profile) PB_HIP(hipEventRecord(ix->ev1, ix->stream));
    hipLaunchKernelGGL(k_select_keys, dim3(nq), dim3(SEL_BLOCK), 0, ix->stream, ix->d_ids, ix->d_qp, ix->d_lists, ix->d_hdrs,
                       ix->d_dropkeys, n_wg, ix->r_ids, ix->r_dist, ix->r_hdr, (uint32_t)PB_MAX_K);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// exhaustive exact pass for the n_sel queries listed in d_qsel
template <int QN, int MAXE>
int launch_exact_co(pb_index *ix, int n_lists, uint32_t n_sel, uint32_t k, const uint32_t *qsel) {
    const size_t cap = (size_t)k + WAVE;
    const size_t lds = (size_t)XC_WAVES * WAVE * XC_PITCH + (size_t)XC_WAVES * QN * cap * sizeof(uint64_t) + (size_t)XC_WAVES * QN * sizeof(int);
    auto kern = k_scan_exact_co<QN, MAXE>;
    PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(n_lists, (n_sel + QN - 1) / QN), dim3(XC_WAVES * WAVE), lds, ix->stream, ix->d_rows, ix->d_norms,
                       ix->n_rows, ix->d_qf, ix->d_qp, qsel, (int)n_sel, ix->d_lut, ix->d_xlists[0], ix->d_xcounts[0],
                       (uint32_t)PB_MAX_K);
    return PB_OK;
}

template <int MAXE>
int launch_exact_co4(pb_index *ix, int n_lists, uint32_t n_sel, uint32_t k, const uint32_t *qsel) {
    const size_t cap = (size_t)k + WAVE;
    const size_t lds = (size_t)XC_WAVES * XC4_IMAGE + (size_t)XC_WAVES * 4 * cap * sizeof(uint64_t) + (size_t)XC_WAVES * 4 * sizeof(int);
    auto kern = k_scan_exact_co4<MAXE>;
    PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(n_lists, (n_sel + 3) / 4), dim3(XC_WAVES * WAVE), lds, ix->stream, ix->d_rows, ix->d_norms, ix->n_rows,
                       ix->d_qf, ix->d_qp, qsel, (int)n_sel, ix->d_xlists[0], ix->d_xcounts[0], (uint32_t)PB_MAX_K);
    return PB_OK;
}

int run_exact(pb_index *ix, uint32_t n_sel, uint32_t k, const uint32_t *qsel = nullptr) {
    if (!qsel) qsel = ix->d_qsel;
    // cosine over 256-byte rows: the coalesced form, two queries per table sweep when there are two (k_scan_exact_co)
    const bool coalesced = ix->metric == 0 && ix->dim == 256 && !ix->env_exact_lane_rows;
    const uint64_t n_tiles = (ix->n_rows + WAVE - 1) / WAVE;
    const uint64_t want = (n_tiles + (coalesced ? XC_WAVES : X_WAVES) - 1) / (coalesced ? XC_WAVES : X_WAVES);
    int n_lists = (int)std::max<uint64_t>(1, std::min<uint64_t>(want, coalesced ? std::min<uint64_t>(X_MAX_WG, 2ull * ix->n_cu) : X_MAX_WG));
    if (coalesced) {
        // every workgroup ends with a sort of its wave lists per query (~20 us): give a wave at least 32 tiles to stream
        // when the queries of the call fill the chip anyway (a 1M-row table ran 2x slower per row than a 10M-row one
        // with 512 workgroups per query pair)
        const int qn_sizing = ix->opt_exact_qn ? ix->opt_exact_qn : (n_sel >= 3 ? 4 : n_sel == 2 ? 2 : 1);
        const uint64_t groups = (n_sel + qn_sizing - 1) / qn_sizing;
        const uint64_t fat = std::max<uint64_t>(1, n_tiles / (32ull * XC_WAVES));
        const uint64_t fill = (2ull * ix->n_cu + groups - 1) / groups;  // workgroups per group that still fill 2 per CU
        n_lists = (int)std::min<uint64_t>((uint64_t)n_lists, std::max<uint64_t>(fat, fill));
        // whole rounds of workgroups (2 per CU at a time): 122 lists x 16 groups = 3.8 rounds cost four
        const uint64_t slots = 2ull * ix->n_cu, total = (uint64_t)n_lists * groups;
        if (total > slots) {
            const uint64_t rounds = total / slots;  // fewer, longer workgroups: each ends with its sorts
            n_lists = (int)std::max<uint64_t>(1, rounds * slots / groups);
        }
    }
    if (ix->opt_profile && ix->opt_path == 1) PB_HIP(hipEventRecord(ix->ev0, ix->stream));
    if (coalesced) {
        hipLaunchKernelGGL(k_make_qf, dim3(n_sel), dim3(256), 0, ix->stream, ix->d_queries, qsel, ix->d_lut, ix->d_qf);
        PB_HIP(hipGetLastError());
        const int qn = ix->opt_exact_qn ? ix->opt_exact_qn : (n_sel >= 3 ? 4 : n_sel == 2 ? 2 : 1);
        int rc;
        if (qn == 4) rc = k <= 128 ? launch_exact_co4<3>(ix, n_lists, n_sel, k, qsel) : launch_exact_co4<5>(ix, n_lists, n_sel, k, qsel);
        else if (k <= 128) rc = qn == 2 ? launch_exact_co<2, 3>(ix, n_lists, n_sel, k, qsel) : launch_exact_co<1, 3>(ix, n_lists, n_sel, k, qsel);
        else rc = qn == 2 ? launch_exact_co<2, 5>(ix, n_lists, n_sel, k, qsel) : launch_exact_co<1, 5>(ix, n_lists, n_sel, k, qsel);
        if (rc) return rc;
    } else {
#define PB_X(MV)                                                                                                   \
    hipLaunchKernelGGL((k_scan_exact<MV>), dim3(n_lists, n_sel), dim3(X_BLOCK), 0, ix->stream, ix->d_rows, ix->d_norms, \
                       ix->n_rows, (int)ix->dim, ix->d_queries, ix->d_qp, qsel, ix->d_lut, ix->d_xlists[0],  \
                       ix->d_xcounts[0], (uint32_t)PB_MAX_K)
    if (ix->metric == 1) PB_X(1);
    else if (ix->metric == 2) PB_X(2);
    else PB_X(0);
#undef PB_X
    }
    PB_HIP(hipGetLastError());
    if (ix->opt_profile && ix->opt_path == 1) PB_HIP(hipEventRecord(ix->ev1, ix->stream));
    int cur = 0;
    for (;;) {
        const int n_groups = (n_lists + M_FANIN - 1) / M_FANIN;
        const int final_out = n_groups == 1;
        hipLaunchKernelGGL(k_merge_lists, dim3(n_groups, n_sel), dim3(M_BLOCK), 0, ix->stream, ix->d_xlists[cur],
                           ix->d_xcounts[cur], n_lists, (uint32_t)PB_MAX_K, k, ix->d_xlists[cur ^ 1],
                           ix->d_xcounts[cur ^ 1], (uint32_t)PB_MAX_K, final_out, ix->d_ids, qsel, ix->r_ids,
                           ix->r_dist, ix->r_hdr, (uint32_t)PB_MAX_K);
        PB_HIP(hipGetLastError());
        if (final_out) break;
        n_lists = n_groups;
        cur ^= 1;
    }
    return PB_OK;
}

int account_profile(pb_index *ix, uint32_t n_queries, uint32_t n_launches) {
    float ms = 0.0f;
    PB_HIP(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
    ix->stats.profiled_launches += n_launches;
    ix->stats.profiled_ms += ms;
    ix->stats.profiled_bytes += (uint64_t)n_queries * ix->n_rows * ix->dim;
    return PB_OK;
}

// concurrent-query path (dim 256): sample pass -> thresholds -> one full pass for all nq queries -> re-score
template <int QT>
void launch_multi(pb_index *ix, bool hist, int grid, uint32_t base, uint32_t nq) {
    const uint8_t *dq = ix->d_queries + (size_t)base * ix->dim;
    const QParams *dp = ix->d_qp + base;
    float *tau = ix->d_tau + base;
    uint64_t *cand = ix->d_cand + (size_t)base * MQ_CAP;
    uint32_t *cnt = ix->d_cand_cnt + base, *gh = ix->d_ghist + (size_t)base * MQ_BINS;
    if (hist)
        hipLaunchKernelGGL((k_scan_multi<QT, true>), dim3(grid), dim3(MQ_WAVES * 64), 0, ix->stream, ix->d_rows, ix->d_sumb,
                           ix->d_denb, ix->n_rows, dq, dp, tau, cand, cnt, gh, (int)nq, (uint32_t)MQ_CAP, 0);
    else
        hipLaunchKernelGGL((k_scan_multi<QT, false>), dim3(grid), dim3(MQ_WAVES * 64), 0, ix->stream, ix->d_rows, ix->d_sumb,
                           ix->d_denb, ix->n_rows, dq, dp, tau, cand, cnt, gh, (int)nq, (uint32_t)MQ_CAP, 0);
}

// queries [base, base + nq) of the staged device arrays; results to the same slots of d_res_*
int run_multi(pb_index *ix, uint32_t nq, uint32_t k, uint32_t base = 0) {
    const int qt = (int)((nq + 15) / 16);
    const uint64_t n_tiles = (ix->n_rows + 15) / 16;
    PB_HIP(hipMemsetAsync(ix->d_ghist + (size_t)base * MQ_BINS, 0, (size_t)nq * MQ_BINS * sizeof(uint32_t), ix->stream));
    PB_HIP(hipMemsetAsync(ix->d_cand_cnt + base, 0, nq * sizeof(uint32_t), ix->stream));
    auto launch = [&](bool hist) {
        const uint64_t tiles = hist ? (n_tiles + MQ_SAMPLE - 1) / MQ_SAMPLE : n_tiles;
        const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((tiles + MQ_WAVES - 1) / MQ_WAVES,
                                                                         (uint64_t)ix->n_cu * (hist ? 1 : ix->opt_mq_wg_per_cu)));
        switch (qt) {
            case 1: launch_multi<1>(ix, hist, grid, base, nq); break;
            case 2: launch_multi<2>(ix, hist, grid, base, nq); break;
            case 3: launch_multi<3>(ix, hist, grid, base, nq); break;
            default: launch_multi<4>(ix, hist, grid, base, nq); break;
        }
    };
    launch(true);
    PB_HIP(hipGetLastError());
    // aim at ~max(4k, 512) candidates per query in the full table (1/32 sample: 16 sampled rows fix tau)
    const uint32_t target_full = std::max<uint32_t>(4 * k, 512);
    const uint32_t target_sample = std::max<uint32_t>(2, (target_full + MQ_SAMPLE - 1) / MQ_SAMPLE);
    hipLaunchKernelGGL(k_mq_pick_tau, dim3(nq), dim3(64), 0, ix->stream, ix->d_ghist + (size_t)base * MQ_BINS, ix->d_qp + base,
                       (int)nq, target_sample, ix->d_tau + base);
    PB_HIP(hipGetLastError());
    if (ix->opt_profile && base == 0) PB_HIP(hipEventRecord(ix->ev0, ix->stream));  // the first pass of a block is timed
    launch(false);
    PB_HIP(hipGetLastError());
    if (ix->opt_profile && base == 0) PB_HIP(hipEventRecord(ix->ev1, ix->stream));
    hipLaunchKernelGGL(k_mq_rescore, dim3(nq), dim3(1024), 0, ix->stream, ix->d_rows, ix->d_ids, ix->d_norms, (int)ix->dim,
                       ix->d_queries + (size_t)base * ix->dim, ix->d_qp + base, ix->d_lut, ix->d_tau + base,
                       ix->d_cand + (size_t)base * MQ_CAP, ix->d_cand_cnt + base, ix->r_ids + (size_t)base * PB_MAX_K,
                       ix->r_dist + (size_t)base * PB_MAX_K, ix->r_hdr + base, (uint32_t)PB_MAX_K);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// A burst of nq <= PIPE_Q staged queries: the sample/threshold passes run per 64-query chunk (a sixteenth of the
// table each), then ONE collect launch in which every workgroup shares its row tiles among 64 * MQ_WG_WAVES
// queries, then the re-scoring of all nq candidate lists.
constexpr int MQ_WG_WAVES = 8;
int run_multi_block(pb_index *ix, uint32_t nq, uint32_t k) {
    const uint64_t n_tiles = (ix->n_rows + 15) / 16;
    PB_HIP(hipMemsetAsync(ix->d_ghist, 0, (size_t)nq * MQ_BINS * sizeof(uint32_t), ix->stream));
    PB_HIP(hipMemsetAsync(ix->d_cand_cnt, 0, nq * sizeof(uint32_t), ix->stream));
    const uint64_t tiles = (n_tiles + MQ_SAMPLE - 1) / MQ_SAMPLE;
    // all 64-query chunks of the burst in one launch (grid.y = chunk).  Four workgroups per CU in total: every
    // workgroup ends by adding its 64 x 256-bin LDS histogram to the global one, and with a full grid per chunk
    // those tens of millions of global atomics, not the sampling, set the time of the pass
    const uint32_t n_chunks = (nq + Q_CHUNK - 1) / Q_CHUNK;
    const int hgrid = (int)std::max<uint64_t>(1, std::min<uint64_t>((tiles + MQ_WAVES - 1) / MQ_WAVES,
                                                                      std::max<uint64_t>(16, 4ull * ix->n_cu / n_chunks)));
    hipLaunchKernelGGL((k_scan_multi<4, true>), dim3(hgrid, n_chunks), dim3(MQ_WAVES * 64), 0, ix->stream,
                       ix->d_rows, ix->d_sumb, ix->d_denb, ix->n_rows, ix->d_queries, ix->d_qp, ix->d_tau, ix->d_cand,
                       ix->d_cand_cnt, ix->d_ghist, (int)nq, (uint32_t)MQ_CAP, 0);
    PB_HIP(hipGetLastError());
    const uint32_t target_full = std::max<uint32_t>(4 * k, 512);
    const uint32_t target_sample = std::max<uint32_t>(2, (target_full + MQ_SAMPLE - 1) / MQ_SAMPLE);
    hipLaunchKernelGGL(k_mq_pick_tau, dim3(nq), dim3(64), 0, ix->stream, ix->d_ghist, ix->d_qp, (int)nq, target_sample, ix->d_tau);
    PB_HIP(hipGetLastError());
    const uint64_t n_steps = (ix->n_rows + 16 * MQ_WG_WAVES - 1) / (16 * MQ_WG_WAVES);  // 16 NWQ rows per step
    const uint32_t groups = (nq + 64 * MQ_WG_WAVES - 1) / (64 * MQ_WG_WAVES);
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(n_steps, (uint64_t)ix->n_cu * ix->opt_mq_wg_per_cu));
    if (ix->opt_profile) PB_HIP(hipEventRecord(ix->ev0, ix->stream));
    hipLaunchKernelGGL((k_scan_multi_wg<MQ_WG_WAVES>), dim3(grid, groups), dim3(MQ_WG_WAVES * 64), 0, ix->stream, ix->d_rows,
                       ix->d_sumb, ix->d_denb, ix->n_rows, ix->d_queries, ix->d_qp, ix->d_tau, ix->d_cand, ix->d_cand_cnt, (int)nq);
    PB_HIP(hipGetLastError());
    if (ix->opt_profile) PB_HIP(hipEventRecord(ix->ev1, ix->stream));
    hipLaunchKernelGGL(k_mq_rescore, dim3(nq), dim3(1024), 0, ix->stream, ix->d_rows, ix->d_ids, ix->d_norms, (int)ix->dim,
                       ix->d_queries, ix->d_qp, ix->d_lut, ix->d_tau, ix->d_cand, ix->d_cand_cnt, ix->r_ids, ix->r_dist,
                       ix->r_hdr, (uint32_t)PB_MAX_K);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// Second chance (pb_scan_kernels.h (4)): queries d_qsel2[0..n_sel) of the staged chunk, thresholds d_tau2[0..n_sel).
// One shared sweep with 64 Ki-entry lists, then exact scoring of every listed row.  Results and headers land in the
// queries' own slots; status 1 = list overflow or fewer than k results (-> exhaustive pass).
bool second_chance_eligible(const pb_index *ix) {
    return ix->metric == 0 && ix->dim == 256 && ix->n_rows >= 4096 && ix->opt_path != 1 && !ix->env_no_second_chance;
}

int run_second_chance(pb_index *ix, uint32_t n_sel) {
    hipLaunchKernelGGL(k_gather_queries, dim3(n_sel), dim3(256), 0, ix->stream, ix->d_queries, ix->d_qp, ix->d_qsel2, (int)ix->dim,
                       ix->d_queries2, ix->d_qp2);
    PB_HIP(hipGetLastError());
    PB_HIP(hipMemsetAsync(ix->d_cand_cnt2, 0, n_sel * sizeof(uint32_t), ix->stream));
    const uint64_t n_tiles = (ix->n_rows + 15) / 16;
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((n_tiles + MQ_WAVES - 1) / MQ_WAVES,
                                                                     (uint64_t)ix->n_cu * ix->opt_mq_wg_per_cu));
    // Would the rows above tau2 fit the 64 Ki-entry lists?  A 1/32 sample answers for the price of a thirty-second
    // of a sweep (bin 0 of the histogram buffer = sampled rows with cos_filter >= tau2).  A query whose estimate is
    // past the list gets an unreachable threshold: its sweep costs nothing and its (empty, "fewer than k") result
    // sends it to the exhaustive pass.  Without this a table with ~1e5 near-ties per query spends its time on
    // global atomics for lists that overflow anyway.
    if (n_tiles >= 4096) {
        PB_HIP(hipMemsetAsync(ix->d_ghist, 0, (size_t)n_sel * MQ_BINS * sizeof(uint32_t), ix->stream));
        const uint64_t stiles = (n_tiles + MQ_SAMPLE - 1) / MQ_SAMPLE;
        const int hgrid = (int)std::max<uint64_t>(1, std::min<uint64_t>((stiles + MQ_WAVES - 1) / MQ_WAVES, (uint64_t)ix->n_cu));
        hipLaunchKernelGGL((k_scan_multi<4, true>), dim3(hgrid), dim3(MQ_WAVES * 64), 0, ix->stream, ix->d_rows, ix->d_sumb, ix->d_denb,
                           ix->n_rows, ix->d_queries2, ix->d_qp2, ix->d_tau2, ix->d_cand2, ix->d_cand_cnt2, ix->d_ghist, (int)n_sel,
                           (uint32_t)MQ_CAP2, 1);
        PB_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_sc_gate, dim3(1), dim3(64), 0, ix->stream, ix->d_ghist, (int)n_sel, (uint32_t)(MQ_CAP2 / MQ_SAMPLE * 3 / 4),
                           ix->d_tau2);
        PB_HIP(hipGetLastError());
    }
#define PB_SC(QTV)                                                                                                        \
    hipLaunchKernelGGL((k_scan_multi<QTV, false>), dim3(grid), dim3(MQ_WAVES * 64), 0, ix->stream, ix->d_rows, ix->d_sumb,  \
                       ix->d_denb, ix->n_rows, ix->d_queries2, ix->d_qp2, ix->d_tau2, ix->d_cand2, ix->d_cand_cnt2,       \
                       (uint32_t *)nullptr, (int)n_sel, (uint32_t)MQ_CAP2, 0)
    switch ((n_sel + 15) / 16) {
        case 1: PB_SC(1); break;
        case 2: PB_SC(2); break;
        case 3: PB_SC(3); break;
        default: PB_SC(4); break;
    }
#undef PB_SC
    PB_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_mq_rescore_big, dim3(n_sel), dim3(1024), 0, ix->stream, ix->d_rows, ix->d_ids, ix->d_norms, (int)ix->dim,
                       ix->d_queries2, ix->d_qp2, ix->d_lut, ix->d_cand2, ix->d_cand_cnt2, (uint32_t)MQ_CAP2, ix->d_qsel2,
                       ix->r_ids, ix->r_dist, ix->r_hdr, (uint32_t)PB_MAX_K);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// Second chance or straight to the exhaustive pass?  The second chance costs one i8-MFMA sweep of the table for the
// chunk (k_scan_multi, ~0.27 us per 1000 rows per 16 queries, plus its 1/32 sample) and the exact re-scoring of lists
// of up to 64 Ki rows per query, and whatever it cannot answer takes the exhaustive pass anyway; the exhaustive pass
// (k_scan_exact_co, two queries per sweep) costs ~0.034 us per 1000 rows per query.  So the second chance pays only on
// large tables, for well-filled chunks, and while it keeps succeeding: its success rate on this index is tracked
// (clustered collections whose near-tie sets overflow the lists fail it query after query) and it is re-tried every
// 32nd time so that a changed collection is noticed.  Measured on the 1M-image end-to-end table: 183 of 574
// uncertified queries rescued at 119 us each against 46 us for the exhaustive pass.
bool second_chance_pays(pb_index *ix, uint32_t n_sc) {
    if (ix->opt_second_chance == 1) return true;
    if (ix->opt_second_chance == 2) return false;
    const double rows = (double)ix->n_rows;
    const double qt = (double)std::min<uint32_t>(4, (n_sc + 15) / 16);
    const double sc_ms = rows * 2.7e-7 * qt * (1.0 + 1.0 / 32) + 0.03 * n_sc;
    const double ex_ms = n_sc * (rows * 3.4e-8 + 0.012);
    const bool pays = sc_ms + (1.0 - (double)ix->sc_success) * ex_ms < 0.9 * ex_ms;
    if (pays) return true;
    if (sc_ms < 0.9 * ex_ms && ++ix->sc_skipped >= 32) {  // worth it at a higher success rate: look again now and then
        ix->sc_skipped = 0;
        return true;
    }
    return false;
}

bool multi_eligible(const pb_index *ix, uint32_t nq) {
    return ix->metric == 0 && fast_dim(ix->dim) && ix->dim == 256 && ix->n_rows >= 65536 &&
           (ix->opt_path == 3 || (ix->opt_path == 0 && nq >= (uint32_t)ix->opt_mq_min_queries));
}

// One chunk (<= Q_CHUNK queries, already in h_stage): run the filter path, check the certificates on the
// host (D2H of the 16-byte headers only), re-run uncertified queries through the exhaustive pass.
// On return d_res_ids / d_res_dist / d_res_hdr hold the final results of the chunk and h_res_hdr mirrors
// the headers.
int search_chunk(pb_index *ix, uint32_t cq, uint32_t k, double max_dist, const float *ck_hint = nullptr,
                 const uint32_t *hint_ncand = nullptr, const float *hint_omax = nullptr, bool host_out = false) {
    const uint32_t d = ix->dim;
    // host_out: the kernels write ids / distances / headers straight into the pinned host arrays (h_res_*); otherwise
    // into d_res_* and only the headers are copied back for the certificate check
    ix->r_ids = host_out ? ix->h_res_ids : ix->d_res_ids;
    ix->r_dist = host_out ? ix->h_res_dist : ix->d_res_dist;
    ix->r_hdr = host_out ? ix->h_res_hdr : ix->d_res_hdr;
    auto wait_headers = [&]() -> int {
        if (!host_out) PB_HIP(hipMemcpyAsync(ix->h_res_hdr, ix->d_res_hdr, cq * sizeof(ResultHdr), hipMemcpyDeviceToHost, ix->stream));
        if (ix->poll_pending) {
            // one-query call: k_select_rescore publishes its results as tagged 16-byte granules in pinned memory
            // (sel_put_granule); a granule counts once its tag is this call's sequence number.  Seeing them spares the
            // end-of-kernel and stream-wait latency (the stream drains behind the caller's back; everything queued later
            // is ordered after it).  Granules that do not come within 20 ms fall back to the stream wait.
            ix->poll_pending = false;
            const uint32_t want = ix->done_seq;
            const uint32_t *g = ix->h_done;
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t spins = 0;
            bool late = false;
            // one snapshot of a granule's four words; valid iff its fourth word, un-mixed, is this call's sequence number
            // (sel_put_granule: a granule carries a check of its own payload, so a torn or half-arrived one is simply not valid yet)
            struct Snap { uint32_t w[4]; };
            auto snap = [&](uint32_t slot, Snap &s) -> bool {
                const volatile uint32_t *p = g + 4 * slot;
                s.w[3] = __atomic_load_n(&g[4 * slot + 3], __ATOMIC_ACQUIRE);
                s.w[0] = p[0]; s.w[1] = p[1]; s.w[2] = p[2];
                return (s.w[3] ^ granule_mix(s.w[0], s.w[1], s.w[2])) == want;
            };
            auto wait_granule = [&](uint32_t slot, Snap &s) -> bool {
                while (!snap(slot, s)) {
                    if (late) return false;
                    __builtin_ia32_pause();
                    if (((++spins & 4095u) == 0u || ix->poll_timeout_us < 100) &&
                        std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(ix->poll_timeout_us)) {
                        late = true;
                        return false;
                    }
                }
                return true;
            };
            auto unpack = [&](const Snap &g0, const Snap &g1) {
                ResultHdr &h = ix->h_res_hdr[0];
                h.count = g0.w[0];
                h.status = g0.w[1];
                h.n_cand = g0.w[2];
                memcpy(&h.o_max, &g1.w[0], 4);
                memcpy(&h.ck, &g1.w[1], 4);
            };
            auto take = [&](uint32_t i, const Snap &s) {
                ix->h_res_ids[i] = (int64_t)(((uint64_t)s.w[1] << 32) | s.w[0]);
                memcpy(&ix->h_res_dist[i], &s.w[2], 4);
            };
            Snap g0, g1, gi;
            bool ok = wait_granule(0, g0) && wait_granule(1, g1);
            if (ok) {
                unpack(g0, g1);
                const uint32_t n = std::min<uint32_t>(ix->h_res_hdr[0].count, PB_MAX_K);
                for (uint32_t i = 0; i < n && ok; ++i) {
                    ok = wait_granule(2 + i, gi);
                    if (!ok) break;
                    take(i, gi);
                }
            }
            if (ok) {
                ix->stamp_timeouts_row = 0;
                return PB_OK;
            }
            // the granules did not arrive in time (a GPU shared with ingest or another process): counted
            // (pb_index_get_stats); three in a row put the next 256 one-query calls on the stream wait, then polling gets
            // another chance.  After the stream wait every granule is there -- and still has to carry this call's number.
            ++ix->stats.stamp_timeouts;
            if (++ix->stamp_timeouts_row >= 3) {
                ix->stamp_timeouts_row = 0;
                ix->no_poll_calls = 256;
            }
            PB_HIP(hipStreamSynchronize(ix->stream));
            PB_CHECK(snap(0, g0) && snap(1, g1), PB_ERR_INTERNAL, "one-query call: result header granules do not carry this call's sequence number after the stream wait");
            unpack(g0, g1);
            const uint32_t n = std::min<uint32_t>(ix->h_res_hdr[0].count, PB_MAX_K);
            for (uint32_t i = 0; i < n; ++i) {
                PB_CHECK(snap(2 + i, gi), PB_ERR_INTERNAL, "one-query call: result granule %u does not carry this call's sequence number after the stream wait", i);
                take(i, gi);
            }
            return PB_OK;
        }
        PB_HIP(hipStreamSynchronize(ix->stream));
        return PB_OK;
    };
    uint8_t *hq = ix->h_stage;
    QParams *hp = reinterpret_cast<QParams *>(ix->h_stage + (size_t)Q_CHUNK * d);
    uint32_t *hsel = reinterpret_cast<uint32_t *>(ix->h_stage + (size_t)Q_CHUNK * (d + sizeof(QParams)));
    // cosine: filter pass + certificate; byte / hamming: coalesced exact-key pass + drop-bound check
    const bool use_dist = ix->metric != 0 && ix->opt_path != 1 && fast_dim(d);
    const bool use_fast = use_dist || (ix->metric == 0 && (ix->opt_path == 0 || ix->opt_path == 2 || ix->opt_path == 3) && fast_dim(d));
    { int rcm = refresh_min_den(ix); if (rcm) return rcm; }
    make_qparams_batch(ix, hq, cq, k, max_dist, hp);
    PB_CT(1);
    const bool default_shape = ix->opt_variant == 0 && ix->opt_waves == F_WAVES && ix->opt_wg_per_cu == 1 && ix->opt_grid == 0;
    if (cq == 1 && d == 256 && ix->metric == 0 && default_shape && (ix->opt_path == 0 || ix->opt_path == 2) && !multi_eligible(ix, cq)) {
        // the reference's call shape (one query per call, engine.rs:363-396) on the filter path: the query bytes and
        // constants are arguments of the filter launch itself (run_fast), which parks them for the kernels behind it
        ix->argq.p = hp[0];
        memcpy(ix->argq.q, hq, 256);
        ix->argq_pending = true;
        if (ix->no_poll_calls) --ix->no_poll_calls;
        if (host_out && !ix->env_no_poll && !ix->no_poll_calls) {  // results go straight to pinned host memory: completion by stamp
            ix->poll_pending = true;
            if (++ix->done_seq == 0) ix->done_seq = 1;
        }
    } else if (cq == 1) {
        // other single-query shapes: a one-wave staging kernel instead of two host-to-device copy commands
        QArg a;
        a.p = hp[0];
        a.dim = d;
        memcpy(a.q, hq, d);
        hipLaunchKernelGGL(k_stage_query, dim3(1), dim3(256), 0, ix->stream, a, ix->d_queries, ix->d_qp);
        PB_HIP(hipGetLastError());
    } else {
        PB_HIP(hipMemcpyAsync(ix->d_queries, hq, (size_t)cq * d, hipMemcpyHostToDevice, ix->stream));
        PB_HIP(hipMemcpyAsync(ix->d_qp, hp, (size_t)cq * sizeof(QParams), hipMemcpyHostToDevice, ix->stream));
    }
    uint32_t n_sel = 0;
    // concurrent-query pass: worth it from ~8 queries on a table large enough for the sample to mean something
    // (the host-buffer entry point routes concurrent bursts through search_block_multi; this chunk path serves
    // the packed/device entry point and small calls)
    const bool use_multi = use_fast && multi_eligible(ix, cq);
    if (use_fast) {
        int rc = use_dist ? run_fast_dist(ix, cq) : (use_multi ? run_multi(ix, cq, k) : run_fast(ix, cq));
        if (rc) {
            ix->poll_pending = ix->argq_pending = false;  // nothing was launched that would stamp / consume them
            return rc;
        }
        PB_CT(3);
        { int rcw = wait_headers(); if (rcw) return rcw; }
        if (!use_dist && !use_multi && ix->last_fused && cq == 1 && ix->h_res_hdr[0].status == 2u) {
            // the one-launch form found more candidates than its last workgroup has threads (status 2): the lists and headers are
            // in memory, the query is parked in slot 0 (workgroup 0 of that launch) -- the selection kernel of its own finishes the call
            hipLaunchKernelGGL(k_select_rescore, dim3(1), dim3(SEL_BLOCK), 0, ix->stream, ix->d_rows, ix->d_ids, ix->d_norms, (int)ix->dim,
                               ix->d_queries, ix->d_qp, ix->d_lut, ix->d_lists, ix->d_hdrs, ix->last_n_wg, ix->r_ids, ix->r_dist, ix->r_hdr,
                               (uint32_t)PB_MAX_K, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (uint32_t)ix->n_rows, ix->filter_tile_rows);
            PB_HIP(hipGetLastError());
            ++ix->stats_refused_fused;
            int rcw = wait_headers();
            if (rcw) return rcw;
        }
        PB_CT(4);
        if (ix->opt_profile) {
            int rc2 = use_multi ? account_profile(ix, 1, 1) : account_profile(ix, cq, (ix->opt_mode == 1 || loop_mode(ix, cq) || (use_dist && ix->opt_mode == 2)) ? 1 : cq);
            if (rc2) return rc2;
        }
        for (uint32_t q = 0; q < cq; ++q)
            if (ix->h_res_hdr[q].status != 0) {
                hsel[n_sel++] = q;
                if (ix->env_trace_cert)
                    fprintf(stderr, "cert fail (chunk path, multi=%d): q=%u count=%u n_cand=%u o_max=%.7f thr0=%.7f\n", (int)use_multi, q,
                            ix->h_res_hdr[q].count, ix->h_res_hdr[q].n_cand, ix->h_res_hdr[q].o_max, hp[q].thr0);
            }
        ix->stats.fast_path += cq - n_sel;
        if (use_multi) ix->stats_multi += cq - n_sel;
    } else {
        for (uint32_t q = 0; q < cq; ++q) hsel[n_sel++] = q;
    }
    // second chance: a query that HAS k results (ck = their smallest exact cosine, from the attempt above or handed
    // in by the burst path) but no certificate gets every row with cos_filter >= ck (1 - 1e-6) - 1.01 m scored exactly
    if (n_sel && !use_dist && (ck_hint || use_fast) && (ck_hint ? ix->metric == 0 && ix->dim == 256 && ix->n_rows >= 4096 && !ix->env_no_second_chance
                                                    : second_chance_eligible(ix))) {
        uint32_t sc_sel[Q_CHUNK], rest[Q_CHUNK];
        float sc_tau[Q_CHUNK];
        uint32_t n_sc = 0, n_rest = 0;
        for (uint32_t i = 0; i < n_sel; ++i) {
            const uint32_t q = hsel[i];
            const float ck = ck_hint ? ck_hint[q] : ix->h_res_hdr[q].ck;
            const float tau2 = ck * (1.0f - 1e-6f) - 1.01f * hp[q].m - 1e-7f;
            // a concurrent-query attempt reports how many rows reached ITS threshold (o_max - m): if that threshold
            // was no lower than tau2 and the count already exceeds the second chance's list, it cannot fit either
            const ResultHdr &h1 = ck_hint ? ResultHdr{0, 1, hint_ncand ? hint_ncand[q] : 0u, hint_omax ? hint_omax[q] : 0.0f, ck}
                                          : ix->h_res_hdr[q];
            const bool hopeless = (ck_hint ? hint_ncand != nullptr : use_multi) && (h1.n_cand & 0x7FFFFFFFu) > (uint32_t)MQ_CAP2 &&
                                  (h1.o_max - hp[q].m) >= tau2;
            if (ck > 0.0f && tau2 > hp[q].thr0 && tau2 > 0.0f && !hopeless) {
                sc_sel[n_sc] = q;
                sc_tau[n_sc++] = tau2;
            } else {
                rest[n_rest++] = q;
            }
        }
        if (n_sc && !second_chance_pays(ix, n_sc)) {
            for (uint32_t i = 0; i < n_sc; ++i) rest[n_rest++] = sc_sel[i];
            n_sc = 0;
            std::sort(rest, rest + n_rest);
            for (uint32_t i = 0; i < n_rest; ++i) hsel[i] = rest[i];
        }
        if (n_sc) {
            PB_HIP(hipMemcpyAsync(ix->d_qsel2, sc_sel, n_sc * sizeof(uint32_t), hipMemcpyHostToDevice, ix->stream));
            PB_HIP(hipMemcpyAsync(ix->d_tau2, sc_tau, n_sc * sizeof(float), hipMemcpyHostToDevice, ix->stream));
            int rc = run_second_chance(ix, n_sc);
            if (rc) return rc;
            { int rcw = wait_headers(); if (rcw) return rcw; }
            const uint32_t n_rest_before = n_rest;
            for (uint32_t i = 0; i < n_sc; ++i) {
                if (ix->h_res_hdr[sc_sel[i]].status != 0) {
                    rest[n_rest++] = sc_sel[i];
                    if (ix->env_trace_cert)
                        fprintf(stderr, "second chance failed: q=%u listed=%u (cap %d) tau2=%.7f\n", sc_sel[i], ix->h_res_hdr[sc_sel[i]].n_cand,
                                MQ_CAP2, sc_tau[i]);
                } else {
                    ++ix->stats.second_chance;
                }
            }
            ix->sc_success = 0.5f * ix->sc_success + 0.5f * (float)(n_sc - (n_rest - n_rest_before)) / (float)n_sc;
            std::sort(rest, rest + n_rest);
            n_sel = n_rest;
            for (uint32_t i = 0; i < n_rest; ++i) hsel[i] = rest[i];
        }
    }
    if (n_sel) {
        PB_HIP(hipMemcpyAsync(ix->d_qsel, hsel, n_sel * sizeof(uint32_t), hipMemcpyHostToDevice, ix->stream));
        int rc = run_exact(ix, n_sel, k);
        if (rc) return rc;
        { int rcw = wait_headers(); if (rcw) return rcw; }
        if (ix->opt_profile && ix->opt_path == 1) {
            int rc2 = account_profile(ix, n_sel, 1);
            if (rc2) return rc2;
        }
        ix->stats.fallback += n_sel;
    }
    ix->stats.queries += cq;
    return PB_OK;
}

// Concurrent-query path for a block of <= PIPE_Q queries: everything is staged and launched back to back, the
// host waits ONCE, then re-runs the (rare) uncertified queries through the exhaustive pass.  Results land in
// h_res_ids / h_res_dist / h_res_hdr slots [0, nq).
int search_block_multi(pb_index *ix, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist,
                       uint32_t *n_failed = nullptr) {
    const uint32_t d = ix->dim;
    uint8_t *hq = ix->h_pipe;
    QParams *hp = reinterpret_cast<QParams *>(ix->h_pipe + (size_t)PIPE_Q * d);
    ix->r_ids = ix->d_res_ids;
    ix->r_dist = ix->d_res_dist;
    ix->r_hdr = ix->d_res_hdr;
    memcpy(hq, queries, (size_t)nq * d);
    { int rcm = refresh_min_den(ix); if (rcm) return rcm; }
    make_qparams_batch(ix, hq, nq, k, max_dist, hp);
    PB_HIP(hipMemcpyAsync(ix->d_queries, hq, (size_t)nq * d, hipMemcpyHostToDevice, ix->stream));
    PB_HIP(hipMemcpyAsync(ix->d_qp, hp, (size_t)nq * sizeof(QParams), hipMemcpyHostToDevice, ix->stream));
    // the burst form queues survivors in LDS (1024 entries, drained between steps); a step of 128 rows x 512
    // queries yields ~65536 * target / N of them, so small tables (where the ~max(4k, 512) wanted candidates are a
    // large fraction of the rows) keep the per-chunk form
    const uint64_t burst_min_rows = 512ull * std::max<uint32_t>(4 * k, 512);
    if (nq > Q_CHUNK && !ix->opt_mq_per_chunk && ix->n_rows >= burst_min_rows) {
        int rc = run_multi_block(ix, nq, k);
        if (rc) return rc;
    } else {
        for (uint32_t base = 0; base < nq; base += Q_CHUNK) {
            int rc = run_multi(ix, std::min(Q_CHUNK, nq - base), k, base);
            if (rc) return rc;
        }
    }
    PB_HIP(hipMemcpyAsync(ix->h_res_hdr, ix->d_res_hdr, nq * sizeof(ResultHdr), hipMemcpyDeviceToHost, ix->stream));
    // only the first k of the PB_MAX_K result slots of each query travel (1.2 MB instead of 3 MB per 1024 queries)
    PB_HIP(hipMemcpy2DAsync(ix->h_res_ids, PB_MAX_K * sizeof(int64_t), ix->d_res_ids, PB_MAX_K * sizeof(int64_t), k * sizeof(int64_t), nq,
                            hipMemcpyDeviceToHost, ix->stream));
    PB_HIP(hipMemcpy2DAsync(ix->h_res_dist, PB_MAX_K * sizeof(float), ix->d_res_dist, PB_MAX_K * sizeof(float), k * sizeof(float), nq,
                            hipMemcpyDeviceToHost, ix->stream));
    PB_HIP(hipStreamSynchronize(ix->stream));
    if (ix->opt_profile) {
        int rc2 = account_profile(ix, 1, 1);
        if (rc2) return rc2;
    }
    std::vector<uint32_t> failed;
    for (uint32_t q = 0; q < nq; ++q)
        if (ix->h_res_hdr[q].status != 0) {
            failed.push_back(q);
            if (ix->env_trace_cert)
                fprintf(stderr, "cert fail (burst path): q=%u count=%u n_cand=%u o_max=%.7f thr0=%.7f\n", q, ix->h_res_hdr[q].count,
                        ix->h_res_hdr[q].n_cand, ix->h_res_hdr[q].o_max, hp[q].thr0);
        }
    if (n_failed) *n_failed = (uint32_t)failed.size();
    ix->stats.queries += nq;
    ix->stats.fast_path += nq - failed.size();
    ix->stats_multi += nq - failed.size();
    // Uncertified queries.  When the second chance is not worth trying (second_chance_pays) they all take the
    // exhaustive pass at once: their queries and parameters are still on the device from the burst, the kernels are
    // queued chunk after chunk with nothing waited for in between, and the final merge writes each result into ITS
    // slot of the pinned host arrays (the burst's results for the other queries are already there).
    const bool coalesced = ix->metric == 0 && ix->dim == 256 && !ix->env_exact_lane_rows;
    if (!failed.empty() && coalesced &&
        (ix->env_no_second_chance || ix->n_rows < 4096 || !second_chance_pays(ix, (uint32_t)std::min<size_t>(Q_CHUNK, failed.size())))) {
        PB_HIP(hipMemcpyAsync(ix->d_qsel, failed.data(), failed.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ix->stream));
        ix->r_ids = ix->h_res_ids;
        ix->r_dist = ix->h_res_dist;
        ix->r_hdr = ix->h_res_hdr;
        for (size_t f0 = 0; f0 < failed.size(); f0 += Q_CHUNK) {
            const uint32_t cq = (uint32_t)std::min<size_t>(Q_CHUNK, failed.size() - f0);
            int rc = run_exact(ix, cq, k, ix->d_qsel + f0);
            if (rc) return rc;
        }
        PB_HIP(hipStreamSynchronize(ix->stream));
        ix->stats.fallback += failed.size();
        return PB_OK;
    }
    // otherwise a chunk at a time through search_chunk (second chance, then the exhaustive pass; uses the chunk workspace at slot 0)
    std::vector<int64_t> keep_ids;
    std::vector<float> keep_dist;
    std::vector<ResultHdr> keep_hdr;
    for (size_t f0 = 0; f0 < failed.size(); f0 += Q_CHUNK) {
        const uint32_t cq = (uint32_t)std::min<size_t>(Q_CHUNK, failed.size() - f0);
        if (keep_hdr.empty()) {  // the fallback overwrites slots [0, cq): save the block's results first
            keep_ids.assign(ix->h_res_ids, ix->h_res_ids + (size_t)nq * PB_MAX_K);
            keep_dist.assign(ix->h_res_dist, ix->h_res_dist + (size_t)nq * PB_MAX_K);
            keep_hdr.assign(ix->h_res_hdr, ix->h_res_hdr + nq);
        }
        for (uint32_t i = 0; i < cq; ++i) memcpy(ix->h_stage + (size_t)i * d, queries + (size_t)failed[f0 + i] * d, d);
        const int saved = ix->opt_path;
        ix->opt_path = 1;
        const uint64_t q_before = ix->stats.queries;
        float ck_hint[Q_CHUNK], omax_hint[Q_CHUNK];  // the burst attempt's k-th exact cosines: the second chance starts from them
        uint32_t ncand_hint[Q_CHUNK];
        for (uint32_t i = 0; i < cq; ++i) {
            ck_hint[i] = keep_hdr[failed[f0 + i]].ck;
            omax_hint[i] = keep_hdr[failed[f0 + i]].o_max;
            ncand_hint[i] = keep_hdr[failed[f0 + i]].n_cand;
        }
        int rc = search_chunk(ix, cq, k, max_dist, ck_hint, ncand_hint, omax_hint);
        ix->opt_path = saved;
        ix->stats.queries = q_before;  // already counted above
        if (rc) return rc;
        PB_HIP(hipMemcpyAsync(ix->h_res_ids, ix->d_res_ids, (size_t)cq * PB_MAX_K * sizeof(int64_t), hipMemcpyDeviceToHost, ix->stream));
        PB_HIP(hipMemcpyAsync(ix->h_res_dist, ix->d_res_dist, (size_t)cq * PB_MAX_K * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
        PB_HIP(hipStreamSynchronize(ix->stream));
        for (uint32_t i = 0; i < cq; ++i) {
            const uint32_t q = failed[f0 + i];
            keep_hdr[q] = ix->h_res_hdr[i];
            memcpy(&keep_ids[(size_t)q * PB_MAX_K], ix->h_res_ids + (size_t)i * PB_MAX_K, PB_MAX_K * sizeof(int64_t));
            memcpy(&keep_dist[(size_t)q * PB_MAX_K], ix->h_res_dist + (size_t)i * PB_MAX_K, PB_MAX_K * sizeof(float));
        }
    }
    if (!keep_hdr.empty()) {
        memcpy(ix->h_res_ids, keep_ids.data(), keep_ids.size() * sizeof(int64_t));
        memcpy(ix->h_res_dist, keep_dist.data(), keep_dist.size() * sizeof(float));
        memcpy(ix->h_res_hdr, keep_hdr.data(), keep_hdr.size() * sizeof(ResultHdr));
    }
    return PB_OK;
}

// results to HOST buffers
int search_locked(pb_index *ix, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist, int64_t *out_ids,
                  float *out_dist, uint32_t *out_count) {
    const uint32_t d = ix->dim;
    if (ix->n_rows == 0) {
        for (uint32_t q = 0; q < nq; ++q) out_count[q] = 0;
        return PB_OK;
    }
    PB_CHECK(ix->n_rows < (1ull << 32), PB_ERR_CAPACITY, "more than 2^32 rows per shard are not supported");
    if (multi_eligible(ix, nq)) {
        for (uint32_t q0 = 0; q0 < nq; q0 += PIPE_Q) {
            const uint32_t cq = std::min(PIPE_Q, nq - q0);
            int rc = search_block_multi(ix, queries + (size_t)q0 * d, cq, k, max_dist);
            if (rc) return rc;
            for (uint32_t q = 0; q < cq; ++q) {
                const uint32_t c = ix->h_res_hdr[q].count;
                out_count[q0 + q] = c;
                memcpy(out_ids + (size_t)(q0 + q) * k, ix->h_res_ids + (size_t)q * PB_MAX_K, c * sizeof(int64_t));
                memcpy(out_dist + (size_t)(q0 + q) * k, ix->h_res_dist + (size_t)q * PB_MAX_K, c * sizeof(float));
            }
        }
        return PB_OK;
    }
    for (uint32_t q0 = 0; q0 < nq; q0 += Q_CHUNK) {
        const uint32_t cq = std::min(Q_CHUNK, nq - q0);
        memcpy(ix->h_stage, queries + (size_t)q0 * d, (size_t)cq * d);
        // results are written by the kernels into h_res_* (host_out): search_chunk's single wait is the call's only one
        int rc = search_chunk(ix, cq, k, max_dist, nullptr, nullptr, nullptr, true);
        if (rc) return rc;
        for (uint32_t q = 0; q < cq; ++q) {
            const uint32_t c = ix->h_res_hdr[q].count;
            out_count[q0 + q] = c;
            memcpy(out_ids + (size_t)(q0 + q) * k, ix->h_res_ids + (size_t)q * PB_MAX_K, c * sizeof(int64_t));
            memcpy(out_dist + (size_t)(q0 + q) * k, ix->h_res_dist + (size_t)q * PB_MAX_K, c * sizeof(float));
        }
    }
    return PB_OK;
}

// Results left in DEVICE memory.  `emit(q0, cq)` is called once the final results of queries [q0, q0 + cq) sit in
// d_res_ids / d_res_dist / d_res_hdr slots [0, cq) (stride PB_MAX_K); it queues a kernel on ix->stream that writes them
// wherever the caller wants them (the all-gather message, plain id / distance / count arrays).
template <class Emit>
int search_to_device(pb_index *ix, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist, Emit emit) {
    const uint32_t d = ix->dim;
    PB_CHECK(ix->n_rows < (1ull << 32), PB_ERR_CAPACITY, "more than 2^32 rows per shard are not supported");
    if (nq > Q_CHUNK && multi_eligible(ix, nq)) {
        // a burst: the concurrent-query path a block at a time; queries that needed the exhaustive pass were
        // patched in the host copies, so those blocks are written back before they are emitted
        for (uint32_t q0 = 0; q0 < nq; q0 += PIPE_Q) {
            const uint32_t cq = std::min(PIPE_Q, nq - q0);
            uint32_t n_failed = 0;
            int rc = search_block_multi(ix, queries + (size_t)q0 * d, cq, k, max_dist, &n_failed);
            if (rc) return rc;
            if (n_failed) {
                PB_HIP(hipMemcpyAsync(ix->d_res_ids, ix->h_res_ids, (size_t)cq * PB_MAX_K * sizeof(int64_t), hipMemcpyHostToDevice, ix->stream));
                PB_HIP(hipMemcpyAsync(ix->d_res_dist, ix->h_res_dist, (size_t)cq * PB_MAX_K * sizeof(float), hipMemcpyHostToDevice, ix->stream));
                PB_HIP(hipMemcpyAsync(ix->d_res_hdr, ix->h_res_hdr, cq * sizeof(ResultHdr), hipMemcpyHostToDevice, ix->stream));
            }
            emit(q0, cq);
            PB_HIP(hipGetLastError());
            PB_HIP(hipStreamSynchronize(ix->stream));  // the next block reuses the staging buffers
        }
        return PB_OK;
    }
    for (uint32_t q0 = 0; q0 < nq; q0 += Q_CHUNK) {
        const uint32_t cq = std::min(Q_CHUNK, nq - q0);
        memcpy(ix->h_stage, queries + (size_t)q0 * d, (size_t)cq * d);
        int rc = search_chunk(ix, cq, k, max_dist);
        if (rc) return rc;
        emit(q0, cq);
        PB_HIP(hipGetLastError());
    }
    PB_HIP(hipStreamSynchronize(ix->stream));
    return PB_OK;
}

// results packed for the all-gather: d_packed[q][0..k) ids, [k..2k) dist bits, [2k] count
int search_packed_locked(pb_index *ix, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist, int64_t *d_packed) {
    const size_t row = 2 * (size_t)k + 1;
    if (ix->n_rows == 0) {  // k_pack_results' encoding of "no results": id INT64_MAX, dist +inf, count 0
        hipLaunchKernelGGL(k_pack_results, dim3(nq), dim3(256), 0, ix->stream, (const int64_t *)nullptr, (const float *)nullptr,
                           (const ResultHdr *)nullptr, (uint32_t)PB_MAX_K, k, d_packed);
        PB_HIP(hipGetLastError());
        PB_HIP(hipStreamSynchronize(ix->stream));
        return PB_OK;
    }
    return search_to_device(ix, queries, nq, k, max_dist, [&](uint32_t q0, uint32_t cq) {
        hipLaunchKernelGGL(k_pack_results, dim3(cq), dim3(256), 0, ix->stream, ix->d_res_ids, ix->d_res_dist, ix->d_res_hdr,
                           (uint32_t)PB_MAX_K, k, d_packed + (size_t)q0 * row);
    });
}

// append rows that are already on the device (or host) at the tail; ids ascending and > last
int append_tail(pb_index *ix, const int64_t *ids, const uint8_t *rows, uint64_t n, hipMemcpyKind kind) {
    PB_CHECK(ix->n_rows + n <= ix->capacity, PB_ERR_CAPACITY, "index full: %llu + %llu > capacity %llu",
             (unsigned long long)ix->n_rows, (unsigned long long)n, (unsigned long long)ix->capacity);
    const size_t d = ix->dim;
    const bool async = ix->opt_append_async && kind == hipMemcpyDeviceToDevice;
    PB_HIP(hipMemcpyAsync(ix->d_rows + ix->n_rows * d, rows, n * d, kind, ix->stream));
    // PB_OPT_APPEND_ASYNC: the device-to-device form returns with its row copy and norms queued on the stream.  The
    // caller's `ids` array is NOT handed to an asynchronous copy (it is pageable memory the caller may free on return):
    // the ids are kept in h_ids and uploaded by refresh_min_den before the next reader of d_ids
    if (async) ix->ids_dirty_lo = std::min(ix->ids_dirty_lo, ix->n_rows);
    else PB_HIP(hipMemcpyAsync(ix->d_ids + ix->n_rows, ids, n * sizeof(int64_t), hipMemcpyHostToDevice, ix->stream));
    int rc = launch_norms(ix, ix->n_rows, n);
    if (rc) return rc;
    if (!async) PB_HIP(hipStreamSynchronize(ix->stream));
    ix->h_ids.insert(ix->h_ids.end(), ids, ids + n);
    ix->n_rows += n;
    return PB_OK;
}

// Out-of-order inserts (ids at or below the current maximum): the new pairs of a call are first appended at the tail in
// call order (append_unsorted), then ONE permutation pass puts the affected suffix of every per-row array back into
// ascending image_id order -- O(suffix) device traffic per CALL.  (Round 1 shifted the tail once per inserted row: a bulk
// load in descending order was quadratic.)
int merge_unsorted_tail(pb_index *ix, uint64_t n_sorted) {
    const uint64_t n_all = ix->n_rows;
    if (n_sorted >= n_all) return PB_OK;
    // new ids with their tail positions, sorted; the affected range starts at the first stored id above the smallest new one
    std::vector<std::pair<int64_t, uint32_t>> fresh;
    fresh.reserve(n_all - n_sorted);
    for (uint64_t p = n_sorted; p < n_all; ++p) fresh.emplace_back(ix->h_ids[p], (uint32_t)p);
    std::sort(fresh.begin(), fresh.end());
    const uint64_t lo = (uint64_t)(std::lower_bound(ix->h_ids.begin(), ix->h_ids.begin() + (ptrdiff_t)n_sorted, fresh.front().first) - ix->h_ids.begin());
    const uint64_t n_aff = n_all - lo;
    std::vector<uint32_t> perm(n_aff);   // perm[i] = current position (relative to lo) of the row that belongs at lo + i
    std::vector<int64_t> merged(n_aff);
    uint64_t a = lo, f = 0, o = 0;
    while (a < n_sorted || f < fresh.size()) {
        const bool take_old = f == fresh.size() || (a < n_sorted && ix->h_ids[a] < fresh[f].first);
        if (take_old) {
            perm[o] = (uint32_t)(a - lo);
            merged[o++] = ix->h_ids[a++];
        } else {
            perm[o] = fresh[f].second - (uint32_t)lo;
            merged[o++] = fresh[f++].first;
        }
    }
    const size_t d = ix->dim;
    uint32_t *d_perm = nullptr;
    uint8_t *d_tmp = nullptr;
    auto body = [&]() -> int {
        PB_HIP(hipMalloc(&d_perm, n_aff * sizeof(uint32_t)));
        PB_HIP(hipMalloc(&d_tmp, n_aff * std::max<size_t>(d, sizeof(int64_t))));
        PB_HIP(hipMemcpyAsync(d_perm, perm.data(), n_aff * sizeof(uint32_t), hipMemcpyHostToDevice, ix->stream));
        auto permute = [&](void *base, size_t elt) -> int {
            uint8_t *p = static_cast<uint8_t *>(base) + lo * elt;
            const uint64_t pieces = n_aff * (elt % 16 == 0 ? elt / 16 : (elt % 4 == 0 ? elt / 4 : elt));
            const int grid = (int)std::min<uint64_t>((pieces + 255) / 256, (uint64_t)ix->n_cu * 16);
            hipLaunchKernelGGL(k_gather_elts, dim3(grid), dim3(256), 0, ix->stream, p, d_perm, n_aff, (uint32_t)elt, d_tmp);
            PB_HIP(hipGetLastError());
            PB_HIP(hipMemcpyAsync(p, d_tmp, n_aff * elt, hipMemcpyDeviceToDevice, ix->stream));
            return PB_OK;
        };
        int rc = permute(ix->d_rows, d);
        if (!rc) rc = permute(ix->d_ids, sizeof(int64_t));
        if (!rc) rc = permute(ix->d_norms, sizeof(float));
        if (!rc) rc = permute(ix->d_sumb, sizeof(int32_t));
        if (!rc) rc = permute(ix->d_denb, sizeof(int32_t));
        if (rc) return rc;
        PB_HIP(hipStreamSynchronize(ix->stream));
        return PB_OK;
    };
    const int rc = body();
    (void)hipFree(d_perm);
    (void)hipFree(d_tmp);
    if (rc) return rc;
    std::copy(merged.begin(), merged.end(), ix->h_ids.begin() + (ptrdiff_t)lo);
    return PB_OK;
}

}  // namespace

extern "C" {

const char *pb_last_error(void) { return pb::tls_error(); }
int pb_version(void) { return 100; }

int pb_device_count(int *n) {
    PB_CHECK(n, PB_ERR_INVALID, "pb_device_count: null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return pb::fail(PB_ERR_HIP, "hipGetDeviceCount -> %s", hipGetErrorString(e));
    }
    *n = c;
    return PB_OK;
}

int pb_index_create(pb_index **out, int device, uint32_t dim, uint64_t capacity_rows) {
    PB_CHECK(out, PB_ERR_INVALID, "pb_index_create: null out pointer");
    *out = nullptr;
    PB_CHECK(dim >= 1 && dim <= 1024, PB_ERR_INVALID, "pb_index_create: dim %u outside 1..1024", dim);
    PB_CHECK(capacity_rows >= 1, PB_ERR_INVALID, "pb_index_create: capacity_rows must be >= 1");
    int n_dev = 0;
    PB_HIP(hipGetDeviceCount(&n_dev));
    PB_CHECK(device >= 0 && device < n_dev, PB_ERR_INVALID, "pb_index_create: device %d of %d", device, n_dev);
    pb::DeviceGuard guard(device);
    PB_CHECK(guard.ok, PB_ERR_HIP, "hipSetDevice(%d) failed", device);
    pb_index *ix = new (std::nothrow) pb_index();
    PB_CHECK(ix, PB_ERR_NOMEM, "out of host memory");
    ix->device = device;
    ix->dim = dim;
    ix->capacity = capacity_rows;
    ix->env_no_fuse = getenv("PB_FUSE") == nullptr;
    ix->env_no_seed = getenv("PB_NO_SEED") != nullptr;
    ix->env_seed_always = getenv("PB_SEED") != nullptr;
    ix->env_no_second_chance = getenv("PB_NO_SECOND_CHANCE") != nullptr;  // diagnostics switches
    ix->env_trace_cert = getenv("PB_TRACE_CERT") != nullptr;
    ix->env_exact_lane_rows = getenv("PB_EXACT_LANE_ROWS") != nullptr;
    ix->env_static_tail = getenv("PB_STATIC_TAIL") != nullptr;
    ix->env_force_tickets = getenv("PB_FORCE_TAIL_TICKETS") != nullptr;
    ix->env_steal_lead = getenv("PB_STEAL_LEAD") ? (uint32_t)atoi(getenv("PB_STEAL_LEAD")) : 0u;
    ix->env_force_steal = getenv("PB_FORCE_STEAL") != nullptr;
    ix->env_no_poll = getenv("PB_NO_POLL") != nullptr;
    if (getenv("PB_POLL_TIMEOUT_US")) ix->poll_timeout_us = std::max<int64_t>(0, atoll(getenv("PB_POLL_TIMEOUT_US")));
    ix->env_loop_static = getenv("PB_LOOP_STATIC") != nullptr;
    make_lut(ix->lut);
    auto body = [&]() -> int {
        hipDeviceProp_t prop;
        PB_HIP(hipGetDeviceProperties(&prop, device));
        ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        {   // the index's own stream at the device's highest priority: its launches are small (an append's copy + norms, a query's
            // filter pass) and must not queue behind the embedders' forward passes when a crawler and queries share the GPU
            int pr_lo = 0, pr_hi = 0;
            PB_HIP(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
            PB_HIP(hipStreamCreateWithPriority(&ix->own_stream, hipStreamNonBlocking, pr_hi));
        }
        ix->stream = ix->own_stream;
        PB_HIP(hipEventCreate(&ix->ev0));
        PB_HIP(hipEventCreate(&ix->ev1));
        // +64 rows of slack: the filter pass clamps, never reads past n_rows-1, but keep tail loads in-bounds
        PB_HIP(hipMalloc(&ix->d_ids, capacity_rows * sizeof(int64_t)));
        PB_HIP(hipMalloc(&ix->d_norms, capacity_rows * sizeof(float)));
        PB_HIP(hipMalloc(&ix->d_sumb, (capacity_rows + 32) * sizeof(int32_t)));
        PB_HIP(hipMalloc(&ix->d_denb, (capacity_rows + 32) * sizeof(int32_t)));
        PB_HIP(hipMemset(ix->d_sumb, 0, (capacity_rows + 32) * sizeof(int32_t)));
        PB_HIP(hipMemset(ix->d_denb, 0, (capacity_rows + 32) * sizeof(int32_t)));
        PB_HIP(hipMalloc(&ix->d_min_den, sizeof(int32_t)));
        PB_HIP(hipMemcpy(ix->d_min_den, &ix->min_den_b, sizeof(int32_t), hipMemcpyHostToDevice));
        PB_HIP(hipMalloc(&ix->d_lut, 256 * sizeof(float)));
        PB_HIP(hipMemcpy(ix->d_lut, ix->lut, 256 * sizeof(float), hipMemcpyHostToDevice));
        { int rcw = alloc_workspace(ix); if (rcw) return rcw; }
        {   // The table: ONE allocation, wherever the driver puts it.  (Round 4 found "slow tables" -- 23.3-23.6 ms per 64 passes over
            // some, 22.7 over others -- read them as a property of the buffer's placement and built a best-of-n allocation behind
            // PB_INDEX_PLACEMENT_TRIES.  Round 5 timed ONE table again and again from process start (profiles/placement_timeline.py,
            // profiles/r05_placement.txt): a table is slow WHILE the kernel driver scrubs memory some process has just released -- 150 GB
            // written and freed make the next two seconds 2.5-6.5 % slower, for every table, and then the same table runs at 22.7 -- and
            // the candidates that the best-of-n probe freed were themselves such a release.  The knob is gone; bench.py waits for a
            // settled step time before its warm-up steps instead.)
            const size_t bytes = (capacity_rows + 64) * (size_t)dim;
            if (hipMalloc(&ix->d_rows, bytes) != hipSuccess) {
                (void)hipGetLastError();
                return pb::fail(PB_ERR_HIP, "pb_index_create: hipMalloc of %zu bytes for the table failed", bytes);
            }
        }
        return PB_OK;
    };
    int rc = body();
    if (rc) {
        free_all(ix);
        delete ix;
        return rc;
    }
    *out = ix;
    return PB_OK;
}

int pb_index_create_metric(pb_index **out, int device, uint32_t dim, uint64_t capacity_rows, int metric) {
    PB_CHECK(metric >= PB_METRIC_COSINE && metric <= PB_METRIC_HAMMING, PB_ERR_INVALID, "pb_index_create_metric: metric %d", metric);
    int rc = pb_index_create(out, device, dim, capacity_rows);
    if (rc) return rc;
    (*out)->metric = metric;
    return PB_OK;
}

int pb_index_destroy(pb_index *ix) {
    if (!ix) return PB_OK;
    {
        pb::DeviceGuard guard(ix->device);
        (void)hipStreamSynchronize(ix->stream);
        free_all(ix);
    }
    delete ix;
    return PB_OK;
}

int pb_index_size(const pb_index *ix, uint64_t *n_rows) {
    PB_CHECK(ix && n_rows, PB_ERR_INVALID, "pb_index_size: null pointer");
    std::lock_guard<std::mutex> lock(ix->mu);
    *n_rows = ix->n_rows;
    return PB_OK;
}

int pb_index_contains(const pb_index *ix, int64_t image_id, int *found) {
    PB_CHECK(ix && found, PB_ERR_INVALID, "pb_index_contains: null pointer");
    std::lock_guard<std::mutex> lock(ix->mu);
    *found = std::binary_search(ix->h_ids.begin(), ix->h_ids.end(), image_id) ? 1 : 0;
    return PB_OK;
}

int pb_index_dim(const pb_index *ix, uint32_t *dim) {
    PB_CHECK(ix && dim, PB_ERR_INVALID, "pb_index_dim: null pointer");
    *dim = ix->dim;
    return PB_OK;
}

int pb_index_append(pb_index *ix, const int64_t *image_ids, const uint8_t *rows, uint64_t n, uint64_t *n_inserted) {
    PB_CHECK(ix, PB_ERR_INVALID, "pb_index_append: null index");
    PB_CHECK(n == 0 || (image_ids && rows), PB_ERR_INVALID, "pb_index_append: null ids/rows");
    std::lock_guard<std::mutex> lock(ix->mu);
    pb::DeviceGuard guard(ix->device);
    if (n_inserted) *n_inserted = 0;
    if (n == 0) return PB_OK;
    const size_t d = ix->dim;
    // common case (ids are fresh last_insert_rowid() values, engine.rs:233): strictly ascending and beyond everything stored
    const int64_t last = ix->h_ids.empty() ? std::numeric_limits<int64_t>::min() : ix->h_ids.back();
    bool ascending = ix->h_ids.empty() ? true : image_ids[0] > last;
    for (uint64_t i = 1; i < n && ascending; ++i) ascending = image_ids[i] > image_ids[i - 1];
    if (ascending) {
        const uint64_t room = ix->capacity - ix->n_rows, take = std::min<uint64_t>(room, n);
        if (take) {
            int rc = append_tail(ix, image_ids, rows, take, hipMemcpyHostToDevice);
            if (rc) return rc;
        }
        if (n_inserted) *n_inserted = take;
        PB_CHECK(take == n, PB_ERR_CAPACITY, "index full: capacity %llu rows (%llu of this call stored)", (unsigned long long)ix->capacity,
                 (unsigned long long)take);
        return PB_OK;
    }
    // general case -- INSERT OR IGNORE: a pair whose image_id is stored already, or appeared earlier in this call, is
    // skipped (first write wins); the others are appended in call order and merged into place in one pass
    { int rcm = refresh_min_den(ix); if (rcm) return rcm; }  // d_ids is about to be permuted: pending uploads first
    std::unordered_set<int64_t> seen;
    std::vector<int64_t> new_ids;
    std::vector<uint8_t> new_rows;
    for (uint64_t i = 0; i < n; ++i) {
        const int64_t id = image_ids[i];
        if (std::binary_search(ix->h_ids.begin(), ix->h_ids.end(), id) || !seen.insert(id).second) continue;
        new_ids.push_back(id);
        new_rows.insert(new_rows.end(), rows + i * d, rows + (i + 1) * d);
    }
    const uint64_t room = ix->capacity - ix->n_rows, take = std::min<uint64_t>(room, new_ids.size());
    const uint64_t n_sorted = ix->n_rows;
    if (take) {
        int rc = append_tail(ix, new_ids.data(), new_rows.data(), take, hipMemcpyHostToDevice);  // unsorted for the moment
        if (!rc) rc = merge_unsorted_tail(ix, n_sorted);
        if (rc) return rc;
    }
    if (n_inserted) *n_inserted = take;
    PB_CHECK(take == new_ids.size(), PB_ERR_CAPACITY, "index full: capacity %llu rows (%llu of this call stored)",
             (unsigned long long)ix->capacity, (unsigned long long)take);
    return PB_OK;
}

int pb_index_append_device(pb_index *ix, const int64_t *image_ids, const uint8_t *d_rows, uint64_t n) {
    PB_CHECK(ix, PB_ERR_INVALID, "pb_index_append_device: null index");
    PB_CHECK(n == 0 || (image_ids && d_rows), PB_ERR_INVALID, "pb_index_append_device: null ids/rows");
    if (n == 0) return PB_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    pb::DeviceGuard guard(ix->device);
    // the embed stage hands over fresh images: ids strictly ascending and beyond everything stored (anything else
    // -- an update, an out-of-order id -- goes through pb_index_append, which implements INSERT OR IGNORE)
    int64_t prev = ix->h_ids.empty() ? std::numeric_limits<int64_t>::min() : ix->h_ids.back();
    for (uint64_t i = 0; i < n; ++i) {
        PB_CHECK(image_ids[i] > prev, PB_ERR_INVALID, "pb_index_append_device: image_ids must be strictly ascending and greater than "
                 "every stored id (id %lld at position %llu)", (long long)image_ids[i], (unsigned long long)i);
        prev = image_ids[i];
    }
    return append_tail(ix, image_ids, d_rows, n, hipMemcpyDeviceToDevice);
}

int pb_index_load(pb_index *ix, const int64_t *image_ids, const uint8_t *rows, uint64_t n) {
    PB_CHECK(ix, PB_ERR_INVALID, "pb_index_load: null index");
    PB_CHECK(n == 0 || (image_ids && rows), PB_ERR_INVALID, "pb_index_load: null ids/rows");
    for (uint64_t i = 1; i < n; ++i)
        PB_CHECK(image_ids[i] > image_ids[i - 1], PB_ERR_INVALID, "pb_index_load: image_ids must be strictly increasing (row %llu)",
                 (unsigned long long)i);
    std::lock_guard<std::mutex> lock(ix->mu);
    pb::DeviceGuard guard(ix->device);
    PB_CHECK(n <= ix->capacity, PB_ERR_CAPACITY, "pb_index_load: %llu rows > capacity %llu", (unsigned long long)n,
             (unsigned long long)ix->capacity);
    PB_HIP(hipStreamSynchronize(ix->stream));
    ix->n_rows = 0;
    ix->h_ids.clear();
    ix->ids_dirty_lo = UINT64_MAX;
    // the error margin's minimum belongs to the rows that are being replaced
    ix->min_den_b = 0x7FFFFFFF;
    ix->min_dirty = false;
    PB_HIP(hipMemcpy(ix->d_min_den, &ix->min_den_b, sizeof(int32_t), hipMemcpyHostToDevice));
    if (n == 0) return PB_OK;
    return append_tail(ix, image_ids, rows, n, hipMemcpyHostToDevice);
}

int pb_index_fill_synthetic(pb_index *ix, uint64_t seed, uint64_t first_row, uint64_t n, int64_t first_id) {
    PB_CHECK(ix, PB_ERR_INVALID, "pb_index_fill_synthetic: null index");
    std::lock_guard<std::mutex> lock(ix->mu);
    pb::DeviceGuard guard(ix->device);
    PB_CHECK(ix->n_rows + n <= ix->capacity, PB_ERR_CAPACITY, "index full");
    PB_CHECK(ix->h_ids.empty() || first_id > ix->h_ids.back(), PB_ERR_INVALID, "first_id must exceed every stored image_id");
    const size_t d = ix->dim;
    PB_CHECK((first_row * d) % 8 == 0 && (ix->n_rows * d) % 8 == 0 && (n * d) % 8 == 0, PB_ERR_INVALID,
             "synthetic fill needs 8-byte aligned row ranges");
    if (n == 0) return PB_OK;
    const uint64_t n_words = n * d / 8;
    const int block = 256;
    const int grid = (int)std::min<uint64_t>((n_words + block - 1) / block, (uint64_t)ix->n_cu * 16);
    hipLaunchKernelGGL(k_fill_synth, dim3(grid), dim3(block), 0, ix->stream, seed, first_row * d / 8, n_words,
                       reinterpret_cast<uint64_t *>(ix->d_rows + ix->n_rows * d));
    PB_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_iota_ids, dim3((int)std::min<uint64_t>((n + block - 1) / block, 4096)), dim3(block), 0, ix->stream,
                       first_id, n, ix->d_ids + ix->n_rows);
    PB_HIP(hipGetLastError());
    int rc = launch_norms(ix, ix->n_rows, n);
    if (rc) return rc;
    PB_HIP(hipStreamSynchronize(ix->stream));
    const size_t old = ix->h_ids.size();
    ix->h_ids.resize(old + n);
    for (uint64_t i = 0; i < n; ++i) ix->h_ids[old + i] = first_id + (int64_t)i;
    ix->n_rows += n;
    return PB_OK;
}

int pb_index_read(const pb_index *ix, uint64_t first, uint64_t n, int64_t *image_ids, uint8_t *rows) {
    PB_CHECK(ix, PB_ERR_INVALID, "pb_index_read: null index");
    std::lock_guard<std::mutex> lock(ix->mu);
    pb::DeviceGuard guard(ix->device);
    PB_CHECK(first <= ix->n_rows && n <= ix->n_rows - first, PB_ERR_INVALID, "pb_index_read: range [%llu, %llu) beyond %llu rows",
             (unsigned long long)first, (unsigned long long)(first + n), (unsigned long long)ix->n_rows);
    PB_HIP(hipStreamSynchronize(ix->stream));
    if (image_ids) memcpy(image_ids, ix->h_ids.data() + first, n * sizeof(int64_t));
    if (rows && n) PB_HIP(hipMemcpy(rows, ix->d_rows + first * ix->dim, n * (size_t)ix->dim, hipMemcpyDeviceToHost));
    return PB_OK;
}

int pb_index_search(pb_index *ix, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist, int64_t *out_ids,
                    float *out_dist, uint32_t *out_count) {
    PB_CHECK(ix, PB_ERR_INVALID, "pb_index_search: null index");
    PB_CHECK(k >= 1 && k <= PB_MAX_K, PB_ERR_INVALID, "pb_index_search: k = %u outside 1..%u", k, PB_MAX_K);
    PB_CHECK(nq == 0 || (queries && out_ids && out_dist && out_count), PB_ERR_INVALID, "pb_index_search: null buffer");
    if (nq == 0) return PB_OK;
    PB_CT_BEGIN();
    std::lock_guard<std::mutex> lock(ix->mu);
    pb::DeviceGuard guard(ix->device);
    PB_CT(0);
    const int rc_s = search_locked(ix, queries, nq, k, max_dist, out_ids, out_dist, out_count);
    PB_CT(6);
#ifdef PB_CALL_TRACE
    if (++g_ct_n % 64 == 0) {
        fprintf(stderr, "call trace (us, mean of 64): device guard %.2f | qparams %.2f | filter launched %.2f | select launched %.2f | results seen %.2f | return %.2f\n",
                g_ct_sum[0] / 64, g_ct_sum[1] / 64, g_ct_sum[2] / 64, g_ct_sum[3] / 64, g_ct_sum[4] / 64, g_ct_sum[6] / 64);
        for (double &x : g_ct_sum) x = 0;
    }
#endif
    return rc_s;
}

int pb_index_search_device(pb_index *ix, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist,
                           int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count) {
    PB_CHECK(ix, PB_ERR_INVALID, "pb_index_search_device: null index");
    PB_CHECK(k >= 1 && k <= PB_MAX_K, PB_ERR_INVALID, "pb_index_search_device: k = %u outside 1..%u", k, PB_MAX_K);
    PB_CHECK(nq == 0 || (queries && d_out_ids && d_out_dist && d_out_count), PB_ERR_INVALID, "null buffer");
    if (nq == 0) return PB_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    pb::DeviceGuard guard(ix->device);
    if (ix->n_rows == 0) {
        hipLaunchKernelGGL(k_export_results, dim3(nq), dim3(256), 0, ix->stream, (const int64_t *)nullptr, (const float *)nullptr,
                           (const ResultHdr *)nullptr, (uint32_t)PB_MAX_K, k, d_out_ids, d_out_dist, d_out_count);
        PB_HIP(hipGetLastError());
        PB_HIP(hipStreamSynchronize(ix->stream));
        return PB_OK;
    }
    // device to device: the results never visit the host (the certificate headers do, 20 bytes per query)
    return search_to_device(ix, queries, nq, k, max_dist, [&](uint32_t q0, uint32_t cq) {
        hipLaunchKernelGGL(k_export_results, dim3(cq), dim3(256), 0, ix->stream, ix->d_res_ids, ix->d_res_dist, ix->d_res_hdr,
                           (uint32_t)PB_MAX_K, k, d_out_ids + (size_t)q0 * k, d_out_dist + (size_t)q0 * k, d_out_count + q0);
    });
}

int pb_index_search_packed(pb_index *ix, const uint8_t *queries, uint32_t nq, uint32_t k, double max_dist, int64_t *d_packed) {
    PB_CHECK(ix, PB_ERR_INVALID, "pb_index_search_packed: null index");
    PB_CHECK(k >= 1 && k <= PB_MAX_K, PB_ERR_INVALID, "pb_index_search_packed: k = %u outside 1..%u", k, PB_MAX_K);
    PB_CHECK(nq == 0 || (queries && d_packed), PB_ERR_INVALID, "pb_index_search_packed: null buffer");
    if (nq == 0) return PB_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    pb::DeviceGuard guard(ix->device);
    return search_packed_locked(ix, queries, nq, k, max_dist, d_packed);
}

int pb_index_set_option(pb_index *ix, int option, int64_t value) {
    PB_CHECK(ix, PB_ERR_INVALID, "pb_index_set_option: null index");
    std::lock_guard<std::mutex> lock(ix->mu);
    switch (option) {
        case PB_OPT_SEARCH_PATH:
            PB_CHECK(value >= 0 && value <= 3, PB_ERR_INVALID, "PB_OPT_SEARCH_PATH: 0..3");
            ix->opt_path = (int)value;
            return PB_OK;
        case PB_OPT_PROFILE:
            ix->opt_profile = value != 0;
            return PB_OK;
        case PB_OPT_STREAM:
            ix->stream = value ? reinterpret_cast<hipStream_t>(value) : ix->own_stream;
            return PB_OK;
        case PB_OPT_SCAN_VARIANT:
            PB_CHECK(value >= 0 && value <= 15 && ((value >> 1) & 3) != 3, PB_ERR_INVALID, "PB_OPT_SCAN_VARIANT: bits 0-3, loads-in-flight field 0..2");
            ix->opt_variant = (int)value;
            return PB_OK;
        case PB_OPT_SCAN_WG_PER_CU:
            PB_CHECK(value >= 1 && value <= 8, PB_ERR_INVALID, "workgroups per CU: 1..8");
            ix->opt_wg_per_cu = (int)value;
            return PB_OK;
        case PB_OPT_MQ_MIN_QUERIES:
            PB_CHECK(value >= 1, PB_ERR_INVALID, "min queries >= 1");
            ix->opt_mq_min_queries = (int)value;
            return PB_OK;
        case PB_OPT_MQ_WG_PER_CU:
            PB_CHECK(value >= 1 && value <= 8, PB_ERR_INVALID, "1..8");
            ix->opt_mq_wg_per_cu = (int)value;
            return PB_OK;
        case PB_OPT_MQ_PER_CHUNK:
            PB_CHECK(value == 0 || value == 1, PB_ERR_INVALID, "0 or 1");
            ix->opt_mq_per_chunk = (int)value;
            return PB_OK;
        case PB_OPT_SCAN_LAUNCH:
            PB_CHECK(value >= 0 && value <= 2, PB_ERR_INVALID, "PB_OPT_SCAN_LAUNCH: 0, 1 or 2");
            ix->opt_mode = (int)value;
            return PB_OK;
        case PB_OPT_APPEND_ASYNC:
            ix->opt_append_async = value != 0;
            return PB_OK;
        case PB_OPT_SECOND_CHANCE:
            PB_CHECK(value >= 0 && value <= 2, PB_ERR_INVALID, "PB_OPT_SECOND_CHANCE: 0 (cost model), 1 (always) or 2 (never)");
            ix->opt_second_chance = (int)value;
            ix->sc_success = 1.0f;
            ix->sc_skipped = 0;
            return PB_OK;
        case PB_OPT_EXACT_QN:
            PB_CHECK(value == 0 || value == 1 || value == 2 || value == 4, PB_ERR_INVALID, "PB_OPT_EXACT_QN: 0 (auto), 1, 2 or 4");
            ix->opt_exact_qn = (int)value;
            return PB_OK;
        case PB_OPT_SCAN_GRID:
            PB_CHECK(value >= 0 && value <= (int64_t)F_MAX_WG, PB_ERR_INVALID, "PB_OPT_SCAN_GRID: 0..%d workgroups", (int)F_MAX_WG);
            ix->opt_grid = (int)value;
            return PB_OK;
        case PB_OPT_SCAN_WAVES:
            PB_CHECK(value == 16 || value == 8 || value == 4, PB_ERR_INVALID, "waves per workgroup: 16, 8 or 4");
            ix->opt_waves = (int)value;
            return PB_OK;
        default:
            return pb::fail(PB_ERR_INVALID, "pb_index_set_option: unknown option %d", option);
    }
}

int pb_index_get_stats(pb_index *ix, pb_scan_stats *out, int reset) {
    PB_CHECK(ix && out, PB_ERR_INVALID, "pb_index_get_stats: null pointer");
    std::lock_guard<std::mutex> lock(ix->mu);
    *out = ix->stats;
    if (reset) ix->stats = pb_scan_stats{};
    return PB_OK;
}

// Host-side G-way merge (after the RCCL all-gather of per-shard results).  Pure C++: runs anywhere.
int pb_topk_merge(const int64_t *ids, const float *dist, const uint32_t *counts, uint32_t n_lists, uint32_t stride,
                  uint32_t k, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    PB_CHECK(out_count, PB_ERR_INVALID, "pb_topk_merge: null out_count");
    PB_CHECK(n_lists == 0 || (ids && dist && counts), PB_ERR_INVALID, "pb_topk_merge: null input");
    PB_CHECK(k == 0 || (out_ids && out_dist), PB_ERR_INVALID, "pb_topk_merge: null output");
    std::vector<uint32_t> pos(n_lists, 0);
    uint32_t n = 0;
    while (n < k) {
        int best = -1;
        for (uint32_t g = 0; g < n_lists; ++g) {
            if (pos[g] >= counts[g] || pos[g] >= stride) continue;
            const size_t i = (size_t)g * stride + pos[g];
            if (best < 0) {
                best = (int)g;
                continue;
            }
            const size_t b = (size_t)best * stride + pos[best];
            if (dist[i] < dist[b] || (dist[i] == dist[b] && ids[i] < ids[b])) best = (int)g;
        }
        if (best < 0) break;
        const size_t b = (size_t)best * stride + pos[best];
        out_ids[n] = ids[b];
        out_dist[n] = dist[b];
        ++pos[best];
        ++n;
    }
    *out_count = n;
    return PB_OK;
}

// merge of all-gathered packed results (HOST memory): gathered[g][q][2k+1] as written by pb_index_search_packed
int pb_topk_merge_packed(const int64_t *gathered, uint32_t n_lists, uint32_t nq, uint32_t k, int64_t *out_ids, float *out_dist,
                         uint32_t *out_count) {
    PB_CHECK(nq == 0 || (gathered && out_ids && out_dist && out_count), PB_ERR_INVALID, "pb_topk_merge_packed: null buffer");
    const size_t row = 2 * (size_t)k + 1;
    std::vector<uint32_t> pos(n_lists);
    for (uint32_t q = 0; q < nq; ++q) {
        std::fill(pos.begin(), pos.end(), 0u);
        uint32_t n = 0;
        while (n < k) {
            int best = -1;
            float bd = 0.f;
            int64_t bi = 0;
            for (uint32_t g = 0; g < n_lists; ++g) {
                const int64_t *p = gathered + ((size_t)g * nq + q) * row;
                const uint32_t cnt = (uint32_t)p[2 * k];
                if (pos[g] >= cnt || pos[g] >= k) continue;
                const uint32_t bits = (uint32_t)p[k + pos[g]];
                float dv;
                memcpy(&dv, &bits, 4);
                const int64_t iv = p[pos[g]];
                if (best < 0 || dv < bd || (dv == bd && iv < bi)) {
                    best = (int)g;
                    bd = dv;
                    bi = iv;
                }
            }
            if (best < 0) break;
            out_ids[(size_t)q * k + n] = bi;
            out_dist[(size_t)q * k + n] = bd;
            ++pos[best];
            ++n;
        }
        out_count[q] = n;
    }
    return PB_OK;
}

int pb_fill_synthetic(int device, uint64_t seed, uint64_t byte_offset, uint64_t nbytes, uint8_t *d_out) {
    PB_CHECK(d_out || nbytes == 0, PB_ERR_INVALID, "pb_fill_synthetic: null output");
    PB_CHECK(byte_offset % 8 == 0 && nbytes % 8 == 0, PB_ERR_INVALID, "pb_fill_synthetic: offset and size must be multiples of 8");
    if (nbytes == 0) return PB_OK;
    pb::DeviceGuard guard(device);
    PB_CHECK(guard.ok, PB_ERR_HIP, "hipSetDevice(%d) failed", device);
    const uint64_t n_words = nbytes / 8;
    const int block = 256;
    const int grid = (int)std::min<uint64_t>((n_words + block - 1) / block, 65536);
    hipLaunchKernelGGL(k_fill_synth, dim3(grid), dim3(block), 0, nullptr, seed, byte_offset / 8, n_words,
                       reinterpret_cast<uint64_t *>(d_out));
    PB_HIP(hipGetLastError());
    PB_HIP(hipStreamSynchronize(nullptr));
    return PB_OK;
}

int pb_fill_synthetic_images(int device, uint64_t seed, uint64_t start, uint64_t n, uint32_t h, uint32_t w, uint8_t *d_out) {
    PB_CHECK(d_out || n == 0, PB_ERR_INVALID, "pb_fill_synthetic_images: null output");
    const uint64_t per = (uint64_t)h * w * 3;
    PB_CHECK(per > 0 && per % 8 == 0, PB_ERR_INVALID, "pb_fill_synthetic_images: h*w*3 = %llu must be a positive multiple of 8",
             (unsigned long long)per);
    if (n == 0) return PB_OK;
    pb::DeviceGuard guard(device);
    PB_CHECK(guard.ok, PB_ERR_HIP, "hipSetDevice(%d) failed", device);
    const uint64_t n_words = n * per / 8;
    const int block = 256;
    const int grid = (int)std::min<uint64_t>((n_words + block - 1) / block, 65536);
    hipLaunchKernelGGL(k_fill_synth_images, dim3(grid), dim3(block), 0, nullptr, seed, start, per, n_words,
                       reinterpret_cast<uint64_t *>(d_out));
    PB_HIP(hipGetLastError());
    PB_HIP(hipStreamSynchronize(nullptr));
    return PB_OK;
}

int pb_fill_synthetic_scenes(int device, uint64_t seed, uint64_t start, uint64_t n, uint32_t h, uint32_t w, uint32_t grid, uint8_t *d_out) {
    PB_CHECK(d_out || n == 0, PB_ERR_INVALID, "pb_fill_synthetic_scenes: null output");
    const uint64_t per = (uint64_t)h * w * 3;
    PB_CHECK(per > 0 && per % 8 == 0, PB_ERR_INVALID, "pb_fill_synthetic_scenes: h*w*3 = %llu must be a positive multiple of 8",
             (unsigned long long)per);
    PB_CHECK(grid >= 1 && h % grid == 0 && w % grid == 0, PB_ERR_INVALID, "pb_fill_synthetic_scenes: %u x %u is not a multiple of the %u x %u grid", h, w,
             grid, grid);
    if (n == 0) return PB_OK;
    pb::DeviceGuard guard(device);
    PB_CHECK(guard.ok, PB_ERR_HIP, "hipSetDevice(%d) failed", device);
    const uint64_t n_words = n * per / 8;
    const int block = 256;
    const int blocks = (int)std::min<uint64_t>((n_words + block - 1) / block, 65536);
    hipLaunchKernelGGL(k_fill_synth_scenes, dim3(blocks), dim3(block), 0, nullptr, seed, start, per, n_words, w * 3, h / grid, w / grid, grid,
                       reinterpret_cast<uint64_t *>(d_out));
    PB_HIP(hipGetLastError());
    PB_HIP(hipStreamSynchronize(nullptr));
    return PB_OK;
}

#ifdef PB_MQ_STAMP
int pb_debug_mq_stamps(unsigned long long *out, int reset) {
    PB_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mq_stamp), 8 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[8] = {0};
        PB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mq_stamp), z, sizeof(z)));
    }
    return PB_OK;
}
#endif

#ifdef PB_SCAN_STAMP
// instrumented build only (profiles/scan_stamps.py): out_filter [F_MAX_WG * 16 * 8], out_sel [16], host_now: this clock now
int pb_debug_scan_stamps(unsigned long long *out_filter, unsigned long long *out_sel, int reset) {
    PB_HIP(hipDeviceSynchronize());
    PB_HIP(hipMemcpyFromSymbol(out_filter, HIP_SYMBOL(pbk::g_scan_stamp), sizeof(unsigned long long) * F_MAX_WG * 16 * 8));
    PB_HIP(hipMemcpyFromSymbol(out_sel, HIP_SYMBOL(pbk::g_sel_stamp), sizeof(unsigned long long) * 16));
    if (reset) {
        void *p = nullptr;
        PB_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(pbk::g_scan_stamp)));
        PB_HIP(hipMemset(p, 0, sizeof(unsigned long long) * F_MAX_WG * 16 * 8));
        PB_HIP(hipDeviceSynchronize());
    }
    return PB_OK;
}
#endif

}  // extern "C"
