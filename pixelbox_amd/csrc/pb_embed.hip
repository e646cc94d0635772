// pb_embed.hip -- host side of the embed half of the C ABI (include/pixelbox_hip.h): the batched
// replacement of `image_hashes::mlhash` (src/image_hashes/efficientnet.rs:31-42).  gfx950 only; no CPU
// fallback.  Weight blob: PBXW0001 (pixelbox_amd/weights.py) -- BN-folded EfficientNet-B0 features +
// Linear(1280, D) exactly as resources/train.py:30-46,167-174 exports them.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "pb_common.h"
#include "pb_embed_kernels.h"
#include "pb_front_band.h"
#include "pb_gemm_p3.h"
#include "pb_gemm_p3_launch.h"
#include "pb_block_small.h"

using namespace pbe;

namespace {

struct Stage {
    int expand, k, stride, cin, cout, repeats;
};
// torchvision efficientnet_b0 inverted-residual setting
const Stage STAGES[7] = {{1, 3, 1, 32, 16, 1},  {6, 3, 2, 16, 24, 2},   {6, 5, 2, 24, 40, 2},  {6, 3, 2, 40, 80, 3},
                         {6, 5, 1, 80, 112, 3}, {6, 5, 2, 112, 192, 4}, {6, 3, 1, 192, 320, 1}};

int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Gemm {  // 1x1 conv as GEMM
    int K = 0, N = 0, Kpad = 0, Npad = 0;
    float *wt = nullptr;    // [Kpad][Npad]
    float *bias = nullptr;  // [Npad]
    void *wt3 = nullptr;    // P3 layers (pb_gemm_p3.h): the weights as three bf16 planes in MFMA fragment order, [ceil(K / 32)][Npad / 16][3][64 lanes] x 16 B
    bool p3 = false;        // the layer's ARITHMETIC: products from bf16 pieces on the bf16 matrix cores (every kernel form of this layer, every
                            // batch size) instead of the f32 MFMA chain; decided by the layer's shape alone when the weights are loaded
    float *wt4 = nullptr;   // fragments by k-step for k_block_small: [K / 16][Npad / 16][kk * 16 + li][e]; null unless asked for
    float *wt2 = nullptr;   // fragment order for k_gemm_t: [chunk of 64 k][Npad / 16][kk][li][s][e] = w[64 chunk + 16 s + 4 kk + e][16 tile + li]; null when K % 16
};
struct Block {
    int cin, cout, e, sq, k, stride;
    bool has_expand, residual;
    Gemm expand, project;
    float *dw_w = nullptr, *dw_b = nullptr;            // [k*k][e], [e]
    f32x4 *dw_wq = nullptr;                            // the same taps per channel quad: [e / 4][k*k] float4 (k_front_band)
    float *dw_wc = nullptr;                            // the same taps per channel: [e][k*k rounded up to 4] (k_block_small)
    int sp = 0;                                        // sq rounded up to 8/16/32/48 (zero-padded rows)
    float *se_w1 = nullptr, *se_b1 = nullptr;          // [sp][e], [sp]
    float *se_w2t = nullptr, *se_b2 = nullptr;         // [sp][e], [e]
};

struct DwGeom {
    int roll;  // 0: strip kernel (k_dwconv), 1: rolling-window kernel (k_dwconv_roll), 2: whole map in LDS (k_dwconv_lds)
    int zsplit, cqpb, slots, n_tiles, strips_per_tile;  // for roll: slots = strips_x, strips_per_tile = rows per band
};

}  // namespace

struct pb_embedder {
    int device = 0;
    uint32_t H = 0, W = 0, D = 0, max_batch = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<void *> allocs;
    uint16_t *stem_w3 = nullptr;  // the stem as bf16 pieces of w / 255 in MFMA fragment order (stem_tile, pb_embed_kernels.h): [2 channel tiles][3 planes][64 lanes][8]
    float *stem_b = nullptr;     // [32]
    std::vector<Block> blocks;
    Gemm head;
    Gemm fc;  // Linear(1280, D) as a GEMM over the pooled features
    // workspace
    uint8_t *d_img = nullptr;
    float *buf_x[2] = {nullptr, nullptr};
    float *buf_e = nullptr, *buf_dw = nullptr, *buf_gate = nullptr, *buf_pool = nullptr;
    long long *buf_part = nullptr;  // SE pooling partial sums, 2^-24 fixed point; buf_part[-1] holds the address of *h_range (se_range_check)
    unsigned *h_range = nullptr;    // pinned host word a kernel raises when an activation leaves the fixed-point domain (|x| >= 128)
    // pb_mlhash / pb_embed_batch with ONE image: input copy + the ~50 launches of the forward + output copies as one replayed hipGraph
    // (a dependent launch costs 3.1 us on a stream and 1.9 us as a graph node: profiles/micro/launch_floor.hip).  Captured at the third
    // one-image call whose predecessor ran no timing loop; dropped when the stream, the picks or anything else it froze changes.
    hipGraphExec_t g1_exec[2] = {nullptr, nullptr};  // [0] with the input copy (pb_mlhash), [1] without (pb_mlhash_image: the resize kernel has left the image in d_img)
    hipStream_t g1_stream = nullptr;   // the stream they were captured on
    uint8_t *g1_in = nullptr;          // pinned: the image
    uint8_t *g1_out_u8 = nullptr;      // pinned: D bytes
    float *g1_out_f32 = nullptr;       // pinned: D floats
    int g1_quiet_calls = 0;            // consecutive one-image calls that ran no timing loop
    unsigned long tune_runs = 0;       // timing loops entered so far (TuneTimer)
    bool g1_off = false;               // PB_NO_GRAPH=1, or a capture failed once
    float *d_out_f32 = nullptr;
    uint8_t *d_out_u8 = nullptr;
    int n_cu = 256;
    hipEvent_t tune_e0 = nullptr, tune_e1 = nullptr;  // timing pair of the per-layer candidate measurements
    // pb_embed_batch's pipeline over chunks: input copy, forward and output copy of consecutive chunks overlap (two slots)
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
    uint8_t *d_img_b = nullptr, *d_out_u8_b = nullptr;
    float *d_out_f32_b = nullptr;
    uint8_t *h_in[2] = {nullptr, nullptr};      // pinned staging of the input chunks (allocated at the first multi-chunk call from pageable memory)
    uint8_t *h_out_u8[2] = {nullptr, nullptr};  // pinned landing buffers of the output copies (a copy into pageable memory blocks the
    float *h_out_f32[2] = {nullptr, nullptr};   // calling thread until the forward it waits for has finished: no overlap at all)
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_fwd[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    int opt_async = 0;    // PB_OPT_EMBED_ASYNC
    int trace_tune = 0;   // PB_TRACE_TUNE (1: chosen forms, 2: every candidate), PB_NO_STEM_FUSION: read once at create
    bool no_stem_fusion = false;
    bool no_resize_fusion = false, resize_attr_set = false;  // PB_NO_RESIZE_FUSION: the two-kernel resize (k_resize_v + k_resize_h) always
    int stem_rpp = 0;  // PB_STEM_RPP: stem rows per phase of k_stem_dw (0: the default)
    // the front of the network over sub-batches (forward_device): blocks [0, front_blocks) run front_sub images at a time, the last of
    // them writing its output map of all images to buf_front.  PB_FRONT_SUB (0: off), PB_FRONT_BLOCKS; PB_OPT_EMBED_FRONT_SUB
    size_t front_blocks = 0, front_out_per_image = 0;
    int front_sub = 0;
    float *buf_front = nullptr;
    // Two half-batches side by side (round 6; forward_device): a forward is a chain of 42 dependent launches, each with its own ramp-up,
    // drain and -- the squeeze-excite kernels, 12 launches, 94 us -- latency-bound stretches where most CUs idle.  From `dual_min`
    // images on the batch is cut in two; the second half runs on a stream and a workspace of its own, so the halves' launches fill
    // each other's tails.  Same kernels on the same per-image values: same bits.  PB_DUAL=0 / PB_OPT_EMBED_DUAL switch it.
    struct WsSet {
        float *buf_x[2] = {nullptr, nullptr}, *buf_e = nullptr, *buf_dw = nullptr, *buf_gate = nullptr, *buf_pool = nullptr, *buf_front = nullptr;
        long long *buf_part = nullptr;
        unsigned *d_se_cnt = nullptr;
    } ws2;
    bool ws2_ready = false;
    int dual_min = 1024;               // measured (profiles/r06_dual.txt): 1024 images 3.71 -> 3.63 ms, 512: 1.921 -> 1.919 (nothing), 128: 0.89 -> 1.03 (worse); 0: off
    hipStream_t dual_stream = nullptr;
    hipEvent_t dual_e0 = nullptr, dual_e1 = nullptr;
    size_t ws_max_x = 0, ws_max_e = 0, ws_max_dw = 0;  // per-image workspace sizes (floats), for the second set
    // k_se_multi (a few images: eight workgroups per image): the units' exchange granules [8 images][64] and arrival counters [8], zero between launches
    unsigned long long *d_se_xchg = nullptr;
    unsigned *d_se_arrive = nullptr;
    int se_multi_max = 8;  // PB_SE_MULTI_MAX (0: always k_se)
    unsigned *d_se_cnt = nullptr;  // [max_batch] arrival counters of the squeeze-excite tails (zero between launches)
    bool fold_se = false;          // PB_FOLD_SE=1: the gates of the first six blocks come from the producing kernels' tails instead of k_se
                                   // launches (measured: +0.04 ms per batch-512 forward and per batch-1 forward -- see SeTail; off by default)
    bool block_attr_set[4] = {false, false, false, false};  // k_block_small's LDS size attribute requested (5 x 5 residual form, 3 x 3 / 320 form)
    bool no_block_fusion = false;  // PB_NO_BLOCK_FUSION: the 4 x 4 blocks as front + k_se + project GEMM (A/B runs)
    bool no_tail_fusion = false;   // PB_NO_TAIL_FUSION: head conv, k_avgpool, FC GEMM and k_tanh_quant as four launches (A/B runs)
    bool no_gemm_stream = false;   // PB_NO_GEMM_STREAM: leave k_gemm_stream out of the per-layer timing loops (A/B runs)
    bool no_gemm_t = false;        // PB_NO_GEMM_T: leave k_gemm_t out of the per-layer timing loops (A/B runs)
    bool no_band_ipw = false;  // PB_NO_BAND_IPW: one (image, band) item per band-kernel workgroup, as before round 6 (A/B runs)
    bool no_band = false, force_band = false;  // PB_NO_BAND / PB_FORCE_BAND: leave out / always take the LDS-ring front kernel where it applies (A/B runs, bit comparisons)
    int tune_pick = 0;    // PB_TUNE_PICK: 0 fastest candidate (default); 1 slowest; 2 a pseudo-random one -- test hook: every form must give the same bits
    uint32_t tune_rng = 12345u;
    int p3e_min_k = 80;   // p3e_layer(): the expand layers of blocks 6-15 (K = 80, 112, 192) are P3 layers too; PB_P3E_MIN_K, PB_NO_P3E (PB_NO_P3 switches both off)
    int p3_min_k = 240;   // p3_layer(): the project layers of blocks 5-15, the head and the Linear are P3 layers (pb_gemm_p3.h); PB_P3_MIN_K, PB_NO_P3
    std::map<std::pair<const void *, long>, std::pair<int, int>> gemm_cfg;  // (layer weights, rows) -> (MR, NR), measured; MR < 0: eight-wave form
    std::map<std::pair<const void *, long>, DwGeom> dw_cfg;
    std::map<std::pair<const void *, long>, int> front_cfg;  // (block, batch) -> 0: expand GEMM + depthwise kernels, else fused kernel config 16 * bands + nc
    size_t part_floats_per_image = 0;      // (layer weights, batch) -> depthwise form, measured
    // the cache keys above are device / host pointers of this embedder; tune_keys names them (block * 8 + kind, 1000 + kind for the
    // head and the Linear) so that the picks can be saved and restored (pb_embed_get_tuning / pb_embed_set_tuning)
    std::vector<std::pair<const void *, uint32_t>> tune_keys;
    double tune_ms = 0.0;  // host time spent in the timing loops (outermost loops only: a front's loop times GEMM loops inside it)
    int tune_depth = 0;
    // pre-processing of image batches (prepare_images): two staging slots -- pinned host block, device source block, descriptor
    // arrays -- so that the host packs and the copy engine moves sub-batch j + 1 while the resize kernels of sub-batch j run;
    // all grow-only.  d_tmp (vertical-pass f32 rows) is used by one stream only.
    uint8_t *h_stage_img[2] = {nullptr, nullptr}, *d_src[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    ResizeDesc *h_desc[2] = {nullptr, nullptr}, *d_desc[2] = {nullptr, nullptr};
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_resized[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};
    float *d_tmp = nullptr;
    size_t d_tmp_cap = 0;
    std::mutex mu;
    // ---- decoder-facing staging (pb_embed_stage_*): decode workers write pixels straight into pinned slot memory
    struct StageSlot {
        uint8_t *h = nullptr, *d = nullptr;  // pinned host block and its device twin
        size_t cap = 0;
        ResizeDesc *h_desc = nullptr, *d_desc = nullptr;
        int state = 0;             // 0 free, 1 open (acquire hands out room), 2 closed (waiting for its writers / being committed),
                                   // 3 discarded by pb_embed_stage_abort with writers still in it (free when the last one releases)
        uint32_t n = 0, gen = 0;   // images handed out; generation (part of every ticket of this filling)
        int writers = 0;           // acquired, not yet released
        size_t bytes = 0, tmp = 0;
        std::vector<uint32_t> w, hgt;
        std::vector<size_t> off;
    };
    StageSlot st[2];
    std::mutex st_mu;
    std::condition_variable st_cv;
    size_t st_bytes_want = 0;          // PB_OPT_EMBED_STAGE_BYTES (0: STAGE_BYTES)
    int st_open = -1, st_closed = -1;  // slot being filled / slot closed and not yet committed
    uint32_t st_gen = 0;
    uint32_t st_abort_seq = 0;         // pb_embed_stage_abort calls so far (a close that waits for writers gives up when it changes)
    bool st_committing = false;        // pb_embed_stage_commit is reading the closed slot (abort leaves that slot to it)
};

namespace {

template <typename T>
int dalloc(pb_embedder *e, T **p, size_t n_elems) {
    void *q = nullptr;
    PB_HIP(hipMalloc(&q, std::max<size_t>(n_elems, 1) * sizeof(T)));
    e->allocs.push_back(q);
    *p = static_cast<T *>(q);
    return PB_OK;
}

int upload(pb_embedder *e, float **dst, const std::vector<float> &src) {
    int rc = dalloc(e, dst, src.size());
    if (rc) return rc;
    PB_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
    return PB_OK;
}

// The per-layer timing loops cache their pick per (layer, bucket of the row / image count): buckets are powers of two,
// so a front end that produces arbitrary batch sizes (BatchingEmbedder, tail batches) triggers at most
// log2(max_batch) + 1 measurements per layer instead of one per distinct size, and the caches stay bounded.  Every
// candidate's eligibility threshold (64, 128, 1024 rows) is a power of two, so a pick is valid for its whole bucket,
// and all candidates give the same bits (tests), so the pick only affects speed.
long tune_bucket(long m) {
    long b = 1;
    while (b < m) b <<= 1;
    return b;
}

// candidate selection of the per-layer timing loops (see pb_embedder::tune_pick)
bool tune_take(pb_embedder *e, float ms, float best_ms) {
    if (e->tune_pick == 1) return best_ms >= 1e29f || ms > best_ms;
    if (e->tune_pick == 2) {
        e->tune_rng = e->tune_rng * 1664525u + 1013904223u;
        return best_ms >= 1e29f || (e->tune_rng >> 16) % 3 == 0;
    }
    return ms < best_ms;
}

// host time of the per-layer timing loops (pb_embed_tune_ms): one of these at the top of every `not measured yet` branch
struct TuneTimer {
    pb_embedder *e;
    std::chrono::steady_clock::time_point t0;
    explicit TuneTimer(pb_embedder *e_) : e(e_), t0(std::chrono::steady_clock::now()) { ++e->tune_depth; ++e->tune_runs; }
    ~TuneTimer() {
        if (--e->tune_depth == 0) e->tune_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};

// torch [N][K] (OI) -> k-major zero-padded [Kpad][Npad] + padded bias
int make_gemm(pb_embedder *e, Gemm *g, const float *w, const float *b, int N, int K, bool pieces = false, bool fragments_by_step = false) {
    g->K = K;
    g->N = N;
    g->Kpad = round_up(K, 16);
    g->Npad = round_up(N, 16);
    std::vector<float> wt((size_t)g->Kpad * g->Npad, 0.0f), bp(g->Npad, 0.0f);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) wt[(size_t)k * g->Npad + n] = w[(size_t)n * K + k];
    for (int n = 0; n < N; ++n) bp[n] = b[n];
    int rc = upload(e, &g->wt, wt);
    if (!rc) rc = upload(e, &g->bias, bp);
    if (!rc && K % 16 == 0) {
        const int chunks = (K + 63) / 64, t16 = g->Npad / 16;
        std::vector<float> w2((size_t)chunks * t16 * 1024, 0.0f);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < N; ++n) {
                const int ch = k / 64, s2 = (k % 64) / 16, kk = (k % 16) / 4, e2 = k % 4;
                w2[((((size_t)ch * t16 + n / 16) * 4 + kk) * 16 + n % 16) * 16 + s2 * 4 + e2] = w[(size_t)n * K + k];
            }
        rc = upload(e, &g->wt2, w2);
        if (!rc && fragments_by_step) {
            // the same fragments with a k-step's 64 lanes contiguous: [k / 16][tile][kk * 16 + li][e] -- for kernels whose lanes
            // load their fragment straight into registers (k_block_small): a wave's request is one contiguous KB
            const int steps = K / 16;
            std::vector<float> w4((size_t)steps * t16 * 256, 0.0f);
            for (int k = 0; k < K; ++k)
                for (int n = 0; n < N; ++n)
                    w4[(((size_t)(k / 16) * t16 + n / 16) * 64 + ((k % 16) / 4) * 16 + n % 16) * 4 + k % 4] = w[(size_t)n * K + k];
            rc = upload(e, &g->wt4, w4);
        }
    }
    if (rc || !pieces) return rc;
    // P3 layer: w = hi + mid + lo exactly, each piece a bf16 (truncation split: 8 + 8 + 8 significand bits), stored as the MFMA
    // fragments of pb_gemm_p3.h: [k-step of 32][16-column tile][plane][lane = kk * 16 + li][j] = piece of w[k = 32 s + 8 kk + j][n = 16 t + li]
    const int steps = (K + 31) / 32, t16 = g->Npad / 16;
    std::vector<uint16_t> w3((size_t)steps * t16 * 3 * 64 * 8, 0);
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto flt = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) {
            const float x = w[(size_t)n * K + k];
            const float r1 = x - flt(bits(x) & 0xFFFF0000u);
            const float r2 = r1 - flt(bits(r1) & 0xFFFF0000u);
            const uint16_t piece[3] = {(uint16_t)(bits(x) >> 16), (uint16_t)(bits(r1) >> 16), (uint16_t)(bits(r2) >> 16)};
            const size_t frag = ((size_t)(k / 32) * t16 + n / 16) * 3;
            const int lane = ((k % 32) / 8) * 16 + n % 16, j = k % 8;
            for (int pl = 0; pl < 3; ++pl) w3[((frag + pl) * 64 + lane) * 8 + j] = piece[pl];
        }
    uint16_t *dw = nullptr;
    rc = dalloc(e, &dw, w3.size());
    if (rc) return rc;
    PB_HIP(hipMemcpy(dw, w3.data(), w3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    g->wt3 = dw;
    g->p3 = true;
    return PB_OK;
}

// Which layers are P3 layers (pb_gemm_p3.h) -- a property of the layer's SHAPE only (never of the batch): deep enough that the
// f32 MFMA chain is what bounds it (K >= 240) and at least five 16-column tiles wide, so that the operand split hides under the
// tile's own matrix instructions (the 40-column project of block 4 is bound by its 126 MB of activations either way and ran
// 44 us as a P3 layer against 39 us on the f32 chain).
bool p3_layer(const pb_embedder *e, int K, int N) { return K >= e->p3_min_k && K % 8 == 0 && (N + 15) / 16 >= 5; }
// EXPAND layers as P3 layers (round 6): K >= 80 -- blocks 6-15 (80 -> 480, 112 -> 672, 192 -> 1152).  The operand is split by the
// CONSUMER: once per workgroup in k_block_small (blocks 12-15, whose f32 expand chain was 38 % of the kernel), once per group and
// k-step in k_mbconv_small (a lane's own 8 k), in the tile's own stream in k_gemm_p3 -- nothing for a producer to write.
bool p3e_layer(const pb_embedder *e, int K, int N) { return K >= e->p3e_min_k && K % 8 == 0 && (N + 15) / 16 >= 5; }

size_t blob_floats(int D) {
    size_t n = 32 * 27 + 32;
    for (const Stage &st : STAGES)
        for (int r = 0; r < st.repeats; ++r) {
            const int cin = r == 0 ? st.cin : st.cout, e = cin * st.expand, sq = std::max(1, cin / 4);
            if (st.expand != 1) n += (size_t)e * cin + e;
            n += (size_t)e * st.k * st.k + e;
            n += (size_t)sq * e + sq + (size_t)e * sq + e;
            n += (size_t)st.cout * e + st.cout;
        }
    n += 1280 * 320 + 1280;
    n += (size_t)D * 1280 + D;
    return n;
}

int load_weights(pb_embedder *e, const uint8_t *blob, size_t len) {
    PB_CHECK(len >= 32 && memcmp(blob, "PBXW0001", 8) == 0, PB_ERR_FORMAT, "weight blob: bad magic (want PBXW0001)");
    uint32_t hdr[4];
    uint64_t nfl;
    memcpy(hdr, blob + 8, 16);
    memcpy(&nfl, blob + 24, 8);
    e->H = hdr[0];
    e->W = hdr[1];
    e->D = hdr[2];
    PB_CHECK(e->D >= 1 && e->D <= 4096, PB_ERR_FORMAT, "weight blob: D = %u outside 1..4096", e->D);
    PB_CHECK(e->H >= 32 && e->W >= 32 && e->H % 32 == 0 && e->W % 32 == 0 && e->H <= 1024 && e->W <= 1024, PB_ERR_FORMAT,
             "weight blob: H x W = %u x %u must be multiples of 32 in 32..1024", e->H, e->W);
    PB_CHECK(nfl == blob_floats((int)e->D) && len == 32 + nfl * 4, PB_ERR_FORMAT, "weight blob: size mismatch");
    const float *p = reinterpret_cast<const float *>(blob + 32);
    int rc;
    {  // stem [32][3][3][3] (OIHW) -> w' = fl(w / 255.0f) (the reference's px / 255, efficientnet.rs:27, folded into the weight: the
       // pixel bytes themselves are exact bf16 operands), split exactly into three bf16 pieces, in the fragment order of stem_tile:
       // lane (li, kq), slot j -> kq < 3: tap (ky = kq, kx = j / 3, ci = j % 3); kq = 3: j < 3: tap (ky = j, kx = 2, ci = 2), else 0
        std::vector<float> b(p + 27 * 32, p + 27 * 32 + 32);
        std::vector<uint16_t> w3((size_t)2 * 3 * 64 * 8, 0);
        auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
        auto flt = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
        for (int c = 0; c < 2; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int li = lane & 15, kq = lane >> 4, co = 16 * c + li;
                    int ky, kx, ci;
                    if (kq < 3) { ky = kq; kx = j / 3; ci = j % 3; }
                    else if (j < 3) { ky = j; kx = 2; ci = 2; }
                    else continue;
                    const float x = p[((co * 3 + ci) * 3 + ky) * 3 + kx] / 255.0f;
                    const float r1 = x - flt(bits(x) & 0xFFFF0000u);
                    const float r2 = r1 - flt(bits(r1) & 0xFFFF0000u);
                    const uint16_t piece[3] = {(uint16_t)(bits(x) >> 16), (uint16_t)(bits(r1) >> 16), (uint16_t)(bits(r2) >> 16)};
                    for (int pl = 0; pl < 3; ++pl) w3[(((size_t)c * 3 + pl) * 64 + lane) * 8 + j] = piece[pl];
                }
        if ((rc = dalloc(e, &e->stem_w3, w3.size()))) return rc;
        PB_HIP(hipMemcpy(e->stem_w3, w3.data(), w3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        if ((rc = upload(e, &e->stem_b, b))) return rc;
        p += 27 * 32 + 32;
    }
    for (const Stage &st : STAGES)
        for (int r = 0; r < st.repeats; ++r) {
            Block bl{};
            bl.cin = r == 0 ? st.cin : st.cout;
            bl.cout = st.cout;
            bl.stride = r == 0 ? st.stride : 1;
            bl.e = bl.cin * st.expand;
            bl.sq = std::max(1, bl.cin / 4);
            bl.k = st.k;
            bl.has_expand = st.expand != 1;
            bl.residual = bl.stride == 1 && bl.cin == bl.cout;
            const int E = bl.e, S = bl.sq, KK = st.k * st.k;
            if (bl.has_expand) {
                if ((rc = make_gemm(e, &bl.expand, p, p + (size_t)E * bl.cin, E, bl.cin, p3e_layer(e, bl.cin, E), bl.stride == 1))) return rc;
                p += (size_t)E * bl.cin + E;
            }
            {  // dw [E][k][k] -> [k*k][E]
                std::vector<float> w((size_t)KK * E), b(p + (size_t)E * KK, p + (size_t)E * KK + E);
                for (int c = 0; c < E; ++c)
                    for (int t = 0; t < KK; ++t) w[(size_t)t * E + c] = p[(size_t)c * KK + t];
                if ((rc = upload(e, &bl.dw_w, w)) || (rc = upload(e, &bl.dw_b, b))) return rc;
                if (E % 4 == 0) {
                    std::vector<float> wq((size_t)KK * E);
                    for (int c = 0; c < E; ++c)
                        for (int t = 0; t < KK; ++t) wq[((size_t)(c / 4) * KK + t) * 4 + (c & 3)] = p[(size_t)c * KK + t];
                    float *dq = nullptr;
                    if ((rc = upload(e, &dq, wq))) return rc;
                    bl.dw_wq = reinterpret_cast<f32x4 *>(dq);
                }
                {  // [E][k*k rounded up to 4]: a channel's taps as 16-byte pieces (k_block_small)
                    const int KKP = (KK + 3) / 4 * 4;
                    std::vector<float> wc((size_t)E * KKP, 0.0f);
                    for (int c = 0; c < E; ++c)
                        for (int t = 0; t < KK; ++t) wc[(size_t)c * KKP + t] = p[(size_t)c * KK + t];
                    if ((rc = upload(e, &bl.dw_wc, wc))) return rc;
                }
                p += (size_t)E * KK + E;
            }
            {
                bl.sp = S <= 8 ? 8 : (S <= 16 ? 16 : (S <= 32 ? 32 : 48));
                PB_CHECK(S <= 48, PB_ERR_FORMAT, "squeeze width %d > 48", S);
                std::vector<float> w1((size_t)bl.sp * E, 0.0f), b1(bl.sp, 0.0f);
                std::copy(p, p + (size_t)S * E, w1.begin());
                std::copy(p + (size_t)S * E, p + (size_t)S * E + S, b1.begin());
                if ((rc = upload(e, &bl.se_w1, w1)) || (rc = upload(e, &bl.se_b1, b1))) return rc;
                p += (size_t)S * E + S;
                std::vector<float> w2t((size_t)bl.sp * E, 0.0f), b2(p + (size_t)E * S, p + (size_t)E * S + E);
                for (int c = 0; c < E; ++c)
                    for (int j = 0; j < S; ++j) w2t[(size_t)j * E + c] = p[(size_t)c * S + j];
                if ((rc = upload(e, &bl.se_w2t, w2t)) || (rc = upload(e, &bl.se_b2, b2))) return rc;
                p += (size_t)E * S + E;
            }
            if ((rc = make_gemm(e, &bl.project, p, p + (size_t)bl.cout * E, bl.cout, E, p3_layer(e, E, bl.cout), true))) return rc;
            p += (size_t)bl.cout * E + bl.cout;
            e->blocks.push_back(bl);
        }
    if ((rc = make_gemm(e, &e->head, p, p + 1280 * 320, 1280, 320, p3_layer(e, 320, 1280)))) return rc;
    p += 1280 * 320 + 1280;
    PB_CHECK(e->D % 4 == 0, PB_ERR_FORMAT, "weight blob: D = %u must be a multiple of 4", e->D);
    if ((rc = make_gemm(e, &e->fc, p, p + (size_t)e->D * 1280, (int)e->D, 1280, p3_layer(e, 1280, (int)e->D)))) return rc;
    for (size_t b = 0; b < e->blocks.size(); ++b) {
        const Block &bl = e->blocks[b];
        const uint32_t base = (uint32_t)b * 8;
        if (bl.has_expand) e->tune_keys.emplace_back(bl.expand.wt, base + 0);
        e->tune_keys.emplace_back(bl.project.wt, base + 1);
        e->tune_keys.emplace_back(bl.dw_w, base + 2);
        e->tune_keys.emplace_back(&bl, base + 3);
    }
    e->tune_keys.emplace_back(e->head.wt, 1000u);
    if (e->head.wt2) e->tune_keys.emplace_back(e->head.wt2, 1001u);
    e->tune_keys.emplace_back(e->fc.wt, 1002u);
    if (e->fc.wt2) e->tune_keys.emplace_back(e->fc.wt2, 1003u);
    return PB_OK;
}

template <int MR, bool GATE, int NW = 4, int PDX = 0>
void launch_gemm_mr(int nr, dim3 grid, hipStream_t st, const float *act, int M, const Gemm &g, const float *gate, int hw,
                    const float *resid, int do_silu, float *out) {
#define PB_G(NRV)                                                                                                    \
    case NRV:                                                                                                        \
        hipLaunchKernelGGL((k_gemm1x1<MR, NRV, GATE, NW, PDX>), grid, dim3(64 * NW), 0, st, act, M, g.K, g.wt, g.Kpad, g.Npad, g.bias, \
                           g.N, gate, hw, resid, do_silu, out);                                                      \
        break;
    switch (nr) {
        PB_G(1) PB_G(2) PB_G(3) PB_G(4) PB_G(5) PB_G(6) PB_G(7) PB_G(8)
    }
#undef PB_G
}

template <bool GATE, int NW, int EPI = 0>
void launch_gemm_t(int nr, hipStream_t st, const float *act, long M, const Gemm &g, const float *gate, int hw, const float *resid,
                   int do_silu, float *out, float scale = 0.f, uint8_t *out_u8 = nullptr) {
    const dim3 grid((unsigned)((M + 16 * NW - 1) / (16 * NW)), (unsigned)(g.Npad / 16 / nr));
#define PB_G(NRV)                                                                                                              \
    case NRV:                                                                                                                  \
        hipLaunchKernelGGL((k_gemm_t<NRV, GATE, NW, EPI>), grid, dim3(64 * NW), 0, st, act, (int)M, g.K, g.wt2, g.Npad / 16, g.bias, g.N, gate, hw, \
                           resid, do_silu, out, scale, out_u8);                                                                \
        break;
    switch (nr) {
        PB_G(1) PB_G(2) PB_G(3) PB_G(4) PB_G(5) PB_G(6) PB_G(7) PB_G(8)
    }
#undef PB_G
}

// Tile choice.  NR (16-column tiles per wave) must divide Npad/16; MR in {4,2,1} (64*MR rows per block).
// The best (MR, NR) depends on the layer shape and on the batch (memory-bound thin layers want big tiles,
// small-M late layers want many small blocks), so it is measured: the first forward with a given row count
// times every candidate on the real buffers (HIP events, 1 warm-up + 2 timed launches each; the outputs are
// simply overwritten with identical values) and the winner is cached per (layer, M).
struct GemmCfg {
    int mr, nr, nw;  // nw = 8: eight waves per block (MR = 1 only), else four
    int pd = 0;      // eight-wave form only: activation prefetch distance 8 / 16 k-steps (0: the default of 4)
    int tform = 0;   // 1: k_gemm_t (fragment-ordered weights, pipelined fragment reads, no masks in the loop), nw = 4 or 8;
                     // 2: k_gemm_stream (whole weight matrix in registers, a wave streams `mr` 16-row tiles of one image), nr = all tiles
};

// k_gemm_stream is built for the gated, thin project layers of the early blocks (K x Npad: 32 x 16, 96 x 32, 144 x 32, 144 x 48, 240 x 48)
bool stream_eligible(const pb_embedder *e, const Gemm &g, long M, const float *gate, int hw, int do_silu) {
    if (e->no_gemm_stream || !g.wt4 || !gate || do_silu || g.K % 16 || M % 16 || hw % 16 || g.N % 4) return false;
    const int ks = g.K / 16, t = g.Npad / 16;
    return (ks == 2 && t == 1) || (ks == 6 && t == 2) || (ks == 9 && (t == 2 || t == 3)) || (ks == 15 && t == 3);
}

void launch_gemm_cfg(pb_embedder *e, GemmCfg c, const float *act, long M, const Gemm &g, const float *gate, int hw,
                     const float *resid, int do_silu, float *out) {
    const int tiles = g.Npad / 16;
    const long rows_per_block = 16L * c.nw * c.mr;
    dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block), (unsigned)(tiles / c.nr));
    const int nr = c.nr;
    if (c.tform == 2) {
        const int ks = g.K / 16;
        const unsigned n_wg = (unsigned)((M / 16 + 4L * c.mr - 1) / (4L * c.mr));
#define PB_GS(KSV, NTV)                                                                                                            \
    do {                                                                                                                           \
        if (resid) hipLaunchKernelGGL((k_gemm_stream<KSV, NTV, true>), dim3(n_wg), dim3(256), 0, e->stream, act, M, g.wt4, tiles, g.bias, g.N, gate, hw, resid, out, c.mr); \
        else hipLaunchKernelGGL((k_gemm_stream<KSV, NTV, false>), dim3(n_wg), dim3(256), 0, e->stream, act, M, g.wt4, tiles, g.bias, g.N, gate, hw, resid, out, c.mr);    \
    } while (0)
        if (ks == 2 && tiles == 1) PB_GS(2, 1);
        else if (ks == 6 && tiles == 2) PB_GS(6, 2);
        else if (ks == 9 && tiles == 2) PB_GS(9, 2);
        else if (ks == 9 && tiles == 3) PB_GS(9, 3);
        else if (ks == 15 && tiles == 3) PB_GS(15, 3);
#undef PB_GS
        return;
    }
    if (c.tform) {
        if (gate) {
            if (c.nw == 8) launch_gemm_t<true, 8>(nr, e->stream, act, M, g, gate, hw, resid, do_silu, out);
            else launch_gemm_t<true, 4>(nr, e->stream, act, M, g, gate, hw, resid, do_silu, out);
        } else {
            if (c.nw == 8) launch_gemm_t<false, 8>(nr, e->stream, act, M, g, gate, hw, resid, do_silu, out);
            else launch_gemm_t<false, 4>(nr, e->stream, act, M, g, gate, hw, resid, do_silu, out);
        }
        return;
    }
    if (c.nw == 1) {  // one wave per 16 x 16 tile, weights from global (a few pixel rows: small batches of the late layers)
        const dim3 tg((unsigned)((M + 15) / 16), (unsigned)tiles);
        if (gate)
            hipLaunchKernelGGL((k_gemm_thin<true>), tg, dim3(64), 0, e->stream, act, (int)M, g.K, g.wt, g.Kpad, g.Npad, g.bias, g.N, gate, hw, resid, do_silu, out);
        else
            hipLaunchKernelGGL((k_gemm_thin<false>), tg, dim3(64), 0, e->stream, act, (int)M, g.K, g.wt, g.Kpad, g.Npad, g.bias, g.N, gate, hw, resid, do_silu, out);
        return;
    }
    if (c.nw == 8) {
#define PB_L8(PDV)                                                                                                      \
    (gate ? launch_gemm_mr<1, true, 8, PDV>(nr, grid, e->stream, act, (int)M, g, gate, hw, resid, do_silu, out)          \
          : launch_gemm_mr<1, false, 8, PDV>(nr, grid, e->stream, act, (int)M, g, gate, hw, resid, do_silu, out))
        PB_L8(0);  // (prefetch distances 8 / 16 were measured on the late layers: no gain)
#undef PB_L8
        return;
    }
#define PB_L(MRV)                                                                                                    \
    (gate ? launch_gemm_mr<MRV, true>(nr, grid, e->stream, act, (int)M, g, gate, hw, resid, do_silu, out)            \
          : launch_gemm_mr<MRV, false>(nr, grid, e->stream, act, (int)M, g, gate, hw, resid, do_silu, out))
    if (c.mr == 4) PB_L(4);
    else if (c.mr == 2) PB_L(2);
    else PB_L(1);
#undef PB_L
}

// ---- P3 layers (pb_gemm_p3.h): one arithmetic, several tile shapes; the shape is measured per (layer, row bucket) like the
// f32 forms' (the shapes give identical bits, so the pick only affects speed).  enc = 4096 * nw + 16 * mr + nr.
P3Args p3_args(const float *act, long M, const Gemm &g, const float *gate, int hw, const float *resid, int do_silu, float *out, float scale = 0.f,
               uint8_t *out_u8 = nullptr) {
    P3Args a;
    a.act = act; a.M = M; a.K = g.K; a.wt3 = g.wt3; a.tiles16 = g.Npad / 16; a.bias = g.bias; a.N = g.N; a.gate = gate; a.hw = hw;
    a.resid = resid; a.do_silu = do_silu; a.out = out; a.scale = scale; a.out_u8 = out_u8;
    return a;
}

int launch_p3_tuned(pb_embedder *e, int epi, const P3Args &a, const void *key_ptr, const char *what) {
    const std::pair<const void *, long> key(key_ptr, tune_bucket(a.M));
    auto it = e->gemm_cfg.find(key);
    if (it == e->gemm_cfg.end()) {
        TuneTimer tt(e);
        int best = 0;
        float best_ms = 1e30f;
        const bool gate = a.gate != nullptr, kt = (a.K & 31) != 0;
        for (int nw : {8, 4, 1})
            for (int mr : {1, 2})
                for (int nr = 8; nr >= 1; --nr) {
                    if (a.tiles16 % nr || !p3_has(nr, mr, nw, epi, gate, kt)) continue;
                    if (nw == 1 && a.M > 1024) continue;             // the one-wave form: a few pixel rows
                    if (nw > 1 && a.M <= 16L * (nw / 2) * mr) continue;  // more than half of the workgroup's rows would be padding
                    p3_launch(nr, mr, nw, epi, e->stream, a);
                    PB_HIP(hipEventRecord(e->tune_e0, e->stream));
                    p3_launch(nr, mr, nw, epi, e->stream, a);
                    p3_launch(nr, mr, nw, epi, e->stream, a);
                    PB_HIP(hipEventRecord(e->tune_e1, e->stream));
                    PB_HIP(hipEventSynchronize(e->tune_e1));
                    PB_HIP(hipGetLastError());
                    float ms = 0.f;
                    PB_HIP(hipEventElapsedTime(&ms, e->tune_e0, e->tune_e1));
                    if (e->trace_tune >= 2) fprintf(stderr, "  %s M%ld K%d N%d: P3 NR%d MR%d NW%d %.1f us\n", what, a.M, a.K, a.N, nr, mr, nw, ms * 500.f);
                    if (tune_take(e, ms, best_ms)) {
                        best_ms = ms;
                        best = 4096 * nw + 16 * mr + nr;
                    }
                }
        if (!best) {  // tiny problems: whatever shape exists
            for (int nw : {1, 4, 8})
                for (int nr = 1; nr <= 8 && !best; ++nr)
                    if (a.tiles16 % nr == 0 && p3_has(nr, 1, nw, epi, gate, kt)) best = 4096 * nw + 16 + nr;
            PB_CHECK(best, PB_ERR_INTERNAL, "no P3 GEMM shape for %d column tiles (epilogue %d)", a.tiles16, epi);
        }
        if (e->trace_tune)
            fprintf(stderr, "%s M%ld K%d N%d%s: P3 best NR%d MR%d NW%d %.1f us = %.1f TFLOP/s\n", what, a.M, a.K, a.N, gate ? " gated" : "", best & 15, (best >> 4) & 15,
                    best >> 12, best_ms * 500.f, 2.0 * (double)a.M * a.K * a.N / (best_ms * 0.5e-3) / 1e12);
        it = e->gemm_cfg.emplace(key, std::make_pair(best, 0)).first;
    }
    const int enc = it->second.first;
    PB_CHECK(p3_launch(enc & 15, (enc >> 4) & 15, enc >> 12, epi, e->stream, a), PB_ERR_INTERNAL, "P3 GEMM shape %d not built", enc);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int launch_gemm_p3(pb_embedder *e, const float *act, long M, const Gemm &g, const float *gate, int hw, const float *resid, int do_silu, float *out) {
    return launch_p3_tuned(e, 0, p3_args(act, M, g, gate, hw, resid, do_silu, out), g.wt, "gemm");
}

int launch_gemm(pb_embedder *e, const float *act, long M, const Gemm &g, const float *gate, int hw, const float *resid,
                int do_silu, float *out) {
    const int tiles = g.Npad / 16;
    const std::pair<const void *, long> key(g.wt, tune_bucket(M));
    auto it = e->gemm_cfg.find(key);
    if (g.p3) return launch_gemm_p3(e, act, M, g, gate, hw, resid, do_silu, out);
    if (it == e->gemm_cfg.end()) {
        TuneTimer tt(e);
        GemmCfg best{1, 1, 4};
        float best_ms = 1e30f;
        const hipEvent_t e0 = e->tune_e0, e1 = e->tune_e1;  // the embedder's own pair: nothing to release on an early return
        for (int nr = 8; nr >= 1; --nr) {
            if (tiles % nr) continue;
            // 0: the eight-wave form of MR = 1; -1: the one-wave form (k_gemm_thin); -104 / -108: k_gemm_t with 4 / 8 waves
            for (int mr : {4, 2, 1, 0, -104, -108, -1, -208, -216, -232, -264}) {
                if (mr <= -200) {  // k_gemm_stream with 8 / 16 / 32 / 64 tiles per wave
                    const int tpw = -mr - 200;
                    if (nr != tiles || !stream_eligible(e, g, M, gate, hw, do_silu) || hw % (16 * tpw) || M / 16 / tpw < 2L * e->n_cu) continue;
                    const GemmCfg c{tpw, nr, 4, 0, 2};
                    launch_gemm_cfg(e, c, act, M, g, gate, hw, resid, do_silu, out);
                    PB_HIP(hipEventRecord(e0, e->stream));
                    launch_gemm_cfg(e, c, act, M, g, gate, hw, resid, do_silu, out);
                    launch_gemm_cfg(e, c, act, M, g, gate, hw, resid, do_silu, out);
                    PB_HIP(hipEventRecord(e1, e->stream));
                    PB_HIP(hipEventSynchronize(e1));
                    PB_HIP(hipGetLastError());
                    float ms = 0.f;
                    PB_HIP(hipEventElapsedTime(&ms, e0, e1));
                    if (e->trace_tune >= 2) fprintf(stderr, "  gemm M%ld K%d N%d: stream form, %d tiles per wave %.1f us\n", M, g.K, g.N, tpw, ms * 500.f);
                    if (tune_take(e, ms, best_ms)) {
                        best_ms = ms;
                        best = c;
                    }
                    continue;
                }
                if (mr > 1 && M <= 64L * (mr / 2)) continue;  // tile taller than the problem
                if ((mr == 0 || mr == -108) && M <= 64) continue;
                if (mr <= -100 && (!g.wt2 || e->no_gemm_t)) continue;
                if (mr == -1 && (nr != 1 || M > 1024)) continue;
                const GemmCfg c{mr > 0 ? mr : 1, nr, mr > 0 ? 4 : (mr == -1 ? 1 : (mr == -104 ? 4 : 8)), 0, mr <= -100 ? 1 : 0};
                launch_gemm_cfg(e, c, act, M, g, gate, hw, resid, do_silu, out);
                PB_HIP(hipEventRecord(e0, e->stream));
                launch_gemm_cfg(e, c, act, M, g, gate, hw, resid, do_silu, out);
                launch_gemm_cfg(e, c, act, M, g, gate, hw, resid, do_silu, out);
                PB_HIP(hipEventRecord(e1, e->stream));
                PB_HIP(hipEventSynchronize(e1));
                PB_HIP(hipGetLastError());
                float ms = 0.f;
                PB_HIP(hipEventElapsedTime(&ms, e0, e1));
                if (e->trace_tune >= 2)
                    fprintf(stderr, "  gemm M%ld K%d N%d: MR%d NR%d NW%d%s %.1f us\n", M, g.K, g.N, c.mr, c.nr, c.nw, c.tform ? " t-form" : "", ms * 500.f);
                if (tune_take(e, ms, best_ms)) {
                    best_ms = ms;
                    best = c;
                }
            }
        }
        if (e->trace_tune)
            fprintf(stderr, "gemm M%ld K%d N%d%s: best %s%d NR%d NW%d%s %.1f us = %.1f TFLOP/s\n", M, g.K, g.N, gate ? " gated" : "",
                    best.tform == 2 ? "stream form, tiles per wave " : "MR", best.mr, best.nr, best.nw, best.tform == 1 ? " t-form" : "", best_ms * 500.f,
                    2.0 * (double)M * g.K * g.N / (best_ms * 0.5e-3) / 1e12);
        // encoding: 100 one-wave form; -1 eight-wave form; 304 / 308 k_gemm_t with 4 / 8 waves; else MR of the four-wave form
        it = e->gemm_cfg.emplace(key, std::make_pair(best.tform == 2 ? 1000 + best.mr : (best.tform ? 300 + best.nw : (best.nw == 1 ? 100 : (best.nw == 8 ? -1 : best.mr))), best.nr)).first;
    }
    const int enc = it->second.first;
    if (enc >= 1000) {  // k_gemm_stream, enc - 1000 tiles per wave
        launch_gemm_cfg(e, GemmCfg{enc - 1000, it->second.second, 4, 0, 2}, act, M, g, gate, hw, resid, do_silu, out);
        PB_HIP(hipGetLastError());
        return PB_OK;
    }
    launch_gemm_cfg(e, GemmCfg{(enc >= 100 || enc < 0) ? 1 : enc, it->second.second, enc >= 300 ? enc - 300 : (enc == 100 ? 1 : (enc < 0 ? 8 : 4)), 0, enc >= 300 ? 1 : 0},
                    act, M, g, gate, hw, resid, do_silu, out);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// strips = rows x ceil(Wo / 4) groups of 4 adjacent output pixels (k_dwconv's register tile)
DwGeom dw_geom(int C, int ho, int wo, int batch, int n_cu) {
    DwGeom g;
    g.roll = 0;
    const int cq = C / 4;
    // channel quads per block: the largest divisor of C/4 that is <= 32 -- keeps the block's filter taps in
    // LDS small (25 taps x 32 quads x 16 B = 12.8 KB) relative to the activations it streams
    g.cqpb = 1;
    for (int dv = 1; dv <= 32 && dv <= cq; ++dv)
        if (cq % dv == 0) g.cqpb = dv;
    g.zsplit = cq / g.cqpb;
    g.slots = std::max(1, 256 / g.cqpb);
    const int n_strips = ho * ((wo + 3) / 4);
    // as few tiles per image as still gives ~4 blocks per CU: every block pays a filter load + two barriers
    const long want = (4L * n_cu + (long)batch * g.zsplit - 1) / ((long)batch * g.zsplit);
    const int cap = std::max(1, n_strips / (g.slots * 2));  // keep >= 2 strips per slot
    g.n_tiles = (int)std::max<long>(1, std::min<long>(std::min(32, cap), want));
    g.strips_per_tile = (n_strips + g.n_tiles - 1) / g.n_tiles;
    g.n_tiles = (n_strips + g.strips_per_tile - 1) / g.strips_per_tile;
    return g;
}
// rolling-window geometry; returns roll = 0 when the map is too narrow to be worth it
DwGeom dw_geom_roll(int C, int k, int ho, int wo, int batch, int n_cu) {
    DwGeom g = dw_geom(C, ho, wo, batch, n_cu);
    const int tx = k == 3 ? 4 : 2;
    if (wo < 16) return g;
    const int cq = C / 4;
    const int strips_x = (wo + tx - 1) / tx;
    if (strips_x > 256) return g;
    int cqpb = 1;
    for (int dv = 1; dv <= cq; ++dv)
        if (cq % dv == 0 && dv * strips_x <= 256 && dv <= 32) cqpb = dv;
    g.roll = 1;
    g.cqpb = cqpb;
    g.zsplit = cq / cqpb;
    g.slots = strips_x;
    const long want = (4L * n_cu + (long)batch * g.zsplit - 1) / ((long)batch * g.zsplit);
    g.n_tiles = (int)std::max<long>(1, std::min<long>(std::min(32, std::max(1, ho / 4)), want));
    g.strips_per_tile = (ho + g.n_tiles - 1) / g.n_tiles;  // rows per band
    g.n_tiles = (ho + g.strips_per_tile - 1) / g.strips_per_tile;
    return g;
}

// whole-map-in-LDS geometry for small maps; returns roll = 0 (strip geometry) when it does not apply
DwGeom dw_geom_lds(int C, int k, int h, int w, int ho, int wo, int batch, int n_cu) {
    DwGeom g = dw_geom(C, ho, wo, batch, n_cu);
    if (h > 16 || w > 16) return g;
    const int pad = (k - 1) / 2, cq = C / 4;
    const size_t per_quad = ((size_t)(h + 2 * pad) * (w + 2 * pad) + (size_t)k * k) * 16;
    int cqb = 0;
    for (int dv = 1; dv <= 32 && dv <= cq; ++dv)
        if (cq % dv == 0 && dv * per_quad <= 40 * 1024) cqb = dv;
    if (!cqb) return g;
    g.roll = 2;
    g.cqpb = cqb;
    g.zsplit = cq / cqb;
    g.slots = std::max(1, std::min(256 / cqb, ho * wo));
    g.n_tiles = 1;
    g.strips_per_tile = 0;
    return g;
}

int launch_dw_geom(pb_embedder *e, const Block &bl, const float *in, int B, int H, int W, float *out, int Ho, int Wo,
                   const DwGeom &g) {
    dim3 grid(g.n_tiles, B, g.zsplit), block(g.cqpb * g.slots);
    if (g.roll == 2) {
        const int pad = (bl.k - 1) / 2;
        const size_t lds2 = ((size_t)(H + 2 * pad) * (W + 2 * pad) + (size_t)bl.k * bl.k) * g.cqpb * 16;
#define PB_DWL(KS, S)                                                                                                \
    hipLaunchKernelGGL((k_dwconv_lds<KS, S>), grid, block, lds2, e->stream, in, H, W, bl.e, bl.dw_w, bl.dw_b, out, Ho, Wo, \
                       e->buf_part, g.cqpb)
        if (bl.k == 3 && bl.stride == 1) PB_DWL(3, 1);
        else if (bl.k == 3 && bl.stride == 2) PB_DWL(3, 2);
        else if (bl.k == 5 && bl.stride == 1) PB_DWL(5, 1);
        else PB_DWL(5, 2);
#undef PB_DWL
        PB_HIP(hipGetLastError());
        return PB_OK;
    }
    const size_t lds = (size_t)bl.k * bl.k * g.cqpb * 16;  // filter taps of this block's channel quads
    if (!g.roll) {
#define PB_DW(KS, S)                                                                                            \
    hipLaunchKernelGGL((k_dwconv<KS, S>), grid, block, lds, e->stream, in, H, W, bl.e, bl.dw_w, bl.dw_b, out, Ho, Wo, \
                       g.strips_per_tile, e->buf_part, g.n_tiles, g.cqpb)
        if (bl.k == 3 && bl.stride == 1) PB_DW(3, 1);
        else if (bl.k == 3 && bl.stride == 2) PB_DW(3, 2);
        else if (bl.k == 5 && bl.stride == 1) PB_DW(5, 1);
        else PB_DW(5, 2);
#undef PB_DW
    } else {
#define PB_DWR(KS, S, TX)                                                                                            \
    hipLaunchKernelGGL((k_dwconv_roll<KS, S, TX>), grid, block, lds, e->stream, in, H, W, bl.e, bl.dw_w, bl.dw_b, out, Ho, \
                       Wo, g.strips_per_tile, e->buf_part, g.n_tiles, g.cqpb)
        if (bl.k == 3 && bl.stride == 1) PB_DWR(3, 1, 4);
        else if (bl.k == 3 && bl.stride == 2) PB_DWR(3, 2, 4);
        else if (bl.k == 5 && bl.stride == 1) PB_DWR(5, 1, 2);
        else PB_DWR(5, 2, 2);
#undef PB_DWR
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// Depthwise launch; the kernel form (strip vs rolling window) is measured per (layer, batch) at first use.
int launch_dw(pb_embedder *e, const Block &bl, const float *in, int B, int H, int W, float *out, int Ho, int Wo,
              DwGeom *used) {
    const std::pair<const void *, long> key(bl.dw_w, tune_bucket(B));
    auto it = e->dw_cfg.find(key);
    if (it == e->dw_cfg.end()) {
        TuneTimer tt(e);
        DwGeom cands[3] = {dw_geom(bl.e, Ho, Wo, B, e->n_cu), dw_geom_roll(bl.e, bl.k, Ho, Wo, B, e->n_cu),
                           dw_geom_lds(bl.e, bl.k, H, W, Ho, Wo, B, e->n_cu)};
        int best = 0;
        if (cands[1].roll || cands[2].roll) {
            float best_ms = 1e30f;
            const hipEvent_t e0 = e->tune_e0, e1 = e->tune_e1;  // the embedder's own pair: nothing to release on an early return
            for (int c = 0; c < 3; ++c) {
                if (c > 0 && !cands[c].roll) continue;  // form not applicable to this layer
                int rc = launch_dw_geom(e, bl, in, B, H, W, out, Ho, Wo, cands[c]);
                if (rc) return rc;
                PB_HIP(hipEventRecord(e0, e->stream));
                for (int rep = 0; rep < 2; ++rep)
                    if ((rc = launch_dw_geom(e, bl, in, B, H, W, out, Ho, Wo, cands[c]))) return rc;
                PB_HIP(hipEventRecord(e1, e->stream));
                PB_HIP(hipEventSynchronize(e1));
                float ms = 0.f;
                PB_HIP(hipEventElapsedTime(&ms, e0, e1));
                if (tune_take(e, ms, best_ms)) {
                    best_ms = ms;
                    best = c;
                }
            }
        }
        it = e->dw_cfg.emplace(key, cands[best]).first;
    }
    *used = it->second;
    return launch_dw_geom(e, bl, in, B, H, W, out, Ho, Wo, it->second);
}

// the squeeze-excite tail of a producing kernel (pb_embed_kernels.h SeTail); sp = 0 when the gates come from k_se
SeTail se_tail(pb_embedder *e, const Block &bl, int Ho, int Wo) {
    SeTail t;
    t.w1 = bl.se_w1; t.b1 = bl.se_b1; t.w2t = bl.se_w2t; t.b2 = bl.se_b2;
    t.gate = e->buf_gate; t.cnt = e->d_se_cnt;
    t.inv_hw = 1.0f / (float)(Ho * Wo);
    t.sp = (e->fold_se && bl.sp <= 16) ? bl.sp : 0;  // small excite weights only (see SeTail)
    return t;
}

// ---- fused MBConv front (expand + depthwise in one kernel, expanded rows in registers) -------------------------
bool front_eligible(const Block &bl) {
    const int kc = bl.expand.Kpad / 16;
    if (bl.expand.p3) return false;  // k_front_roll expands on the f32 chain
    if (!bl.has_expand || bl.e % 48 || bl.expand.Kpad % 16 || bl.cin % 4) return false;
    return (bl.k == 3 && bl.stride == 2 && (kc == 1 || kc == 3)) || (bl.k == 3 && bl.stride == 1 && kc == 2) ||
           (bl.k == 5 && bl.stride == 2 && kc == 2) || (bl.k == 5 && bl.stride == 1 && kc == 3);
}

int front_strips(const Block &bl, int Wo) {
    const int ow = (16 - bl.k) / bl.stride + 1;
    return (Wo + ow - 1) / ow;
}

// cfg = 16 * n_bands + nc: n_bands row bands per strip, nc (1 or 3) 16-channel tiles per wave
int launch_front(pb_embedder *e, const Block &bl, int cfg, const float *x, int B, int H, int W, float *out, int Ho, int Wo) {
    const int n_bands = cfg >> 4, nc = cfg & 15;
    const int n_strips = front_strips(bl, Wo);
    const int rows_per_band = (Ho + n_bands - 1) / n_bands;
    const dim3 grid((n_strips * n_bands + 3) / 4, B, bl.e / (16 * nc)), block(256);
    const int kc = bl.expand.Kpad / 16;
#define PB_FR1(KS, S, KC, NCV)                                                                                             \
    hipLaunchKernelGGL((k_front_roll<KS, S, KC, NCV>), grid, block, 0, e->stream, x, H, W, bl.cin, bl.expand.wt, bl.expand.Npad, \
                       bl.expand.bias, bl.dw_w, bl.dw_b, bl.e, out, Ho, Wo, e->buf_part, n_strips, n_bands, rows_per_band)
#define PB_FR(KS, S, KC)             \
    do {                             \
        if (nc == 1) PB_FR1(KS, S, KC, 1); \
        else PB_FR1(KS, S, KC, 3);   \
    } while (0)
    if (bl.k == 3 && bl.stride == 2 && kc == 1) PB_FR(3, 2, 1);
    else if (bl.k == 3 && bl.stride == 2 && kc == 3) PB_FR(3, 2, 3);
    else if (bl.k == 3 && bl.stride == 1 && kc == 2) PB_FR(3, 1, 2);
    else if (bl.k == 5 && bl.stride == 2 && kc == 2) PB_FR(5, 2, 2);
    else if (bl.k == 5 && bl.stride == 1 && kc == 3) PB_FR(5, 1, 3);
    else PB_CHECK(false, PB_ERR_INVALID, "launch_front: no fused kernel for k%d s%d Kpad %d", bl.k, bl.stride, bl.expand.Kpad);
#undef PB_FR
#undef PB_FR1
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// ---- fused MBConv front for small maps (k_mbconv_small): cfg = 0x1000 + 256 * mr + 16 * nr + 8 * regs + log2(groups per workgroup)
bool small_eligible(const Block &bl, int H, int W, int nr, int mr) {
    const int P = H * W;
    // a P3 expand layer: the operands-in-registers form only (its P3E instances: k_mbconv_small, K = 80 / 112 / 192 on 8 x 8 and 4 x 4 maps)
    if (bl.expand.p3 && !(mr == 1 && (bl.expand.Kpad == 80 || bl.expand.Kpad == 112 || bl.expand.Kpad == 192) && bl.cin % 8 == 0)) return false;
    if (!bl.has_expand || H != W || bl.e % (16 * nr) || bl.expand.Kpad % 16 || bl.cin % 4) return false;
    if (!((P == 256 && mr == 4) || (P == 64 && mr == 1) || (P == 16 && mr == 1))) return false;  // two row tiles per wave never won
    if (P == 256) return (bl.k == 5 && bl.stride == 1) || (bl.k == 3 && bl.stride == 2);
    return (bl.k == 3 && bl.stride == 1) || (bl.k == 5 && bl.stride == 1) || (bl.k == 5 && bl.stride == 2);
}

// the k-step counts the operands-in-registers form is instantiated for (EfficientNet-B0 widths 40 / 80 / 112 / 192)
bool small_regs_form(const Block &bl, int mr) {
    const int ns = bl.expand.Kpad / 16;
    return mr == 4 ? ns == 3 : (ns == 5 || ns == 7 || ns == 12);
}

size_t small_lds_bytes(const Block &bl, int H, int W, int nr, int mr) {
    const int nt = 16 * nr, pad = (bl.k - 1) / 2, g = 64 * mr / (H * W);
    const size_t w_floats = bl.expand.p3 ? (size_t)((bl.expand.Kpad / 16 + 1) / 2) * nr * 768 : (size_t)bl.expand.Kpad * (nt + 4);
    return (w_floats + (size_t)g * (H + 2 * pad) * ((W + 2 * pad) | 1) * nt + (size_t)bl.k * bl.k * nt + 2 * nt) * sizeof(float);
}

template <int KS, int S, int NR, int MR, int NS, bool P3E = false>
int launch_small_t(pb_embedder *e, const Block &bl, int gpw, const float *x, int B, int H, int W, float *out, int Ho, int Wo) {
    const size_t lds = small_lds_bytes(bl, H, W, NR, MR);
    auto kern = k_mbconv_small<KS, S, NR, MR, NS, P3E>;
    if (lds > 48 * 1024)
        PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int G = 64 * MR / (H * W), n_groups = (B + G - 1) / G;
    hipLaunchKernelGGL(kern, dim3((n_groups + gpw - 1) / gpw, 1, bl.e / (16 * NR)), dim3(256), lds, e->stream, x, H, W, bl.cin, bl.expand.wt,
                       bl.expand.Kpad, bl.expand.Npad, bl.expand.bias, bl.dw_w, bl.dw_b, bl.e, out, Ho, Wo, e->buf_part, B, gpw,
                       reinterpret_cast<const u32x4 *>(bl.expand.wt3), bl.expand.Npad / 16);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// NS = k-steps held in registers (Kpad / 16 for the EfficientNet-B0 widths 40 / 80 / 112 / 192), 0 = streaming form
template <int KS, int S, int NR, int MR>
int launch_small_ns(pb_embedder *e, const Block &bl, int gpw, bool regs, const float *x, int B, int H, int W, float *out, int Ho, int Wo) {
    const int ns = regs ? bl.expand.Kpad / 16 : 0;
#define PB_SMN(NSV) launch_small_t<KS, S, NR, MR, NSV>(e, bl, gpw, x, B, H, W, out, Ho, Wo)
    if constexpr (MR == 1) {
        if (bl.expand.p3) {  // the layer's arithmetic is P3: only the P3E instances compute it (small_eligible admits nothing else)
            const int nsp = bl.expand.Kpad / 16;
            if (nsp == 5) return launch_small_t<KS, S, NR, MR, 5, true>(e, bl, gpw, x, B, H, W, out, Ho, Wo);
            if (nsp == 7) return launch_small_t<KS, S, NR, MR, 7, true>(e, bl, gpw, x, B, H, W, out, Ho, Wo);
            if (nsp == 12) return launch_small_t<KS, S, NR, MR, 12, true>(e, bl, gpw, x, B, H, W, out, Ho, Wo);
            return pb::fail(PB_ERR_INTERNAL, "launch_small: no P3 expand instance for K = %d", bl.expand.Kpad);
        }
    } else if (bl.expand.p3) {
        return pb::fail(PB_ERR_INTERNAL, "launch_small: no P3 expand instance for MR = %d", MR);
    }
    if constexpr (MR == 4) {
        if (ns == 3) return PB_SMN(3);
    } else {
        if (ns == 5) return PB_SMN(5);
        if (ns == 7) return PB_SMN(7);
        if (ns == 12) return PB_SMN(12);
    }
    return PB_SMN(0);
#undef PB_SMN
}

int launch_small(pb_embedder *e, const Block &bl, int cfg, const float *x, int B, int H, int W, float *out, int Ho, int Wo) {
    const int mr = (cfg >> 8) & 15, nr = (cfg >> 4) & 15, gpw = 1 << (cfg & 7);
    const bool regs = cfg & 8;  // activation operands of a group held in registers (NS > 0) instead of streamed
#define PB_SM2(KS, S, MRV) (nr == 2 ? launch_small_ns<KS, S, 2, MRV>(e, bl, gpw, regs, x, B, H, W, out, Ho, Wo) : launch_small_ns<KS, S, 3, MRV>(e, bl, gpw, regs, x, B, H, W, out, Ho, Wo))
#define PB_SM(KS, S) PB_SM2(KS, S, 1)
    if (mr == 4) return (bl.k == 5) ? PB_SM2(5, 1, 4) : PB_SM2(3, 2, 4);
    if (bl.k == 3 && bl.stride == 1) return PB_SM(3, 1);
    if (bl.k == 5 && bl.stride == 1) return PB_SM(5, 1);
    return PB_SM(5, 2);
#undef PB_SM
#undef PB_SM2
}

// ---- fused MBConv front with the expanded rows in an LDS ring (k_front_band, pb_front_band.h): cfg = 0x2000 + n_bands
struct BandShape {
    int ks, s, cin, wt, rps;
};
// the instantiated shapes: EfficientNet-B0 blocks 1-5 at 128 x 128 input (maps 64 / 32 / 16 wide)
const BandShape BAND_SHAPES[] = {{3, 2, 16, 4, 2}, {3, 1, 24, 2, 2}, {5, 2, 24, 2, 4}, {5, 1, 40, 1, 4}, {3, 2, 40, 1, 8}};

const BandShape *band_shape(const Block &bl, int H, int W) {
    if (!bl.has_expand || !bl.dw_wq || bl.e % 16 || H != W || bl.expand.p3) return nullptr;
    for (const BandShape &bs : BAND_SHAPES)
        if (bs.ks == bl.k && bs.s == bl.stride && bs.cin == bl.cin && bs.wt * 16 == W) return &bs;
    return nullptr;
}

template <int KS, int S, int CIN, int WT, int RPS>
int launch_band_t(pb_embedder *e, const Block &bl, int n_bands_cfg, const float *x, int B, float *out) {
    const int n_bands = n_bands_cfg & 63, ipw = 1 << (n_bands_cfg >> 6);  // items a workgroup walks (k_front_band ipw)
    using G = FrontBandGeom<KS, S, WT, RPS>;
    auto kern = k_front_band<KS, S, CIN, WT, RPS>;
    if (G::LDS_BYTES > 48 * 1024)
        PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
    const unsigned n_items = (unsigned)B * (unsigned)n_bands, n_wgi = (n_items + ipw - 1) / ipw;
    hipLaunchKernelGGL(kern, dim3((n_wgi + 7) / 8 * 8 * (bl.e / 16)), dim3(256), G::LDS_BYTES, e->stream, x, bl.expand.wt, bl.expand.Npad,
                       bl.expand.bias, bl.dw_wq, bl.dw_b, bl.e, out, e->buf_part, n_bands, G::Wo / n_bands, n_items,
                       se_tail(e, bl, G::Wo, G::Wo), ipw);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int launch_band(pb_embedder *e, const Block &bl, const BandShape &bs, int n_bands, const float *x, int B, float *out) {
#define PB_BD(KS, S, CIN, WT, RPS) \
    if (bs.ks == KS && bs.s == S && bs.cin == CIN && bs.wt == WT) return launch_band_t<KS, S, CIN, WT, RPS>(e, bl, n_bands, x, B, out)
    PB_BD(3, 2, 16, 4, 2);
    PB_BD(3, 1, 24, 2, 2);
    PB_BD(5, 2, 24, 2, 4);
    PB_BD(5, 1, 40, 1, 4);
    PB_BD(3, 2, 40, 1, 8);
#undef PB_BD
    return pb::fail(PB_ERR_INVALID, "launch_band: no kernel for k%d s%d", bs.ks, bs.s);
}

int launch_gemm(pb_embedder *e, const float *act, long M, const Gemm &g, const float *gate, int hw, const float *resid,
                int do_silu, float *out);
int launch_dw(pb_embedder *e, const Block &bl, const float *in, int B, int H, int W, float *out, int Ho, int Wo, DwGeom *used);

// expand + depthwise of one block: the fused kernel (per row-band count) and the two-kernel path are timed on
// the real buffers at first use per (block, batch); returns the number of SE partial tiles written to buf_part
int run_front(pb_embedder *e, const Block &bl, const float *x, int n, int H, int W, int Ho, int Wo, int *n_part_tiles, bool *folded) {
    *folded = false;
    const std::pair<const void *, long> key(bl.expand.wt, tune_bucket(n));
    auto separate = [&](int *tiles) -> int {
        int rc = launch_gemm(e, x, (long)n * H * W, bl.expand, nullptr, 1, nullptr, 1, e->buf_e);
        if (rc) return rc;
        DwGeom g;
        rc = launch_dw(e, bl, e->buf_e, n, H, W, e->buf_dw, Ho, Wo, &g);
        *tiles = g.n_tiles;
        return rc;
    };
    const int n_strips = front_strips(bl, Wo);
    auto bands_used = [&](int nb) { const int rpb = (Ho + nb - 1) / nb; return (Ho + rpb - 1) / rpb; };
    auto it = e->front_cfg.find(key);
    if (it == e->front_cfg.end()) {
        TuneTimer tt(e);
        int tiles = 0;
        int rc = separate(&tiles);  // also warms the GEMM / depthwise selections
        if (rc) return rc;
        const hipEvent_t e0 = e->tune_e0, e1 = e->tune_e1;  // the embedder's own pair: nothing to release on an early return
        auto time_it = [&](int nb, float *ms) -> int {
            PB_HIP(hipEventRecord(e0, e->stream));
            for (int rep = 0; rep < 2; ++rep) {
                int r2 = nb ? launch_front(e, bl, nb, x, n, H, W, e->buf_dw, Ho, Wo) : separate(&tiles);
                if (r2) return r2;
            }
            PB_HIP(hipEventRecord(e1, e->stream));
            PB_HIP(hipEventSynchronize(e1));
            PB_HIP(hipEventElapsedTime(ms, e0, e1));
            return PB_OK;
        };
        int best = 0;
        float best_ms = 0.f;
        if ((rc = time_it(0, &best_ms))) return rc;
        const float sep_ms = best_ms;
        for (int nc : {3, 1})
            for (int nb : {1, 2, 4, 8}) {
                if (!front_eligible(bl) || bands_used(nb) != nb || (nb > 1 && Ho / nb < 4) ||
                    (size_t)n_strips * nb * bl.e > e->part_floats_per_image)
                    continue;
                const int cfg = 16 * nb + nc;
                float ms = 0.f;
                if ((rc = launch_front(e, bl, cfg, x, n, H, W, e->buf_dw, Ho, Wo))) return rc;  // warm-up
                if ((rc = time_it(cfg, &ms))) return rc;
                if (e->trace_tune) fprintf(stderr, "front k%d s%d e%d n%d: bands %d nc %d %.1f us (separate %.1f)\n", bl.k, bl.stride, bl.e, n, nb, nc, ms * 500.f, sep_ms * 500.f);
                if (tune_take(e, ms, best_ms)) {
                    best_ms = ms;
                    best = cfg;
                }
            }
        for (int mr : {1, 4})
            for (int nr : {3, 2})
                for (int lg : {0, 1, 2, 9, 10}) {  // 8 + lg: operands-in-registers form (needs >= 2 groups per workgroup to pay)
                    if (!small_eligible(bl, H, W, nr, mr) || small_lds_bytes(bl, H, W, nr, mr) > 100 * 1024) continue;
                    if ((lg & 8) && !small_regs_form(bl, mr)) continue;
                    if (bl.expand.p3 && !(lg & 8) && lg != 0) continue;  // a P3 layer always runs the operands-in-registers form: the plain cfgs alias it
                    const int cfg = 0x1000 + 256 * mr + 16 * nr + lg;
                    float ms = 0.f;
                    if ((rc = launch_small(e, bl, cfg, x, n, H, W, e->buf_dw, Ho, Wo))) return rc;  // warm-up
                    PB_HIP(hipEventRecord(e0, e->stream));
                    for (int rep = 0; rep < 2; ++rep)
                        if ((rc = launch_small(e, bl, cfg, x, n, H, W, e->buf_dw, Ho, Wo))) return rc;
                    PB_HIP(hipEventRecord(e1, e->stream));
                    PB_HIP(hipEventSynchronize(e1));
                    PB_HIP(hipEventElapsedTime(&ms, e0, e1));
                    if (e->trace_tune)
                        fprintf(stderr, "front k%d s%d e%d n%d: small-map fused mr %d nr %d groups/wg %d%s %.1f us (separate %.1f)\n", bl.k, bl.stride, bl.e, n,
                                mr, nr, 1 << (lg & 7), (lg & 8) ? " regs" : "", ms * 500.f, sep_ms * 500.f);
                    if (tune_take(e, ms, best_ms)) {
                        best_ms = ms;
                        best = cfg;
                    }
                }
        if (const BandShape *bs = band_shape(bl, H, W)) {
            const int steps = Ho / bs->rps;
            for (int nb : {1, 2, 4, 8, 16})
                for (int lg_ipw : {0, 1, 2, 3}) {  // items a workgroup walks: 1, 2, 4, 8 (round 6; one only with a squeeze-excite tail)
                    if (steps % nb || (size_t)nb * bl.e > e->part_floats_per_image || e->no_band) continue;
                    if (lg_ipw && (e->fold_se || e->no_band_ipw || (long)n * nb < (1L << lg_ipw) * 2L * e->n_cu / (bl.e / 16) + 1)) continue;
                    const int cfgb = nb + 64 * lg_ipw;
                    float ms = 0.f;
                    if ((rc = launch_band(e, bl, *bs, cfgb, x, n, e->buf_dw))) return rc;  // warm-up
                    PB_HIP(hipEventRecord(e0, e->stream));
                    for (int rep = 0; rep < 2; ++rep)
                        if ((rc = launch_band(e, bl, *bs, cfgb, x, n, e->buf_dw))) return rc;
                    PB_HIP(hipEventRecord(e1, e->stream));
                    PB_HIP(hipEventSynchronize(e1));
                    PB_HIP(hipEventElapsedTime(&ms, e0, e1));
                    if (e->trace_tune)
                        fprintf(stderr, "front k%d s%d e%d n%d: LDS-ring band kernel, bands %d, %d item(s) per workgroup %.1f us (separate %.1f)\n", bl.k, bl.stride,
                                bl.e, n, nb, 1 << lg_ipw, ms * 500.f, sep_ms * 500.f);
                    if (tune_take(e, ms, best_ms) || (e->force_band && !lg_ipw)) {
                        if (e->force_band && best >= 0x2000 && ms >= best_ms) continue;
                        best_ms = ms;
                        best = 0x2000 + cfgb;
                    }
                }
        }
        it = e->front_cfg.emplace(key, best).first;
    }
    if (it->second >= 0x2000) {
        *n_part_tiles = (it->second - 0x2000) & 63;
        *folded = e->fold_se && bl.sp <= 16;  // the band kernel's last workgroup per image has written the gate
        return launch_band(e, bl, *band_shape(bl, H, W), it->second - 0x2000, x, n, e->buf_dw);
    }
    if (it->second >= 0x1000) {
        *n_part_tiles = 1;
        return launch_small(e, bl, it->second, x, n, H, W, e->buf_dw, Ho, Wo);
    }
    if (it->second) {
        *n_part_tiles = n_strips * (it->second >> 4);
        return launch_front(e, bl, it->second, x, n, H, W, e->buf_dw, Ho, Wo);
    }
    return separate(n_part_tiles);
}

// forward for n images already on the device; results to device buffers
// ---- a whole MBConv block of a 4 x 4 map in one kernel (k_block_small, pb_block_small.h)
template <int KS, int CIN, int E, int COUT, int HW, int G, int SP, bool RESID, bool P3, bool P3E = false>
int launch_block_t(pb_embedder *e, const Block &bl, const float *x, int n, float *out) {
    using GEO = BlockGeom<KS, CIN, E, COUT, HW, G, SP, P3>;
    auto kern = k_block_small<KS, CIN, E, COUT, HW, G, SP, RESID, 0, P3, P3E>;
    static_assert(GEO::LDS_BYTES <= 160 * 1024, "one CU's LDS");
    bool &attr_set = e->block_attr_set[(RESID ? 0 : 1) + (P3E ? 2 : 0)];
    if (!attr_set) {  // once per embedder and instantiation (the attribute is per device function and context)
        PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::LDS_BYTES));
        attr_set = true;
    }
    BlockW w{};
    w.we2 = bl.expand.wt4; w.we3 = bl.expand.wt3; w.be = bl.expand.bias; w.dwc = bl.dw_wc; w.bd = bl.dw_b;
    w.w1 = bl.se_w1; w.b1 = bl.se_b1; w.w2t = bl.se_w2t; w.b2 = bl.se_b2;
    w.wp2 = bl.project.wt4; w.wp3 = bl.project.wt3; w.bp = bl.project.bias; w.nt16 = bl.project.Npad / 16;
    w.range_slot = e->buf_part - 1;
    hipLaunchKernelGGL(kern, dim3((n + G - 1) / G), dim3(512), GEO::LDS_BYTES, e->stream, x, w, out, n);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// 0: not a shape the fused kernel is built for
int block_shape(const pb_embedder *e, const Block &bl, int H, int W) {
    if (e->no_block_fusion || !bl.has_expand || bl.stride != 1 || !bl.expand.wt4 || !bl.project.wt4 || !bl.dw_wc) return 0;
    if (bl.expand.p3 && !bl.project.p3) return 0;  // (a combination only an A/B switch produces: not instantiated)
    if (H == 4 && W == 4 && bl.cin == 192 && bl.e == 1152 && bl.sp == 48) {
        if (bl.k == 5 && bl.cout == 192 && bl.residual) return 1;
        if (bl.k == 3 && bl.cout == 320 && !bl.residual) return 2;
    }
    return 0;
}

int launch_block(pb_embedder *e, const Block &bl, int shape, const float *x, int n, float *out) {
    // the project phase follows the layer's arithmetic (P3 or the f32 chain)
    // the expand and project phases follow their layers' arithmetic (P3 or the f32 chain)
    if (shape == 1) {
        if (bl.expand.p3) return launch_block_t<5, 192, 1152, 192, 4, 2, 48, true, true, true>(e, bl, x, n, out);
        return bl.project.p3 ? launch_block_t<5, 192, 1152, 192, 4, 2, 48, true, true>(e, bl, x, n, out)
                             : launch_block_t<5, 192, 1152, 192, 4, 2, 48, true, false>(e, bl, x, n, out);
    }
    if (shape == 2) {
        if (bl.expand.p3) return launch_block_t<3, 192, 1152, 320, 4, 2, 48, false, true, true>(e, bl, x, n, out);
        return bl.project.p3 ? launch_block_t<3, 192, 1152, 320, 4, 2, 48, false, true>(e, bl, x, n, out)
                             : launch_block_t<3, 192, 1152, 320, 4, 2, 48, false, false>(e, bl, x, n, out);
    }
    return PB_ERR_INTERNAL;
}

// squeeze-excite gates of n images from the pooled sums in buf_part -> buf_gate (k_se)
int launch_se(pb_embedder *e, const Block &bl, int n_tiles, int n, int hw, int se_qp) {
    // a few images: eight workgroups per image (k_se_multi: same bits, a workgroup reads an eighth of the weights)
    if (n <= e->se_multi_max && bl.e / 4 <= 320 && e->d_se_xchg) {
#define PB_SEM(SPV)                                                                                                              \
    hipLaunchKernelGGL((k_se_multi<SPV>), dim3(SEM_WG, n), dim3(((bl.e / 4 + 63) / 64) * 64), 0, e->stream, e->buf_part, n_tiles, bl.e, \
                       1.0f / (float)hw, bl.se_w1, bl.se_b1, bl.se_w2t, bl.se_b2, e->buf_gate, e->d_se_xchg, e->d_se_arrive)
        if (bl.sp == 8) PB_SEM(8);
        else if (bl.sp == 16) PB_SEM(16);
        else if (bl.sp == 32) PB_SEM(32);
        else PB_SEM(48);
#undef PB_SEM
        PB_HIP(hipGetLastError());
        return PB_OK;
    }
#define PB_SE1(SPV, IMGV)                                                                                                  \
    hipLaunchKernelGGL((k_se<SPV, IMGV>), dim3((n + (IMGV) - 1) / (IMGV)), dim3(se_qp * ((SPV) < 16 ? 1 : (SPV) / 16)), 0, e->stream, \
                       e->buf_part, n_tiles, bl.e, 1.0f / (float)hw, bl.se_w1, bl.se_b1, bl.se_w2t, bl.se_b2, e->buf_gate, se_qp, n)
    // the widest layers (2 x 48 x 1152 weights = 442 KB per block) run two images per block from 64 images
    // on: measured 24 -> 19 us per launch at batch 512; the narrower ones lose more parallelism than they
    // save traffic (9 -> 13 us) and keep one image per block
    if (bl.sp == 8) PB_SE1(8, 1);
    else if (bl.sp == 16) PB_SE1(16, 1);
    else if (bl.sp == 32) PB_SE1(32, 1);
    else if (n >= 64) PB_SE1(48, 2);
    else PB_SE1(48, 1);
#undef PB_SE1
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// stem + the first `nb` blocks (all of them: nb = blocks.size()) of n images.  The last of those blocks writes to `out_last` when
// that is given, otherwise the blocks ping-pong over buf_x as always; *x_out is where the next block finds its input.
int forward_blocks(pb_embedder *e, const uint8_t *d_rgb, int n, size_t nb, float *out_last, const float **x_out, int *H_out, int *W_out);
int forward_rest(pb_embedder *e, const float *x_in, int n, size_t first, int H, int W, uint8_t *d_u8, float *d_f32);

// The leading blocks' maps are the big ones (128 x 128 input, batch 512: the depthwise outputs of blocks 0-4 are 268, 201, 302, 75
// and 126 MB, each written by one kernel and read back by the next).  With front_sub > 0 (PB_OPT_EMBED_FRONT_SUB; off by default: it
// measured slower, see pb_embed_create) stem .. block front_blocks - 1 run over sub-batches whose maps would stay inside the 256 MiB
// Infinity Cache between their writer and their reader, the rest of the network over the whole batch.  A host-side loop: every
// kernel computes an image's values from that image alone, so the bits are those of the one-pass form.
int forward_one(pb_embedder *e, const uint8_t *d_rgb, int n, uint8_t *d_u8, float *d_f32);
int ws2_alloc(pb_embedder *e);

void ws_swap(pb_embedder *e) {
    pb_embedder::WsSet &w = e->ws2;
    std::swap(e->buf_x[0], w.buf_x[0]); std::swap(e->buf_x[1], w.buf_x[1]); std::swap(e->buf_e, w.buf_e); std::swap(e->buf_dw, w.buf_dw);
    std::swap(e->buf_gate, w.buf_gate); std::swap(e->buf_pool, w.buf_pool); std::swap(e->buf_front, w.buf_front);
    std::swap(e->buf_part, w.buf_part); std::swap(e->d_se_cnt, w.d_se_cnt);
}

int forward_device(pb_embedder *e, const uint8_t *d_rgb, int n, uint8_t *d_u8, float *d_f32) {
    if (e->dual_min <= 0 || n < e->dual_min || e->tune_depth > 0) return forward_one(e, d_rgb, n, d_u8, d_f32);
    int rc;
    if (!e->ws2_ready && (rc = ws2_alloc(e))) return rc;
    const int n0 = (n + 1) / 2, n1 = n - n0;
    const uint8_t *rgb1 = d_rgb + (size_t)n0 * e->H * e->W * 3;
    uint8_t *u81 = d_u8 + (size_t)n0 * e->D;
    float *f321 = d_f32 ? d_f32 + (size_t)n0 * e->D : nullptr;
    hipStream_t s0 = e->stream;
    // the second half starts where the caller's stream stands NOW (its input is there), not behind the first half
    PB_HIP(hipEventRecord(e->dual_e0, s0));
    const unsigned long before = e->tune_runs;
    if ((rc = forward_one(e, d_rgb, n0, d_u8, d_f32))) return rc;  // first half, on the caller's stream
    // a call that ran timing loops (the per-layer picks of this batch bucket) finishes one half after the other: a timing loop
    // must have the device to itself.  So does a split whose halves fall into different buckets.
    if (e->tune_runs != before || tune_bucket(n0) != tune_bucket(n1)) return forward_one(e, rgb1, n1, u81, f321);
    PB_HIP(hipStreamWaitEvent(e->dual_stream, e->dual_e0, 0));
    ws_swap(e);
    e->stream = e->dual_stream;
    rc = forward_one(e, rgb1, n1, u81, f321);
    e->stream = s0;
    ws_swap(e);
    if (rc) {
        (void)hipStreamSynchronize(e->dual_stream);
        return rc;
    }
    PB_HIP(hipEventRecord(e->dual_e1, e->dual_stream));
    PB_HIP(hipStreamWaitEvent(s0, e->dual_e1, 0));
    return PB_OK;
}

// the second workspace: what one half of max_batch images needs (the first half keeps using the embedder's own buffers)
int ws2_alloc(pb_embedder *e) {
    pb::DeviceGuard guard(e->device);
    const size_t B = (e->max_batch + 1) / 2;
    pb_embedder::WsSet &w = e->ws2;
    int rc;
    if ((rc = dalloc(e, &w.buf_x[0], B * e->ws_max_x)) || (rc = dalloc(e, &w.buf_x[1], B * e->ws_max_x))) return rc;
    if ((rc = dalloc(e, &w.buf_e, B * e->ws_max_e)) || (rc = dalloc(e, &w.buf_dw, B * e->ws_max_dw))) return rc;
    if ((rc = dalloc(e, &w.buf_part, B * e->part_floats_per_image + 2)) || (rc = dalloc(e, &w.buf_gate, B * std::max<size_t>(1152, e->D)))) return rc;
    PB_HIP(hipMemcpy(w.buf_part, e->buf_part - 2, 2 * sizeof(long long), hipMemcpyDeviceToDevice));  // the header: the range word's address
    w.buf_part += 2;
    if ((rc = dalloc(e, &w.buf_pool, B * 1280)) || (rc = dalloc(e, &w.d_se_cnt, B))) return rc;
    PB_HIP(hipMemset(w.d_se_cnt, 0, B * sizeof(unsigned)));
    if (e->front_blocks && (rc = dalloc(e, &w.buf_front, B * e->front_out_per_image))) return rc;
    PB_HIP(hipStreamCreateWithFlags(&e->dual_stream, hipStreamNonBlocking));
    PB_HIP(hipEventCreateWithFlags(&e->dual_e0, hipEventDisableTiming));
    PB_HIP(hipEventCreateWithFlags(&e->dual_e1, hipEventDisableTiming));
    e->ws2_ready = true;
    return PB_OK;
}

int forward_one(pb_embedder *e, const uint8_t *d_rgb, int n, uint8_t *d_u8, float *d_f32) {
    int H = 0, W = 0, rc;
    const float *x = nullptr;
    const size_t fb = std::min<size_t>(e->front_blocks, e->blocks.size());
    const int sub = e->front_sub;
    if (fb > 0 && sub > 0 && n > sub && e->buf_front) {
        for (int s = 0; s < n; s += sub) {
            const int m = std::min(sub, n - s);
            if ((rc = forward_blocks(e, d_rgb + (size_t)s * e->H * e->W * 3, m, fb, e->buf_front + (size_t)s * e->front_out_per_image, &x, &H, &W))) return rc;
        }
        return forward_rest(e, e->buf_front, n, fb, H, W, d_u8, d_f32);
    }
    if ((rc = forward_blocks(e, d_rgb, n, e->blocks.size(), nullptr, &x, &H, &W))) return rc;
    return forward_rest(e, x, n, e->blocks.size(), H, W, d_u8, d_f32);
}

int run_block(pb_embedder *e, const Block &bl, const float *x, float *out, int n, int H, int W, bool stem_fused, int stem_bands);

int forward_blocks(pb_embedder *e, const uint8_t *d_rgb, int n, size_t nb, float *out_last, const float **x_out, int *H_out, int *W_out) {
    int H = (int)e->H / 2, W = (int)e->W / 2;
    // arrival counters of the squeeze-excite tails: every tail leaves them zero; cleared anyway (a failed launch must not poison the next forward)
    if (e->fold_se) PB_HIP(hipMemsetAsync(e->d_se_cnt, 0, (size_t)n * sizeof(unsigned), e->stream));
    // the first block (no expansion, 3x3 stride 1, 32 channels, no residual) takes its depthwise conv fused into the
    // stem when the 3-row LDS ring fits (input width <= 320); identical bits either way
    const Block &b0 = e->blocks.front();
    // bands of 16 output rows (2 stem rows of halo each), fewer rows per band for small batches (<= 32 partial tiles)
    int rpb = 16;
    while (rpb > 4 && (long)n * ((H + rpb - 1) / rpb) < 2L * e->n_cu && (H + rpb / 2 - 1) / (rpb / 2) <= 32) rpb /= 2;
    // stem rows per phase (pb_embed_kernels.h, k_stem_dw): 2 where the 4-slot ring still leaves three workgroups per CU
    auto stem_lds = [&](int r) { return (size_t)(r + 2) * (W + 2) * 36 * sizeof(float) + (size_t)(2 * rpb + 5) * (e->W * 3 + 4); };
    const int rpp = e->stem_rpp ? e->stem_rpp : (3 * stem_lds(2) <= 160 * 1024 ? 2 : 1);
    const size_t fused_lds = stem_lds(rpp);
    const bool fuse_stem = !b0.has_expand && b0.k == 3 && b0.stride == 1 && b0.e == 32 && fused_lds <= 60 * 1024 &&
                           (H + rpb - 1) / rpb <= 32 && !e->no_stem_fusion;
    int stem_bands = 0;
    if (fuse_stem) {
        stem_bands = (H + rpb - 1) / rpb;
        auto go = [&](auto kern) -> int {
            if (fused_lds > 48 * 1024)
                PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds));
            hipLaunchKernelGGL(kern, dim3(stem_bands, n), dim3(256), fused_lds, e->stream, d_rgb, n, (int)e->H, (int)e->W,
                               reinterpret_cast<const u32x4s *>(e->stem_w3), e->stem_b, b0.dw_w, b0.dw_b, e->buf_dw, e->buf_part, stem_bands, rpb,
                               se_tail(e, b0, H, W));
            return PB_OK;
        };
        const int rc_s = rpp == 1 ? go(k_stem_dw<1>) : go(k_stem_dw<2>);
        if (rc_s) return rc_s;
        PB_HIP(hipGetLastError());
    } else {
        // one block per output row, taps from LDS-staged input rows (W is a multiple of 32 and <= 1024: 37 KB at most)
        const int grid = (int)std::min<long>((long)n * H, (long)e->n_cu * 16);
        hipLaunchKernelGGL(k_stem, dim3(grid), dim3(256), (size_t)3 * (e->W * 3 + 4), e->stream, d_rgb, n,
                           (int)e->H, (int)e->W, reinterpret_cast<const u32x4s *>(e->stem_w3), e->stem_b, e->buf_x[0]);
        PB_HIP(hipGetLastError());
    }
    int cur = 0;
    const float *x = e->buf_x[0];
    for (size_t bi = 0; bi < nb; ++bi) {
        const Block &bl = e->blocks[bi];
        float *out = (bi + 1 == nb && out_last) ? out_last : e->buf_x[cur ^ 1];
        int rc = run_block(e, bl, x, out, n, H, W, fuse_stem && bi == 0, stem_bands);
        if (rc) return rc;
        x = out;
        cur ^= 1;
        H = (H + bl.stride - 1) / bl.stride;
        W = (W + bl.stride - 1) / bl.stride;
    }
    *x_out = x;
    *H_out = H;
    *W_out = W;
    return PB_OK;
}

// one MBConv block of n images: x [n][H][W][cin] -> out [n][Ho][Wo][cout]; stem_fused: the depthwise output and the pooled sums of
// block 0 are already in buf_dw / buf_part (k_stem_dw)
int run_block(pb_embedder *e, const Block &bl, const float *x, float *out, int n, int H, int W, bool stem_fused, int stem_bands) {
    const int Ho = (H + bl.stride - 1) / bl.stride, Wo = (W + bl.stride - 1) / bl.stride;
    // the unfused form of the block: front (expand + depthwise + pooled sums), squeeze-excite gates, project GEMM
    auto run_unfused = [&]() -> int {
        int rc;
        int part_tiles = 0;
        bool folded = false;  // the gate was computed in the tail of the kernel that produced the pooled sums
        if (stem_fused) {
            part_tiles = stem_bands;  // depthwise output and SE partials are already in buf_dw / buf_part
            folded = e->fold_se && bl.sp <= 16;  // ... and the gate, written by the band that completed the image
        } else if (bl.has_expand) {
            if ((rc = run_front(e, bl, x, n, H, W, Ho, Wo, &part_tiles, &folded))) return rc;
        } else {
            DwGeom g0;
            if ((rc = launch_dw(e, bl, x, n, H, W, e->buf_dw, Ho, Wo, &g0))) return rc;
            part_tiles = g0.n_tiles;
        }
        struct { int n_tiles; } g{part_tiles};
        // block = (channel quads rounded up to a wave multiple) x (groups of 16 squeeze units)
        const int se_qp = ((bl.e / 4 + 63) / 64) * 64;
        const long Mo = (long)n * Ho * Wo;
        if (!folded && (rc = launch_se(e, bl, g.n_tiles, n, Ho * Wo, se_qp))) return rc;
        if ((rc = launch_gemm(e, e->buf_dw, Mo, bl.project, e->buf_gate, Ho * Wo, bl.residual ? x : nullptr, 0,
                              out)))
            return rc;
        return PB_OK;
    };
    int rc;
    if (const int shape = block_shape(e, bl, H, W)) {
        // one kernel for the whole block, or the unfused form -- identical bits; which is faster depends on the batch (the
        // fused kernel runs two images per CU at a fixed ~110 us: it wins from about one workgroup per CU on), so both are
        // timed once per (block, batch bucket)
        const std::pair<const void *, long> key(&bl, tune_bucket(n));
        auto it = e->front_cfg.find(key);
        if (it == e->front_cfg.end()) {
            TuneTimer tt(e);
            if ((rc = run_unfused())) return rc;  // lets the unfused kernels pick their own forms first
            float best_ms = 1e30f;
            int best = 0;
            for (int cand = 0; cand < 2; ++cand) {
                // one untimed run first: a kernel's first launch carries its code load (the first of the three 5 x 5 blocks
                // kept losing to the unfused form by exactly that)
                if ((rc = cand ? launch_block(e, bl, shape, x, n, out) : run_unfused())) return rc;
                PB_HIP(hipEventRecord(e->tune_e0, e->stream));
                for (int rep = 0; rep < 2; ++rep)
                    if ((rc = cand ? launch_block(e, bl, shape, x, n, out) : run_unfused())) return rc;
                PB_HIP(hipEventRecord(e->tune_e1, e->stream));
                PB_HIP(hipEventSynchronize(e->tune_e1));
                float ms = 0.f;
                PB_HIP(hipEventElapsedTime(&ms, e->tune_e0, e->tune_e1));
                if (e->trace_tune) fprintf(stderr, "block e%d k%d n%d: %s %.1f us\n", bl.e, bl.k, n, cand ? "one kernel" : "front + se + project", ms * 500.f);
                if (tune_take(e, ms, best_ms)) {
                    best_ms = ms;
                    best = cand;
                }
            }
            it = e->front_cfg.emplace(key, best).first;
        }
        if ((rc = it->second ? launch_block(e, bl, shape, x, n, out) : run_unfused())) return rc;
    } else if ((rc = run_unfused())) {
        return rc;
    }
    return PB_OK;
}

int forward_rest(pb_embedder *e, const float *x_in, int n, size_t first, int H, int W, uint8_t *d_u8, float *d_f32) {
    // the ping-pong continues on whichever of buf_x the input is not
    const float *xr = x_in;
    for (size_t bi = first; bi < e->blocks.size(); ++bi) {
        const Block &bl = e->blocks[bi];
        float *out = xr == e->buf_x[0] ? e->buf_x[1] : e->buf_x[0];
        int rc = run_block(e, bl, xr, out, n, H, W, false, 0);
        if (rc) return rc;
        xr = out;
        H = (H + bl.stride - 1) / bl.stride;
        W = (W + bl.stride - 1) / bl.stride;
    }
    const float *const xh = xr;
    const long M = (long)n * H * W;
    if (H * W == 16 && e->head.p3 && e->fc.p3 && !e->no_tail_fusion) {
        // P3 head + Linear with the same fused epilogues (pool in the head's accumulators; tanh + quantiser behind the Linear)
        int rc = launch_p3_tuned(e, 1, p3_args(xh, M, e->head, nullptr, 16, nullptr, 1, e->buf_pool, 1.0f / 16.0f), e->head.wt2, "head + pool");
        if (rc) return rc;
        return launch_p3_tuned(e, 2, p3_args(e->buf_pool, n, e->fc, nullptr, 1, nullptr, 0, d_f32, 0.f, d_u8), e->fc.wt2, "linear + tanh + quantiser");
    }
    if (H * W == 16 && e->head.wt2 && e->fc.wt2 && !e->head.p3 && !e->fc.p3 && !e->no_tail_fusion) {
        // 4 x 4 final map: the head conv pools in its epilogue (a wave's 16 rows are one image) and the Linear finishes with
        // tanh + the u8 quantiser: three launches (k_avgpool, the FC GEMM's own pass, k_tanh_quant) and the head's [M][1280]
        // round trip less; same arithmetic in the same order (bit-identical)
        const std::pair<const void *, long> key(e->head.wt2, tune_bucket(M));
        auto it = e->gemm_cfg.find(key);
        auto launch_head = [&](int nr, int nw) {
            if (nw == 8) launch_gemm_t<false, 8, 1>(nr, e->stream, xh, M, e->head, nullptr, 16, nullptr, 1, e->buf_pool, 1.0f / 16.0f);
            else launch_gemm_t<false, 4, 1>(nr, e->stream, xh, M, e->head, nullptr, 16, nullptr, 1, e->buf_pool, 1.0f / 16.0f);
        };
        if (it == e->gemm_cfg.end()) {
            TuneTimer tt(e);
            int best_nr = 1, best_nw = 4;
            float best_ms = 1e30f;
            const int tiles = e->head.Npad / 16;
            for (int nr = 8; nr >= 1; --nr) {
                if (tiles % nr) continue;
                for (int nw : {8, 4}) {
                    if (nw == 8 && M <= 64) continue;
                    launch_head(nr, nw);
                    PB_HIP(hipEventRecord(e->tune_e0, e->stream));
                    launch_head(nr, nw);
                    launch_head(nr, nw);
                    PB_HIP(hipEventRecord(e->tune_e1, e->stream));
                    PB_HIP(hipEventSynchronize(e->tune_e1));
                    PB_HIP(hipGetLastError());
                    float ms = 0.f;
                    PB_HIP(hipEventElapsedTime(&ms, e->tune_e0, e->tune_e1));
                    if (e->trace_tune >= 2) fprintf(stderr, "  head + pool M%ld: NR%d NW%d %.1f us\n", M, nr, nw, ms * 500.f);
                    if (tune_take(e, ms, best_ms)) {
                        best_ms = ms;
                        best_nr = nr;
                        best_nw = nw;
                    }
                }
            }
            if (e->trace_tune) fprintf(stderr, "head + pool M%ld: best NR%d NW%d %.1f us\n", M, best_nr, best_nw, best_ms * 500.f);
            it = e->gemm_cfg.emplace(key, std::make_pair(best_nw, best_nr)).first;
        }
        launch_head(it->second.second, it->second.first);
        PB_HIP(hipGetLastError());
        launch_gemm_t<false, 4, 2>(1, e->stream, e->buf_pool, n, e->fc, nullptr, 1, nullptr, 0, d_f32, 0.f, d_u8);
        PB_HIP(hipGetLastError());
        return PB_OK;
    }
    int rc = launch_gemm(e, xh, M, e->head, nullptr, 1, nullptr, 1, e->buf_e);
    if (rc) return rc;
    hipLaunchKernelGGL(k_avgpool, dim3((1280 + 255) / 256, n), dim3(256), 0, e->stream, e->buf_e, H * W, 1280,
                       1.0f / (float)(H * W), e->buf_pool);
    PB_HIP(hipGetLastError());
    if ((rc = launch_gemm(e, e->buf_pool, n, e->fc, nullptr, 1, nullptr, 0, e->buf_gate))) return rc;  // pre-tanh [n][D]
    {
        const long tot = (long)n * e->D;
        hipLaunchKernelGGL(k_tanh_quant, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, e->stream, e->buf_gate, tot, d_f32,
                           d_u8);
        PB_HIP(hipGetLastError());
    }
    return PB_OK;
}

void destroy(pb_embedder *e) {
    if (e->dual_stream) (void)hipStreamDestroy(e->dual_stream);
    if (e->dual_e0) (void)hipEventDestroy(e->dual_e0);
    if (e->dual_e1) (void)hipEventDestroy(e->dual_e1);
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->h_range) (void)hipHostFree(e->h_range);
    for (auto &ge : e->g1_exec)
        if (ge) (void)hipGraphExecDestroy(ge);
    if (e->g1_in) (void)hipHostFree(e->g1_in);
    if (e->g1_out_u8) (void)hipHostFree(e->g1_out_u8);
    if (e->g1_out_f32) (void)hipHostFree(e->g1_out_f32);
    for (auto &sl : e->st) {
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.h_desc) (void)hipHostFree(sl.h_desc);
        (void)hipFree(sl.d);
        (void)hipFree(sl.d_desc);
    }
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(e->d_src[i]);
        (void)hipFree(e->d_desc[i]);
        if (e->h_stage_img[i]) (void)hipHostFree(e->h_stage_img[i]);
        if (e->h_desc[i]) (void)hipHostFree(e->h_desc[i]);
        if (e->ev_copied[i]) (void)hipEventDestroy(e->ev_copied[i]);
        if (e->ev_resized[i]) (void)hipEventDestroy(e->ev_resized[i]);
    }
    (void)hipFree(e->d_tmp);
    if (e->tune_e0) (void)hipEventDestroy(e->tune_e0);
    for (int i = 0; i < 2; ++i) {
        if (e->ev_in[i]) (void)hipEventDestroy(e->ev_in[i]);
        if (e->ev_fwd[i]) (void)hipEventDestroy(e->ev_fwd[i]);
        if (e->ev_out[i]) (void)hipEventDestroy(e->ev_out[i]);
    }
    for (int i = 0; i < 2; ++i) {
        if (e->h_in[i]) (void)hipHostFree(e->h_in[i]);
        if (e->h_out_u8[i]) (void)hipHostFree(e->h_out_u8[i]);
        if (e->h_out_f32[i]) (void)hipHostFree(e->h_out_f32[i]);
    }
    if (e->h2d_stream) (void)hipStreamDestroy(e->h2d_stream);
    if (e->d2h_stream) (void)hipStreamDestroy(e->d2h_stream);
    if (e->tune_e1) (void)hipEventDestroy(e->tune_e1);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
}

// efficientnet.rs:20 on the device for a BATCH of host RGB8 images of individual sizes: resize_to_fill(W, H, Triangle) of image
// i into d_dst[i][H][W][3] (image 0.25.x semantics; kernels in pb_embed_kernels.h; the tests compare with a CPU restatement bit
// for bit).  The reference does this per image on a crawler worker (crawler.rs:68-119 -> indexed_image.rs:71 ->
// efficientnet.rs:19-29); here the batch is cut into sub-batches of <= STAGE_BYTES of source pixels (and <= STAGE_TMP_FLOATS of
// vertical-pass scratch) and pipelined over two
// staging slots: the host packs sub-batch j + 1 into pinned memory (split over four threads) while the copy engine moves
// sub-batch j and the resize kernels -- TWO launches per sub-batch, over a descriptor array -- run on the embedder's stream.
// Nothing waits for the GPU except a slot's reuse; the kernels are queued on e->stream, so a forward pass queued after this
// call is ordered behind them.  On an error the caller drains (drain_streams).
constexpr size_t STAGE_BYTES = 48u << 20;
constexpr uint32_t STAGE_MAX_IMAGES = 1024;
constexpr size_t STAGE_TMP_FLOATS = 64u << 20;  // 256 MB of vertical-pass scratch per sub-batch (48 MB of 256 x 256 sources need 96 MB)

void drain_streams(pb_embedder *e) {
    if (e->dual_stream) (void)hipStreamSynchronize(e->dual_stream);
    (void)hipStreamSynchronize(e->h2d_stream);
    (void)hipStreamSynchronize(e->stream);
    (void)hipStreamSynchronize(e->d2h_stream);
}

// copy into a staging block with streaming (non-temporal) stores: the block is read next by the copy engine, not by this
// core, and a cached store would first fetch every destination line (one more pass over host memory, which is what bounds
// the packing: profiles/ingest_probe.py).  dst is 16-byte aligned (staging offsets are multiples of 16).
void stream_copy(uint8_t *dst, const uint8_t *src, size_t n) {
    typedef long long v2di __attribute__((vector_size(16)));
    size_t i = 0;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        for (; i + 64 <= n; i += 64) {
            v2di a, b, c, d;
            memcpy(&a, src + i, 16); memcpy(&b, src + i + 16, 16); memcpy(&c, src + i + 32, 16); memcpy(&d, src + i + 48, 16);
            __builtin_nontemporal_store(a, reinterpret_cast<v2di *>(dst + i));
            __builtin_nontemporal_store(b, reinterpret_cast<v2di *>(dst + i + 16));
            __builtin_nontemporal_store(c, reinterpret_cast<v2di *>(dst + i + 32));
            __builtin_nontemporal_store(d, reinterpret_cast<v2di *>(dst + i + 48));
        }
        __builtin_ia32_sfence();
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
}

// copies of many buffers into one block, split over up to four threads by bytes
void pack_parallel(uint8_t *dst, const uint8_t *const *src, const size_t *bytes, const size_t *off, uint32_t n) {
    size_t total = 0;
    for (uint32_t i = 0; i < n; ++i) total += bytes[i];
    constexpr int NTH = 4;
    if (total < (4u << 20) || n < 2) {
        for (uint32_t i = 0; i < n; ++i) stream_copy(dst + off[i], src[i], bytes[i]);
        return;
    }
    // thread t takes images [cut[t], cut[t + 1]): contiguous runs of about total / NTH bytes
    uint32_t cut[NTH + 1];
    cut[0] = 0;
    size_t acc = 0;
    int t = 1;
    for (uint32_t i = 0; i < n && t < NTH; ++i) {
        acc += bytes[i];
        if (acc >= total * t / NTH) cut[t++] = i + 1;
    }
    for (; t <= NTH; ++t) cut[t] = n;
    auto run = [&](uint32_t i0, uint32_t i1) {
        for (uint32_t i = i0; i < i1; ++i) stream_copy(dst + off[i], src[i], bytes[i]);
    };
    std::thread th[NTH - 1];
    int started = 0;
    try {
        for (int k = 1; k < NTH; ++k) {
            if (cut[k] >= cut[k + 1]) continue;
            th[started] = std::thread(run, cut[k], cut[k + 1]);
            ++started;
        }
    } catch (...) {  // no thread to be had: this thread copies what the missing ones would have
        for (int k = 0; k < started; ++k) th[k].join();
        run(0, n);
        return;
    }
    run(cut[0], cut[1]);
    for (int k = 0; k < started; ++k) th[k].join();
}

// the two passes of resize_to_fill for m staged images: one fused launch when every image's row of vertical sums fits the LDS asked
// for (k_resize_fused), else the vertical pass into the f32 scratch and the horizontal pass from it (k_resize_v, k_resize_h) -- same bytes
size_t resize_fused_lds(const pb_embedder *e, const ResizeDesc *h_desc, uint32_t m) {  // 0: the two-kernel path
    size_t span = 0;  // the widest image that is resampled bounds every row's column span
    for (uint32_t i = 0; i < m; ++i)
        if (h_desc[i].resample) span = std::max<size_t>(span, h_desc[i].w);
    const size_t lds = std::max<size_t>(span * 3 * sizeof(float), 16);
    return (!e->no_resize_fusion && lds <= 96 * 1024) ? lds : 0;
}
int launch_resize(pb_embedder *e, const uint8_t *d_src, const ResizeDesc *d_desc, const ResizeDesc *h_desc, uint32_t m, uint32_t max_w, size_t tmp_total,
                  uint8_t *d_dst) {
    const uint32_t W = e->W, H = e->H;
    const size_t lds = resize_fused_lds(e, h_desc, m);
    if (lds) {
        if (lds > 48 * 1024 && !e->resize_attr_set) {
            PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resize_fused), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            e->resize_attr_set = true;
        }
        hipLaunchKernelGGL(k_resize_fused, dim3(H, m), dim3(256), lds, e->stream, d_src, d_desc, W, H, d_dst);
        PB_HIP(hipGetLastError());
        return PB_OK;
    }
    if (tmp_total) {
        hipLaunchKernelGGL(k_resize_v, dim3((max_w + 255) / 256, H, m), dim3(256), 0, e->stream, d_src, d_desc, e->d_tmp);
        PB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_resize_h, dim3((W * H + 255) / 256, m), dim3(256), 0, e->stream, d_src, e->d_tmp, d_desc, W, H, d_dst);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// geometry of `resize_to_fill(W, H)` for a w x h source: src/math/utils.rs resize_dimensions(.., fill = true) + src/dynimage.rs
// resize_to_fill's centre crop (efficientnet.rs:20)
int resize_geometry(uint32_t W, uint32_t H, uint32_t w, uint32_t h, ResizeDesc *out) {
    const double wratio = (double)W / (double)w, hratio = (double)H / (double)h;
    const double ratio = wratio > hratio ? wratio : hratio;
    const double a = std::round((double)w * ratio), b2 = std::round((double)h * ratio);
    ResizeDesc d{};
    d.w = w; d.h = h;
    d.w2 = a < 1.0 ? 1u : (uint32_t)a;
    d.h2 = b2 < 1.0 ? 1u : (uint32_t)b2;
    PB_CHECK(d.w2 >= W && d.h2 >= H, PB_ERR_INVALID, "resize_to_fill: %ux%u does not cover %ux%u", d.w2, d.h2, W, H);
    d.cx = d.cy = 0;
    if ((uint64_t)W * d.h2 > (uint64_t)d.w2 * H) d.cy = (d.h2 - H) / 2;  // centre crop along the dimension that overshoots
    else d.cx = (d.w2 - W) / 2;
    d.resample = (d.w2 == w && d.h2 == h) ? 0u : 1u;  // imageops::resize: same dimensions -> copy
    *out = d;
    return PB_OK;
}

int prepare_images(pb_embedder *e, const uint8_t *const *rgb, const uint32_t *widths, const uint32_t *heights, uint32_t n, uint8_t *d_dst) {
    const uint32_t W = e->W, H = e->H;
    for (int b = 0; b < 2; ++b) {
        if (!e->ev_copied[b]) PB_HIP(hipEventCreateWithFlags(&e->ev_copied[b], hipEventDisableTiming));
        if (!e->ev_resized[b]) PB_HIP(hipEventCreateWithFlags(&e->ev_resized[b], hipEventDisableTiming));
        if (!e->h_desc[b]) PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_desc[b]), STAGE_MAX_IMAGES * sizeof(ResizeDesc), hipHostMallocDefault));
        if (!e->d_desc[b]) PB_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_desc[b]), STAGE_MAX_IMAGES * sizeof(ResizeDesc)));
    }
    std::vector<size_t> bytes(STAGE_MAX_IMAGES), off(STAGE_MAX_IMAGES);
    int slot = 0;
    for (uint32_t i0 = 0; i0 < n; slot ^= 1) {
        // ---- the sub-batch [i0, i1): descriptors, offsets, scratch need
        ResizeDesc *hd = e->h_desc[slot];
        if (e->stage_used[slot]) PB_HIP(hipEventSynchronize(e->ev_copied[slot]));  // its pinned blocks have been read
        size_t src_total = 0, tmp_total = 0;
        uint32_t i1 = i0, max_w = 1;
        for (; i1 < n && i1 - i0 < STAGE_MAX_IMAGES; ++i1) {
            const uint32_t w = widths[i1], h = heights[i1];
            PB_CHECK(rgb[i1], PB_ERR_INVALID, "image %u: null pointer", i1);
            PB_CHECK(w >= 1 && h >= 1 && w <= 65535 && h <= 65535, PB_ERR_INVALID, "image %u: size %ux%u outside 1..65535", i1, w, h);
            const size_t sb = (size_t)w * h * 3;
            if (i1 > i0 && src_total + sb > STAGE_BYTES) break;
            // the vertical pass's scratch is H * w * 3 floats per image -- 4 H / h times the source bytes: a sub-batch of wide, short
            // images is cut by ITS size too (a single image over the cap still gets a sub-batch of its own, as with the source bytes)
            if (i1 > i0 && tmp_total + (size_t)H * w * 3 > STAGE_TMP_FLOATS) break;
            ResizeDesc d;
            { int rcd = resize_geometry(W, H, w, h, &d); if (rcd) return rcd; }
            d.slot = i1;
            d.src_off = src_total;
            d.tmp_off = tmp_total;
            bytes[i1 - i0] = sb;
            off[i1 - i0] = src_total;
            src_total += (sb + 15) & ~(size_t)15;
            if (d.resample) tmp_total += ((size_t)H * w * 3 + 3) & ~(size_t)3;
            max_w = std::max(max_w, w);
            hd[i1 - i0] = d;
        }
        const uint32_t m = i1 - i0;
        // ---- staging blocks (grow-only; a block in use by the GPU is not freed: drain first)
        if (src_total > e->stage_cap[slot]) {
            drain_streams(e);
            if (e->h_stage_img[slot]) (void)hipHostFree(e->h_stage_img[slot]);
            (void)hipFree(e->d_src[slot]);
            e->h_stage_img[slot] = nullptr;
            e->d_src[slot] = nullptr;
            e->stage_cap[slot] = 0;
            const size_t cap = std::max(src_total, STAGE_BYTES);
            PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_stage_img[slot]), cap, hipHostMallocDefault));
            PB_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_src[slot]), cap));
            e->stage_cap[slot] = cap;
        }
        if (tmp_total > e->d_tmp_cap && !resize_fused_lds(e, hd, m)) {  // (the fused resize needs no scratch image)
            drain_streams(e);
            (void)hipFree(e->d_tmp);
            e->d_tmp = nullptr;
            e->d_tmp_cap = 0;
            PB_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_tmp), tmp_total * sizeof(float)));
            e->d_tmp_cap = tmp_total;
        }
        // ---- pack (host), copy (copy engine), resize (embedder's stream).  (Transferring images that already sit in pinned
        // memory one by one instead was measured and is slower -- 112 k against 182 k images/s at 256 x 256, 45 k against 49 k at
        // 640 x 480: a copy command per image costs more than the packing pass saves.)
        {
            const auto tp0 = std::chrono::steady_clock::now();
            pack_parallel(e->h_stage_img[slot], rgb + i0, bytes.data(), off.data(), m);
            if (e->trace_tune >= 3)
                fprintf(stderr, "prepare_images: sub-batch of %u images, %.1f MB packed in %.3f ms\n", m, src_total / 1e6,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
        }
        if (n == 1) {
            // one image (pb_mlhash_image): nothing to overlap, so the copies go on the embedder's stream itself -- no event hand-over
            // between two streams in front of the resize kernel (the events are still recorded: a later batch call waits on them)
            PB_HIP(hipMemcpyAsync(e->d_src[slot], e->h_stage_img[slot], src_total, hipMemcpyHostToDevice, e->stream));
            PB_HIP(hipMemcpyAsync(e->d_desc[slot], hd, m * sizeof(ResizeDesc), hipMemcpyHostToDevice, e->stream));
            PB_HIP(hipEventRecord(e->ev_copied[slot], e->stream));
            e->stage_used[slot] = true;
        } else {
            if (e->stage_used[slot]) PB_HIP(hipStreamWaitEvent(e->h2d_stream, e->ev_resized[slot], 0));  // the kernels that read d_src[slot] last
            PB_HIP(hipMemcpyAsync(e->d_src[slot], e->h_stage_img[slot], src_total, hipMemcpyHostToDevice, e->h2d_stream));
            PB_HIP(hipMemcpyAsync(e->d_desc[slot], hd, m * sizeof(ResizeDesc), hipMemcpyHostToDevice, e->h2d_stream));
            PB_HIP(hipEventRecord(e->ev_copied[slot], e->h2d_stream));
            e->stage_used[slot] = true;
            PB_HIP(hipStreamWaitEvent(e->stream, e->ev_copied[slot], 0));
        }
        { int rcr = launch_resize(e, e->d_src[slot], e->d_desc[slot], hd, m, max_w, tmp_total, d_dst); if (rcr) return rcr; }
        PB_HIP(hipEventRecord(e->ev_resized[slot], e->stream));
        i0 = i1;
    }
    return PB_OK;
}

}  // namespace

extern "C" {

int pb_embed_create(pb_embedder **out, int device, const void *weights_blob, size_t blob_len, uint32_t max_batch) {
    PB_CHECK(out, PB_ERR_INVALID, "pb_embed_create: null out pointer");
    *out = nullptr;
    PB_CHECK(weights_blob, PB_ERR_INVALID, "pb_embed_create: null weight blob");
    PB_CHECK(max_batch >= 1 && max_batch <= 8192, PB_ERR_INVALID, "pb_embed_create: max_batch %u outside 1..8192", max_batch);
    int n_dev = 0;
    PB_HIP(hipGetDeviceCount(&n_dev));
    PB_CHECK(device >= 0 && device < n_dev, PB_ERR_INVALID, "pb_embed_create: device %d of %d", device, n_dev);
    pb::DeviceGuard guard(device);
    PB_CHECK(guard.ok, PB_ERR_HIP, "hipSetDevice(%d) failed", device);
    pb_embedder *e = new (std::nothrow) pb_embedder();
    PB_CHECK(e, PB_ERR_NOMEM, "out of host memory");
    e->device = device;
    e->max_batch = max_batch;
    if (const char *pk = getenv("PB_P3_MIN_K")) e->p3_min_k = atoi(pk);  // A/B runs; the default is part of the arithmetic's definition
    if (const char *pk = getenv("PB_P3E_MIN_K")) e->p3e_min_k = atoi(pk);
    if (getenv("PB_NO_P3E")) e->p3e_min_k = 1 << 30;
    if (getenv("PB_NO_P3")) e->p3_min_k = e->p3e_min_k = 1 << 30;
    if (const char *tp = getenv("PB_TUNE_PICK")) e->tune_pick = atoi(tp);
    if (const char *tt = getenv("PB_TRACE_TUNE")) e->trace_tune = tt[0] == '3' ? 3 : (tt[0] == '2' ? 2 : 1);  // 3: + host-side staging times
    e->no_stem_fusion = getenv("PB_NO_STEM_FUSION") != nullptr;
    e->g1_off = getenv("PB_NO_GRAPH") != nullptr;
    e->no_resize_fusion = getenv("PB_NO_RESIZE_FUSION") != nullptr;
    if (const char *v = getenv("PB_STEM_RPP")) e->stem_rpp = atoi(v) == 1 ? 1 : 2;
    e->fold_se = getenv("PB_FOLD_SE") != nullptr;
    if (const char *v = getenv("PB_SE_MULTI_MAX")) e->se_multi_max = std::min(8, std::max(0, atoi(v)));
    if (const char *v = getenv("PB_DUAL")) e->dual_min = atoi(v);  // images from which a forward runs as two concurrent halves (0: never)
    e->no_gemm_t = getenv("PB_NO_GEMM_T") != nullptr;
    e->no_tail_fusion = getenv("PB_NO_TAIL_FUSION") != nullptr;
    e->no_block_fusion = getenv("PB_NO_BLOCK_FUSION") != nullptr;
    e->no_gemm_stream = getenv("PB_NO_GEMM_STREAM") != nullptr;
    e->no_band = getenv("PB_NO_BAND") != nullptr;
    e->no_band_ipw = getenv("PB_NO_BAND_IPW") != nullptr;
    e->force_band = getenv("PB_FORCE_BAND") != nullptr;
    auto body = [&]() -> int {
        hipDeviceProp_t prop;
        PB_HIP(hipGetDeviceProperties(&prop, device));
        e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        PB_HIP(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
        e->stream = e->own_stream;
        PB_HIP(hipEventCreate(&e->tune_e0));
        PB_HIP(hipStreamCreateWithFlags(&e->h2d_stream, hipStreamNonBlocking));
        PB_HIP(hipStreamCreateWithFlags(&e->d2h_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            PB_HIP(hipEventCreateWithFlags(&e->ev_in[i], hipEventDisableTiming));
            PB_HIP(hipEventCreateWithFlags(&e->ev_fwd[i], hipEventDisableTiming));
            PB_HIP(hipEventCreateWithFlags(&e->ev_out[i], hipEventDisableTiming));
        }
        PB_HIP(hipEventCreate(&e->tune_e1));
        int rc = load_weights(e, static_cast<const uint8_t *>(weights_blob), blob_len);
        if (rc) return rc;
        // workspace sized for max_batch images (NHWC f32)
        const size_t B = max_batch, h2 = e->H / 2, w2 = e->W / 2;
        size_t max_x = h2 * w2 * 32, max_e = 0, max_dw = 0, max_part = 0;
        size_t h = h2, w = w2;
        for (const Block &bl : e->blocks) {
            const size_t ho = (h + bl.stride - 1) / bl.stride, wo = (w + bl.stride - 1) / bl.stride;
            if (bl.has_expand) max_e = std::max(max_e, h * w * bl.e);
            max_dw = std::max(max_dw, ho * wo * bl.e);
            max_x = std::max(max_x, ho * wo * (size_t)bl.cout);
            // SE partials: the depthwise kernels use at most 32 tiles / bands, the fused front one per (strip, band)
            max_part = std::max(max_part, std::max<size_t>(32, (size_t)front_strips(bl, (int)wo) * 8) * bl.e);
            h = ho;
            w = wo;
        }
        max_e = std::max(max_e, h * w * 1280);
        {   // front sub-batches: the leading blocks up to the last whose depthwise output is >= 128 KB per image (blocks 0-4 at
            // 128 x 128), in sub-batches whose largest block working set (input + depthwise output + output map) fits ~100 MB
            size_t hh = h2, ww = w2, fb = 0, per = 0, ws = 0;
            std::vector<size_t> out_per, ws_per;
            for (const Block &bl : e->blocks) {
                const size_t ho = (hh + bl.stride - 1) / bl.stride, wo = (ww + bl.stride - 1) / bl.stride;
                out_per.push_back(ho * wo * (size_t)bl.cout);
                ws_per.push_back((hh * ww * (size_t)bl.cin + ho * wo * (size_t)bl.e + ho * wo * (size_t)bl.cout) * sizeof(float));
                if (ho * wo * (size_t)bl.e * sizeof(float) >= 128 * 1024) fb = out_per.size();
                hh = ho;
                ww = wo;
            }
            if (const char *v = getenv("PB_FRONT_BLOCKS")) fb = std::min<size_t>((size_t)std::max(0, atoi(v)), e->blocks.size());
            for (size_t i = 0; i < fb; ++i) ws = std::max(ws, ws_per[i]);
            per = fb ? out_per[fb - 1] : 0;
            // measured (profiles/r06_front_sub.txt): at 128 x 128, batch 512, every split is SLOWER than one pass (sub 256: +0.03 ms, 128: +0.16,
            // 64: +0.54 over 1.98) -- the front kernels are not paced by where their maps come from, and smaller grids pay their fill and
            // drain more often.  So the default is off; the option stays for other shapes and for A/B runs.
            (void)ws;
            int sub = 0;
            if (const char *v = getenv("PB_FRONT_SUB")) sub = std::max(0, atoi(v));
            e->front_blocks = fb;
            e->front_out_per_image = per;
            e->front_sub = sub;
            if (fb && (rc = dalloc(e, &e->buf_front, B * per))) return rc;
        }
        e->ws_max_x = max_x; e->ws_max_e = max_e; e->ws_max_dw = max_dw;
        if ((rc = dalloc(e, &e->d_img, B * e->H * e->W * 3))) return rc;
        if ((rc = dalloc(e, &e->buf_x[0], B * max_x)) || (rc = dalloc(e, &e->buf_x[1], B * max_x))) return rc;
        if ((rc = dalloc(e, &e->buf_e, B * max_e)) || (rc = dalloc(e, &e->buf_dw, B * max_dw))) return rc;
        e->part_floats_per_image = max_part;
        if ((rc = dalloc(e, &e->buf_part, B * max_part + 2)) || (rc = dalloc(e, &e->buf_gate, B * std::max<size_t>(1152, e->D)))) return rc;
        {  // 16-byte header in front of the partial sums: the address of the range word, for the kernels that write them
            PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_range), sizeof(unsigned), hipHostMallocMapped));
            *e->h_range = 0u;
            void *d_flag = nullptr;
            PB_HIP(hipHostGetDevicePointer(&d_flag, e->h_range, 0));
            const long long hdr[2] = {0, (long long)reinterpret_cast<uintptr_t>(d_flag)};
            PB_HIP(hipMemcpy(e->buf_part, hdr, sizeof(hdr), hipMemcpyHostToDevice));
            e->buf_part += 2;
        }
        if ((rc = dalloc(e, &e->buf_pool, B * 1280))) return rc;
        if ((rc = dalloc(e, &e->d_se_cnt, B))) return rc;
        PB_HIP(hipMemset(e->d_se_cnt, 0, B * sizeof(unsigned)));
        if ((rc = dalloc(e, &e->d_se_xchg, (size_t)8 * 64)) || (rc = dalloc(e, &e->d_se_arrive, (size_t)8))) return rc;
        PB_HIP(hipMemset(e->d_se_xchg, 0, (size_t)8 * 64 * sizeof(unsigned long long)));
        PB_HIP(hipMemset(e->d_se_arrive, 0, 8 * sizeof(unsigned)));
        if ((rc = dalloc(e, &e->d_out_f32, B * e->D)) || (rc = dalloc(e, &e->d_out_u8, B * e->D))) return rc;
        for (int i = 0; i < 2; ++i) {
            PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_out_u8[i]), B * e->D, hipHostMallocDefault));
            PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_out_f32[i]), B * e->D * sizeof(float), hipHostMallocDefault));
        }
        if ((rc = dalloc(e, &e->d_img_b, B * e->H * e->W * 3)) || (rc = dalloc(e, &e->d_out_f32_b, B * e->D)) || (rc = dalloc(e, &e->d_out_u8_b, B * e->D))) return rc;
        return PB_OK;
    };
    int rc = body();
    if (rc) {
        destroy(e);
        delete e;
        return rc;
    }
    *out = e;
    return PB_OK;
}

int pb_embed_destroy(pb_embedder *e) {
    if (!e) return PB_OK;
    {
        pb::DeviceGuard guard(e->device);
        (void)hipStreamSynchronize(e->stream);
        destroy(e);
    }
    delete e;
    return PB_OK;
}

int pb_embed_info(const pb_embedder *e, uint32_t *h, uint32_t *w, uint32_t *d, uint32_t *max_batch) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_info: null embedder");
    if (h) *h = e->H;
    if (w) *w = e->W;
    if (d) *d = e->D;
    if (max_batch) *max_batch = e->max_batch;
    return PB_OK;
}

// After a stream wait: did a kernel of the forward passes behind it see a depthwise output outside the domain of the fixed-point
// squeeze-excite sums (se_range_check, pb_embed_common.h)?  Such a batch's hashes are wrong; the call fails instead of returning them.
static int range_status(pb_embedder *e) {
    if (e->h_range && __atomic_load_n(e->h_range, __ATOMIC_ACQUIRE)) {
        __atomic_store_n(e->h_range, 0u, __ATOMIC_RELEASE);
        PB_CHECK(false, PB_ERR_RANGE,
                 "embed: a depthwise activation of 128 or more left the domain of the 2^-24 fixed-point squeeze-excite sums; the batch's outputs are not valid "
                 "(with PB_OPT_EMBED_ASYNC the batch may be an earlier one on this embedder)");
    }
    return PB_OK;
}

int pb_embed_batch_device(pb_embedder *e, const uint8_t *d_rgb, uint32_t n, uint8_t *d_out_u8, float *d_out_f32) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_batch_device: null embedder");
    PB_CHECK(n <= e->max_batch, PB_ERR_INVALID, "pb_embed_batch_device: n = %u > max_batch %u", n, e->max_batch);
    PB_CHECK(n == 0 || (d_rgb && d_out_u8), PB_ERR_INVALID, "pb_embed_batch_device: null buffer");
    if (n == 0) return PB_OK;
    std::lock_guard<std::mutex> lock(e->mu);
    pb::DeviceGuard guard(e->device);
    int rc = forward_device(e, d_rgb, (int)n, d_out_u8, d_out_f32);
    if (rc) return rc;
    // Synchronous by default: on return the outputs are complete, whatever stream the consumer uses (the index
    // owns a stream of its own).  PB_OPT_EMBED_ASYNC = 1 returns with the forward pass queued on the embedder's
    // stream instead: the consumer must then run on that same stream (PB_OPT_EMBED_STREAM + PB_OPT_STREAM) or
    // wait for it.
    if (!e->opt_async) {
        PB_HIP(hipStreamSynchronize(e->stream));
        return range_status(e);
    }
    // Queued: the range flag belongs to work whose wait is the caller's, so this call neither reads nor clears it (until round 6 it
    // returned -- and consumed -- the flag of EARLIER batches: an error for a call whose own work was still in flight, and the batch
    // that raised it reported to whichever call came next).  The caller asks pb_embed_check_range after its stream wait.
    return PB_OK;
}

int pb_embed_check_range(pb_embedder *e) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_check_range: null embedder");
    std::lock_guard<std::mutex> lock(e->mu);
    return range_status(e);
}

// host copy split over a few threads (one thread moves ~10 GB/s: 2.5 ms for a chunk of 512 images, longer than its forward)
static void parallel_copy(uint8_t *dst, const uint8_t *src, size_t bytes) {
    constexpr int NTH = 4;
    if (bytes < (4u << 20)) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t part = (bytes / NTH + 4095) & ~(size_t)4095;
    std::thread th[NTH - 1];
    int started = 0;
    try {
        for (int i = 1; i < NTH; ++i) {
            const size_t off = (size_t)i * part;
            if (off >= bytes) break;
            th[started] = std::thread([=] { memcpy(dst + off, src + off, std::min(part, bytes - off)); });
            ++started;
        }
    } catch (...) {  // no thread to be had: this thread copies the rest itself
        for (int i = 0; i < started; ++i) th[i].join();
        const size_t done = (size_t)(started + 1) * part;
        memcpy(dst, src, std::min(part, bytes));
        if (done < bytes) memcpy(dst + done, src + done, bytes - done);
        return;
    }
    memcpy(dst, src, std::min(part, bytes));
    for (int i = 0; i < started; ++i) th[i].join();
}

static void g1_drop(pb_embedder *e) {
    for (auto &ge : e->g1_exec) {
        if (ge) (void)hipGraphExecDestroy(ge);
        ge = nullptr;
    }
    e->g1_quiet_calls = 0;
}

// One image through the replayed graph.  rgb = null: the image is in d_img already (queued on the embedder's stream by the resize).
// *used = 0: no graph (not yet / not possible), the caller takes the plain path.
static int one_image_graph(pb_embedder *e, const uint8_t *rgb, uint8_t *out_u8, float *out_f32, int *used) {
    *used = 0;
    const size_t img_bytes = (size_t)e->H * e->W * 3;
    const int kind = rgb ? 0 : 1;
    if ((e->g1_exec[0] || e->g1_exec[1]) && e->g1_stream != e->stream) g1_drop(e);
    if (!e->g1_exec[kind]) {
        if (e->g1_quiet_calls < 2 || e->fold_se) return PB_OK;  // the kernel forms of this bucket are picked and warm by then
        if (!e->g1_in) {
            PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->g1_in), img_bytes, hipHostMallocDefault));
            PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->g1_out_u8), e->D, hipHostMallocDefault));
            PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->g1_out_f32), e->D * sizeof(float), hipHostMallocDefault));
        }
        // capture: the same calls the plain path makes, on the same stream.  Anything that cannot be captured (a timing loop that
        // was not expected, a runtime that refuses a call) ends the capture, switches the graph off for this embedder and leaves
        // the call to the plain path -- never an error of its own.
        const unsigned long tuned_before = e->tune_runs;
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
            (void)hipGetLastError();
            e->g1_off = true;
            return PB_OK;
        }
        bool ok = kind == 1 || hipMemcpyAsync(e->d_img, e->g1_in, img_bytes, hipMemcpyHostToDevice, e->stream) == hipSuccess;
        ok = ok && forward_device(e, e->d_img, 1, e->d_out_u8, e->d_out_f32) == PB_OK;
        ok = ok && hipMemcpyAsync(e->g1_out_u8, e->d_out_u8, e->D, hipMemcpyDeviceToHost, e->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(e->g1_out_f32, e->d_out_f32, e->D * sizeof(float), hipMemcpyDeviceToHost, e->stream) == hipSuccess;
        const hipError_t ec = hipStreamEndCapture(e->stream, &g);
        ok = ok && ec == hipSuccess && g != nullptr && e->tune_runs == tuned_before;
        if (ok) ok = hipGraphInstantiate(&e->g1_exec[kind], g, nullptr, nullptr, 0) == hipSuccess;
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        if (!ok) {
            e->g1_exec[kind] = nullptr;
            e->g1_off = true;
            (void)hipStreamSynchronize(e->stream);
            return PB_OK;
        }
        e->g1_stream = e->stream;
    }
    *used = 1;
    if (kind == 0) memcpy(e->g1_in, rgb, img_bytes);
    auto body = [&]() -> int {
        PB_HIP(hipGraphLaunch(e->g1_exec[kind], e->stream));
        PB_HIP(hipStreamSynchronize(e->stream));
        return range_status(e);
    };
    const int rc = body();
    if (rc) {
        (void)hipStreamSynchronize(e->stream);
        return rc;
    }
    memcpy(out_u8, e->g1_out_u8, e->D);
    if (out_f32) memcpy(out_f32, e->g1_out_f32, e->D * sizeof(float));
    return PB_OK;
}

int pb_embed_batch(pb_embedder *e, const uint8_t *rgb, uint32_t n, uint8_t *out_u8, float *out_f32) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_batch: null embedder");
    PB_CHECK(n == 0 || (rgb && out_u8), PB_ERR_INVALID, "pb_embed_batch: null buffer");
    if (n == 0) return PB_OK;
    std::lock_guard<std::mutex> lock(e->mu);
    pb::DeviceGuard guard(e->device);
    const size_t img_bytes = (size_t)e->H * e->W * 3;
    // A call of more than max_batch images goes through a two-slot pipeline: the input copy of chunk i + 1 (its own stream) and
    // the output copy of chunk i - 1 (a third stream) run beside the forward pass of chunk i.  (Cutting a single chunk in two
    // to hide half of its copies was measured and is slower: 148 k against 168 k images/s at 512 -- the half-batches' forwards
    // lose more than the hidden copies save.)
    if (n == 1 && !e->g1_off) {
        int used = 0;
        const int rc = one_image_graph(e, rgb, out_u8, out_f32, &used);
        if (used) return rc;  // else: not captured (yet): the plain path below
    }
    if (n <= e->max_batch) {  // one chunk: nothing to overlap, one stream, no events
        const unsigned long tuned_before = e->tune_runs;
        auto body = [&]() -> int {
            PB_HIP(hipMemcpyAsync(e->d_img, rgb, n * img_bytes, hipMemcpyHostToDevice, e->stream));
            int rc = forward_device(e, e->d_img, (int)n, e->d_out_u8, e->d_out_f32);
            if (rc) return rc;
            PB_HIP(hipMemcpyAsync(out_u8, e->d_out_u8, (size_t)n * e->D, hipMemcpyDeviceToHost, e->stream));
            if (out_f32) PB_HIP(hipMemcpyAsync(out_f32, e->d_out_f32, (size_t)n * e->D * sizeof(float), hipMemcpyDeviceToHost, e->stream));
            PB_HIP(hipStreamSynchronize(e->stream));
            return range_status(e);
        };
        const int rc = body();
        if (rc) (void)hipStreamSynchronize(e->stream);  // a copy may still be reading `rgb` / writing the outputs: not after the call has returned
        if (n == 1) e->g1_quiet_calls = (!rc && e->tune_runs == tuned_before) ? e->g1_quiet_calls + 1 : 0;
        return rc;
    }
    const uint32_t chunk = e->max_batch;
    uint8_t *d_in[2] = {e->d_img, e->d_img_b}, *d_u8[2] = {e->d_out_u8, e->d_out_u8_b};
    float *d_f[2] = {e->d_out_f32, e->d_out_f32_b};
    uint32_t first[2] = {0, 0}, count[2] = {0, 0};  // the chunk whose outputs a slot's pinned buffers hold (or will hold)
    // A copy from pageable memory blocks the calling thread and, beside a running forward, moves ~9 GB/s: the pipeline then
    // runs at the copy's pace.  Pageable input is therefore staged through two pinned buffers by a host copy split over four
    // threads (while the previous chunk's forward runs) and transferred from there; pinned or registered caller memory is
    // transferred directly.
    bool stage = true;
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, rgb) == hipSuccess) stage = attr.type == hipMemoryTypeUnregistered;
        else (void)hipGetLastError();  // unknown to the runtime: plain host memory
    }
    if (stage && !e->h_in[0]) {
        for (int i = 0; i < 2; ++i) PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_in[i]), (size_t)e->max_batch * img_bytes, hipHostMallocDefault));
    }
    // outputs land in pinned memory (a plain DMA transfer that the calling thread does not wait for) and are handed to the
    // caller's arrays when their slot is reused, or at the end
    auto hand_over = [&](int slot) -> int {
        if (!count[slot]) return PB_OK;
        PB_HIP(hipEventSynchronize(e->ev_out[slot]));
        memcpy(out_u8 + (size_t)first[slot] * e->D, e->h_out_u8[slot], (size_t)count[slot] * e->D);
        if (out_f32) memcpy(out_f32 + (size_t)first[slot] * e->D, e->h_out_f32[slot], (size_t)count[slot] * e->D * sizeof(float));
        count[slot] = 0;
        return PB_OK;
    };
    // one chunk's share of the pipeline; a failing call leaves through the drain below (a transfer may still be reading the
    // caller's buffer: the call must not return while it does)
    auto run_chunk = [&](int slot, uint32_t i0, uint32_t c) -> int {
        int rc = hand_over(slot);  // the slot's previous outputs have arrived (its input was consumed before them)
        if (rc) return rc;
        const uint8_t *src = rgb + i0 * img_bytes;
        if (stage) {  // pageable caller memory: into the slot's pinned buffer first
            parallel_copy(e->h_in[slot], src, c * img_bytes);
            src = e->h_in[slot];
        }
        PB_HIP(hipMemcpyAsync(d_in[slot], src, c * img_bytes, hipMemcpyHostToDevice, e->h2d_stream));
        PB_HIP(hipEventRecord(e->ev_in[slot], e->h2d_stream));
        PB_HIP(hipStreamWaitEvent(e->stream, e->ev_in[slot], 0));
        if ((rc = forward_device(e, d_in[slot], (int)c, d_u8[slot], d_f[slot]))) return rc;
        PB_HIP(hipEventRecord(e->ev_fwd[slot], e->stream));
        PB_HIP(hipStreamWaitEvent(e->d2h_stream, e->ev_fwd[slot], 0));
        PB_HIP(hipMemcpyAsync(e->h_out_u8[slot], d_u8[slot], (size_t)c * e->D, hipMemcpyDeviceToHost, e->d2h_stream));
        if (out_f32)
            PB_HIP(hipMemcpyAsync(e->h_out_f32[slot], d_f[slot], (size_t)c * e->D * sizeof(float), hipMemcpyDeviceToHost, e->d2h_stream));
        PB_HIP(hipEventRecord(e->ev_out[slot], e->d2h_stream));
        first[slot] = i0;
        count[slot] = c;
        return PB_OK;
    };
    int slot = 0, rc = PB_OK;
    for (uint32_t i0 = 0; i0 < n && !rc; i0 += chunk, slot ^= 1) rc = run_chunk(slot, i0, std::min(chunk, n - i0));
    if (!rc) rc = hand_over(slot);
    if (!rc) rc = hand_over(slot ^ 1);
    if (!rc) rc = range_status(e);
    if (rc) {  // drain: nothing of this call may still be in flight when it returns
        (void)hipStreamSynchronize(e->h2d_stream);
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamSynchronize(e->d2h_stream);
    }
    return rc;
}

int pb_mlhash(pb_embedder *e, const uint8_t *rgb, uint8_t *out, size_t out_len) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_mlhash: null embedder");
    PB_CHECK(out_len >= e->D, PB_ERR_INVALID, "pb_mlhash: out_len %zu < D = %u", out_len, e->D);
    return pb_embed_batch(e, rgb, 1, out, nullptr);
}

int pb_embed_batch_images(pb_embedder *e, const uint8_t *const *rgb, const uint32_t *widths, const uint32_t *heights, uint32_t n,
                          uint8_t *out_u8, float *out_f32) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_batch_images: null embedder");
    PB_CHECK(n == 0 || (rgb && widths && heights && out_u8), PB_ERR_INVALID, "pb_embed_batch_images: null buffer");
    std::lock_guard<std::mutex> lock(e->mu);
    pb::DeviceGuard guard(e->device);
    // chunks of max_batch images through two input slots: the staging + resize of chunk c + 1 is queued while the forward pass
    // of chunk c runs (the host's packing and the copy engine's transfers hide under it); outputs land in pinned memory on a
    // third stream and are handed over when their slot comes round again, as in pb_embed_batch
    if (n == 1 && !e->g1_off && (e->g1_exec[1] || e->g1_quiet_calls >= 2)) {  // pb_mlhash_image: resize, then the forward as one replayed graph
        int used = 0;
        int rc1 = prepare_images(e, rgb, widths, heights, 1, e->d_img);
        if (!rc1) rc1 = one_image_graph(e, nullptr, out_u8, out_f32, &used);
        if (rc1 || used) {
            if (rc1) drain_streams(e);
            return rc1;
        }
    }
    const unsigned long tuned_before = e->tune_runs;
    uint8_t *d_in[2] = {e->d_img, e->d_img_b}, *d_u8[2] = {e->d_out_u8, e->d_out_u8_b};
    float *d_f[2] = {e->d_out_f32, e->d_out_f32_b};
    uint32_t first[2] = {0, 0}, count[2] = {0, 0};
    auto hand_over = [&](int slot) -> int {
        if (!count[slot]) return PB_OK;
        PB_HIP(hipEventSynchronize(e->ev_out[slot]));
        memcpy(out_u8 + (size_t)first[slot] * e->D, e->h_out_u8[slot], (size_t)count[slot] * e->D);
        if (out_f32) memcpy(out_f32 + (size_t)first[slot] * e->D, e->h_out_f32[slot], (size_t)count[slot] * e->D * sizeof(float));
        count[slot] = 0;
        return PB_OK;
    };
    auto run_chunk = [&](int slot, uint32_t i0, uint32_t c) -> int {
        int rc = hand_over(slot);  // the slot's previous outputs have arrived (so its forward pass, which read d_in[slot], is over)
        if (rc) return rc;
        if ((rc = prepare_images(e, rgb + i0, widths + i0, heights + i0, c, d_in[slot]))) return rc;
        if ((rc = forward_device(e, d_in[slot], (int)c, d_u8[slot], d_f[slot]))) return rc;
        PB_HIP(hipEventRecord(e->ev_fwd[slot], e->stream));
        PB_HIP(hipStreamWaitEvent(e->d2h_stream, e->ev_fwd[slot], 0));
        PB_HIP(hipMemcpyAsync(e->h_out_u8[slot], d_u8[slot], (size_t)c * e->D, hipMemcpyDeviceToHost, e->d2h_stream));
        if (out_f32)
            PB_HIP(hipMemcpyAsync(e->h_out_f32[slot], d_f[slot], (size_t)c * e->D * sizeof(float), hipMemcpyDeviceToHost, e->d2h_stream));
        PB_HIP(hipEventRecord(e->ev_out[slot], e->d2h_stream));
        first[slot] = i0;
        count[slot] = c;
        return PB_OK;
    };
    int slot = 0, rc = PB_OK;
    for (uint32_t i0 = 0; i0 < n && !rc; i0 += e->max_batch, slot ^= 1) rc = run_chunk(slot, i0, std::min(e->max_batch, n - i0));
    if (!rc) rc = hand_over(slot);
    if (!rc) rc = hand_over(slot ^ 1);
    if (!rc) rc = range_status(e);
    if (rc) drain_streams(e);  // nothing of this call may still be in flight when it returns
    if (n == 1) e->g1_quiet_calls = (!rc && e->tune_runs == tuned_before) ? e->g1_quiet_calls + 1 : 0;
    return rc;
}

// ---- decoder-facing staging (include/pixelbox_hip.h: pb_embed_stage_*).  Reference loop being replaced: crawler.rs:68-119 ->
// indexed_image.rs:47-91 (decode into a buffer of the decoder's own, then hash one image at a time).
int pb_embed_stage_acquire(pb_embedder *e, uint32_t w, uint32_t h, uint8_t **pixels, uint64_t *ticket) {
    PB_CHECK(e && pixels && ticket, PB_ERR_INVALID, "pb_embed_stage_acquire: null pointer");
    PB_CHECK(w >= 1 && h >= 1 && w <= 65535 && h <= 65535, PB_ERR_INVALID, "pb_embed_stage_acquire: size %ux%u outside 1..65535", w, h);
    const size_t sb = ((size_t)w * h * 3 + 15) & ~(size_t)15, tmp = (size_t)e->H * w * 3;
    std::unique_lock<std::mutex> lk(e->st_mu);
    if (e->st_open < 0) {  // open the next slot (they alternate); it may still be in its commit
        const int s = e->st_closed == 0 ? 1 : (e->st_closed == 1 ? 0 : (int)(e->st_gen & 1u));
        e->st_cv.wait(lk, [&] { return e->st[s].state == 0 || e->st_open >= 0; });
        if (e->st_open < 0) {
            pb_embedder::StageSlot &sl = e->st[s];
            const size_t want = e->st_bytes_want ? e->st_bytes_want : STAGE_BYTES;
            if (!sl.h || sl.cap < want) {  // first use, or PB_OPT_EMBED_STAGE_BYTES asked for more (the slot is free: nobody holds room in it)
                pb::DeviceGuard guard(e->device);
                PB_CHECK(guard.ok, PB_ERR_HIP, "hipSetDevice(%d) failed", e->device);
                if (sl.h) (void)hipHostFree(sl.h);
                (void)hipFree(sl.d);
                sl.h = nullptr; sl.d = nullptr; sl.cap = 0;
                PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&sl.h), want, hipHostMallocDefault));
                PB_HIP(hipMalloc(reinterpret_cast<void **>(&sl.d), want));
                if (!sl.h_desc) PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&sl.h_desc), STAGE_MAX_IMAGES * sizeof(ResizeDesc), hipHostMallocDefault));
                if (!sl.d_desc) PB_HIP(hipMalloc(reinterpret_cast<void **>(&sl.d_desc), STAGE_MAX_IMAGES * sizeof(ResizeDesc)));
                sl.cap = want;
            }
            sl.state = 1;
            sl.n = 0;
            sl.bytes = sl.tmp = 0;
            sl.writers = 0;
            sl.gen = ++e->st_gen;
            sl.w.clear(); sl.hgt.clear(); sl.off.clear();
            e->st_open = s;
        }
    }
    pb_embedder::StageSlot &sl = e->st[e->st_open];
    const uint32_t lim = std::min<uint32_t>(e->max_batch, STAGE_MAX_IMAGES);
    if (sl.n == 0 && sb > sl.cap) {  // one image larger than the block: the (empty) slot grows to hold it
        pb::DeviceGuard guard(e->device);
        PB_CHECK(guard.ok, PB_ERR_HIP, "hipSetDevice(%d) failed", e->device);
        (void)hipHostFree(sl.h);
        (void)hipFree(sl.d);
        sl.h = nullptr; sl.d = nullptr; sl.cap = 0;
        PB_HIP(hipHostMalloc(reinterpret_cast<void **>(&sl.h), sb, hipHostMallocDefault));
        PB_HIP(hipMalloc(reinterpret_cast<void **>(&sl.d), sb));
        sl.cap = sb;
    }
    if (sl.n >= lim || sl.bytes + sb > sl.cap || (sl.n > 0 && sl.tmp + tmp > STAGE_TMP_FLOATS)) return PB_STAGE_FULL;
    *pixels = sl.h + sl.bytes;
    *ticket = ((uint64_t)sl.gen << 32) | ((uint64_t)e->st_open << 16) | sl.n;
    sl.w.push_back(w); sl.hgt.push_back(h); sl.off.push_back(sl.bytes);
    sl.bytes += sb;
    sl.tmp += tmp;
    ++sl.n;
    ++sl.writers;
    return PB_OK;
}

int pb_embed_stage_release(pb_embedder *e, uint64_t ticket) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_stage_release: null embedder");
    const int s = (int)((ticket >> 16) & 0xFFFFu);
    std::lock_guard<std::mutex> lk(e->st_mu);
    PB_CHECK(s < 2 && e->st[s].gen == (uint32_t)(ticket >> 32) && e->st[s].state != 0 && e->st[s].writers > 0, PB_ERR_INVALID,
             "pb_embed_stage_release: stale ticket");
    if (--e->st[s].writers == 0 && e->st[s].state == 3) {  // the last writer of a discarded batch: the slot is free again
        e->st[s].state = 0;
        e->st[s].n = 0;
    }
    e->st_cv.notify_all();
    return PB_OK;
}

int pb_embed_stage_abort(pb_embedder *e) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_stage_abort: null embedder");
    std::lock_guard<std::mutex> lk(e->st_mu);
    ++e->st_abort_seq;
    auto discard = [&](int s) {
        pb_embedder::StageSlot &sl = e->st[s];
        if (sl.state == 0) return;
        if (sl.writers > 0) {
            sl.state = 3;  // its writers still hold pointers into the block: nobody gets room in it before the last release
        } else {
            sl.state = 0;
            sl.n = 0;
        }
    };
    if (e->st_open >= 0) {
        discard(e->st_open);
        e->st_open = -1;
    }
    if (e->st_closed >= 0 && !e->st_committing) {
        discard(e->st_closed);
        e->st_closed = -1;
    }
    e->st_cv.notify_all();
    return PB_OK;
}

int pb_embed_stage_close(pb_embedder *e, uint32_t *n, uint32_t *generation, uint32_t *widths, uint32_t *heights, const uint8_t **pixels) {
    PB_CHECK(e && n, PB_ERR_INVALID, "pb_embed_stage_close: null pointer");
    std::unique_lock<std::mutex> lk(e->st_mu);
    PB_CHECK(e->st_closed < 0, PB_ERR_INVALID, "pb_embed_stage_close: the batch closed before has not been committed (or aborted)");
    *n = 0;
    if (generation) *generation = 0;
    if (e->st_open < 0) return PB_OK;
    const int s = e->st_open;
    pb_embedder::StageSlot &sl = e->st[s];
    sl.state = 2;
    e->st_open = -1;
    e->st_closed = s;
    const uint32_t seq = e->st_abort_seq;
    e->st_cv.wait(lk, [&] { return sl.writers == 0 || e->st_abort_seq != seq; });
    if (e->st_abort_seq != seq) return PB_STAGE_ABORTED;  // pb_embed_stage_abort has taken the batch
    *n = sl.n;
    if (generation) *generation = sl.gen;
    for (uint32_t i = 0; i < sl.n; ++i) {
        if (widths) widths[i] = sl.w[i];
        if (heights) heights[i] = sl.hgt[i];
        if (pixels) pixels[i] = sl.h + sl.off[i];
    }
    if (sl.n == 0) {  // nothing in it: free again at once
        sl.state = 0;
        e->st_closed = -1;
        e->st_cv.notify_all();
    }
    return PB_OK;
}

int pb_embed_stage_commit(pb_embedder *e, uint8_t *out_u8, const uint8_t **d_out_u8) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_stage_commit: null embedder");
    int s;
    {
        std::lock_guard<std::mutex> lk(e->st_mu);
        s = e->st_closed;
        if (s >= 0) e->st_committing = true;
    }
    if (d_out_u8) *d_out_u8 = e->d_out_u8;
    if (s < 0) return PB_OK;
    pb_embedder::StageSlot &sl = e->st[s];
    int rc;
    {
        std::lock_guard<std::mutex> lock(e->mu);
        pb::DeviceGuard guard(e->device);
        auto body = [&]() -> int {
            const uint32_t m = sl.n, W = e->W, H = e->H;
            size_t tmp_total = 0;
            uint32_t max_w = 1;
            for (uint32_t i = 0; i < m; ++i) {
                ResizeDesc d;
                int rcd = resize_geometry(W, H, sl.w[i], sl.hgt[i], &d);
                if (rcd) return rcd;
                d.slot = i;
                d.src_off = sl.off[i];
                d.tmp_off = tmp_total;
                if (d.resample) tmp_total += ((size_t)H * sl.w[i] * 3 + 3) & ~(size_t)3;
                max_w = std::max(max_w, sl.w[i]);
                sl.h_desc[i] = d;
            }
            if (tmp_total > e->d_tmp_cap && !resize_fused_lds(e, sl.h_desc, m)) {
                drain_streams(e);
                (void)hipFree(e->d_tmp);
                e->d_tmp = nullptr;
                e->d_tmp_cap = 0;
                PB_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_tmp), tmp_total * sizeof(float)));
                e->d_tmp_cap = tmp_total;
            }
            // ONE transfer of the block the decoders wrote (no packing pass), then the two resize kernels and the forward pass
            PB_HIP(hipMemcpyAsync(sl.d, sl.h, sl.bytes, hipMemcpyHostToDevice, e->stream));
            PB_HIP(hipMemcpyAsync(sl.d_desc, sl.h_desc, m * sizeof(ResizeDesc), hipMemcpyHostToDevice, e->stream));
            { int rcr = launch_resize(e, sl.d, sl.d_desc, sl.h_desc, m, max_w, tmp_total, e->d_img); if (rcr) return rcr; }
            int rcf = forward_device(e, e->d_img, (int)m, e->d_out_u8, e->d_out_f32);
            if (rcf) return rcf;
            if (out_u8) PB_HIP(hipMemcpyAsync(out_u8, e->d_out_u8, (size_t)m * e->D, hipMemcpyDeviceToHost, e->stream));
            PB_HIP(hipStreamSynchronize(e->stream));
            return range_status(e);
        };
        rc = body();
        if (rc) drain_streams(e);
    }
    std::lock_guard<std::mutex> lk(e->st_mu);  // the block has been read (or the call failed): the slot is free again either way
    sl.state = 0;
    sl.n = 0;
    e->st_closed = -1;
    e->st_committing = false;
    e->st_cv.notify_all();
    return rc;
}

int pb_embed_batch_images_device(pb_embedder *e, const uint8_t *const *rgb, const uint32_t *widths, const uint32_t *heights, uint32_t n,
                                 uint8_t *out_u8, const uint8_t **d_out_u8) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_batch_images_device: null embedder");
    PB_CHECK(d_out_u8, PB_ERR_INVALID, "pb_embed_batch_images_device: null device-pointer output");
    PB_CHECK(n <= e->max_batch, PB_ERR_INVALID, "pb_embed_batch_images_device: n = %u > max_batch %u", n, e->max_batch);
    PB_CHECK(n == 0 || (rgb && widths && heights), PB_ERR_INVALID, "pb_embed_batch_images_device: null buffer");
    *d_out_u8 = e->d_out_u8;
    if (n == 0) return PB_OK;
    std::lock_guard<std::mutex> lock(e->mu);
    pb::DeviceGuard guard(e->device);
    auto body = [&]() -> int {
        int rc = prepare_images(e, rgb, widths, heights, n, e->d_img);
        if (rc) return rc;
        if ((rc = forward_device(e, e->d_img, (int)n, e->d_out_u8, e->d_out_f32))) return rc;
        if (out_u8) PB_HIP(hipMemcpyAsync(out_u8, e->d_out_u8, (size_t)n * e->D, hipMemcpyDeviceToHost, e->stream));
        PB_HIP(hipStreamSynchronize(e->stream));
        return range_status(e);
    };
    const int rc = body();
    if (rc) drain_streams(e);
    return rc;
}

int pb_mlhash_image(pb_embedder *e, const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out, size_t out_len) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_mlhash_image: null embedder");
    PB_CHECK(out_len >= e->D, PB_ERR_INVALID, "pb_mlhash_image: out_len %zu < D = %u", out_len, e->D);
    return pb_embed_batch_images(e, &rgb, &width, &height, 1, out, nullptr);
}

int pb_resize_to_fill(pb_embedder *e, const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out_rgb) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_resize_to_fill: null embedder");
    PB_CHECK(out_rgb, PB_ERR_INVALID, "pb_resize_to_fill: null output");
    std::lock_guard<std::mutex> lock(e->mu);
    pb::DeviceGuard guard(e->device);
    auto body = [&]() -> int {
        int rc = prepare_images(e, &rgb, &width, &height, 1, e->d_img);
        if (rc) return rc;
        PB_HIP(hipMemcpyAsync(out_rgb, e->d_img, (size_t)e->H * e->W * 3, hipMemcpyDeviceToHost, e->stream));
        PB_HIP(hipStreamSynchronize(e->stream));
        return PB_OK;
    };
    const int rc = body();
    if (rc) drain_streams(e);
    return rc;
}

int pb_pinned_alloc(void **out, size_t bytes) {
    PB_CHECK(out, PB_ERR_INVALID, "pb_pinned_alloc: null out pointer");
    *out = nullptr;
    void *p = nullptr;
    PB_HIP(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
    *out = p;
    return PB_OK;
}

int pb_pinned_free(void *p) {
    if (p) PB_HIP(hipHostFree(p));
    return PB_OK;
}

int pb_embed_tune_ms(pb_embedder *e, double *ms) {
    PB_CHECK(e && ms, PB_ERR_INVALID, "pb_embed_tune_ms: null pointer");
    std::lock_guard<std::mutex> lock(e->mu);
    *ms = e->tune_ms;
    return PB_OK;
}

namespace {
// serialised picks: header {magic "PBTN", format, H, W, D, n_entries} then entries {map, key id, bucket, v[6]}.  `format` changes
// whenever an encoding of gemm_cfg / dw_cfg / front_cfg changes meaning (a new kernel form, a renumbered shape).
constexpr uint32_t TUNE_MAGIC = 0x4E544250u, TUNE_FORMAT = 6u;
struct TuneHdr {
    uint32_t magic, format, H, W, D, n;
    uint32_t p3_min_k, reserved;  // which layers are P3 layers is part of what a gemm entry's numbers MEAN (format 6)
};
struct TuneEntry {
    uint32_t map, key;
    int64_t bucket;
    int32_t v[6];
};
uint32_t tune_key_id(const pb_embedder *e, const void *p) {
    for (const auto &kv : e->tune_keys)
        if (kv.first == p) return kv.second;
    return 0xFFFFFFFFu;
}
const void *tune_key_ptr(const pb_embedder *e, uint32_t id) {
    for (const auto &kv : e->tune_keys)
        if (kv.second == id) return kv.first;
    return nullptr;
}
}  // namespace

int pb_embed_get_tuning(pb_embedder *e, uint8_t *out, size_t cap, size_t *len) {
    PB_CHECK(e && len, PB_ERR_INVALID, "pb_embed_get_tuning: null pointer");
    std::lock_guard<std::mutex> lock(e->mu);
    std::vector<TuneEntry> ent;
    for (const auto &kv : e->gemm_cfg) {
        TuneEntry t{0u, tune_key_id(e, kv.first.first), kv.first.second, {kv.second.first, kv.second.second, 0, 0, 0, 0}};
        if (t.key != 0xFFFFFFFFu) ent.push_back(t);
    }
    for (const auto &kv : e->dw_cfg) {
        const DwGeom &g = kv.second;
        TuneEntry t{1u, tune_key_id(e, kv.first.first), kv.first.second, {g.roll, g.zsplit, g.cqpb, g.slots, g.n_tiles, g.strips_per_tile}};
        if (t.key != 0xFFFFFFFFu) ent.push_back(t);
    }
    for (const auto &kv : e->front_cfg) {
        TuneEntry t{2u, tune_key_id(e, kv.first.first), kv.first.second, {kv.second, 0, 0, 0, 0, 0}};
        if (t.key != 0xFFFFFFFFu) ent.push_back(t);
    }
    // by (map, layer, bucket), not by the maps' pointer order: two embedders with the same picks serialise to the same bytes
    std::sort(ent.begin(), ent.end(), [](const TuneEntry &a, const TuneEntry &b) {
        return a.map != b.map ? a.map < b.map : (a.key != b.key ? a.key < b.key : a.bucket < b.bucket);
    });
    const size_t need = sizeof(TuneHdr) + ent.size() * sizeof(TuneEntry);
    *len = need;
    if (!out) return PB_OK;
    PB_CHECK(cap >= need, PB_ERR_INVALID, "pb_embed_get_tuning: buffer of %zu bytes, %zu needed", cap, need);
    const TuneHdr h{TUNE_MAGIC, TUNE_FORMAT, e->H, e->W, e->D, (uint32_t)ent.size(), (uint32_t)e->p3_min_k, 0u};
    memcpy(out, &h, sizeof h);
    if (!ent.empty()) memcpy(out + sizeof h, ent.data(), ent.size() * sizeof(TuneEntry));
    return PB_OK;
}

int pb_embed_set_tuning(pb_embedder *e, const uint8_t *data, size_t len) {
    PB_CHECK(e && data, PB_ERR_INVALID, "pb_embed_set_tuning: null pointer");
    std::lock_guard<std::mutex> lock(e->mu);
    g1_drop(e);  // the one-image graph froze the picks it was captured with
    TuneHdr h;
    PB_CHECK(len >= sizeof h, PB_ERR_FORMAT, "tuning data: %zu bytes is shorter than its header", len);
    memcpy(&h, data, sizeof h);
    PB_CHECK(h.magic == TUNE_MAGIC && h.format == TUNE_FORMAT, PB_ERR_FORMAT, "tuning data: not a PBTN block of this library build (format %u, want %u)",
             h.format, TUNE_FORMAT);
    PB_CHECK(h.H == e->H && h.W == e->W && h.D == e->D, PB_ERR_FORMAT, "tuning data: made for %u x %u -> %u, this embedder is %u x %u -> %u", h.H, h.W,
             h.D, e->H, e->W, e->D);
    PB_CHECK(len == sizeof h + (size_t)h.n * sizeof(TuneEntry), PB_ERR_FORMAT, "tuning data: %zu bytes for %u entries", len, h.n);
    std::vector<TuneEntry> ent(h.n);
    if (h.n) memcpy(ent.data(), data + sizeof h, (size_t)h.n * sizeof(TuneEntry));
    PB_CHECK(h.p3_min_k == (uint32_t)e->p3_min_k, PB_ERR_FORMAT, "tuning data: made with P3 layers from K = %u on, this embedder has %d", h.p3_min_k, e->p3_min_k);
    // Every entry is checked before any is accepted (all or nothing): the numbers go straight into launch geometry, and a stale or
    // damaged block must not become a grid that leaves output columns unwritten (a column-tile count that does not divide the
    // layer's) or a shape that is not built (which would fail every later call of that bucket).
    auto find_gemm = [&](const void *p, bool *frag_key) -> const Gemm * {
        auto hit = [&](const Gemm &g) { if (g.wt == p) { *frag_key = false; return true; } if (g.wt2 && g.wt2 == p) { *frag_key = true; return true; } return false; };
        for (const Block &bl : e->blocks) {
            if (bl.has_expand && hit(bl.expand)) return &bl.expand;
            if (hit(bl.project)) return &bl.project;
        }
        if (hit(e->head)) return &e->head;
        if (hit(e->fc)) return &e->fc;
        return nullptr;
    };
    for (const TuneEntry &t : ent) {
        PB_CHECK(t.map <= 2u && tune_key_ptr(e, t.key) && t.bucket >= 1 && (t.bucket & (t.bucket - 1)) == 0, PB_ERR_FORMAT,
                 "tuning data: entry for layer %u / map %u does not belong to this model", t.key, t.map);
        if (t.map == 0) {
            bool frag = false;
            const Gemm *g = find_gemm(tune_key_ptr(e, t.key), &frag);
            PB_CHECK(g, PB_ERR_FORMAT, "tuning data: GEMM entry for key %u, which is not a GEMM layer", t.key);
            const int tiles = g->Npad / 16;
            if (g->p3) {  // enc = 4096 nw + 16 mr + nr, for the store epilogue (key wt) or the head / Linear epilogues (key wt2)
                const int nr = t.v[0] & 15, mr = (t.v[0] >> 4) & 15, nw = t.v[0] >> 12;
                const int epi = !frag ? 0 : (g == &e->head ? 1 : 2);
                bool built = false;
                for (int gate = 0; gate < 2 && !built; ++gate) built = p3_has(nr, mr, nw, epi, gate != 0, (g->K & 31) != 0);
                PB_CHECK(nr >= 1 && tiles % nr == 0 && built, PB_ERR_FORMAT, "tuning data: P3 GEMM shape NR%d MR%d NW%d is not built for layer %u (%d column tiles)", nr,
                         mr, nw, t.key, tiles);
            } else if (frag) {  // the pooling head of the f32 chain: (waves, column tiles per workgroup)
                PB_CHECK((t.v[0] == 4 || t.v[0] == 8) && t.v[1] >= 1 && t.v[1] <= 8 && tiles % t.v[1] == 0, PB_ERR_FORMAT,
                         "tuning data: head shape NW%d NR%d does not fit layer %u", t.v[0], t.v[1], t.key);
            } else {  // the f32 forms: (form code, column tiles per workgroup)
                const int c = t.v[0], nr = t.v[1];
                const bool form = c == 1 || c == 2 || c == 4 || c == -1 || c == 100 || c == 304 || c == 308 || c == 1008 || c == 1016 || c == 1032 || c == 1064;
                PB_CHECK(form && nr >= 1 && nr <= 8 && tiles % nr == 0 && (c < 1000 || nr == tiles) && (c != 100 || nr == 1) && ((c != 304 && c != 308) || g->wt2),
                         PB_ERR_FORMAT, "tuning data: GEMM form %d with %d column tiles per workgroup does not fit layer %u (%d tiles)", c, nr, t.key, tiles);
            }
        } else if (t.map == 1) {  // depthwise geometry: ranges of the three kernels' launch shapes
            bool in_range = t.v[0] >= 0 && t.v[0] <= 2;
            for (int j = 1; j < 6; ++j) in_range = in_range && t.v[j] >= 0 && t.v[j] <= 65536;
            // (the launch is grid (n_tiles, B, zsplit) x block cqpb * slots in all three kernels)
            PB_CHECK(in_range && (long)t.v[2] * t.v[3] >= 1 && (long)t.v[2] * t.v[3] <= 1024 && t.v[1] >= 1 && t.v[4] >= 1,
                     PB_ERR_FORMAT, "tuning data: depthwise geometry of layer %u out of range", t.key);
        } else {
            const int c = t.v[0];
            const bool small = c >= 0x1000 && c < 0x2000 && (((c >> 8) & 15) == 1 || ((c >> 8) & 15) == 4) && (((c >> 4) & 15) == 2 || ((c >> 4) & 15) == 3) && (c & 7) <= 2;
            // k_front_band: 0x2000 + bands per image (1, 2, 4, 8, 16) + 64 lg(items a workgroup walks: 1, 2, 4, 8)
            const int bnb = (c - 0x2000) & 63, bip = (c - 0x2000) >> 6;
            const bool band = c > 0x2000 && bip >= 0 && bip <= 3 && bnb >= 1 && bnb <= 16 && (bnb & (bnb - 1)) == 0;
            PB_CHECK(c == 0 || c == 1 || small || band || (c > 1 && c < 0x1000), PB_ERR_FORMAT, "tuning data: front form %d of layer %u is not a form of this build", c, t.key);
        }
    }
    for (const TuneEntry &t : ent) {
        const std::pair<const void *, long> key(tune_key_ptr(e, t.key), (long)t.bucket);
        if (t.map == 0) e->gemm_cfg[key] = std::make_pair(t.v[0], t.v[1]);
        else if (t.map == 1) {
            DwGeom g;
            g.roll = t.v[0]; g.zsplit = t.v[1]; g.cqpb = t.v[2]; g.slots = t.v[3]; g.n_tiles = t.v[4]; g.strips_per_tile = t.v[5];
            e->dw_cfg[key] = g;
        } else e->front_cfg[key] = t.v[0];
    }
    return PB_OK;
}

int pb_embed_set_option(pb_embedder *e, int option, int64_t value) {
    PB_CHECK(e, PB_ERR_INVALID, "pb_embed_set_option: null embedder");
    std::lock_guard<std::mutex> lock(e->mu);
    if (option == PB_OPT_EMBED_STREAM) {
        e->stream = value ? reinterpret_cast<hipStream_t>(value) : e->own_stream;
        g1_drop(e);  // the one-image graph was captured on the other stream
        return PB_OK;
    }
    if (option == PB_OPT_EMBED_ASYNC) {
        PB_CHECK(value == 0 || value == 1, PB_ERR_INVALID, "PB_OPT_EMBED_ASYNC: 0 or 1");
        e->opt_async = (int)value;
        return PB_OK;
    }
    if (option == PB_OPT_EMBED_STAGE_BYTES) {
        PB_CHECK(value >= (1 << 20) && value <= (1ll << 31), PB_ERR_INVALID, "PB_OPT_EMBED_STAGE_BYTES: 1 MB .. 2 GB");
        std::lock_guard<std::mutex> lk(e->st_mu);
        e->st_bytes_want = ((size_t)value + 4095) & ~(size_t)4095;  // taken by a slot the next time it is opened empty
        return PB_OK;
    }
    if (option == PB_OPT_EMBED_DUAL) {
        PB_CHECK(value >= 0 && value <= (int64_t)1 << 20, PB_ERR_INVALID, "PB_OPT_EMBED_DUAL: 0 (off) or the batch size from which a forward runs as two concurrent halves");
        e->dual_min = (int)value;
        return PB_OK;
    }
    if (option == PB_OPT_EMBED_FRONT_SUB) {
        PB_CHECK(value >= 0 && value <= (int64_t)e->max_batch, PB_ERR_INVALID, "PB_OPT_EMBED_FRONT_SUB: 0 (off) .. max_batch images");
        e->front_sub = (int)value;
        return PB_OK;
    }
    return pb::fail(PB_ERR_INVALID, "pb_embed_set_option: unknown option %d", option);
}

#ifdef PB_SM_STAMP_E
int pb_debug_small_stamps(unsigned long long *out, int reset) {  // out: 65536 * 12
    PB_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sm_stamp), 65536 * 12 * sizeof(unsigned long long)));
    if (reset) {
        void *p = nullptr;
        PB_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_sm_stamp)));
        PB_HIP(hipMemset(p, 0, 65536 * 12 * sizeof(unsigned long long)));
        PB_HIP(hipDeviceSynchronize());
    }
    return PB_OK;
}
#endif

}  // extern "C"
