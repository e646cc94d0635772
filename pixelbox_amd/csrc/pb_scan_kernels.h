// pb_scan_kernels.h -- gfx950 device code for the u8 cosine-distance top-k scan.
//
// Reference semantics (PixelBox): src/engine.rs:572-588 (cosine_distance), :608-622 (f64 widening),
// :375-390 (WHERE dist < ? ORDER BY dist ASC LIMIT k).  DESIGN.md section "scan" explains the two
// paths built here:
//   (1) k_scan_filter  -- the HBM-bound pass: per-row EXACT INTEGER dot / norms with v_dot4_u32_u8,
//       an f32 cosine that is within M_GLOB of the reference's f32 cosine for every possible row,
//       per-wave candidate buffers in LDS pruned by a ballot radix-select while streaming and closed by bitonic
//       networks, one sorted list per workgroup plus the count of rows it evaluated;  k_select_rescore then re-scores
//       the few candidates with the reference's exact sequential f32 arithmetic and certifies that no other row can
//       belong to the top-k -- and that every row of the table was evaluated exactly once.
//   (2) k_scan_exact / k_merge_lists -- exhaustive exact scan (every row re-scored); used when the
//       certificate of (1) fails, for dims (1) does not cover, or when forced (PB_OPT_SEARCH_PATH).
// All keys are u64 "smaller is better": (order-preserving score bits << 32) | row position; rows are
// stored in ascending image_id order, so row position breaks ties exactly like (dist, image_id).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace pbk {

constexpr int WAVE = 64;
// ---- filter pass geometry ----
constexpr int F_WAVES = 8;               // default: 8 waves per workgroup, ONE workgroup per CU (measured best)
constexpr int F_BLOCK = F_WAVES * WAVE;
constexpr int F_KW = 32;                 // entries a wave keeps after a prune
constexpr int F_CAPW = 128;              // per-wave LDS buffer (entries); prune when > CAPW - 64
constexpr int F_KWG = 32;                // entries per workgroup list
constexpr int F_MAX_WG = 512;
constexpr int SEL_BLOCK = 1024;
constexpr int SEL_MAX_CAND = 1024;
constexpr int SEL_BINS = 1024;
// ---- exact pass geometry ----
constexpr int X_BLOCK = 256;
constexpr int X_WAVES = X_BLOCK / WAVE;
constexpr int X_MAX_WG = 512;
constexpr int X_MAXE = 5;  // (PB_MAX_K + 64) / 64
constexpr int M_FANIN = 16;  // lists merged per workgroup by k_merge_lists
constexpr int M_BLOCK = 1024;
constexpr int M_SORT = 4096;  // M_FANIN * PB_MAX_K

// Rigorous bound on |cos_reference_f32 - cos_filter| for any two byte vectors of length <= 1024,
// see DESIGN.md "error budget": 2*gamma_n + 4u from the sequential f32 folds, 6u*sqrt(n)*(1/|x|+1/|y|)
// from the rounded de-quantisation table (|x| >= sqrt(n)/255), 1e-6 for the filter's own f32 steps.
// n = 256: 3.1e-5 + 1.8e-4 + 1e-6.  The constant below is used for every n <= 1024 that the filter
// pass accepts (gamma_n grows to 1.2e-4 at n = 1024, the table term does not depend on n).
constexpr float M_GLOB = 4.0e-4f;  // ceiling of the per-query margin QParams::m (worst case over all byte vectors)

struct QParams {
    double max_dist;     // WHERE dist < ?
    float sqrt_sa;       // sqrt(fold(x^2)) of the de-quantised query, reference f32 arithmetic
    float den_a;         // exact integer sum (2a-255)^2 as f32
    float thr0;          // filter pass: rows with cos_filter < thr0 are never collected
    float c_floor;       // every row with cos_ref < c_floor fails `dist < max_dist`
    int32_t sum_a;       // integer sum of query bytes
    int32_t floor_is_filter;  // thr0 == c_floor - m (rows below thr0 are provably filtered out)
    uint32_t k;
    float m;             // |cos_filter - cos_ref| <= m for this query against every stored row
};

// one query with its constants as a kernel argument (k_stage_query)
struct QArg {
    QParams p;
    uint32_t dim;
    uint32_t pad_;
    uint8_t q[1024];
};

struct ListHdr {
    uint32_t count;
    float dropped;  // every row this workgroup saw and did not list has cos_filter <= dropped (0: none)
    uint32_t rows_seen;  // rows this workgroup evaluated for the query (k_scan_filter): the certificate of k_select_rescore also
                         // requires that the workgroups' counts add up to the table -- the partition of the table over
                         // workgroups is dynamic in some launch forms, and a tile nobody read must cost time, not an answer
    uint32_t sig_lo, sig_hi;  // sum over the tiles this workgroup evaluated of (tile number + 1)^2, mod 2^64 (round 6): the counts alone
                              // are a NECESSARY condition -- a tile read twice while another full tile is skipped adds up too -- the
                              // sums of squares beside them must also come to the closed form for tiles 0 .. n - 1 (tile_sig_expected)
};
// sum_{s = 0}^{n - 1} (s + 1)^2 mod 2^64 = n (n + 1) (2 n + 1) / 6 mod 2^64 (the division before the reduction: one factor of 2 and
// one of 3 are taken out of the three factors first)
__host__ __device__ inline uint64_t tile_sig_expected(uint64_t n) {
    uint64_t a = n, b = n + 1, c = 2 * n + 1;
    if ((a & 1) == 0) a >>= 1; else b >>= 1;  // n (n + 1) is even
    if (a % 3 == 0) a /= 3; else if (b % 3 == 0) b /= 3; else c /= 3;  // one of n, n + 1, 2 n + 1 is a multiple of 3
    return a * b * c;
}

struct ResultHdr {
    uint32_t count;
    uint32_t status;  // 0 = certified, 1 = needs the exhaustive pass
    uint32_t n_cand;
    float o_max;
    float ck;  // smallest exact cosine among the count results when count == k (a lower bound of the true k-th), else -1
};

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot4(a, b, c, false);
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// all-reduce (sum) over groups of LPR consecutive lanes; LPR in {1,2,4,8,16,32,64}
template <int LPR>
__device__ __forceinline__ int group_sum(int v) {
    if constexpr (LPR >= 2) v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]  (lane ^ 1)
    if constexpr (LPR >= 4) v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]  (lane ^ 2)
    if constexpr (LPR == 8) v += dpp_mov<0x141>(v);  // row_half_mirror: quads hold equal sums -> lane ^ 4 class
    if constexpr (LPR >= 16) {
        v += dpp_mov<0x124>(v);  // row_ror:4
        v += dpp_mov<0x128>(v);  // row_ror:8
    }
    if constexpr (LPR >= 32) v += __shfl_xor(v, 16);
    if constexpr (LPR >= 64) v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ uint32_t sortable_f32(float f) {
    uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unsortable_f32(uint32_t s) {
    uint32_t b = (s & 0x80000000u) ? (s ^ 0x80000000u) : ~s;
    return __uint_as_float(b);
}
// filter keys: larger cosine = better = smaller key. cos > 0 only, so ~bits is order-reversing.
__device__ __forceinline__ uint64_t filter_key(float cs, uint32_t row) {
    return ((uint64_t)(~__float_as_uint(cs)) << 32) | row;
}
__device__ __forceinline__ float filter_key_cos(uint64_t key) { return __uint_as_float(~(uint32_t)(key >> 32)); }

// Keep the K smallest of buf[0..cnt) (wave-private LDS, all keys distinct), compacted to buf[0..K).
// Ballot radix-select: 32 steps on the score word, then (only if scores tie at the cut) on the row word.
// MAXE = ceil(max cnt / 64).  Requires cnt >= K.  Returns the K-th smallest key.
template <int MAXE>
__device__ __forceinline__ uint64_t wave_keep_smallest(uint64_t *buf, int cnt, int K) {
    const int lane = lane_id();
    uint32_t hi[MAXE], lo[MAXE];
    uint32_t act = 0;
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
        const int i = lane + 64 * j;
        const uint64_t e = (i < cnt) ? buf[i] : ~0ull;
        hi[j] = (uint32_t)(e >> 32);
        lo[j] = (uint32_t)e;
        act |= (i < cnt) ? (1u << j) : 0u;
    }
    const uint32_t valid = act;
    int remaining = K;  // the K-th smallest is the `remaining`-th smallest of the active set
    int nact = cnt;
    uint32_t pre_hi = 0, pre_lo = 0;
    for (int bit = 31; bit >= 0; --bit) {
        int z = 0;
        uint32_t zmask = 0;
#pragma unroll
        for (int j = 0; j < MAXE; ++j) {
            const bool isz = ((act >> j) & 1u) && !((hi[j] >> bit) & 1u);
            z += __popcll(__ballot(isz));
            zmask |= isz ? (1u << j) : 0u;
        }
        if (remaining <= z) {
            act = zmask;
            nact = z;
        } else {
            remaining -= z;
            act &= ~zmask;
            nact -= z;
            pre_hi |= 1u << bit;
        }
    }
    // active = entries whose score word equals pre_hi
    if (nact > remaining) {
        for (int bit = 31; bit >= 0; --bit) {
            int z = 0;
            uint32_t zmask = 0;
#pragma unroll
            for (int j = 0; j < MAXE; ++j) {
                const bool isz = ((act >> j) & 1u) && !((lo[j] >> bit) & 1u);
                z += __popcll(__ballot(isz));
                zmask |= isz ? (1u << j) : 0u;
            }
            if (remaining <= z) {
                act = zmask;
            } else {
                remaining -= z;
                act &= ~zmask;
                pre_lo |= 1u << bit;
            }
        }
    } else {
        // every entry with score word pre_hi is kept; the K-th smallest is the largest row among them
        uint32_t mx = 0;
#pragma unroll
        for (int j = 0; j < MAXE; ++j) mx = ((act >> j) & 1u) && lo[j] > mx ? lo[j] : mx;
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mx, off);
            mx = o > mx ? o : mx;
        }
        pre_lo = mx;
    }
    int base = 0;
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
        const bool keep = ((valid >> j) & 1u) && (hi[j] < pre_hi || (hi[j] == pre_hi && lo[j] <= pre_lo));
        const uint64_t m = __ballot(keep);
        if (keep) buf[base + mbcnt(m)] = ((uint64_t)hi[j] << 32) | lo[j];
        base += __popcll(m);
    }
    return ((uint64_t)pre_hi << 32) | pre_lo;
}

// ---- bitonic networks over ONE u64 key per lane (the ends of the filter passes: a wave's <= 64 buffered keys are sorted in
// 21 compare-exchange steps, two sorted 32-key lists are merged in 6).  A ballot radix-select of the same keys costs 32+
// dependent bit-steps (2.3 us per wave, 7.5 us for the workgroup's 256 keys -- stamps: profiles/r04_scan_stamps.txt); the
// networks run at DPP speed for partner distances 1, 2 and 8 and pay an LDS-crossbar permute for 4, 16 and 32.
template <int D>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (D == 1) return (uint32_t)dpp_mov<0xB1>((int)v);        // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return (uint32_t)dpp_mov<0x4E>((int)v);   // quad_perm [2,3,0,1]
    else if constexpr (D == 8) return (uint32_t)dpp_mov<0x128>((int)v);  // row_ror:8 = lane ^ 8 within a row of 16
    else return (uint32_t)__shfl_xor((int)v, D);
}
// compare-exchange with lane ^ D: this lane keeps the smaller key of the pair iff keep_min
template <int D>
__device__ __forceinline__ uint64_t lane_cmpx(uint64_t key, bool keep_min) {
    const uint32_t ph = lane_xor<D>((uint32_t)(key >> 32)), pl = lane_xor<D>((uint32_t)key);
    const uint64_t p = ((uint64_t)ph << 32) | pl;
    return ((p < key) == keep_min) ? p : key;
}
// lanes hold a bitonic sequence (ascending then descending) -> ascending over the 64 lanes
__device__ __forceinline__ uint64_t wave_merge64(uint64_t key) {
    int lane = lane_id();
    asm volatile("" : "+v"(lane));  // the lane predicates below are made HERE: hoisted out of a filter kernel's query loop they
                                    // stay live as ~20 SGPR pairs across its streaming loop, which then spills (-2 % at 10M rows)
    key = lane_cmpx<32>(key, (lane & 32) == 0);
    key = lane_cmpx<16>(key, (lane & 16) == 0);
    key = lane_cmpx<8>(key, (lane & 8) == 0);
    key = lane_cmpx<4>(key, (lane & 4) == 0);
    key = lane_cmpx<2>(key, (lane & 2) == 0);
    key = lane_cmpx<1>(key, (lane & 1) == 0);
    return key;
}
// any 64 keys -> ascending over the 64 lanes
__device__ __forceinline__ uint64_t wave_sort64(uint64_t key) {
    int lane = lane_id();
    asm volatile("" : "+v"(lane));  // as in wave_merge64
#define PB_CX(D, K) key = lane_cmpx<D>(key, ((lane & (D)) == 0) == (((lane >> (K)) & 1) == 0))
    PB_CX(1, 1);
    PB_CX(2, 2); PB_CX(1, 2);
    PB_CX(4, 3); PB_CX(2, 3); PB_CX(1, 3);
    PB_CX(8, 4); PB_CX(4, 4); PB_CX(2, 4); PB_CX(1, 4);
    PB_CX(16, 5); PB_CX(8, 5); PB_CX(4, 5); PB_CX(2, 5); PB_CX(1, 5);
#undef PB_CX
    return wave_merge64(key);
}
// A wave's buffered keys (cnt <= 64, distinct) -> its F_KW smallest, ascending, in buf[0..F_KW) (unused slots ~0).
// Returns the smallest key it discarded (~0: none); cnt becomes min(cnt, F_KW).
__device__ __forceinline__ uint64_t wave_finish_list(uint64_t *buf, int &cnt) {
    const int lane = lane_id();
    uint64_t key = lane < cnt ? buf[lane] : ~0ull;
    key = wave_sort64(key);
    const uint64_t first_out = __shfl((unsigned long long)key, F_KW);
    if (lane < F_KW) buf[lane] = key;
    cnt = cnt < F_KW ? cnt : F_KW;
    return first_out;
}
// Wave 0 merges the NW sorted wave lists (bufs[w * pitch .. + F_KW), ~0-padded) into the workgroup's F_KWG best, ascending
// over lanes 0..31 of the result; *n_out their number, *first_out the smallest key discarded here (~0: none).
template <int NW>
__device__ __forceinline__ uint64_t wave_merge_lists(const uint64_t *bufs, int pitch, int *n_out, uint64_t *first_out) {
    static_assert(F_KW == 32 && F_KWG == 32, "two 32-key lists fill one wave");
    const int lane = lane_id();
    uint64_t oth[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) oth[w] = bufs[(size_t)w * pitch + (63 - lane < F_KW ? 63 - lane : 0)];  // lanes >= 32: list w reversed
    uint64_t key = lane < F_KW ? bufs[lane] : ~0ull;
    uint64_t out_min = ~0ull;
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        key = wave_merge64(lane < F_KW ? key : oth[w]);
        const uint64_t o = __shfl((unsigned long long)key, F_KW);
        out_min = o < out_min ? o : out_min;
    }
    *n_out = __popcll(__ballot(lane < F_KWG && key != ~0ull));
    *first_out = out_min;
    return key;
}

// ------------------------------------------------------------------------------------------------
// exact reference arithmetic for one row (engine.rs:575-587); row_norm = sqrt(fold(x^2)) of the row,
// precomputed at append time by k_row_norms with the same arithmetic.  s_lut/s_qf live in LDS.
__device__ __forceinline__ float ref_fold_dot(const uint8_t *__restrict__ row, const float *s_qf,
                                              const float *s_lut, int d) {
    float dot = 0.0f;
    int i = 0;
    if ((d & 15) == 0) {
        for (; i < d; i += 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(row + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float x = s_lut[(w[c] >> (8 * b)) & 0xFF];
                    const float p = s_qf[i + 4 * c + b] * x;
                    dot = dot + p;
                }
            }
        }
    } else {
        for (; i < d; ++i) {
            const float p = s_qf[i] * s_lut[row[i]];
            dot = dot + p;
        }
    }
    return dot;
}

// The same fold for a 256-byte row with ALL sixteen 16-byte loads requested before the first use: the candidates of a
// re-scoring kernel are random rows in HBM, and the loop above (runtime d, one load per trip) pays one memory round
// trip per 16 bytes -- sixteen dependent trips per candidate on the latency-bound tail of every query.  Same operations
// in the same order: bit-identical.
__device__ __forceinline__ float ref_fold_dot256(const uint8_t *__restrict__ row, const float *s_qf, const float *s_lut) {
    uint4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const uint4 *>(row + 16 * j);
    float dot = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float x = s_lut[(w[c] >> (8 * b)) & 0xFF];
                const float p = s_qf[16 * j + 4 * c + b] * x;
                dot = dot + p;
            }
        }
    }
    return dot;
}
// The same fold for a 256-byte row by EIGHT consecutive lanes (lane `part` = threadIdx & 7 owns bytes 32 part .. 32 part + 31,
// already loaded: v0, v1): the 256 de-quantisations (an LDS look-up each) and products are the long part of a re-scoring and
// do not depend on each other, so each lane makes 32 of them; the sum itself stays the reference's one chain -- the group walks
// its eight parts in order, every lane adding its own products to what it holds, and after each part the sums move one lane
// up (row_shr:1), so that lane p starts part p from the sum through part p - 1.  Same operations in the same order on the
// value that counts: bit-identical.  All eight lanes must be active; the result is lane 7's.
__device__ __forceinline__ float ref_fold_dot256_by8(const uint4 v0, const uint4 v1, const float *s_qf, const float *s_lut, int part) {
    float p[32];
    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 qf = *reinterpret_cast<const float4 *>(s_qf + 32 * part + 4 * c);
        p[4 * c + 0] = qf.x * s_lut[w[c] & 0xFF];
        p[4 * c + 1] = qf.y * s_lut[(w[c] >> 8) & 0xFF];
        p[4 * c + 2] = qf.z * s_lut[(w[c] >> 16) & 0xFF];
        p[4 * c + 3] = qf.w * s_lut[w[c] >> 24];
    }
    float a = 0.0f;
#pragma unroll
    for (int ph = 0; ph < 8; ++ph) {
#pragma unroll
        for (int i = 0; i < 32; ++i) a = a + p[i];
        if (ph < 7) a = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x111, 0xF, 0xF, true));  // row_shr:1
    }
    return a;
}
__device__ __forceinline__ float ref_fold_dot_any(const uint8_t *__restrict__ row, const float *s_qf, const float *s_lut, int d) {
    return d == 256 ? ref_fold_dot256(row, s_qf, s_lut) : ref_fold_dot(row, s_qf, s_lut, d);
}

__device__ __forceinline__ float ref_distance(float dot, float sqrt_sa, float row_norm, float *cs_out) {
    const float magnitude = sqrt_sa * row_norm;  // engine.rs:581
    if (magnitude < 1e-6f) {                     // engine.rs:582-584
        *cs_out = 0.0f;
        return 0.0f;
    }
    const float cs = dot / magnitude;  // engine.rs:586 (correctly rounded: hipcc default)
    *cs_out = cs;
    const float m = fmaxf(cs, 1e-6f);  // f32::max
    const float r = 1.0f / m;
    return r - 1.0f;  // engine.rs:587
}

// sqrt(fold(x*x)) per row, row-per-lane (engine.rs:580-581, the `hash_b` half)
__global__ void k_row_norms(const uint8_t *__restrict__ rows, uint64_t first, uint64_t n, int d,
                            const float *__restrict__ lut, float *__restrict__ norms,
                            int32_t *__restrict__ sum_b, int32_t *__restrict__ den_b, int32_t *__restrict__ min_den) {
    __shared__ float s_lut[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_lut[i] = lut[i];
    __syncthreads();
    int32_t my_min = 0x7FFFFFFF;  // smallest sum (2b-255)^2 seen: feeds the per-query error margin
    for (uint64_t r = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < first + n;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *row = rows + r * (uint64_t)d;
        float acc = 0.0f;
        int32_t sb = 0, sb2 = 0;
        auto take = [&](uint32_t w) __attribute__((always_inline)) {  // four bytes, in order: the fold stays sequential (engine.rs:580)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int v = (int)((w >> (8 * b)) & 0xFF);
                const float x = s_lut[v];
                const float p = x * x;
                acc = acc + p;
                sb += v;
                sb2 += v * v;
            }
        };
        if (d == 256) {
            // a lane owns a row (the fold is sequential), but it asks for the row's sixteen 16-byte pieces back to back: the
            // eight requests that fall into one 128-byte line merge on their way to memory.  Byte loads at a 256-byte stride
            // fetched the table 5.8 times over (profiles/r03_scan_pmc.json: 14.96 GB for 2.56 GB of rows)
            uint4 q[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) q[j] = *reinterpret_cast<const uint4 *>(row + 16 * j);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                take(q[j].x); take(q[j].y); take(q[j].z); take(q[j].w);
            }
        } else if ((d & 15) == 0) {
            for (int i = 0; i < d; i += 16) {
                const uint4 q = *reinterpret_cast<const uint4 *>(row + i);
                take(q.x); take(q.y); take(q.z); take(q.w);
            }
        } else {
            for (int i = 0; i < d; ++i) {
                const int v = row[i];
                const float x = s_lut[v];
                const float p = x * x;
                acc = acc + p;
                sb += v;
                sb2 += v * v;
            }
        }
        norms[r] = sqrtf(acc);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt default)
        sum_b[r] = sb;                                   // exact integers for the multi-query pass
        const int32_t den = 4 * sb2 - 1020 * sb + 65025 * d;  // = sum (2b-255)^2
        den_b[r] = den;
        my_min = den < my_min ? den : my_min;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const int32_t o = __shfl_xor(my_min, off);
        my_min = o < my_min ? o : my_min;
    }
    if ((threadIdx.x & 63) == 0 && my_min != 0x7FFFFFFF) atomicMin(min_den, my_min);
}

// ------------------------------------------------------------------------------------------------
// (1) HBM-bound filter pass.  LPR lanes share one row (16 B each); one wave-instruction = 64/LPR rows.
// LOOPQ: the workgroup answers `nq_loop` queries one after the other in ONE launch -- for every query it streams its
// rows again (a full pass over the table per query, ~0.4 ms apart: nothing of the previous pass is left in any cache
// at 10M rows), but the launch gap and the ramp-up / tail of one launch per query (~12 us, a fifth of the pass over a
// 1.25M-row shard) are paid once per batch.  Its own template instance, so that profilers list it apart from the
// one-query launches.
// ARGQ (single-query calls, 256-byte rows): the query and its constants arrive as a kernel ARGUMENT (QArg256 by value, read
// through the kernarg segment) instead of being staged in device memory by a copy or a staging kernel first -- the filter
// launch is then the first command of the call; workgroup 0 parks them in slot 0 of the staged arrays for the kernels
// that follow (k_select_rescore, the fallbacks).
#ifdef PB_SCAN_STAMP
// instrumented build (profiles/scan_stamps.py): wall-clock stamps (s_memrealtime, 100 MHz) of the one-query filter launch,
// [wave of the grid][8], and of k_select_rescore, [16]
__device__ unsigned long long g_scan_stamp[F_MAX_WG * 16 * 8];
__device__ unsigned long long g_sel_stamp[16];
#define PB_STAMP(slot)                                                                                          \
    do {                                                                                                        \
        if (ARGQ && lane == 0) g_scan_stamp[((size_t)blockIdx.x * NW + wave) * 8 + (slot)] = wall_clock64();     \
    } while (0)
// (slots 0 and 7: thread 0; the others: the last thread, whose wave re-scores nothing -- a stamp is a global store, and the
// stamping wave's next wait for its loads also waits for that store)
#define PB_SEL_STAMP(slot)                                             \
    do {                                                               \
        if (blockIdx.x == 0 && threadIdx.x == (((slot) == 0 || (slot) == 7) ? 0 : SEL_BLOCK - 1)) g_sel_stamp[slot] = wall_clock64(); \
    } while (0)
#else
#define PB_STAMP(slot) do { } while (0)
#define PB_SEL_STAMP(slot) do { } while (0)
#endif
struct QArg256 {
    QParams p;
    uint8_t q[256];
};
#ifndef PB_DYN_DEN
#define PB_DYN_DEN 8
#endif
#ifndef PB_DYN_CH
#define PB_DYN_CH 4
#endif
constexpr int DYN_DEN = PB_DYN_DEN;  // k_scan_filter DYN: 1/DYN_DEN of the table is handed out dynamically
constexpr int DYN_CH = PB_DYN_CH;    // super-tiles per ticket
constexpr int DYN_REGIONS = 32;     // ticket counters
constexpr int DYN_CTR_STRIDE = 64;  // uint32 between counters (256 B: different L2 channels)
// STEAL (one-query launch, with WGT): a workgroup's own share (tickets below sg.S) is 7/8 of an equal split; the rest of the
// table is a pool, cut into DYN_REGIONS regions (one per group of 8 consecutive workgroups = one workgroup of every XCD each)
// and handed out in CHUNKS of 2^sg.shift tiles through one device-scope counter per region.  A chunk is requested by ONE
// wave for the whole workgroup, a chunk ahead of its use (the wave that draws the first ticket of chunk c - 1 asks for chunk c:
// one atomic per 8+ tiles instead of one per wave and tile, issued before that wave's own loads, so its round trip hides
// behind them), and published to the other waves through an LDS table indexed by the workgroup's chunk number -- a wave whose
// ticket falls into a chunk that is not there yet spins on LDS, never on memory.  Fast workgroups (XCDs 1 and 3 finish a
// 1M-row pass ~8 % before 0 and 6, profiles/r04_scan_stamps.txt) take more chunks.  k_select_rescore zeroes the counters.
struct StealGeo {
    uint32_t S;        // tickets of a workgroup's own share (tile = ticket * grid + workgroup)
    uint32_t shift;    // log2(tiles per chunk)
    uint32_t n_reg;    // regions of the pool
    uint32_t lead;     // a chunk is requested `lead` chunks ahead of its first ticket (1 or 2)
    unsigned long long per_reg;  // tiles per region (a multiple of the chunk)
};
constexpr int ST_MAXC = 256;  // chunks a workgroup can take (the host picks the chunk size so that a region has fewer)
// WGT (with LOOPQ): the waves of a workgroup take the workgroup's super-tiles by tickets from an LDS counter instead of
// fixed strides, and the per-wave buffers are double-buffered by query parity.  The workgroup list of query i is merged
// by wave 0 AFTER the barrier that ends query i while the other waves are already streaming query i + 1 -- wave 0 simply
// takes fewer tickets of that query -- so the merge (and the second barrier that protected the buffers) leaves the
// critical path: it is ~1 % of a pass over 10M rows but ~5 % of one over a 1.25M-row shard (8-GPU strong scaling).
// FUSE (round 6; the one-query launch with WGT): ONE launch per one-query call.  A workgroup stores its list and header write-through
// (sc1), drains them, and takes a number at a device-scope arrival counter; the workgroup whose number is the last runs the
// candidate selection, exact re-scoring and certificate (select_rescore_body<512, true>) itself, reading the others' lists with
// sc1 loads -- the hand-off form "the workgroup whose add came last" of the MI355X guide (one workgroup per CU, 4- and 8-byte sc1
// stores, drained before the add; the adding wave loads after its add has returned, the other waves of its workgroup after an
// LDS word it then sets).  Waves that finished early wait on that LDS word (s_final) to learn whether their workgroup is the one.
// Against filter launch + k_select_rescore launch: one kernel launch of host time and the launch boundary less per call.
struct SelArgs;
template <int BLK, bool FUSED>
__device__ __forceinline__ void select_rescore_body(const SelArgs &A, const int q, const QParams &P, const uint8_t *__restrict__ qbytes);
struct FuseArgs {  // (a copy of SelArgs' fields: SelArgs is complete only further down)
    const int64_t *ids;
    const float *norms;
    const float *lut;
    int64_t *out_ids;
    float *out_dist;
    ResultHdr *out_hdr;
    uint32_t *done_flag;
    uint32_t *arrive;  // the arrival counter (zero between launches: the last workgroup resets it)
    uint32_t out_stride, done_seq, tile_rows;
};
template <bool FUSE>
__device__ __forceinline__ void fused_select(const uint8_t *rows, uint64_t n_rows, const uint64_t *lists, const ListHdr *hdrs, uint32_t *tail_ctr,
                                             const QArg256 &qarg, const FuseArgs &fa);

template <int LPR, int U = 8, bool NT = true, int NW = F_WAVES, int MAPB = 0, bool LOOPQ = false, bool ARGQ = false, bool DYN = false,
          bool WGT = false, int HS = 0, bool STEAL = false, bool FUSE = false>
__global__ __launch_bounds__(NW * WAVE) void k_scan_filter(const uint8_t *__restrict__ rows, uint64_t n_rows,
                                                         const uint8_t *queries, const QParams *qp,
                                                         uint64_t *__restrict__ lists,
                                                         ListHdr *__restrict__ hdrs, int q_base, int nq_loop,
                                                         uint8_t *stage_q, QParams *stage_p, const QArg256 qarg,
                                                         uint32_t *tail_ctr = nullptr, const StealGeo sg = StealGeo{},
                                                         const FuseArgs fa = FuseArgs{}) {
    static_assert(!FUSE || (ARGQ && WGT && NW == 8), "FUSE: the one-query launch with workgroup tickets");
    static_assert(!ARGQ || (LPR == 16 && !LOOPQ), "ARGQ: one 256-byte query per launch");
    static_assert(!STEAL || (WGT && ARGQ), "STEAL: the one-query launch with workgroup tickets");
    static_assert(!DYN || (ARGQ && MAPB == 0), "DYN: the one-query launch only");
    static_assert(!WGT || ((LOOPQ || ARGQ) && MAPB == 0 && !DYN), "WGT: the looped launch, or the one-query launch with static shares");
    constexpr int D = LPR * 16;
    constexpr int RPT = WAVE / LPR;                // rows per wave-instruction
    // U = loads in flight per lane (U KiB per wave); NT = non-temporal loads (the table is streamed once per query)
    constexpr int ROWS_IT = U * RPT;               // rows per wave-iteration
    constexpr int ROUNDS = (U + LPR - 1) / LPR;    // evaluation rounds (one row per lane each)
    constexpr int NPAR = WGT ? 2 : 1;
    __shared__ uint64_t s_buf[NPAR][NW][F_CAPW];
    __shared__ float s_drop[NPAR][NW];
    __shared__ uint32_t s_seen[NPAR][NW];
    __shared__ unsigned long long s_sig[NPAR][NW];
    __shared__ uint32_t s_ticket[2];
    __shared__ uint32_t s_node[8];  // one-query launch: arrival counters of the list-merging tree (4 pairs, 2 quads, 1 root)
    __shared__ uint32_t s_final;    // FUSE: 0 not known yet, 1 another workgroup arrives last, 2 this one does: its waves run the selection
    __shared__ uint32_t s_turn;     // STEAL: the chunk number whose request may go out next
    __shared__ uint32_t s_chunk[STEAL ? ST_MAXC : 1];  // STEAL: 1 + the region counter's answer for the workgroup's c-th chunk (0: not there yet)

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: loop bounds and LDS bases stay in SGPRs
    PB_STAMP(0);
    const int sub = lane % LPR;
    const int g = lane / LPR;
    if constexpr (WGT) {
        if (threadIdx.x < 2) s_ticket[threadIdx.x] = 0u;
        if (threadIdx.x < 8) s_node[threadIdx.x] = 0u;
        if (threadIdx.x == 8) s_turn = 0u;
        if (threadIdx.x == 9) s_final = 0u;
        if constexpr (STEAL)
            for (int i = threadIdx.x; i < ST_MAXC; i += NW * WAVE) s_chunk[i] = 0u;
        __syncthreads();
    }
  for (int qi = 0; qi < (LOOPQ ? nq_loop : 1); ++qi) {
    const int par = WGT ? (qi & 1) : 0;
    uint64_t *buf = s_buf[par][wave];
    // the other parity's counter served query qi - 1 (every wave is past that query's barrier) and serves qi + 1
    if (WGT && threadIdx.x == 0) s_ticket[par ^ 1] = 0u;
    const int q = q_base + (LOOPQ ? qi : (int)blockIdx.y);
    QParams P;
    uint4 qv;
    if constexpr (ARGQ) {
        P = qarg.p;
        qv = *reinterpret_cast<const uint4 *>(qarg.q + sub * 16);
        if (blockIdx.x == 0 && threadIdx.x < 16) {
            reinterpret_cast<uint4 *>(stage_q)[threadIdx.x] = *reinterpret_cast<const uint4 *>(qarg.q + threadIdx.x * 16);
            if (threadIdx.x == 0) stage_p[0] = qarg.p;
        }
    } else {
        P = qp[q];
        qv = *reinterpret_cast<const uint4 *>(queries + (size_t)q * D + sub * 16);
    }

    float thr = P.thr0;
    float dropped = 0.0f;
    int cnt = 0;
    uint32_t rows_seen = 0;  // scalar: rows of the tiles this wave evaluated
    unsigned long long tile_sig = 0;  // scalar: sum of (tile number + 1)^2 over them, mod 2^64 (ListHdr::sig_*)
    const int k_num = 65025 * D - 510 * P.sum_a;  // num = 4P - 510*S + k_num
    const int k_den = 65025 * D;

    const uint64_t n_super = (n_rows + ROWS_IT - 1) / ROWS_IT;
    const uint64_t stride = (uint64_t)gridDim.x * NW;
    // super-tile -> wave mapping: MAPB = 0: workgroup fastest (adjacent 8-KiB pieces on different CUs),
    // 1: wave fastest (a workgroup reads NW adjacent pieces)
    const uint64_t first = MAPB ? (uint64_t)blockIdx.x * NW + wave : (uint64_t)wave * gridDim.x + blockIdx.x;
    // DYN (one-query call): the last 1/DYN_DEN of the table is handed out in DYN_CH-tile chunks through DYN_REGIONS
    // ticket counters (one per group of 8 consecutive workgroups = one per XCD each), so that a wave which finished its
    // static share early takes more of the tail instead of idling; the ticket for the next chunk is requested before
    // the current chunk's loads, so its latency hides behind them.  k_select_rescore zeroes the counters afterwards.
    uint64_t s = first;
    uint64_t n_static = n_super, dyn_lo = 0, dyn_hi = 0, dyn_cur = 0;
    uint32_t dyn_left = 0, dyn_pend = 0;
    bool dyn_on = false, dyn_done = false, dyn_req = false;
    uint32_t *dyn_ctr = nullptr;
    if constexpr (DYN) {
        n_static = (n_super - n_super / DYN_DEN) / stride * stride;
        const uint32_t groups = (gridDim.x + 7) >> 3;
        const uint32_t n_reg = groups < (uint32_t)DYN_REGIONS ? groups : (uint32_t)DYN_REGIONS;  // every region has takers
        const uint64_t per_reg = ((n_super - n_static + n_reg - 1) / n_reg + DYN_CH - 1) / DYN_CH * DYN_CH;
        const uint32_t reg = (blockIdx.x >> 3) % n_reg;
        dyn_lo = n_static + (uint64_t)reg * per_reg;
        dyn_hi = dyn_lo + per_reg < n_super ? dyn_lo + per_reg : n_super;
        dyn_ctr = tail_ctr + reg * DYN_CTR_STRIDE;
        // the first ticket is requested when the wave has ONE static tile left (below), not here: 2 048 waves asking 32
        // counters at kernel start queue up for ~4.5 us, which a pass over a small table never earns back
        if (s >= n_static || s + stride >= n_static) {
            if (lane == 0) dyn_pend = atomicAdd(dyn_ctr, 1u);
            dyn_req = true;
        }
    }
    auto next_dyn = [&]() -> uint64_t {
        if (dyn_left) {
            --dyn_left;
            return ++dyn_cur;
        }
        if (dyn_done) return ~0ull;
        const uint32_t t = __builtin_amdgcn_readfirstlane(dyn_pend);
        const uint64_t s0 = dyn_lo + (uint64_t)t * DYN_CH;
        if (s0 >= dyn_hi) {
            dyn_done = true;
            return ~0ull;
        }
        if (lane == 0) dyn_pend = atomicAdd(dyn_ctr, 1u);
        dyn_cur = s0;
        const uint64_t rest = dyn_hi - s0 - 1;
        dyn_left = rest < (uint64_t)(DYN_CH - 1) ? (uint32_t)rest : (uint32_t)(DYN_CH - 1);
        return s0;
    };
    if constexpr (DYN) {
        if (s >= n_static) {
            dyn_on = true;
            s = next_dyn();
        }
    }
    uint32_t wg_pend = 0;
    // STEAL state: the region of this workgroup, and the chunk request this wave has in flight (st_rq, for chunk st_rc)
    uint64_t st_lo = 0, st_hi = 0;
    uint32_t *st_ctr = nullptr;
    uint32_t st_rq = 0, st_rc = 0;
    bool st_req = false;
    // ticket -> tile; asks for a chunk on the way if this ticket is the one that does (see above)
    auto st_resolve = [&](uint32_t t) -> uint64_t {
        const uint32_t cht = 1u << sg.shift, from = sg.S - sg.lead * cht;
        st_req = t >= from && ((t - from) & (cht - 1u)) == 0u && ((t - from) >> sg.shift) < (uint32_t)ST_MAXC;
        if (st_req) {
            st_rc = (t - from) >> sg.shift;
            // Requests of a workgroup go out ONE AT A TIME, in chunk order: the request for chunk c waits until chunk c - 1 has
            // been published (s_turn).  The answers of a region's counter then grow with c, so the first chunk that lies
            // past the region's end is followed by nothing valid -- which is what lets a wave leave at its first invalid
            // tile.  (Tickets are drawn in order but RESOLVED in any order; requests issued straight from the trigger tickets
            // could overtake each other, chunk c come back past the end and chunk c + 1 not, and its tiles were never
            // read: 2-3 % of calls over 2.1M rows missed a row that way before this wait was there.)
            while (__hip_atomic_load(&s_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != st_rc) __builtin_amdgcn_s_sleep(1);
            if (lane == 0) st_rq = atomicAdd(st_ctr, 1u);
        }
        if (t < sg.S) return (uint64_t)t * gridDim.x + blockIdx.x;
        const uint32_t j = t - sg.S, c = j >> sg.shift;
        if (c >= (uint32_t)ST_MAXC) return ~0ull;  // past any region's last chunk (host geometry)
        uint32_t e;
        while ((e = __hip_atomic_load(&s_chunk[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0u) __builtin_amdgcn_s_sleep(1);
        const uint64_t tile = st_lo + ((uint64_t)(e - 1u) << sg.shift) + (j & (cht - 1u));
        return tile < st_hi ? tile : ~0ull;
    };
    auto st_publish = [&]() {
        if (st_req) {
            const uint32_t v = __builtin_amdgcn_readfirstlane(st_rq);
            if (lane == 0) {
                __hip_atomic_store(&s_chunk[st_rc], v + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&s_turn, st_rc + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // the answer is in: next request
            }
            st_req = false;
        }
    };
    if constexpr (WGT) {
        if (lane == 0) wg_pend = atomicAdd(&s_ticket[par], 1u);
        if constexpr (STEAL) {
            const uint32_t reg = (blockIdx.x >> 3) % sg.n_reg;
            st_lo = (uint64_t)sg.S * gridDim.x + (uint64_t)reg * sg.per_reg;
            st_hi = st_lo + sg.per_reg < n_super ? st_lo + sg.per_reg : n_super;
            st_ctr = tail_ctr + reg * DYN_CTR_STRIDE;
            s = st_resolve((uint32_t)__builtin_amdgcn_readfirstlane(wg_pend));
            if (s == ~0ull) st_publish();
        } else {
            s = (uint64_t)__builtin_amdgcn_readfirstlane(wg_pend) * gridDim.x + blockIdx.x;
        }
    }
    PB_STAMP(1);
#ifdef PB_SCAN_STAMP
    bool first_tile = true;
#endif
    for (; (DYN || STEAL) ? (s != ~0ull) : (s < n_super);) {
        if constexpr (WGT) {
            if (lane == 0) wg_pend = atomicAdd(&s_ticket[par], 1u);  // the next ticket, requested ahead of this tile's loads
        }
        uint64_t s_eval = s;  // the tile this turn evaluates (= the tile it was handed, unless a fault is injected)
#ifdef PB_FAULT_DOUBLE_TILE  // fault injection: tile PB_FAULT_DOUBLE_TILE + 1 is never read, tile PB_FAULT_DOUBLE_TILE is read twice -- the row
        if (s_eval == (uint64_t)(PB_FAULT_DOUBLE_TILE) + 1) s_eval = (uint64_t)(PB_FAULT_DOUBLE_TILE);  // counts still add up to the table; the sums of squares do not
#endif
        const uint64_t row0 = s_eval * ROWS_IT;
#ifdef PB_FAULT_SKIP_TILE  // fault injection (profiles/r04_scan_stamps.txt): one tile is counted as nobody's
        if (s != (uint64_t)PB_FAULT_SKIP_TILE)
#endif
        rows_seen += row0 < n_rows ? (uint32_t)(n_rows - row0 < (uint64_t)ROWS_IT ? n_rows - row0 : (uint64_t)ROWS_IT) : 0u;
        if (row0 < n_rows) tile_sig += (unsigned long long)(s_eval + 1) * (unsigned long long)(s_eval + 1);
        uint4 b[U];
        auto load_row = [&](int u) {
            uint64_t r = row0 + (uint64_t)(u * RPT + g);
            r = r < n_rows ? r : n_rows - 1;
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 *src = reinterpret_cast<const u32x4 *>(rows + r * D + sub * 16);
            u32x4 t;
            if constexpr (NT) t = __builtin_nontemporal_load(src);
            else t = *src;
            b[u] = make_uint4(t.x, t.y, t.z, t.w);
        };
        // HS > 0: HS loads leave first, each of the others behind the evaluation of the load HS places before it (pinned with
        // scheduling barriers); 0: all U requested in source order and hipcc places them.  Over a table that streams from HBM
        // four-then-one-by-one is the fastest (64 passes over 10M rows, one box, TB/s: 2: 5.74, 3: 6.64, 4: 7.17, 5: 7.13, 6: 7.09,
        // all 8 at once: 7.04; hipcc's own placement was 4 + 4 in round 3 and became 6 + 2 when the code after the loop changed:
        // 7.17 -> 7.09); over a table that sits in the Infinity Cache more in flight is better (1M rows: 49.7 us with 4, 47.5 left
        // to hipcc) -- so the instances for large tables pin 4 and the one-query launch for small tables does not
        constexpr int H = (HS > 0 && HS < U) ? HS : U;
#pragma unroll
        for (int u = 0; u < H; ++u) load_row(u);
        if constexpr (STEAL) {  // the chunk this wave asked for (before these loads) goes to the workgroup's table
            __builtin_amdgcn_sched_barrier(0);
            st_publish();
        }
        if constexpr (H < U) __builtin_amdgcn_sched_barrier(0);
        int sp[ROUNDS], ss[ROUNDS], sq[ROUNDS];
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) sp[rd] = ss[rd] = sq[rd] = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t p = dot4(b[u].x, qv.x, 0), sm = dot4(b[u].x, 0x01010101u, 0), s2 = dot4(b[u].x, b[u].x, 0);
            p = dot4(b[u].y, qv.y, p), sm = dot4(b[u].y, 0x01010101u, sm), s2 = dot4(b[u].y, b[u].y, s2);
            p = dot4(b[u].z, qv.z, p), sm = dot4(b[u].z, 0x01010101u, sm), s2 = dot4(b[u].z, b[u].z, s2);
            p = dot4(b[u].w, qv.w, p), sm = dot4(b[u].w, 0x01010101u, sm), s2 = dot4(b[u].w, b[u].w, s2);
            const int tp = group_sum<LPR>((int)p), ts = group_sum<LPR>((int)sm), tq = group_sum<LPR>((int)s2);
            const bool mine = (u % LPR) == sub;
            sp[u / LPR] = mine ? tp : sp[u / LPR];
            ss[u / LPR] = mine ? ts : ss[u / LPR];
            sq[u / LPR] = mine ? tq : sq[u / LPR];
            if constexpr (H < U) {
                if (u + H < U) load_row(u + H);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int u = rd * LPR + sub;
            const uint64_t r = row0 + (uint64_t)(u * RPT + g);
            const bool valid = (u < U) && (r < n_rows);
            const int num = 4 * sp[rd] - 510 * ss[rd] + k_num;
            const int den_b = 4 * sq[rd] - 1020 * ss[rd] + k_den;  // = sum (2b-255)^2 >= D
            float cs = (float)num * __builtin_amdgcn_rsqf((float)den_b * P.den_a);
            cs = valid ? cs : -1.0f;
            const bool pass = cs >= thr;
            const uint64_t m = __ballot(pass);
            if (m) {
                if (pass) buf[cnt + mbcnt(m)] = filter_key(cs, (uint32_t)r);
                cnt += __popcll(m);
                if (cnt > F_CAPW - WAVE) {
                    const uint64_t kth = wave_keep_smallest<F_CAPW / WAVE>(buf, cnt, F_KW);
                    cnt = F_KW;
                    thr = filter_key_cos(kth);
                    dropped = thr;
                }
            }
        }
#ifdef PB_SCAN_STAMP
        if (first_tile) {
            PB_STAMP(2);
            first_tile = false;
        }
#endif
        if constexpr (DYN) {
            if (!dyn_on) {
                s += stride;
                if (!dyn_req && s + stride >= n_static) {  // the tile about to be loaded is the last static one (or past it)
                    if (lane == 0) dyn_pend = atomicAdd(dyn_ctr, 1u);
                    dyn_req = true;
                }
                if (s >= n_static) {
                    dyn_on = true;
                    s = next_dyn();
                }
            } else {
                s = next_dyn();
            }
        } else if constexpr (STEAL) {
            s = st_resolve((uint32_t)__builtin_amdgcn_readfirstlane(wg_pend));
            if (s == ~0ull) st_publish();  // leaving: the chunk it asked for on the way out is still owed to the others
        } else if constexpr (WGT) {
            s = (uint64_t)__builtin_amdgcn_readfirstlane(wg_pend) * gridDim.x + blockIdx.x;
        } else {
            s += stride;
        }
    }
    PB_STAMP(3);
    {   // the wave's list: its F_KW best, sorted (cnt <= 64 here: the loop prunes above that)
        const uint64_t first_out = wave_finish_list(buf, cnt);
        if (first_out != ~0ull) dropped = filter_key_cos(first_out);  // >= thr: everything buffered had passed it
    }
    if constexpr (ARGQ && WGT && NW == 8) {
        // One query per launch: no barrier and no single merging wave -- an arrival tree.  A wave leaves its sorted list and
        // its bound in LDS and takes a number at its pair's node; the SECOND to arrive merges the two lists (one bitonic
        // merge, 0.25 us) and carries the result to the next node.  Behind the workgroup's last wave there are its own
        // sort and three merges instead of a barrier and seven (stamps: 2.1 -> ~1.2 us).
        int slot = wave;          // where the list this wave carries sits in s_buf
        float carried = dropped;  // the bound that goes with it
        uint32_t seen = rows_seen;  // and the rows behind it
        unsigned long long sig = tile_sig;
        uint64_t key = ~0ull;
        bool last = true;
#pragma unroll
        for (int lv = 0; lv < 3; ++lv) {
            if (lane == 0) {
                s_drop[par][slot] = carried;
                s_seen[par][slot] = seen;
                s_sig[par][slot] = sig;
            }
            uint32_t pos = 0;
            // release: this wave's list and bound are in LDS before its number is taken; acquire: the partner's are read after
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0)
                pos = __hip_atomic_fetch_add(&s_node[(lv == 0 ? 0 : (lv == 1 ? 4 : 6)) + (wave >> (lv + 1))], 1u, __ATOMIC_ACQ_REL,
                                             __HIP_MEMORY_SCOPE_WORKGROUP);
            pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (pos == 0u) {  // the partner is still streaming: it will find this list here
                last = false;
                break;
            }
            const int lo = (wave >> (lv + 1)) << (lv + 1), hi = lo + (1 << lv);
            key = wave_merge64(lane < F_KW ? s_buf[par][lo][lane] : s_buf[par][hi][63 - lane]);
            const uint64_t first_out = __shfl((unsigned long long)key, F_KW);
            carried = fmaxf(s_drop[par][lo], s_drop[par][hi]);
            seen = s_seen[par][lo] + s_seen[par][hi];
            sig = s_sig[par][lo] + s_sig[par][hi];
            if (first_out != ~0ull) carried = fmaxf(carried, filter_key_cos(first_out));
            if (lv < 2 && lane < F_KW) s_buf[par][lo][lane] = key;
            slot = lo;
        }
        PB_STAMP(4);
        if (last) {
            const int total = __popcll(__ballot(lane < F_KWG && key != ~0ull));
            uint64_t *out = lists + ((size_t)q * gridDim.x + blockIdx.x) * F_KWG;
            if constexpr (FUSE) {
                // write-through stores (the reader is another CU of this launch), drained before the arrival number is taken
                if (lane < F_KWG) __hip_atomic_store(out + lane, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t *h32 = reinterpret_cast<uint32_t *>(hdrs + ((size_t)q * gridDim.x + blockIdx.x));
                if (lane < 5) {
                    const uint32_t v = lane == 0 ? (uint32_t)total : (lane == 1 ? __float_as_uint(carried) : (lane == 2 ? seen : (lane == 3 ? (uint32_t)sig : (uint32_t)(sig >> 32))));
                    __hip_atomic_store(h32 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t pos = 0;
                if (lane == 0) pos = __hip_atomic_fetch_add(fa.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
                if (lane == 0) __hip_atomic_store(&s_final, pos + 1u == gridDim.x ? 2u : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
            if (lane < F_KWG) out[lane] = key;  // every slot: the unused ones hold ~0, which is how k_select_rescore tells them
            if (lane == 0) {
                ListHdr h;
                h.count = (uint32_t)total;
                h.dropped = carried;
                h.rows_seen = seen;
                h.sig_lo = (uint32_t)sig;
                h.sig_hi = (uint32_t)(sig >> 32);
                hdrs[(size_t)q * gridDim.x + blockIdx.x] = h;
            }
            }
            PB_STAMP(6);
        }
        if constexpr (FUSE) {
            uint32_t v;
            while ((v = __hip_atomic_load(&s_final, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0u) __builtin_amdgcn_s_sleep(2);
            if (v == 2u) fused_select<FUSE>(rows, n_rows, lists, hdrs, tail_ctr, qarg, fa);  // wave-uniform over the whole workgroup
        }
    } else {
    if (lane == 0) {
        s_drop[par][wave] = dropped;
        s_seen[par][wave] = rows_seen;
        s_sig[par][wave] = tile_sig;
    }
    PB_STAMP(4);
    __syncthreads();
    PB_STAMP(5);
    if (wave == 0) {
        // workgroup list: the F_KWG best of the NW sorted wave lists, by wave 0 (NW - 1 bitonic merges of 32 + 32 keys)
        int total;
        uint64_t first_out;
        const uint64_t mykey = wave_merge_lists<NW>(&s_buf[par][0][0], F_CAPW, &total, &first_out);
        float drop = lane < NW ? s_drop[par][lane] : 0.0f;
        for (int off = 8; off >= 1; off >>= 1) drop = fmaxf(drop, __shfl_xor(drop, off));
        drop = __shfl(drop, 0);
        if (first_out != ~0ull) drop = fmaxf(drop, filter_key_cos(first_out));
        uint64_t *out = lists + ((size_t)q * gridDim.x + blockIdx.x) * F_KWG;
        if (lane < F_KWG) out[lane] = mykey;  // every slot: the unused ones hold ~0, which is how k_select_rescore tells them
        if (lane == 0) {
            ListHdr h;
            h.count = (uint32_t)total;
            h.dropped = drop;
            uint32_t seen = 0;
            unsigned long long sig = 0;
            for (int w = 0; w < NW; ++w) {
                seen += s_seen[par][w];
                sig += s_sig[par][w];
            }
            h.rows_seen = seen;
            h.sig_lo = (uint32_t)sig;
            h.sig_hi = (uint32_t)(sig >> 32);
            hdrs[(size_t)q * gridDim.x + blockIdx.x] = h;
        }
        PB_STAMP(6);
    }
    }
    if constexpr (LOOPQ && !WGT) __syncthreads();  // the wave buffers are reused by the next query (WGT: the other parity's are)
  }
}

// ------------------------------------------------------------------------------------------------
// (1c) byte_distance / hamming_distance (engine.rs:590-604) as a coalesced, HBM-bound pass over a `phashes`-like
// table: the same streaming skeleton as k_scan_filter (LPR lanes share a row, 16 B each, U loads in flight), with
// v_sad_u8 / popcount as the inner op.  These distances are EXACT in the integer domain (a sum below 2^24 and one
// correctly rounded divide -- the u8 wrap of hamming included), so the keys (dist bits, row) are final: no
// re-scoring.  A wave keeps its F_KW best keys and the best key it dropped; k_select_keys merges the workgroup
// lists and the result is the reference's iff its k-th key is smaller than every dropped key (else: exhaustive).
template <int LPR, int METRIC, int U = 8, int NW = F_WAVES>
__global__ __launch_bounds__(NW * WAVE) void k_scan_dist(const uint8_t *__restrict__ rows, uint64_t n_rows,
                                                       const uint8_t *__restrict__ queries,
                                                       const QParams *__restrict__ qp, uint64_t *__restrict__ lists,
                                                       ListHdr *__restrict__ hdrs, uint64_t *__restrict__ drop_keys,
                                                       int q_base, int nq_loop) {
    constexpr int D = LPR * 16;
    constexpr int RPT = WAVE / LPR;
    constexpr int ROWS_IT = U * RPT;
    constexpr int ROUNDS = (U + LPR - 1) / LPR;
    __shared__ uint64_t s_buf[NW][F_CAPW];
    __shared__ uint64_t s_drop[NW];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: loop bounds and LDS bases stay in SGPRs
    const int sub = lane % LPR;
    const int g = lane / LPR;
    uint64_t *buf = s_buf[wave];
  // nq_loop queries one after the other in this launch (one table pass each), as k_scan_filter's LOOPQ form
  for (int qi = 0; qi < nq_loop; ++qi) {
    const int q = q_base + qi;
    const QParams P = qp[q];
    const uint4 qv = *reinterpret_cast<const uint4 *>(queries + (size_t)q * D + sub * 16);
    uint64_t thr_key = ~0ull;  // keep keys < thr_key
    uint64_t dropped = ~0ull;  // smallest key this wave dropped
    int cnt = 0;
    const float denom = METRIC == 1 ? 255.0f * (float)D : 8.0f * (float)D;
    const uint64_t n_super = (n_rows + ROWS_IT - 1) / ROWS_IT;
    const uint64_t stride = (uint64_t)gridDim.x * NW;
    for (uint64_t s = (uint64_t)wave * gridDim.x + blockIdx.x; s < n_super; s += stride) {
        const uint64_t row0 = s * ROWS_IT;
        uint4 b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t r = row0 + (uint64_t)(u * RPT + g);
            r = r < n_rows ? r : n_rows - 1;
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows + r * D + sub * 16));
            b[u] = make_uint4(t.x, t.y, t.z, t.w);
        }
        int sv[ROUNDS];
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) sv[rd] = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t a;
            if constexpr (METRIC == 1) {
                a = __builtin_amdgcn_sad_u8(qv.x, b[u].x, 0u);
                a = __builtin_amdgcn_sad_u8(qv.y, b[u].y, a);
                a = __builtin_amdgcn_sad_u8(qv.z, b[u].z, a);
                a = __builtin_amdgcn_sad_u8(qv.w, b[u].w, a);
            } else {
                a = __popc(qv.x ^ b[u].x) + __popc(qv.y ^ b[u].y) + __popc(qv.z ^ b[u].z) + __popc(qv.w ^ b[u].w);
            }
            const int tot = group_sum<LPR>((int)a);
            sv[u / LPR] = ((u % LPR) == sub) ? tot : sv[u / LPR];
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int u = rd * LPR + sub;
            const uint64_t r = row0 + (uint64_t)(u * RPT + g);
            const bool valid = (u < U) && (r < n_rows);
            const uint32_t acc = (uint32_t)sv[rd];
            const float dist = METRIC == 1 ? (float)acc / denom : (float)(acc & 0xFFu) / denom;  // engine.rs:592 / :603
            uint64_t key = ~0ull;
            if (valid && (double)dist < P.max_dist) key = ((uint64_t)sortable_f32(dist) << 32) | (uint32_t)r;
            const bool pass = key < thr_key;
            const uint64_t m = __ballot(pass);
            if (m) {
                if (pass) buf[cnt + mbcnt(m)] = key;
                cnt += __popcll(m);
                if (cnt > F_CAPW - WAVE) {
                    thr_key = wave_keep_smallest<F_CAPW / WAVE>(buf, cnt, F_KW);  // = the largest key kept
                    cnt = F_KW;
                    dropped = thr_key + 1 < dropped ? thr_key + 1 : dropped;  // everything dropped is > thr_key
                    thr_key = thr_key + 1;                                      // keep only keys <= the kept maximum
                }
            }
        }
    }
    {   // the wave's list: its F_KW best, sorted (cnt <= 64 here)
        const uint64_t first_out = wave_finish_list(buf, cnt);
        dropped = first_out < dropped ? first_out : dropped;
    }
    if (lane == 0) s_drop[wave] = dropped;
    __syncthreads();
    if (wave == 0) {
    int total;
    uint64_t first_out;
    const uint64_t mykey = wave_merge_lists<NW>(&s_buf[0][0], F_CAPW, &total, &first_out);
    uint64_t drop = first_out;
    for (int w = 0; w < NW; ++w) drop = s_drop[w] < drop ? s_drop[w] : drop;
    uint64_t *out = lists + ((size_t)q * gridDim.x + blockIdx.x) * F_KWG;
    if (lane < total) out[lane] = mykey;
    if (lane == 0) {
        ListHdr h;
        h.count = (uint32_t)total;
        h.dropped = 0.0f;
        h.rows_seen = 0u;
        h.sig_lo = h.sig_hi = 0u;
        hdrs[(size_t)q * gridDim.x + blockIdx.x] = h;
        drop_keys[(size_t)q * gridDim.x + blockIdx.x] = drop;  // every key this workgroup saw and did not list is >= drop
    }
    }
    if (qi + 1 < nq_loop) __syncthreads();  // the wave buffers are reused by the next query
  }
}

// ------------------------------------------------------------------------------------------------
// in-LDS bitonic sort of n (power of two) u64 keys, ascending; all threads of the block participate
__device__ __forceinline__ void block_bitonic_sort(uint64_t *s, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = s[i], b = s[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// (1b) candidate selection from the workgroup lists, exact re-scoring, certificate.  One block per query.
// This kernel is the latency tail of every query (one workgroup, a chain of dependent phases), so it is written to make
// as few round trips as it can: every thread fetches its share of the list slots ONCE (all loads in flight together; an
// unused slot holds ~0) and serves both the lower bound (the k-th largest list head, by counting) and the candidate selection
// from registers; a candidate's row, norm and image_id are requested together, its 256 look-ups and products are made by eight
// lanes (ref_fold_dot256_by8); and up to SEL_RANK_MAX candidates are ordered by counting (each candidate counts the keys
// below its own from LDS: one barrier) instead of a 36-barrier bitonic network -- 29 us in round 1, 13 us now
// (profiles/r04_scan_stamps.txt).
// One-query calls answered into pinned host memory (done_flag != nullptr) publish their results as 16-byte GRANULES
// {payload[3], tag}: one store instruction of one lane each, so a granule arrives whole, and the host accepts a granule once
// its tag (fourth word ^ granule_mix(payload), below) equals the call's sequence number -- no fence between the result stores and a
// separate completion flag (which cost the store round trip over PCIe, ~2 us at the end of every query), no flag.  Granule 0:
// {count, status, n_cand}, 1: {o_max, ck, 0}, 2 + i: {id low, id high, distance} of result i.
// A granule is SELF-VALIDATING: its fourth word is tag ^ granule_mix(payload), so a reader that sees the words of two
// different stores in one granule -- the store split on its way, or the tag word landing first, neither of which the architecture
// rules out even though a 16-byte store of one lane has been observed to arrive whole on gfx950 -- finds a word that is not its
// call's sequence number and keeps polling; stale payload under a fresh tag passes with probability 2^-32.  The host takes one
// snapshot of the four words, checks it, and uses the payload of THAT snapshot (search_chunk, pb_scan.hip).
__host__ __device__ __forceinline__ uint32_t granule_mix(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u;
    h ^= (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h = (h << 13) | (h >> 19);
    h ^= (c + 0x165667B1u) * 0xC2B2AE35u;
    return h ^ (h >> 16);
}
__device__ __forceinline__ void sel_put_granule(uint32_t *base, uint32_t slot, uint32_t a, uint32_t b, uint32_t c, uint32_t tag) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 g;
    g.x = a; g.y = b; g.z = c; g.w = tag ^ granule_mix(a, b, c);
    *reinterpret_cast<u32x4 *>(base + 4 * (size_t)slot) = g;
}
constexpr int SEL_RANK_MAX = 256;
// Everything k_select_rescore reads and writes besides the query (which comes from memory there and from the kernel argument in the
// filter launch's own last workgroup: k_scan_filter FUSE).
struct SelArgs {
    const uint8_t *rows;
    const int64_t *ids;
    const float *norms;
    const float *lut;
    const uint64_t *lists;
    const ListHdr *hdrs;
    int64_t *out_ids;
    float *out_dist;
    ResultHdr *out_hdr;
    uint32_t *tail_ctr;
    uint32_t *done_flag;
    int d, n_lists;
    uint32_t out_stride, done_seq, n_rows, tile_rows;
};
// BLK threads (1 024: the kernel of its own; 512: the filter launch's last workgroup).  FUSED: the lists and headers were written by
// OTHER workgroups of the same launch with write-through (sc1) stores, drained before each workgroup took its arrival number, and
// are read here with sc1 loads (MI355X guide, inter-workgroup hand-offs: "the workgroup whose add came last"); the thread count
// bounds what the body can take -- the k-th-largest count over n_lists * ceil(k / n_lists) <= BLK list heads (the host checks before
// it picks this form) and one candidate per thread (more than BLK candidates: status 2, the host runs the kernel of its own).
template <int BLK, bool FUSED>
__device__ __forceinline__ void select_rescore_body(const SelArgs &A, const int q, const QParams &P, const uint8_t *__restrict__ qbytes) {
    constexpr int SLOTS = F_MAX_WG * F_KWG / BLK;  // list slots per thread
    const uint8_t *__restrict__ rows = A.rows;
    const int64_t *__restrict__ ids = A.ids;
    const float *__restrict__ norms = A.norms;
    const int d = A.d, n_lists = A.n_lists;
    int64_t *__restrict__ out_ids = A.out_ids;
    float *__restrict__ out_dist = A.out_dist;
    uint32_t *done_flag = A.done_flag;
    const uint32_t done_seq = A.done_seq, out_stride = A.out_stride;
    // the filter launch handed out its tail through these counters (k_scan_filter DYN / STEAL): clear them
    if (A.tail_ctr && (FUSED || blockIdx.x == 0) && threadIdx.x < DYN_REGIONS) A.tail_ctr[threadIdx.x * DYN_CTR_STRIDE] = 0u;
    __shared__ float s_lut[256];
    __shared__ float s_qf[1024];
    __shared__ uint8_t s_qf_late[BLK < 1024 ? 1024 - BLK : 1];
    __shared__ __attribute__((aligned(16))) float s_top[SEL_BINS];  // the first j_top filter cosines of every list (-1: unused slot)
    __shared__ __attribute__((aligned(16))) uint64_t s_key[SEL_MAX_CAND];   // candidates (filter keys), then exact keys
    __shared__ float s_red[BLK / WAVE];
    __shared__ float s_ck[BLK / WAVE];
    __shared__ uint32_t s_u[8];
    __shared__ unsigned long long s_sig64;
    const int tid = threadIdx.x;
    PB_SEL_STAMP(0);
    const uint64_t *ql = A.lists + (size_t)q * n_lists * F_KWG;
    const ListHdr *qh = A.hdrs + (size_t)q * n_lists;
    auto ld_key = [&](int i) -> uint64_t {
        if constexpr (FUSED) return __hip_atomic_load(ql + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return ql[i];
    };
    auto ld_u32 = [&](const uint32_t *p2) -> uint32_t {
        if constexpr (FUSED) return __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *p2;
    };

    // ---- one round trip: this thread's list slots (slot i = list i / 32, entry i % 32; an unused slot holds ~0) and, on the
    //      first n_lists threads, a list's `dropped` bound
    const int total_slots = n_lists * F_KWG;
    uint64_t my_key[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int i = tid + j * BLK;
        my_key[j] = i < total_slots ? ld_key(i) : ~0ull;
    }
    float dmax = 0.0f;
    uint32_t seen = 0u;             // summed below: must come to the whole table
    unsigned long long sig = 0ull;  // ... and to the tiles' closed form
    if (tid < n_lists) {
        const uint32_t *h32 = reinterpret_cast<const uint32_t *>(qh + tid);  // {count, dropped, rows_seen, sig_lo, sig_hi}
        dmax = __uint_as_float(ld_u32(h32 + 1));
        seen = ld_u32(h32 + 2);
        sig = ((unsigned long long)ld_u32(h32 + 4) << 32) | ld_u32(h32 + 3);
    }
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_key);  // lower-bound counts per list head (s_key holds candidates only later)
    for (int i = tid; i < SEL_MAX_CAND; i += BLK) s_cnt[i] = 0u;
    const uint8_t qbyte = tid < d ? qbytes[tid] : (uint8_t)0;
    if (BLK < 1024)
        for (int i = tid + BLK; i < d; i += BLK) s_qf_late[i - BLK] = qbytes[i];  // (d > BLK: the bytes beyond the first BLK, via LDS)
    s_lut[tid & 255] = A.lut[tid & 255];
    if (tid < 8) s_u[tid] = 0;
    if (tid == 8) s_sig64 = 0ull;
    __syncthreads();
    PB_SEL_STAMP(1);
    if (tid < d) s_qf[tid] = s_lut[qbyte];
    if (BLK < 1024)
        for (int i = tid + BLK; i < d; i += BLK) s_qf[i] = s_lut[s_qf_late[i - BLK]];

    // ---- lower bound LB on the k-th largest filter cosine: the k-th largest among the first j entries of every (sorted)
    //      list, j = ceil(k / n_lists) -- the k-th largest of a subset is <= the k-th largest of all.  Found by counting:
    //      the T = n_lists * j values (< k + n_lists <= 768) go to LDS, thread t counts the values above and equal to its own.
    const int j_need = (int)((P.k + n_lists - 1) / n_lists);
    const int j_top = j_need < F_KWG ? j_need : F_KWG;  // (fewer than k listed entries in all: LB = thr0 below)
    const int n_topset = n_lists * j_top;
    uint32_t n_top = 0;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int i = tid + j * BLK;
        const int e = i % F_KWG;
        if (i < total_slots && e < j_top) {
            const bool live = my_key[j] != ~0ull;
            s_top[(i / F_KWG) * j_top + e] = live ? filter_key_cos(my_key[j]) : -1.0f;
            n_top += live ? 1u : 0u;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        dmax = fmaxf(dmax, __shfl_xor(dmax, off));
        n_top += (uint32_t)__shfl_xor((int)n_top, off);
        seen += (uint32_t)__shfl_xor((int)seen, off);
        sig += (unsigned long long)__shfl_xor((long long)sig, off);
    }
    if ((tid & 63) == 0) {
        s_red[tid >> 6] = dmax;
        if (n_top) atomicAdd(&s_u[0], n_top);
        if (seen) atomicAdd(&s_u[4], seen);
        if (sig) atomicAdd(&s_sig64, sig);
    }
    for (int i = n_topset + tid; i < ((n_topset + 15) & ~15); i += BLK) s_top[i] = -1.0f;  // pad to whole batches of reads
    __syncthreads();
    dmax = 0.0f;
    for (int w = 0; w < BLK / WAVE; ++w) dmax = fmaxf(dmax, s_red[w]);
    const uint32_t top_total = s_u[0];
    float lb = P.thr0;
    if (top_total >= P.k) {  // uniform
        // all 1 024 threads count: the values' indices are padded to a power of two tp, thread t counts for value t % tp over
        // part t / tp of the set (256 list heads: four parts of 64) and adds {above, equal} as two 16-bit halves into s_cnt
        int lg_tp = 6;
        while ((1 << lg_tp) < n_topset) ++lg_tp;
        const int vi = tid & ((1 << lg_tp) - 1), part = tid >> lg_tp, n_part = BLK >> lg_tp;
        const int len = (((n_topset + n_part - 1) / n_part) + 15) & ~15;  // values per part, whole batches of four 16-byte reads
        if (vi < n_topset) {
            const float v = s_top[vi];
            uint32_t above = 0, same = 0;
            const int u_end = (part + 1) * len < n_topset ? (part + 1) * len : n_topset;  // (s_top is padded with -1 to a multiple of 16)
            for (int u = part * len; u < u_end; u += 16) {  // four 16-byte LDS reads in flight (one per trip is a chain of LDS latencies)
                float4 o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = *reinterpret_cast<const float4 *>(s_top + u + 4 * c);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    above += (o[c].x > v ? 1u : 0u) + (o[c].y > v ? 1u : 0u) + (o[c].z > v ? 1u : 0u) + (o[c].w > v ? 1u : 0u);
                    same += (o[c].x == v ? 1u : 0u) + (o[c].y == v ? 1u : 0u) + (o[c].z == v ? 1u : 0u) + (o[c].w == v ? 1u : 0u);
                }
            }
            if (above | same) atomicAdd(&s_cnt[vi], above | (same << 16));
        }
        __syncthreads();
        if (tid < n_topset) {
            const uint32_t c = s_cnt[tid], above = c & 0xFFFFu, same = c >> 16;
            if (above < P.k && P.k <= above + same) s_u[1] = __float_as_uint(s_top[tid]);  // ties write the same word
        }
        __syncthreads();
        lb = fmaxf(__uint_as_float(s_u[1]), P.thr0);
    }
    const float cut = lb - 2.0f * P.m;
    PB_SEL_STAMP(2);

    // ---- candidates: every listed entry with cos_filter >= cut (from the registers loaded above)
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        if (my_key[j] != ~0ull && filter_key_cos(my_key[j]) >= cut) {
            const uint32_t pos = atomicAdd(&s_u[2], 1u);
            if (pos < (BLK < SEL_MAX_CAND ? BLK : SEL_MAX_CAND)) s_key[pos] = my_key[j];
        }
    }
    __syncthreads();
    const uint32_t n_cand_raw = s_u[2];
    constexpr uint32_t CAND_CAP = BLK < SEL_MAX_CAND ? BLK : SEL_MAX_CAND;  // one candidate per thread
    const int n_cand = n_cand_raw < CAND_CAP ? (int)n_cand_raw : (int)CAND_CAP;
    const bool overflow = n_cand_raw > CAND_CAP;
    PB_SEL_STAMP(3);
#ifdef PB_SCAN_STAMP
    if (blockIdx.x == 0 && threadIdx.x == BLK - 1) {
        g_sel_stamp[8] = n_cand_raw;
        g_sel_stamp[9] = __float_as_uint(lb);
        g_sel_stamp[10] = __float_as_uint(cut);
    }
#endif

    // ---- exact re-scoring, one candidate per thread (reference arithmetic, engine.rs:575-587); the row, its norm and
    //      its image_id are requested together
    uint64_t xkey = ~0ull;
    float my_cs = -2.0f, my_dist = 0.0f;
    int64_t my_id = 0;
    bool filtered = false;
    float nrm_c = 0.0f;
    if (tid < n_cand) {  // requested before the rows, consumed after them
        const uint32_t r = (uint32_t)s_key[tid];
        nrm_c = norms[r];
        my_id = ids[r];
    }
    if (d == 256) {
        // eight lanes per candidate (ref_fold_dot256_by8), 128 candidates per pass; the rows of the first two passes are
        // requested together (one round trip for up to 256 candidates); the sums go to thread `candidate` through LDS
        constexpr int PER = BLK / 8;
        constexpr int NCT = 2048 / BLK;  // candidates per lane group and trip: the rows of 256 candidates are requested together either way
        const int part = tid & 7;
        for (int c0 = tid >> 3; c0 < n_cand; c0 += NCT * PER) {
            uint4 ra[NCT][2];
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                const int cj = c0 + j * PER;
                const uint8_t *rj = rows + (uint64_t)(uint32_t)s_key[cj < n_cand ? cj : c0] * 256 + 32 * part;
                ra[j][0] = *reinterpret_cast<const uint4 *>(rj);
                ra[j][1] = *reinterpret_cast<const uint4 *>(rj + 16);
            }
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                const int cj = c0 + j * PER;
                if (cj < n_cand) {  // uniform over the eight lanes
                    const float dj = ref_fold_dot256_by8(ra[j][0], ra[j][1], s_qf, s_lut, part);
                    if (part == 7) s_top[cj] = dj;
                }
                __builtin_amdgcn_sched_barrier(0);  // (two folds interleaved need twice the product registers: scratch)
            }
        }
        __syncthreads();
    }
    if (tid < n_cand) {
        const uint32_t r = (uint32_t)s_key[tid];
        const float nrm = nrm_c;
        const float dot = d == 256 ? s_top[tid] : ref_fold_dot(rows + (uint64_t)r * d, s_qf, s_lut, d);
        float cs;
        const float dist = ref_distance(dot, P.sqrt_sa, nrm, &cs);
        my_cs = cs;
        my_dist = dist;
        if ((double)dist < P.max_dist) xkey = ((uint64_t)sortable_f32(dist) << 32) | r;
        else filtered = true;
    }
    __syncthreads();
    PB_SEL_STAMP(4);
    // largest exact cosine among candidates rejected by `dist < max_dist` (monotone: everything at or
    // below it is rejected too)
    float cfilt = filtered ? my_cs : -2.0f;
    for (int off = 32; off >= 1; off >>= 1) cfilt = fmaxf(cfilt, __shfl_xor(cfilt, off));
    if ((tid & 63) == 0) s_red[tid >> 6] = cfilt;
    int nsort = 64;
    while (nsort < n_cand) nsort <<= 1;
    if (tid < nsort) s_key[tid] = xkey;
    const uint64_t vm = __ballot(xkey != ~0ull);
    if ((tid & 63) == 0 && vm) atomicAdd(&s_u[3], (uint32_t)__popcll(vm));
    __syncthreads();
    cfilt = -2.0f;
    for (int w = 0; w < BLK / WAVE; ++w) cfilt = fmaxf(cfilt, s_red[w]);
    const uint32_t n_valid = s_u[3];
    const uint32_t n_out = n_valid < P.k ? n_valid : P.k;
    float ck = 3.0f;  // smallest exact cosine among the n_out selected
    if (n_cand <= SEL_RANK_MAX) {
        // order by counting: keys are distinct (row in the low word), rank = number of exact keys below mine
        if (tid < n_cand && xkey != ~0ull) {
            uint32_t rank = 0;
            for (int j = 0; j < n_cand; j += 16) {  // eight 16-byte LDS reads in flight (slots up to nsort >= 64 hold ~0)
                ulonglong2 o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = *reinterpret_cast<const ulonglong2 *>(s_key + j + 2 * c);
#pragma unroll
                for (int c = 0; c < 8; ++c) rank += (o[c].x < xkey ? 1u : 0u) + (o[c].y < xkey ? 1u : 0u);
            }
            if (rank < n_out) {
                if (done_flag) {
                    sel_put_granule(done_flag, 2 + rank, (uint32_t)my_id, (uint32_t)((uint64_t)my_id >> 32), __float_as_uint(my_dist), done_seq);
                } else {
                    out_ids[(size_t)q * out_stride + rank] = my_id;
                    out_dist[(size_t)q * out_stride + rank] = my_dist;
                }
                ck = my_cs;
            }
        }
    } else {
        block_bitonic_sort(s_key, nsort);
        if (tid < (int)n_out) {
            const uint64_t key = s_key[tid];
            const uint32_t r = (uint32_t)key;
            const int64_t id = ids[r];
            if (done_flag) {
                sel_put_granule(done_flag, 2 + tid, (uint32_t)id, (uint32_t)((uint64_t)id >> 32), (uint32_t)__float_as_uint(unsortable_f32((uint32_t)(key >> 32))), done_seq);
            } else {
                out_ids[(size_t)q * out_stride + tid] = id;
                out_dist[(size_t)q * out_stride + tid] = unsortable_f32((uint32_t)(key >> 32));
            }
        }
        if (n_out > 0) {
            const uint64_t kth_key = s_key[n_out - 1];
            // thread `tid` owns candidate tid (unsorted order): selected iff its exact key <= kth_key
            if (tid < n_cand && xkey <= kth_key) ck = my_cs;
        }
    }
    // ---- certificate (DESIGN.md "certificate"): exact cosine of the k-th result vs. the best any
    //      unexamined row could reach
    PB_SEL_STAMP(5);
    for (int off = 32; off >= 1; off >>= 1) ck = fminf(ck, __shfl_xor(ck, off));
    if ((tid & 63) == 0) s_ck[tid >> 6] = ck;
    __syncthreads();
    PB_SEL_STAMP(6);
    if (tid == 0) {
        ck = 3.0f;
        for (int w = 0; w < BLK / WAVE; ++w) ck = fminf(ck, s_ck[w]);
        const float o_max = fmaxf(fmaxf(cut, dmax), P.thr0) + P.m;  // no unexamined row's exact cos reaches this
        // the rows the workgroups report add up to the table (u32: a table is < 2^32 rows).  A NECESSARY condition only: it catches a
        // tile that nobody read or that two workgroups read (the failure class of a dynamic partition), not a tile read twice
        // while another full tile was skipped -- which the partitions here cannot produce (a ticket or chunk number is handed
        // out once by an atomic; what went wrong in round 4 was a chunk past the region's end hiding a valid one: a lost tile)
        // ... and (round 6) the sums of squared tile numbers come to the closed form for tiles 0 .. n_tiles - 1: a tile read twice while
        // another is skipped keeps the row count (full tiles hold the same number of rows) and breaks this one
        bool ok = !overflow && s_u[4] == A.n_rows && s_sig64 == tile_sig_expected(((uint64_t)A.n_rows + (A.tile_rows - 1)) / A.tile_rows);
        if (n_out == P.k) {
            ok = ok && (o_max <= ck * (1.0f - 1e-6f));
        } else {
            // fewer than k pass the filter among the candidates: every unexamined row must fail it too
            ok = ok && ((P.floor_is_filter && o_max <= P.c_floor) || (o_max <= cfilt));
        }
        ResultHdr h;
        h.count = n_out;
        // FUSED with more candidates than threads: 2 = "the lists are in memory: run k_select_rescore" (the host does)
        h.status = ok ? 0u : ((FUSED && overflow && n_cand_raw <= (uint32_t)SEL_MAX_CAND) ? 2u : 1u);
        h.n_cand = n_cand_raw;
        h.o_max = o_max;
        h.ck = n_out == P.k ? ck : -1.0f;
        if (done_flag) {
            sel_put_granule(done_flag, 0, h.count, h.status, h.n_cand, done_seq);
            sel_put_granule(done_flag, 1, __float_as_uint(h.o_max), __float_as_uint(h.ck), 0u, done_seq);
        } else {
            A.out_hdr[q] = h;
        }
        PB_SEL_STAMP(7);
    }
}

__global__ __launch_bounds__(SEL_BLOCK) void k_select_rescore(
    const uint8_t *__restrict__ rows, const int64_t *__restrict__ ids, const float *__restrict__ norms,
    int d, const uint8_t *__restrict__ queries, const QParams *__restrict__ qp,
    const float *__restrict__ lut, const uint64_t *__restrict__ lists, const ListHdr *__restrict__ hdrs,
    int n_lists, int64_t *__restrict__ out_ids, float *__restrict__ out_dist, ResultHdr *__restrict__ out_hdr,
    uint32_t out_stride, uint32_t *tail_ctr = nullptr, uint32_t *done_flag = nullptr, uint32_t done_seq = 0, uint32_t n_rows = 0,
    uint32_t tile_rows = 32) {
    SelArgs A;
    A.rows = rows; A.ids = ids; A.norms = norms; A.lut = lut; A.lists = lists; A.hdrs = hdrs; A.out_ids = out_ids; A.out_dist = out_dist;
    A.out_hdr = out_hdr; A.tail_ctr = tail_ctr; A.done_flag = done_flag; A.d = d; A.n_lists = n_lists; A.out_stride = out_stride;
    A.done_seq = done_seq; A.n_rows = n_rows; A.tile_rows = tile_rows;
    const int q = blockIdx.x;
    const QParams P = qp[q];
    select_rescore_body<SEL_BLOCK, false>(A, q, P, queries + (size_t)q * d);
}

template <bool FUSE>
__device__ __forceinline__ void fused_select(const uint8_t *rows, uint64_t n_rows, const uint64_t *lists, const ListHdr *hdrs, uint32_t *tail_ctr,
                                             const QArg256 &qarg, const FuseArgs &fa) {
    if constexpr (FUSE) {
        SelArgs A;
        A.rows = rows; A.ids = fa.ids; A.norms = fa.norms; A.lut = fa.lut; A.lists = lists; A.hdrs = hdrs; A.out_ids = fa.out_ids;
        A.out_dist = fa.out_dist; A.out_hdr = fa.out_hdr; A.tail_ctr = tail_ctr; A.done_flag = fa.done_flag; A.d = 256;
        A.n_lists = (int)gridDim.x; A.out_stride = fa.out_stride; A.done_seq = fa.done_seq; A.n_rows = (uint32_t)n_rows; A.tile_rows = fa.tile_rows;
        if (threadIdx.x == 0) __hip_atomic_store(fa.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every workgroup has arrived: zero for the next launch
        select_rescore_body<F_BLOCK, true>(A, 0, qarg.p, qarg.q);
    }
}

// (1d) merge of the k_scan_dist workgroup lists: sort all listed keys, take the first k; exact iff the k-th key is
// below every workgroup's `drop` bound (fewer than k results: iff nothing was dropped at all).  One block per query.
__global__ __launch_bounds__(SEL_BLOCK) void k_select_keys(const int64_t *__restrict__ ids, const QParams *__restrict__ qp,
                                                           const uint64_t *__restrict__ lists, const ListHdr *__restrict__ hdrs,
                                                           const uint64_t *__restrict__ drop_keys, int n_lists,
                                                           int64_t *__restrict__ out_ids, float *__restrict__ out_dist,
                                                           ResultHdr *__restrict__ out_hdr, uint32_t out_stride) {
    constexpr int CAP = F_MAX_WG * F_KWG / 2;  // 8192 keys: 256 workgroup lists
    __shared__ uint64_t s_key[CAP];
    __shared__ uint64_t s_min[SEL_BLOCK / WAVE];
    __shared__ uint32_t s_n;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const QParams P = qp[q];
    if (tid == 0) s_n = 0;
    __syncthreads();
    uint64_t dmin = ~0ull;
    const bool too_many = n_lists * F_KWG > CAP;
    for (int l = tid; l < n_lists; l += SEL_BLOCK) {
        const uint64_t dk = drop_keys[(size_t)q * n_lists + l];
        dmin = dk < dmin ? dk : dmin;
    }
    for (int i = tid; i < CAP; i += SEL_BLOCK) {
        const int l = i / F_KWG, j = i % F_KWG;
        uint64_t key = ~0ull;
        if (l < n_lists && j < (int)hdrs[(size_t)q * n_lists + l].count) key = lists[((size_t)q * n_lists + l) * F_KWG + j];
        s_key[i] = key;
        if (key != ~0ull) atomicAdd(&s_n, 1u);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const uint64_t o = __shfl_xor((unsigned long long)dmin, off);
        dmin = o < dmin ? o : dmin;
    }
    if ((tid & 63) == 0) s_min[tid >> 6] = dmin;
    block_bitonic_sort(s_key, CAP);
    const uint32_t n_valid = s_n;
    const uint32_t n_out = n_valid < P.k ? n_valid : P.k;
    if (tid < (int)n_out) {
        const uint64_t key = s_key[tid];
        out_ids[(size_t)q * out_stride + tid] = ids[(uint32_t)key];
        out_dist[(size_t)q * out_stride + tid] = unsortable_f32((uint32_t)(key >> 32));
    }
    if (tid == 0) {
        uint64_t dm = ~0ull;
        for (int w = 0; w < SEL_BLOCK / WAVE; ++w) dm = s_min[w] < dm ? s_min[w] : dm;
        bool ok = !too_many;
        if (n_out == P.k) ok = ok && s_key[n_out - 1] < dm;
        else ok = ok && dm == ~0ull;
        ResultHdr h;
        h.count = n_out;
        h.status = ok ? 0u : 1u;
        h.n_cand = n_valid;
        h.o_max = 0.0f;
        h.ck = -1.0f;
        out_hdr[q] = h;
    }
}

// ------------------------------------------------------------------------------------------------
// (2) exhaustive exact scan: one row per lane, every row re-scored with the reference arithmetic.
// qsel[blockIdx.y] = index of the query to run (the host compacts the queries that need this pass).
// METRIC 0: cosine_distance (engine.rs:572-588); 1: byte_distance (engine.rs:590-592); 2: hamming_distance
// (engine.rs:594-604, including its `.sum::<u8>()` wrap-around).  The integer sums of 1 and 2 are exact in f32
// (<= 255 * 1024 < 2^24), so only the final correctly rounded divide touches floating point.
__device__ __forceinline__ float ref_byte_or_hamming(const uint8_t *__restrict__ row, const uint8_t *__restrict__ q,
                                                      int d, int metric) {
    uint32_t acc = 0;
    int i = 0;
    for (; i + 4 <= d; i += 4) {
        uint32_t a, b;
        __builtin_memcpy(&a, q + i, 4);
        __builtin_memcpy(&b, row + i, 4);
        acc = metric == 1 ? __builtin_amdgcn_sad_u8(a, b, acc) : acc + __popc(a ^ b);
    }
    for (; i < d; ++i) {
        const uint32_t a = q[i], b = row[i];
        acc += metric == 1 ? (a > b ? a - b : b - a) : __popc(a ^ b);
    }
    if (metric == 1) return (float)acc / (255.0f * (float)d);
    return (float)(acc & 0xFFu) / (8.0f * (float)d);
}

template <int METRIC>
__global__ __launch_bounds__(X_BLOCK) void k_scan_exact(
    const uint8_t *__restrict__ rows, const float *__restrict__ norms, uint64_t n_rows, int d,
    const uint8_t *__restrict__ queries, const QParams *__restrict__ qp, const uint32_t *__restrict__ qsel,
    const float *__restrict__ lut, uint64_t *__restrict__ lists, uint32_t *__restrict__ list_counts,
    uint32_t list_stride) {
    __shared__ float s_lut[256];
    __shared__ float s_qf[1024];
    __shared__ __attribute__((aligned(4))) uint8_t s_qb[1024];
    __shared__ uint64_t s_buf[X_WAVES][X_MAXE * WAVE];
    __shared__ int s_cnt[X_WAVES];
    const int q = (int)qsel[blockIdx.y];
    const QParams P = qp[q];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: loop bounds and LDS bases stay in SGPRs
    for (int i = threadIdx.x; i < 256; i += X_BLOCK) s_lut[i] = lut[i];
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += X_BLOCK) {
        s_qb[i] = queries[(size_t)q * d + i];
        s_qf[i] = s_lut[s_qb[i]];
    }
    __syncthreads();
    uint64_t *buf = s_buf[wave];
    const int K = (int)P.k;
    int cnt = 0;
    uint64_t thr_key = ~0ull;  // keep keys < thr_key
    const uint64_t n_tiles = (n_rows + WAVE - 1) / WAVE;
    const uint64_t stride = (uint64_t)gridDim.x * X_WAVES;
    for (uint64_t t = (uint64_t)wave * gridDim.x + blockIdx.x; t < n_tiles; t += stride) {
        const uint64_t r = t * WAVE + lane;
        uint64_t key = ~0ull;
        if (r < n_rows) {
            float dist;
            if constexpr (METRIC == 0) {
                const float dot = ref_fold_dot(rows + r * (uint64_t)d, s_qf, s_lut, d);
                float cs;
                dist = ref_distance(dot, P.sqrt_sa, norms[r], &cs);
            } else {
                dist = ref_byte_or_hamming(rows + r * (uint64_t)d, s_qb, d, METRIC);
            }
            if ((double)dist < P.max_dist) key = ((uint64_t)sortable_f32(dist) << 32) | (uint32_t)r;
        }
        bool pass = key < thr_key;
        uint64_t m = __ballot(pass);
        if (m) {
            // prune only when the newcomers do not fit (capacity K + 64), not after every insertion (k_scan_exact_co)
            if (cnt + (int)__popcll(m) > K + WAVE) {
                thr_key = wave_keep_smallest<X_MAXE>(buf, cnt, K);
                cnt = K;
                pass = key < thr_key;
                m = __ballot(pass);
            }
            if (pass) buf[cnt + mbcnt(m)] = key;
            cnt += __popcll(m);
        }
    }
    if (cnt > K) {
        wave_keep_smallest<X_MAXE>(buf, cnt, K);
        cnt = K;
    }
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    // workgroup list = the K best of the 4 wave lists (sorted): bitonic sort of <= 4*(K) padded keys
    __shared__ uint64_t s_sort[X_WAVES * 256];
    int total = 0;
    int offs[X_WAVES];
    for (int w = 0; w < X_WAVES; ++w) {
        offs[w] = total;
        total += s_cnt[w];
    }
    int nsort = 64;
    while (nsort < total) nsort <<= 1;
    for (int i = threadIdx.x; i < nsort; i += X_BLOCK) s_sort[i] = ~0ull;
    __syncthreads();
    for (int w = 0; w < X_WAVES; ++w)
        for (int i = threadIdx.x; i < s_cnt[w]; i += X_BLOCK) s_sort[offs[w] + i] = s_buf[w][i];
    block_bitonic_sort(s_sort, nsort);
    const int n_out = total < K ? total : K;
    uint64_t *out = lists + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * list_stride;
    for (int i = threadIdx.x; i < n_out; i += X_BLOCK) out[i] = s_sort[i];
    if (threadIdx.x == 0) list_counts[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (uint32_t)n_out;
}

// engine.rs:576 without the table: L[v] = fl(fl(fl(v / 255) * 2) - 1) for v = 0..255 from three vector instructions.
// t = fl(v / 255) is reproduced by a split reciprocal, t = fma(v, R_HI, fl(v * R_LO)) with R_HI + R_LO = 1/255 to 48
// bits -- checked against the correctly rounded quotient for ALL 256 byte values (tests/test_oracle.py pins the two
// constants the same way; the kernel's results are compared with the oracle bit for bit) -- and fl(2 t - 1) is one fma
// (2 t is exact).  A table gather costs an LDS access at a random bank per byte; in the exhaustive pass, which
// de-quantises every byte of the table, that gather was the bound.
__device__ __forceinline__ float dequant_exact(float v) {
    const float R_HI = 0x1.010102p-8f, R_LO = -0x1.fdfdfep-33f;
    const float t = __builtin_fmaf(v, R_HI, v * R_LO);
    return __builtin_fmaf(t, 2.0f, -1.0f);
}

// (2b) the exhaustive pass for 256-byte rows, COALESCED, for QN queries at once.  k_scan_exact above lets every lane
// walk its own row (64 lanes x 16 B at a 256-B stride per load: 1.3 TB/s).  Here a wave streams a tile of 64 rows =
// 16 KiB with sixteen 1-KiB wave loads (16 B per lane, non-temporal: the table is read once), parks it in a
// wave-private LDS image with a 272-byte row pitch (the 16-B pad makes both the 16-B stores of the staging order and the
// lane-per-row 16-B reads conflict-free), and then every lane folds ITS row exactly as the reference does
// (engine.rs:575-587: table de-quantisation, dot = fl(dot + fl(a*b)) left to right) -- once for each of the QN queries
// of the group, so a row is fetched, staged and de-quantised once per QN queries.  The next tile's loads are in flight
// (64 registers) while the current one is folded.  Query values are broadcast LDS reads, and the
// de-quantisation is arithmetic (dequant_exact: the first version gathered from the 256-entry table in LDS and was
// bound by that gather, 0.88 ms per 10M-row sweep).  What bounds it now: 4 + 2 QN vector instructions per byte
// against the HBM stream.
// Output = k_scan_exact's: one sorted list of <= k exact keys per (query, workgroup) for k_merge_lists.
constexpr int XC_WAVES = 4;
constexpr int XC_PITCH = 272;
template <int QN, int MAXE>  // QN = 1 or 2
__global__ __launch_bounds__(XC_WAVES * WAVE, 2) void k_scan_exact_co(
    const uint8_t *__restrict__ rows, const float *__restrict__ norms, uint64_t n_rows,
    const float *__restrict__ qf, const QParams *__restrict__ qp, const uint32_t *__restrict__ qsel, int n_sel,
    const float *__restrict__ lut, uint64_t *__restrict__ lists, uint32_t *__restrict__ list_counts, uint32_t list_stride) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    // layout: [XC_WAVES][64 * XC_PITCH] tiles (queries in their pad bytes) | [XC_WAVES][QN][cap] key buffers | counts
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot0 = (int)blockIdx.y * QN;
    int K = 0;
    QParams P[QN];
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        const int sl = slot0 + j < n_sel ? slot0 + j : n_sel - 1;  // a short last group repeats its last query (never written)
        P[j] = qp[qsel[sl]];
        K = (int)P[j].k;
    }
    const int cap = K + WAVE;
    uint8_t *tile = s_dyn + (size_t)wave * (WAVE * XC_PITCH);
    uint64_t *s_keys = reinterpret_cast<uint64_t *>(s_dyn + (size_t)XC_WAVES * WAVE * XC_PITCH);
    uint64_t *buf[QN];
    int cnt[QN];
    uint64_t thr_key[QN];
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        buf[j] = s_keys + ((size_t)wave * QN + j) * cap;
        cnt[j] = 0;
        thr_key[j] = ~0ull;
    }
    // the de-quantised queries live in the 16 pad bytes behind every row of the tile images: value i of query j in
    // float i % 4 of the pad of row i / 4 of tile j (the parking stores never touch the pads)
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        const int i = threadIdx.x;  // 256 threads, 256 values
        const float v = qf[(size_t)(slot0 + j < n_sel ? slot0 + j : n_sel - 1) * 256 + i];
        *reinterpret_cast<float *>(s_dyn + (size_t)j * (WAVE * XC_PITCH) + (size_t)(i >> 2) * XC_PITCH + 256 + 4 * (i & 3)) = v;
    }
    __syncthreads();

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t n_tiles = (n_rows + WAVE - 1) / WAVE;
    const uint64_t stride = (uint64_t)gridDim.x * XC_WAVES;
    const uint64_t last_chunk = (n_rows * 256 - 16) / 16;  // last 16-byte piece inside the table
    auto request = [&](uint64_t t, u32x4 (&v)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint64_t c = t * 1024 + (uint64_t)j * 64 + lane;  // 16-byte piece index
            c = c < last_chunk ? c : last_chunk;
            v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows) + c);
        }
    };
    uint64_t t = (uint64_t)wave * gridDim.x + blockIdx.x;
    u32x4 nxt[16];
    if (t < n_tiles) request(t, nxt);
    for (; t < n_tiles; t += stride) {
        // park the tile: piece j of lane l is row 4 j + l / 16, bytes [16 (l % 16), +16)
#pragma unroll
        for (int j = 0; j < 16; ++j)
            *reinterpret_cast<u32x4 *>(tile + (size_t)(4 * j + (lane >> 4)) * XC_PITCH + 16 * (lane & 15)) = nxt[j];
        const uint64_t tn = t + stride;
        if (tn < n_tiles) request(tn, nxt);
        const uint64_t r = t * WAVE + lane;
        const float nrm = norms[r < n_rows ? r : n_rows - 1];
        float dot[QN];
#pragma unroll
        for (int j = 0; j < QN; ++j) dot[j] = 0.0f;
        const uint8_t *myrow = tile + (size_t)lane * XC_PITCH;
        // The sixteen 16-byte pieces of the row, two per loop trip, software-pipelined by hand: the LDS read of a piece
        // and the scalar loads of its 16 QN query values are issued one piece ahead.  Scalar loads return out of
        // order, so any wait for one of them is `lgkmcnt(0)` -- a wait for EVERYTHING outstanding; the empty asm
        // statements "use" a piece's operands BEFORE the next piece's loads are issued, which pins that wait to a
        // point where nothing else is in flight (left to itself the compiler waited at the top of every piece for the
        // loads it had just issued: ~35 % of the loop).
        // The loads and the wait are inline asm so that they stay where they are written: left to the compiler the
        // prefetch is sunk back to the top of the next trip, next to its use (tried with plain loads + empty asm
        // "uses" + a memory clobber: all re-rolled).  asm volatile statements keep their order; the wait statement
        // passes the piece through ("+" operands), so every fold instruction depends on it.
        // The row in sixteen 16-byte pieces; with a piece come the 16 QN query values it meets, as 4 QN broadcast LDS
        // reads (every lane the same address: conflict-free) from the pad bytes of the tile images, where the queries
        // were parked at kernel start.  LDS returns in order, and these loads are asm so that they stay one piece
        // ahead of their use.  Two other ways to supply the query values were measured on the 10M-row sweep
        // (profiles/exact_probe.py; no supply at all, i.e. garbage registers: 0.50 ms at QN = 1, 0.56 ms at QN = 2):
        // s_load_dwordx16 per piece from a global array 0.69 / 1.13 ms, pipelined or not (scalar-cache latency and, at
        // QN = 2, scalar registers); v_readlane broadcasts from 4 QN resident registers 0.81 / 1.33 ms (a readlane and
        // its wait states per multiply).
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        struct Piece {
            u32x4 w;
            f32x4 q[QN][4];
        };
        const uint32_t lds_row = (uint32_t)(uintptr_t)myrow;  // low 32 bits of a shared-aperture address = LDS offset
        const uint32_t lds_q = (uint32_t)(uintptr_t)s_dyn + 256u;
        // `other` is the piece about to be folded: it rides through the first load as an in/out operand, so its fold
        // cannot be scheduled ahead of the loads
        auto fetch = [&](Piece &pc, int c, Piece &other) {
            // ONE statement for all loads of the piece (separate statements get spread out: the query reads were sunk
            // behind the fold they should run under); outputs are early-clobber, the address registers stay live
            const uint32_t aw = lds_row + 16u * (uint32_t)c, aq = lds_q + (uint32_t)(4 * c) * XC_PITCH;
            if constexpr (QN == 1)
                asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %7 offset:272\n\t"
                             "ds_read_b128 %3, %7 offset:544\n\tds_read_b128 %4, %7 offset:816"
                             : "=&v"(pc.w), "=&v"(pc.q[0][0]), "=&v"(pc.q[0][1]), "=&v"(pc.q[0][2]), "=&v"(pc.q[0][3]), "+v"(other.w)
                             : "v"(aw), "v"(aq));
            else
                asm volatile("ds_read_b128 %0, %10\n\tds_read_b128 %1, %11\n\tds_read_b128 %2, %11 offset:272\n\t"
                             "ds_read_b128 %3, %11 offset:544\n\tds_read_b128 %4, %11 offset:816\n\t"
                             "ds_read_b128 %5, %11 offset:17408\n\tds_read_b128 %6, %11 offset:17680\n\t"
                             "ds_read_b128 %7, %11 offset:17952\n\tds_read_b128 %8, %11 offset:18224"
                             : "=&v"(pc.w), "=&v"(pc.q[0][0]), "=&v"(pc.q[0][1]), "=&v"(pc.q[0][2]), "=&v"(pc.q[0][3]), "=&v"(pc.q[1][0]),
                               "=&v"(pc.q[1][1]), "=&v"(pc.q[1][2]), "=&v"(pc.q[1][3]), "+v"(other.w)
                             : "v"(aw), "v"(aq));
        };
        static_assert(XC_PITCH == 272 && WAVE * XC_PITCH == 17408, "the offsets in the asm above are written out");
        // the running sums ride through the wait: it cannot be hoisted above the fold of the piece before it
        auto settle = [&](Piece &pc) {
            if constexpr (QN == 1)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc.w), "+v"(pc.q[0][0]), "+v"(pc.q[0][1]), "+v"(pc.q[0][2]), "+v"(pc.q[0][3]), "+v"(dot[0]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc.w), "+v"(pc.q[0][0]), "+v"(pc.q[0][1]), "+v"(pc.q[0][2]), "+v"(pc.q[0][3]),
                             "+v"(pc.q[1][0]), "+v"(pc.q[1][1]), "+v"(pc.q[1][2]), "+v"(pc.q[1][3]), "+v"(dot[0]), "+v"(dot[1]));
        };
        auto fold = [&](const Piece &pc) {
#if defined(PB_XC_ABL) && (PB_XC_ABL == 1)
            dot[0] += (float)pc.w[0];  // ablation: no fold
            return;
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float x = dequant_exact((float)((pc.w[e] >> (8 * b)) & 0xFF));  // v_cvt_f32_ubyteN + 3 VALU, no LDS gather
#pragma unroll
                    for (int j = 0; j < QN; ++j) {
                        // asm: left to the compiler pairs of these are packed (v_pk_mul_f32 / v_pk_add_f32) at the price
                        // of register moves, and a packed instruction costs 1.7 issue slots (profiles/r02_valu_rate.txt)
                        float p;
                        asm("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(pc.q[j][e][b]), "v"(x));
                        asm("v_add_f32 %0, %1, %2" : "=v"(dot[j]) : "v"(dot[j]), "v"(p));
                    }
                }
            }
        };
        Piece pa, pb;
        pb.w = u32x4{0, 0, 0, 0};
        fetch(pa, 0, pb);
#pragma unroll 1
        for (int c = 0; c < 16; c += 2) {
            settle(pa);
            fetch(pb, c + 1, pa);
            fold(pa);
            settle(pb);
            fetch(pa, c + 2 < 16 ? c + 2 : 15, pb);
            fold(pb);
        }
        settle(pa);  // nothing of ours may be in flight when compiler-tracked LDS traffic resumes
#pragma unroll
        for (int j = 0; j < QN; ++j) {
            float cs;
            const float dist = ref_distance(dot[j], P[j].sqrt_sa, nrm, &cs);
            uint64_t key = ~0ull;
            if (r < n_rows && (double)dist < P[j].max_dist) key = ((uint64_t)sortable_f32(dist) << 32) | (uint32_t)r;
            bool pass = key < thr_key[j];
            uint64_t m = __ballot(pass);
            if (m) {
                // Prune only when the newcomers do not fit (capacity K + 64): pruning after every insertion -- what
                // "if (cnt > K) prune" amounts to once the buffer holds K keys -- ran the 64-step radix select some
                // K ln(rows / K) ~ 400 times per wave; this way it runs once per ~64 insertions.
                if (cnt[j] + (int)__popcll(m) > cap) {
                    thr_key[j] = wave_keep_smallest<MAXE>(buf[j], cnt[j], K);
                    cnt[j] = K;
                    pass = key < thr_key[j];
                    m = __ballot(pass);
                }
                if (pass) buf[j][cnt[j] + mbcnt(m)] = key;
                cnt[j] += __popcll(m);
            }
        }
    }
    // workgroup list per query = the K best of the XC_WAVES wave lists, sorted (the tile area is free now)
#pragma unroll
    for (int j = 0; j < QN; ++j)
        if (cnt[j] > K) {
            wave_keep_smallest<MAXE>(buf[j], cnt[j], K);
            cnt[j] = K;
        }
    int *s_cnt = reinterpret_cast<int *>(s_keys + (size_t)XC_WAVES * QN * cap);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < QN; ++j) s_cnt[wave * QN + j] = cnt[j];
    }
    __syncthreads();
    uint64_t *s_sort = reinterpret_cast<uint64_t *>(s_dyn);  // XC_WAVES * 256 keys <= 8 KiB
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        if (slot0 + j >= n_sel) break;
        int total = 0, offs[XC_WAVES];
        for (int w = 0; w < XC_WAVES; ++w) {
            offs[w] = total;
            total += s_cnt[w * QN + j];
        }
        int nsort = 64;
        while (nsort < total) nsort <<= 1;
        __syncthreads();
        for (int i = threadIdx.x; i < nsort; i += XC_WAVES * WAVE) s_sort[i] = ~0ull;
        __syncthreads();
        for (int w = 0; w < XC_WAVES; ++w)
            for (int i = threadIdx.x; i < s_cnt[w * QN + j]; i += XC_WAVES * WAVE) s_sort[offs[w] + i] = s_keys[((size_t)w * QN + j) * cap + i];
        block_bitonic_sort(s_sort, nsort);
        const int n_out = total < K ? total : K;
        uint64_t *out = lists + ((size_t)(slot0 + j) * gridDim.x + blockIdx.x) * list_stride;
        for (int i = threadIdx.x; i < n_out; i += XC_WAVES * WAVE) out[i] = s_sort[i];
        if (threadIdx.x == 0) list_counts[(size_t)(slot0 + j) * gridDim.x + blockIdx.x] = (uint32_t)n_out;
    }
}
// (2c) the same pass for FOUR queries per sweep.  The fold is bound by vector-instruction issue (profiles/r02_valu_rate.txt),
// and of its 4.45 + 2 QN issue slots per byte (conversion 1.45, de-quantisation 3, multiply + add per query) the first
// 4.45 are shared by the queries of a group: 4.2 slots per byte and query at QN = 2, 3.1 at QN = 4.  Sixteen query
// values per query and 16-byte piece are 64 registers per piece in flight, twice (one piece ahead), so the table
// stream gives up half of its registers: a tile is parked and folded in two column HALVES (bytes [0,128) and [128,256) of
// its 64 rows: eight 1-KiB wave loads per half, each covering 8 full 128-byte lines; 144-byte row pitch in LDS, 16 pad
// bytes holding the queries as in k_scan_exact_co), the running sums stay in registers across the halves.  Half the
// bytes in flight per wave is enough here: at four folds per byte the sweep needs a third of the HBM rate.
constexpr int XC4_PITCH = 144;
constexpr int XC4_IMAGE = WAVE * XC4_PITCH;  // 9216
template <int MAXE>
__global__ __launch_bounds__(XC_WAVES * WAVE, 2) void k_scan_exact_co4(
    const uint8_t *__restrict__ rows, const float *__restrict__ norms, uint64_t n_rows,
    const float *__restrict__ qf, const QParams *__restrict__ qp, const uint32_t *__restrict__ qsel, int n_sel,
    uint64_t *__restrict__ lists, uint32_t *__restrict__ list_counts, uint32_t list_stride) {
    constexpr int QN = 4;
    static_assert(QN <= XC_WAVES, "one tile image per query for the parked query values");
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    // layout: [XC_WAVES] half-tile images (queries in their pad bytes) | [XC_WAVES][QN][cap] key buffers | counts
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot0 = (int)blockIdx.y * QN;
    int K = 0;
    QParams P[QN];
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        const int sl = slot0 + j < n_sel ? slot0 + j : n_sel - 1;  // a short last group repeats its last query (never written)
        P[j] = qp[qsel[sl]];
        K = (int)P[j].k;
    }
    const int cap = K + WAVE;
    uint8_t *tile = s_dyn + (size_t)wave * XC4_IMAGE;
    uint64_t *s_keys = reinterpret_cast<uint64_t *>(s_dyn + (size_t)XC_WAVES * XC4_IMAGE);
    uint64_t *buf[QN];
    int cnt[QN];
    uint64_t thr_key[QN];
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        buf[j] = s_keys + ((size_t)wave * QN + j) * cap;
        cnt[j] = 0;
        thr_key[j] = ~0ull;
    }
    // value i of query j: float i % 4 of the pad of row i / 4 of image j
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        const int i = threadIdx.x;  // 256 threads, 256 values
        const float v = qf[(size_t)(slot0 + j < n_sel ? slot0 + j : n_sel - 1) * 256 + i];
        *reinterpret_cast<float *>(s_dyn + (size_t)j * XC4_IMAGE + (size_t)(i >> 2) * XC4_PITCH + 128 + 4 * (i & 3)) = v;
    }
    __syncthreads();

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const uint64_t n_tiles = (n_rows + WAVE - 1) / WAVE;
    const uint64_t stride = (uint64_t)gridDim.x * XC_WAVES;
    const uint64_t last_chunk = (n_rows * 256 - 16) / 16;  // last 16-byte piece inside the table
    // half h of tile t: load j of lane l is row 8 j + l / 8 of the tile, bytes [128 h + 16 (l % 8), +16)
    auto request = [&](uint64_t t, int h, u32x4 (&v)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t c = (t * WAVE + (uint64_t)(8 * j + (lane >> 3))) * 16 + (uint64_t)(8 * h + (lane & 7));  // 16-byte piece index
            c = c < last_chunk ? c : last_chunk;
            v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows) + c);
        }
    };
    struct Piece {
        u32x4 w;
        f32x4 q[QN][4];
    };
    const uint32_t lds_row = (uint32_t)(uintptr_t)(tile + (size_t)lane * XC4_PITCH);  // low 32 bits of a shared-aperture address = LDS offset
    const uint32_t lds_q = (uint32_t)(uintptr_t)s_dyn + 128u;
    uint64_t t = (uint64_t)wave * gridDim.x + blockIdx.x;
    u32x4 nxt[8];
    if (t < n_tiles) request(t, 0, nxt);
    for (; t < n_tiles; t += stride) {
        const uint64_t r = t * WAVE + lane;
        const float nrm = norms[r < n_rows ? r : n_rows - 1];
        float dot[QN];
#pragma unroll
        for (int j = 0; j < QN; ++j) dot[j] = 0.0f;
        // the loads of a piece (row bytes + 16 broadcast reads of query values) in ONE asm statement, one piece ahead of
        // their use; `other` (the piece about to be folded) rides through as an in/out operand (see k_scan_exact_co)
        auto fetch = [&](Piece &pc, int h, int c, Piece &other) {
            const uint32_t aw = lds_row + 16u * (uint32_t)c, aq = lds_q + (uint32_t)(32 * h + 4 * c) * XC4_PITCH;
            asm volatile("ds_read_b128 %0, %18\n\t"
                         "ds_read_b128 %1, %19\n\tds_read_b128 %2, %19 offset:144\n\tds_read_b128 %3, %19 offset:288\n\tds_read_b128 %4, %19 offset:432\n\t"
                         "ds_read_b128 %5, %19 offset:9216\n\tds_read_b128 %6, %19 offset:9360\n\tds_read_b128 %7, %19 offset:9504\n\tds_read_b128 %8, %19 offset:9648\n\t"
                         "ds_read_b128 %9, %19 offset:18432\n\tds_read_b128 %10, %19 offset:18576\n\tds_read_b128 %11, %19 offset:18720\n\tds_read_b128 %12, %19 offset:18864\n\t"
                         "ds_read_b128 %13, %19 offset:27648\n\tds_read_b128 %14, %19 offset:27792\n\tds_read_b128 %15, %19 offset:27936\n\tds_read_b128 %16, %19 offset:28080"
                         : "=&v"(pc.w), "=&v"(pc.q[0][0]), "=&v"(pc.q[0][1]), "=&v"(pc.q[0][2]), "=&v"(pc.q[0][3]), "=&v"(pc.q[1][0]),
                           "=&v"(pc.q[1][1]), "=&v"(pc.q[1][2]), "=&v"(pc.q[1][3]), "=&v"(pc.q[2][0]), "=&v"(pc.q[2][1]), "=&v"(pc.q[2][2]),
                           "=&v"(pc.q[2][3]), "=&v"(pc.q[3][0]), "=&v"(pc.q[3][1]), "=&v"(pc.q[3][2]), "=&v"(pc.q[3][3]), "+v"(other.w)
                         : "v"(aw), "v"(aq));
        };
        static_assert(XC4_PITCH == 144 && XC4_IMAGE == 9216, "the offsets in the asm above are written out");
        // the wait, and behind it two empty statements that the rest of the piece and the running sums pass through
        // (an asm statement takes 30 operands): nothing of the fold can be scheduled above the wait
        auto settle = [&](Piece &pc) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc.w), "+v"(pc.q[0][0]), "+v"(pc.q[0][1]), "+v"(pc.q[0][2]), "+v"(pc.q[0][3]),
                         "+v"(pc.q[1][0]), "+v"(pc.q[1][1]), "+v"(pc.q[1][2]), "+v"(pc.q[1][3]), "+v"(dot[0]), "+v"(dot[1]));
            asm volatile("" : "+v"(pc.q[2][0]), "+v"(pc.q[2][1]), "+v"(pc.q[2][2]), "+v"(pc.q[2][3]), "+v"(pc.q[3][0]), "+v"(pc.q[3][1]),
                         "+v"(pc.q[3][2]), "+v"(pc.q[3][3]), "+v"(dot[2]), "+v"(dot[3]));
        };
        auto fold = [&](const Piece &pc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float x = dequant_exact((float)((pc.w[e] >> (8 * b)) & 0xFF));
#pragma unroll
                    for (int j = 0; j < QN; ++j) {
#ifdef PB_XC4_PLAIN
                        const float p = pc.q[j][e][b] * x;
                        dot[j] = dot[j] + p;
#else
                        // asm: left to the compiler pairs of these are packed (v_pk_mul_f32 / v_pk_add_f32) at the price
                        // of register moves, and a packed instruction costs 1.7 issue slots (profiles/r02_valu_rate.txt)
                        float p;
                        asm("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(pc.q[j][e][b]), "v"(x));
                        asm("v_add_f32 %0, %1, %2" : "=v"(dot[j]) : "v"(dot[j]), "v"(p));
#endif
                    }
                }
            }
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // park the half-tile: load j of lane l is row 8 j + l / 8, bytes [16 (l % 8), +16) of the half
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<u32x4 *>(tile + (size_t)(8 * j + (lane >> 3)) * XC4_PITCH + 16 * (lane & 7)) = nxt[j];
            if (h == 0) request(t, 1, nxt);
            else if (t + stride < n_tiles) request(t + stride, 0, nxt);
            asm volatile("" ::: "memory");  // the asm reads below are not memory operations to the compiler
            Piece pa, pb;
            pb.w = u32x4{0, 0, 0, 0};
            fetch(pa, h, 0, pb);
#pragma unroll 1
            for (int c = 0; c < 8; c += 2) {
                settle(pa);
                fetch(pb, h, c + 1, pa);
                fold(pa);
                settle(pb);
                fetch(pa, h, c + 2 < 8 ? c + 2 : 7, pb);
                fold(pb);
            }
            settle(pa);  // nothing of ours may be in flight when compiler-tracked LDS traffic resumes
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < QN; ++j) {
            float cs;
            const float dist = ref_distance(dot[j], P[j].sqrt_sa, nrm, &cs);
            uint64_t key = ~0ull;
            if (r < n_rows && (double)dist < P[j].max_dist) key = ((uint64_t)sortable_f32(dist) << 32) | (uint32_t)r;
            bool pass = key < thr_key[j];
            uint64_t m = __ballot(pass);
            if (m) {
                if (cnt[j] + (int)__popcll(m) > cap) {  // lazy pruning, as in k_scan_exact_co
                    thr_key[j] = wave_keep_smallest<MAXE>(buf[j], cnt[j], K);
                    cnt[j] = K;
                    pass = key < thr_key[j];
                    m = __ballot(pass);
                }
                if (pass) buf[j][cnt[j] + mbcnt(m)] = key;
                cnt[j] += __popcll(m);
            }
        }
    }
    // workgroup list per query = the K best of the XC_WAVES wave lists, sorted (the tile area is free now)
#pragma unroll
    for (int j = 0; j < QN; ++j)
        if (cnt[j] > K) {
            wave_keep_smallest<MAXE>(buf[j], cnt[j], K);
            cnt[j] = K;
        }
    int *s_cnt = reinterpret_cast<int *>(s_keys + (size_t)XC_WAVES * QN * cap);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < QN; ++j) s_cnt[wave * QN + j] = cnt[j];
    }
    __syncthreads();
    uint64_t *s_sort = reinterpret_cast<uint64_t *>(s_dyn);  // XC_WAVES * 256 keys <= 8 KiB
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        if (slot0 + j >= n_sel) break;
        int total = 0, offs[XC_WAVES];
        for (int w = 0; w < XC_WAVES; ++w) {
            offs[w] = total;
            total += s_cnt[w * QN + j];
        }
        int nsort = 64;
        while (nsort < total) nsort <<= 1;
        __syncthreads();
        for (int i = threadIdx.x; i < nsort; i += XC_WAVES * WAVE) s_sort[i] = ~0ull;
        __syncthreads();
        for (int w = 0; w < XC_WAVES; ++w)
            for (int i = threadIdx.x; i < s_cnt[w * QN + j]; i += XC_WAVES * WAVE) s_sort[offs[w] + i] = s_keys[((size_t)w * QN + j) * cap + i];
        block_bitonic_sort(s_sort, nsort);
        const int n_out = total < K ? total : K;
        uint64_t *out = lists + ((size_t)(slot0 + j) * gridDim.x + blockIdx.x) * list_stride;
        for (int i = threadIdx.x; i < n_out; i += XC_WAVES * WAVE) out[i] = s_sort[i];
        if (threadIdx.x == 0) list_counts[(size_t)(slot0 + j) * gridDim.x + blockIdx.x] = (uint32_t)n_out;
    }
}
// de-quantised queries of the selected slots for k_scan_exact_co: qf[slot][i] = lut[queries[qsel[slot]][i]]
__global__ void k_make_qf(const uint8_t *__restrict__ queries, const uint32_t *__restrict__ qsel, const float *__restrict__ lut,
                          float *__restrict__ qf) {
    qf[(size_t)blockIdx.x * 256 + threadIdx.x] = lut[queries[(size_t)qsel[blockIdx.x] * 256 + threadIdx.x]];
}

// merge up to M_FANIN sorted lists of <= k exact keys into one (k smallest, sorted). grid = (n_groups, nq).
// When `final_out` is set (last level) the result is written as ids / dists / count instead of keys.
__global__ __launch_bounds__(M_BLOCK) void k_merge_lists(
    const uint64_t *__restrict__ in_lists, const uint32_t *__restrict__ in_counts, int n_in,
    uint32_t in_stride, uint32_t k, uint64_t *__restrict__ out_lists, uint32_t *__restrict__ out_counts,
    uint32_t out_stride, int final_out, const int64_t *__restrict__ ids, const uint32_t *__restrict__ qsel,
    int64_t *__restrict__ out_ids, float *__restrict__ out_dist, ResultHdr *__restrict__ out_hdr,
    uint32_t res_stride) {
    __shared__ uint64_t s_sort[M_SORT];
    __shared__ uint32_t s_off[M_FANIN + 1];
    const int qi = blockIdx.y;
    const int grp = blockIdx.x;
    const int n_groups = gridDim.x;
    const int l0 = grp * M_FANIN;
    const int l1 = l0 + M_FANIN < n_in ? l0 + M_FANIN : n_in;
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int l = l0; l < l1; ++l) {
            s_off[l - l0] = acc;
            acc += in_counts[(size_t)qi * n_in + l];
        }
        s_off[l1 - l0] = acc;
    }
    __syncthreads();
    const int total = (int)s_off[l1 - l0];
    int nsort = 64;
    while (nsort < total) nsort <<= 1;
    for (int i = threadIdx.x; i < nsort; i += M_BLOCK) s_sort[i] = ~0ull;
    __syncthreads();
    for (int l = l0; l < l1; ++l) {
        const int c = (int)(s_off[l - l0 + 1] - s_off[l - l0]);
        const uint64_t *src = in_lists + ((size_t)qi * n_in + l) * in_stride;
        for (int i = threadIdx.x; i < c; i += M_BLOCK) s_sort[s_off[l - l0] + i] = src[i];
    }
    block_bitonic_sort(s_sort, nsort);
    const int n_out = total < (int)k ? total : (int)k;
    if (!final_out) {
        uint64_t *dst = out_lists + ((size_t)qi * n_groups + grp) * out_stride;
        for (int i = threadIdx.x; i < n_out; i += M_BLOCK) dst[i] = s_sort[i];
        if (threadIdx.x == 0) out_counts[(size_t)qi * n_groups + grp] = (uint32_t)n_out;
    } else {
        const int q = (int)qsel[qi];
        for (int i = threadIdx.x; i < n_out; i += M_BLOCK) {
            const uint64_t key = s_sort[i];
            out_ids[(size_t)q * res_stride + i] = ids[(uint32_t)key];
            out_dist[(size_t)q * res_stride + i] = unsortable_f32((uint32_t)(key >> 32));
        }
        if (threadIdx.x == 0) {
            ResultHdr h;
            h.count = (uint32_t)n_out;
            h.status = 0;
            h.n_cand = 0;
            h.o_max = 0.0f;
            h.ck = -1.0f;
            out_hdr[q] = h;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// (3) concurrent queries: ONE pass over the table serves up to 64 queries (dim 256).  The integer dots of a
// 16-row x 16-query tile come from v_mfma_i32_16x16x64_i8 (u8 -> s8 by XOR 0x80 on both operands:
// sum (a-128)(b-128) = P - 128(A+B) + 16384 D, hence num = 4*acc + 2(A+B) - 511 D), per-row integer norms from
// side tables filled at append time.  Two launches: HIST = true visits every MQ_SAMPLE-th tile and histograms
// cos_filter per query (LDS, then global); k_mq_pick_tau turns that into a per-query threshold tau expected to
// admit a few hundred rows; HIST = false streams the whole table once and appends every (row, query) with
// cos_filter >= tau[q] to that query's candidate list.  k_mq_rescore then does what k_select_rescore does:
// exact re-scoring, sort, certificate (every unlisted row has cos_filter < tau => cos_ref < tau + M_GLOB).
// The sampling only steers efficiency; correctness rests on the certificate, failures fall back.
constexpr int MQ_MAXQ = 64;
constexpr int MQ_WAVES = 4;
constexpr int MQ_BINS = 256;
// The sample pass's histogram of cos_filter (it only steers tau: efficiency, never correctness).  Rounds 1-4 used 256 equal bins over
// [0, 1]: the queries of a DENSE cluster -- a k-th best cosine of 0.993-0.9995 with thousands of rows at 0.99+, 11 % of the end-to-end
// leg's queries (profiles/e2e_uncertified_probe.py) -- found their whole candidate set in the top bin, were gated as "near-duplicate
// floods" and took the exhaustive pass although their tie sets hold 100-106 rows.  Round 5: equal bins of 1/128 below 0.875 (112 of
// them) and LOGARITHMIC bins in 1 - cos above it -- eight per octave from 2^-3 down to 2^-21 (144 of them), read straight off the
// float's exponent and top three mantissa bits -- so that the resolution follows the density where embeddings cluster.
__host__ __device__ __forceinline__ int mq_bin(float cs) {
    if (cs < 0.875f) {
        const int b = (int)(cs * 128.0f);
        return b < 0 ? 0 : b;
    }
    float u = 1.0f - cs;
    u = u > 4.76837158203125e-07f ? u : 4.76837158203125e-07f;  // 2^-21
#ifdef __HIP_DEVICE_COMPILE__
    const int key = (int)(__float_as_uint(u) >> 20);
#else
    uint32_t ub;
    memcpy(&ub, &u, 4);
    const int key = (int)(ub >> 20);
#endif
    const int b = 112 + (991 - key);  // u just below 2^-3: key 991 -> bin 112; u = 2^-21: key 848 -> bin 255
    return b < 112 ? 112 : (b > 255 ? 255 : b);
}
// lower cosine edge of bin b: every value in bins >= b is >= this
__host__ __device__ __forceinline__ float mq_bin_edge(int b) {
    if (b < 112) return (float)b / 128.0f;
    const uint32_t hi = (uint32_t)(1104 - b) << 20;  // upper edge of the bin in 1 - cos
#ifdef __HIP_DEVICE_COMPILE__
    return 1.0f - __uint_as_float(hi);
#else
    float f;
    memcpy(&f, &hi, 4);
    return 1.0f - f;
#endif
}
constexpr int MQ_CAP = 4096;
constexpr int MQ_SAMPLE = 32;
constexpr int MQ_HPITCH = 65;  // LDS histogram: [bin pair][query] with a 65-word pitch, two 16-bit counters per word
                               // (a workgroup samples far fewer than 65536 rows)
constexpr int MQ_LDROW = 272;  // 256 B row + 16 B pad: conflict-light ds_read_b128 in MFMA operand order

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int QT, bool HIST>
__global__ __launch_bounds__(MQ_WAVES * WAVE) void k_scan_multi(
    const uint8_t *__restrict__ rows, const int32_t *__restrict__ sum_b, const int32_t *__restrict__ den_b,
    uint64_t n_rows, const uint8_t *queries, const QParams *qp, const float *tau, uint64_t *cand, uint32_t *cand_cnt,
    uint32_t *ghist, int n_q, uint32_t cap, int count_mode) {
    // count_mode (HIST only): instead of a histogram, count the sampled rows with cos_filter >= tau[q] into bin 0
    // (the second chance asks "would the rows above tau2 fit the list?" before paying for a full sweep)
    constexpr int D = 256;
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[MQ_WAVES][16 * MQ_LDROW];
    __shared__ uint32_t s_hist[HIST ? (MQ_BINS / 2) * MQ_HPITCH : 1];  // two 16-bit counters per word
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: loop bounds and LDS bases stay in SGPRs
    const int li = lane & 15, kq = lane >> 4;
    // blockIdx.y selects a chunk of MQ_MAXQ queries (a burst's sample passes run as ONE launch); single-chunk
    // launches have gridDim.y == 1
    {
        const int chunk0 = (int)blockIdx.y * MQ_MAXQ;
        queries += (size_t)chunk0 * D;
        qp += chunk0;
        tau += chunk0;
        cand += (size_t)chunk0 * cap;
        cand_cnt += chunk0;
        ghist += (size_t)chunk0 * MQ_BINS;
        n_q = (n_q - chunk0) < MQ_MAXQ ? (n_q - chunk0) : MQ_MAXQ;
    }
    if constexpr (HIST) {
        for (int i = threadIdx.x; i < (MQ_BINS / 2) * MQ_HPITCH; i += blockDim.x) s_hist[i] = 0;
        __syncthreads();
    }
    // query fragments (B operand): lane (j = li, kq) holds bytes [64 s + 16 kq, +16) of query 16 qt + j, as s8
    i32x4 bq[QT][4];
    float den_a[QT], q_tau[QT], g_tau[QT], rs_a[QT];
    int sum_a[QT], sa2[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int q = qt * 16 + li;
        q = q < n_q ? q : n_q - 1;  // padded columns repeat the last query (their results are never read)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            i32x4 v = *reinterpret_cast<const i32x4 *>(queries + (size_t)q * D + 64 * s + 16 * kq);
            v.x ^= 0x80808080; v.y ^= 0x80808080; v.z ^= 0x80808080; v.w ^= 0x80808080;
            bq[qt][s] = v;
        }
        den_a[qt] = qp[q].den_a;
        sum_a[qt] = qp[q].sum_a;
        q_tau[qt] = (HIST && !count_mode) ? qp[q].thr0 : tau[q];
        sa2[qt] = 2 * sum_a[qt];
        g_tau[qt] = q_tau[qt] * __builtin_amdgcn_sqrtf(den_a[qt]);  // tau > 0
        rs_a[qt] = __builtin_amdgcn_rsqf(den_a[qt]);
    }
    uint8_t *tile = s_tile[wave];
    const uint64_t n_tiles = (n_rows + 15) / 16;
    const uint64_t stride = (uint64_t)gridDim.x * MQ_WAVES;
    // software pipeline: the 4 KiB of tile t+1 are requested before tile t is computed
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto tile_of = [&](uint64_t t) { return HIST ? t * MQ_SAMPLE : t; };
    // the per-row integer norms of the tile travel with it (a dependent load after the MFMAs would expose a
    // full memory round trip per tile); the side tables have 32 entries of zeroed slack past capacity
    auto issue = [&](uint64_t tt, u32x4 (&dst)[4], i32x4 &sbv, i32x4 &dbv) {
        const uint64_t r0 = tt * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t r = r0 + (uint64_t)(4 * j + kq);
            r = r < n_rows ? r : n_rows - 1;
            dst[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows + r * D + li * 16));
        }
        sbv = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(sum_b + r0 + 4 * (uint64_t)kq));
        dbv = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(den_b + r0 + 4 * (uint64_t)kq));
    };
    // two tiles in flight per wave (register slots 0 / 1): with one, a wave's next 4 KiB arrive later than it
    // finishes the current tile and the pass runs at memory LATENCY
    u32x4 ld[2][4];
    i32x4 sb_n[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, db_n[2] = {{1, 1, 1, 1}, {1, 1, 1, 1}};
    auto valid = [&](uint64_t tv) { return tv < n_tiles && tile_of(tv) < n_tiles; };
    auto process = [&](uint64_t t, u32x4 (&lds)[4], i32x4 &sbn, i32x4 &dbn) __attribute__((always_inline)) {
        const uint64_t tt = tile_of(t);
        const uint64_t row0 = tt * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<u32x4 *>(tile + (4 * j + kq) * MQ_LDROW + li * 16) = lds[j];
        i32x4 sb = sbn, db = dbn;
        {
            const uint64_t tn = t + 2 * stride;  // this slot's next tile
            if (valid(tn)) issue(tile_of(tn), lds, sbn, dbn);
        }
        i32x4 acc[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[qt] = (i32x4){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // A operand: lane (i = li, kq) holds bytes [64 s + 16 kq, +16) of row i
            i32x4 a = *reinterpret_cast<const i32x4 *>(tile + li * MQ_LDROW + 64 * s + 16 * kq);
            a.x ^= 0x80808080; a.y ^= 0x80808080; a.z ^= 0x80808080; a.w ^= 0x80808080;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
                acc[qt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bq[qt][s], acc[qt], 0, 0, 0);
        }
        // result: lane holds rows row0 + 4 kq + r (r = 0..3) of query column li
        const uint64_t rbase = row0 + 4 * (uint64_t)kq;
        // rows past the end of the table: the side-table slack may hold anything; make the divisor harmless
        db.x = (rbase + 0 < n_rows) ? db.x : 1;
        db.y = (rbase + 1 < n_rows) ? db.y : 1;
        db.z = (rbase + 2 < n_rows) ? db.z : 1;
        db.w = (rbase + 3 < n_rows) ? db.w : 1;
        if constexpr (HIST) {
            // the histogram only steers tau (efficiency, not correctness), so cos_filter is formed the cheap way:
            // one multiply by the row's 1/sqrt(den_b) and one by the query's 1/sqrt(den_a); bins are laid out
            // bin-major with a 65-word pitch so that the 16 query columns of a wave hit different LDS banks
            float rsb[4];
            int cr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rsb[r] = (rbase + r < n_rows) ? __builtin_amdgcn_rsqf((float)db[r]) : 0.0f;
                cr[r] = 2 * sb[r] - 511 * D;
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int q = qt * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int num = 4 * acc[qt][r] + sa2[qt] + cr[r];
                    const float cs = (float)num * rsb[r] * rs_a[qt];
                    if (cs >= q_tau[qt] && q < n_q) {
                        const int bin = count_mode ? 0 : mq_bin(cs);
                        atomicAdd(&s_hist[(bin >> 1) * MQ_HPITCH + q], 1u << (16 * (bin & 1)));
                    }
                }
            }
        } else {
            // cheap conservative pre-test in 5 instructions per (row, query): num >= tau*sqrt(den_a) * sqrt(den_b) * (1-1e-6)
            // is implied by cos_filter >= tau; only the rare survivors evaluate cos_filter itself
            float wr[4];
            int cr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wr[r] = (rbase + r < n_rows) ? __builtin_amdgcn_sqrtf((float)db[r]) * (1.0f - 1e-6f) : 3.0e38f;
                cr[r] = 2 * sb[r] - 511 * D;
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int q = qt * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int num = 4 * acc[qt][r] + sa2[qt] + cr[r];
                    if ((float)num >= g_tau[qt] * wr[r]) {
                        const float cs = (float)num * __builtin_amdgcn_rsqf((float)db[r] * den_a[qt]);
                        if (cs >= q_tau[qt] && q < n_q) {
                            const uint32_t idx = atomicAdd(&cand_cnt[q], 1u);
                            if (idx < cap) cand[(size_t)q * cap + idx] = filter_key(cs, (uint32_t)(rbase + r));
                        }
                    }
                }
            }
        }
    };
    uint64_t t = (uint64_t)wave * gridDim.x + blockIdx.x;
    if (valid(t)) issue(tile_of(t), ld[0], sb_n[0], db_n[0]);
    if (valid(t + stride)) issue(tile_of(t + stride), ld[1], sb_n[1], db_n[1]);
    while (valid(t)) {
        process(t, ld[0], sb_n[0], db_n[0]);
        t += stride;
        if (!valid(t)) break;
        process(t, ld[1], sb_n[1], db_n[1]);
        t += stride;
    }
    if constexpr (HIST) {
        __syncthreads();
        for (int i = threadIdx.x; i < QT * 16 * MQ_BINS; i += blockDim.x) {
            const int q = i / MQ_BINS, bin = i % MQ_BINS;  // ghist stays query-major
            const uint32_t v = (s_hist[(bin >> 1) * MQ_HPITCH + q] >> (16 * (bin & 1))) & 0xFFFFu;
            if (v) atomicAdd(&ghist[i], v);
        }
    }
}

// Collect pass for a large burst: a workgroup of NWQ waves walks the table ONCE for 64 * NWQ queries.  The row tile
// (16 NWQ rows per step, four 16-byte loads per thread, s8-converted on the way) is staged in LDS, double-buffered
// with one barrier per step and the global loads two steps ahead, and every wave multiplies it by ITS OWN 64
// queries (4 x 4 query fragments resident in registers).  Table bytes per query drop from N*D/64 to N*D/(64 NWQ),
// which moves the pass from the HBM roof to the i8-MFMA roof (at the clock the chip holds in an MFMA-dense loop).
// Per (row, query) the survivor test is ONE v_fma_f32 plus a share of a max/compare:
//     cos_filter >= tau  <=  num >= g W            with g = tau sqrt(den_a), W = sqrt(den_b) (1 - 1e-6)
//                        <=  (4 acc' + cr) / W >= g - 1      acc' = acc + floor(2A/4),  cr = 2B - 511 D
// * the per-query term rides in the accumulator: the MFMA chain starts from MAGIC + floor(2A[q] / 4);
// * MAGIC = 0x4B400000 is the bit pattern of 1.5 * 2^23, so the i32 result READ AS A FLOAT is 12582912 + acc'
//   (exact up to 2^24, an over-estimate beyond: still conservative), which makes the i32 -> f32 convert free;
// * the per-row factors 4/W and (cr - 4 * 12582912)/W are computed once per step by the first threads (in f64,
//   rounded once) and shared through LDS;
// * the "- 1" covers the dropped remainder of 2A/4 (<= 2/W <= 0.125) and the f32 roundings (<= 0.6).
// The rare survivors are re-tested exactly in integers and queued in LDS; the queue is drained with global atomics
// between steps, so no wave waits for an atomic's return inside the MFMA stream.  The MFMAs of tile t+1 are issued
// before the tests of tile t.  grid = (workgroups, ceil(n_q / (64 NWQ))).
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float max3_plain(float a, float b, float c) {
    // v_max3_f32 without the quieting v_max x, x that fmaxf() puts in front of every operand (no NaNs here)
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_plain(float a, float b, float c) {
    // one v_fma_f32: keeps hipcc from SLP-packing the tests into v_pk_fma_f32, which issues slower beside MFMAs
    float d;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

#ifdef PB_MQ_STAMP
__device__ unsigned long long g_mq_stamp[8];
#endif
template <int NWQ>
__global__ __launch_bounds__(NWQ * WAVE) void k_scan_multi_wg(
    const uint8_t *__restrict__ rows, const int32_t *__restrict__ sum_b, const int32_t *__restrict__ den_b,
    uint64_t n_rows, const uint8_t *__restrict__ queries, const QParams *__restrict__ qp,
    const float *__restrict__ tau, uint64_t *__restrict__ cand, uint32_t *__restrict__ cand_cnt, int n_q) {
    constexpr int D = 256;
    constexpr int LPT = 4;             // 16-byte loads per thread per step
    constexpr int TR = 4 * NWQ * LPT;  // rows per step
    constexpr int NT = TR / 16;        // 16-row MFMA tiles per step
    constexpr int QT = 4;
    constexpr int MAGIC = 0x4B400000;  // bits of 12582912.0f
    constexpr int QCAP = 1024;         // survivor queue entries
    static_assert(TR % 32 == 0 && TR <= NWQ * WAVE, "step shape");
    // (A 496-byte pitch with a per-row shift that makes every ds_read_b128 lane group hit sixteen distinct 16-byte slots -- the 272-byte
    // pitch has one two-way conflict per group, 8 LDS cycles per read instead of 4 -- was measured in round 6: 2.44 ms against 2.41.
    // The operand reads cost 0.55 ms of the pass (profiles/r06_burst_collect.txt), but not through the LDS array's cycles.)
    constexpr int WROW = MQ_LDROW;
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[2][TR * WROW];
    __shared__ __attribute__((aligned(16))) float s_iw[2][TR];  // per row: 4 / W
    __shared__ __attribute__((aligned(16))) float s_c2[2][TR];  // (cr - 4 * 12582912) / W
    __shared__ __attribute__((aligned(8))) float s_f4[2][TR / 4][2];  // per group of 4 rows: {4 / min W, (max cr - 4 * 12582912) / min W}: one ds_read_b64
    __shared__ i32x4 s_qacc[QCAP];  // survivor queue: accumulator quad, query, first row of the quad
    __shared__ uint32_t s_qq[QCAP], s_qrow[QCAP];
    // Round 6: the queue is eight SEGMENTS, one per wave, and a wave's fill count lives in a scalar register: appending is a ballot, a
    // lane count and three LDS stores -- no LDS atomic whose return the wave would wait for, with the matrix pipe idle, on ~0.8 tiles
    // of every step (PB_MQ_STAMP: the waves of a step differ by how many of their tiles held a survivor, and all eight wait at the
    // step's barrier for the unluckiest: 1 225 of a step's 6 900 clocks).  s_qn publishes the counts for drain(); s_qflag asks for one.
    constexpr int QSEG = QCAP / NWQ;
    __shared__ uint32_t s_qn[NWQ];
    __shared__ uint32_t s_qflag;
    uint32_t wq_cnt = 0;  // entries in this wave's segment (uniform over the wave)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;  // (a scalar wave index measured 2.5 % slower here: 2.82 vs 2.75 ms)
    const int li = lane & 15, kq = lane >> 4;
    const int a_off = li * WROW + 16 * kq;  // this lane's A-fragment bytes of k slice 0 inside a 16-row tile
    const int qbase = (int)blockIdx.y * (NWQ * 64) + wave * 64;
    const bool active = qbase < n_q;  // a wave whose 64 queries lie past n_q only helps loading
    if (tid < NWQ) s_qn[tid] = 0;
    if (tid == 0) s_qflag = 0;
    i32x4 bq[QT][4];
    i32x4 cinit[QT];
    float gthr[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int q = qbase + qt * 16 + li;
        q = q < n_q ? q : n_q - 1;  // padded columns repeat the last query (their results are never appended)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            i32x4 v = *reinterpret_cast<const i32x4 *>(queries + (size_t)q * D + 64 * s + 16 * kq);
            v.x ^= 0x80808080; v.y ^= 0x80808080; v.z ^= 0x80808080; v.w ^= 0x80808080;
            bq[qt][s] = v;
        }
        const int c0 = MAGIC + (2 * qp[q].sum_a) / 4;  // sum_a >= 0
        cinit[qt] = (i32x4){c0, c0, c0, c0};
        gthr[qt] = tau[q] * __builtin_amdgcn_sqrtf(qp[q].den_a) - 1.0f;  // tau > 0
    }
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t n_steps = (n_rows + TR - 1) / TR;
    const int ld_row = tid >> 4, ld_col = (tid & 15) * 16;  // thread's first row of the step; the others are + 4 NWQ j
    u32x4 ld[2][LPT];
    int ld_sb[2] = {0, 0}, ld_db[2] = {1, 1};
    auto issue = [&](uint64_t stp, u32x4 (&dst)[LPT], int &sb, int &db) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            uint64_t r = stp * TR + (uint64_t)(ld_row + j * 4 * NWQ);
            r = r < n_rows ? r : n_rows - 1;
            dst[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows + r * D + ld_col));
        }
        if (tid < TR) {
            uint64_t rr = stp * TR + (uint64_t)tid;
            rr = rr < n_rows ? rr : n_rows - 1;
            sb = sum_b[rr];
            db = den_b[rr];
        }
    };
    auto stage_tile = [&](int buf, const u32x4 (&src)[LPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            u32x4 v = src[j];
            v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
            *reinterpret_cast<u32x4 *>(&s_tile[buf][(ld_row + j * 4 * NWQ) * WROW + ld_col]) = v;
        }
    };
    // The per-row and per-group factors of a step (the first TR threads: two waves).  Round 6: (1) called at the TOP of the step before,
    // not beside the tile staging in front of the barrier -- there the other six waves waited at the barrier while these two worked
    // (PB_MQ_STAMP: 1 243 of a step's 7 214 clocks spent waiting at the barrier); here the work overlaps the MFMAs of the waves these
    // two share their SIMDs with.  The factor buffer written is the one the step before last read, behind a barrier.  (2) ONE
    // v_rcp_f32 per row and per group instead of four correctly rounded divisions (-fno-fast-math: ~10 dependent instructions each).
    // v_rcp_f32 is within 1 ulp; the two quotients take it with DIRECTED margins -- 4 / W from rcp * (1 + 2^-21) >= 1 / W, and
    // (cr - 4 * 12582912) / W, whose numerator is always negative, from rcp * (1 - 2^-21) <= 1 / W -- so both terms of t can only be
    // over-estimated against the divisions they replace (by <= 2^-20 of ~2.1e4 each on ordinary rows: 0.02; more rows pass the
    // first stage by that margin, none fewer) and the threshold's "- 1" keeps covering what it covered.
    auto stage_factors = [&](uint64_t stp, int buf, int sb, int db) __attribute__((always_inline)) {
        if (tid < TR) {  // TR is a multiple of 64: whole waves
            // the 4 rows a lane tests together (one accumulator quad) share ONE pair of factors: the smallest W and the
            // largest cr of the group bound every row's t from above, (4 acc' + cr_r) / W_r <= (4 max acc' + max cr) / min W
            // when the numerator is >= 0 (a negative one fails the test either way)
            const bool ok = stp * TR + (uint64_t)tid < n_rows;  // rows past the end can never pass
            float w = ok ? __builtin_amdgcn_sqrtf((float)db) * (1.0f - 1e-6f) : 3.0e38f;
            int cr = ok ? 2 * sb - 511 * D : -0x40000000;
            constexpr float UP = 1.0f + 4.76837158203125e-7f, DN = 1.0f - 4.76837158203125e-7f;  // 1 +- 2^-21
            const float rw = __builtin_amdgcn_rcpf(w);
            s_iw[buf][tid] = ok ? 4.0f * (rw * UP) : 0.0f;  // the row's own factors: the second, per-row stage of the test
            s_c2[buf][tid] = ok ? (float)(cr - 50331648) * (rw * DN) : -3.0e38f;
            w = fminf(w, __shfl_xor(w, 1));
            w = fminf(w, __shfl_xor(w, 2));
            const int c1 = __shfl_xor(cr, 1);
            cr = cr > c1 ? cr : c1;
            const int c2 = __shfl_xor(cr, 2);
            cr = cr > c2 ? cr : c2;
            // |cr - 4 * 12582912| < 2^26 loses <= 2 units in the conversion, i.e. <= 2 / W <= 0.125 more in t -- with the 0.125 of
            // the dropped remainder and the roundings of the products and of the fma still inside the "- 1" of the threshold.
            // Every lane of a group computes the group's pair (same inputs, same bits) and one stores it: no divergent branch.
            const bool any_ok = w < 1.0e38f;
            const float rg = __builtin_amdgcn_rcpf(w);
            const float iw4 = any_ok ? 4.0f * (rg * UP) : 0.0f;
            const float c24 = any_ok ? (float)(cr - 50331648) * (rg * DN) : -3.0e38f;
            if ((tid & 3) == 0) {
                *reinterpret_cast<float2 *>(&s_f4[buf][tid >> 2][0]) = make_float2(iw4, c24);
            }
        }
    };
    auto stage = [&](uint64_t stp, int buf, const u32x4 (&src)[LPT], int sb, int db) __attribute__((always_inline)) {
        stage_tile(buf, src);
        stage_factors(stp, buf, sb, db);
    };
    // exact integer re-test of one queued (row, query) and append to the query's candidate list
    auto retest_append = [&](int acc_bits, int q, uint32_t row) __attribute__((always_inline)) {
        const int sa2 = 2 * qp[q].sum_a;
        const int num = 4 * (acc_bits - MAGIC - sa2 / 4) + sa2 + 2 * sum_b[row] - 511 * D;
        const float cs = (float)num * __builtin_amdgcn_rsqf((float)den_b[row] * qp[q].den_a);
        if (cs >= tau[q]) {
            const uint32_t idx = atomicAdd(&cand_cnt[q], 1u);
            if (idx < MQ_CAP) cand[(size_t)q * MQ_CAP + idx] = filter_key(cs, row);
        }
    };
    // A fragments of one 16-row tile (the lane's 16 bytes of each of the four 64-byte k slices) and the factors of the lane's row
    // group: read ONE TILE AHEAD of their use (round 6), so no block of the step's pipeline starts by waiting for LDS
    auto load_a = [&](const uint8_t *tile, i32x4 (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#if defined(PB_MQ_ABL) && PB_MQ_ABL == 3  // ablation 3: no tests AND no LDS operand reads (a resident register instead)
            a[s] = bq[0][s];
#else
            a[s] = *reinterpret_cast<const i32x4 *>(tile + a_off + 64 * s);
#endif
        }
    };
    auto load_f = [&](int buf, int tl, float (&f)[2]) __attribute__((always_inline)) {
        const float2 v = *reinterpret_cast<const float2 *>(&s_f4[buf][4 * tl + kq][0]);
        f[0] = v.x;
        f[1] = v.y;
    };
    auto mfma_a = [&](const i32x4 (&a)[4], i32x4 (&acc)[QT]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
#if defined(PB_MQ_ABL) && PB_MQ_ABL == 2  // ablation 2: no MFMAs (the operand reads stay)
                acc[qt] = s == 0 ? cinit[qt] : (i32x4){acc[qt][0] + a[s][0], acc[qt][1], acc[qt][2], acc[qt][3]};
#else
                acc[qt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], bq[qt][s], s == 0 ? cinit[qt] : acc[qt], 0, 0, 0);
#endif
            }
        }
    };
    auto test_tile = [&](const i32x4 (&acc)[QT], int buf, int tl, uint64_t stp, const float (&f)[2]) __attribute__((always_inline)) {
        // lane holds rows rbase + r (r = 0..3) of query column li
        const int rl = 16 * tl + 4 * kq;
        const float iw = f[0], c2 = f[1];  // s_f4 of the lane's row group (load_f)
        float dq[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            // the accumulators read as floats are 12582912 + acc' (positive): their float order is their integer order, so
            // the largest of the quad's four rows is picked first and tested ONCE with the group's factors
            const float m = max3_plain(max3_plain(__int_as_float(acc[qt][0]), __int_as_float(acc[qt][1]), __int_as_float(acc[qt][2])),
                                       __int_as_float(acc[qt][3]), 0.0f);
            dq[qt] = fma_plain(m, iw, c2);
        }
        // four compares whose masks are OR-ed on the scalar side (no subtractions, no max over the four)
        const bool hit[QT] = {dq[0] >= gthr[0], dq[1] >= gthr[1], dq[2] >= gthr[2], dq[3] >= gthr[3]};
        const bool any = hit[0] | hit[1] | hit[2] | hit[3];
        // ONE rarely-taken branch per tile.  It only QUEUES the lane's accumulator quad (4 rows of one query) in
        // LDS, in a handful of instructions: with 8 waves meeting at a barrier every step, and some wave of a
        // step's 64 wave-tiles nearly always holding a survivor, whatever this branch costs is paid by the whole
        // workgroup on almost every step.  The exact re-test (which also discards the quad's non-survivors: it
        // implies the test above) and the append to the candidate lists happen in drain().
        // (a branch of the WHOLE wave: every lane keeps the same wq_cnt)
        if (__builtin_amdgcn_ballot_w64(any) != 0) {
            // second stage, per row with the row's own factors (what the test was before the group bound): on a table whose
            // neighbouring rows have very different norms the group bound alone would pass whole quads and flood the queue.
            // Straight-line per query tile (four fma, four compares, no short-circuit cascade of exec-mask branches); a query tile
            // none of whose lanes passed the first stage is skipped on the scalar side.
            const uint32_t row0 = (uint32_t)(stp * TR) + (uint32_t)rl;
            const f32x4_t iwr = *reinterpret_cast<const f32x4_t *>(&s_iw[buf][rl]);
            const f32x4_t c2r = *reinterpret_cast<const f32x4_t *>(&s_c2[buf][rl]);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (__builtin_amdgcn_ballot_w64(hit[qt]) == 0) continue;
                const int q = qbase + qt * 16 + li;
#ifdef PB_MQ_NO_STAGE2  // test-of-the-test build: the group bound alone
                bool pass = hit[qt];
#else
                const bool p0 = fma_plain(__int_as_float(acc[qt][0]), iwr[0], c2r[0]) >= gthr[qt];
                const bool p1 = fma_plain(__int_as_float(acc[qt][1]), iwr[1], c2r[1]) >= gthr[qt];
                const bool p2 = fma_plain(__int_as_float(acc[qt][2]), iwr[2], c2r[2]) >= gthr[qt];
                const bool p3 = fma_plain(__int_as_float(acc[qt][3]), iwr[3], c2r[3]) >= gthr[qt];
                bool pass = hit[qt] & (p0 | p1 | p2 | p3);
#endif
                pass = pass & (q < n_q);
                const uint64_t pm = __builtin_amdgcn_ballot_w64(pass);
                if (pm != 0) {
                    const uint32_t slot = wq_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                    if (pass) {
                        if (slot < (uint32_t)QSEG) {
                            const uint32_t e = (uint32_t)wave * QSEG + slot;
                            s_qacc[e] = acc[qt];
                            s_qq[e] = (uint32_t)q;
                            s_qrow[e] = row0;
                        } else {
                            // segment full: only a burst of (near-)duplicates of a query inside a step or two does
                            // that.  Mark the list as overflowed; the query takes the exhaustive pass.
                            atomicOr(&cand_cnt[q], 0x80000000u);
                        }
                    }
                    wq_cnt += (uint32_t)__builtin_popcountll(pm);
                }
            }
            if (lane == 0) {
                s_qn[wave] = wq_cnt < (uint32_t)QSEG ? wq_cnt : (uint32_t)QSEG;
                if (wq_cnt >= (uint32_t)(QSEG / 2)) s_qflag = 1u;
            }
        }
    };
    // drain the survivor queue (called by all threads between two barriers)
    auto drain = [&]() __attribute__((always_inline)) {
        for (int w = 0; w < NWQ; ++w) {
            const uint32_t n = s_qn[w];
            for (uint32_t i = tid; i < 4 * n; i += NWQ * WAVE) {
                const uint32_t e = (uint32_t)w * QSEG + (i >> 2), r = i & 3;
                const uint32_t row = s_qrow[e] + r;
                if ((uint64_t)row < n_rows) retest_append(reinterpret_cast<const int *>(&s_qacc[e])[r], (int)s_qq[e], row);
            }
        }
        __syncthreads();
        if (tid < NWQ) s_qn[tid] = 0;
        if (tid == 0) s_qflag = 0;
        wq_cnt = 0;
        __syncthreads();
    };
#ifdef PB_MQ_STAMP
    unsigned long long st_c = 0, st_s = 0, st_b = 0, st_n = 0;
#endif
    // one step: loads for step + 2 go out, step is computed from LDS buffer `buf`, step + 1 (requested one step ago)
    // is staged into the other buffer, barrier
    auto step = [&](uint64_t stp, int buf, u32x4 (&ld_far)[LPT], int &sb_far, int &db_far, const u32x4 (&ld_near)[LPT], int sb_near,
                    int db_near) __attribute__((always_inline)) {
        const uint64_t s1 = stp + gridDim.x, s2 = s1 + gridDim.x;
#ifdef PB_MQ_STAMP
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        // the drain request of the step before (written before the barrier that ended it): read here, beside the first operand reads,
        // and acted on after block 0 -- the first point of the step at which a test could append -- so the step does not begin with
        // two LDS round trips one after the other (flag, then operands)
        const uint32_t dflag = s_qflag;
        if (s1 < n_steps) stage_factors(s1, buf ^ 1, sb_near, db_near);
        i32x4 acc0[QT], acc1[QT];
        i32x4 ae[4], ao[4];  // A fragments of the even / odd tile in flight
        float fe[2], fo[2];  // factors of the even / odd tile under test
        const uint8_t *tb = &s_tile[buf][0];
        if (active) {
            load_a(tb, ae);
            // ld_far's registers were staged one step ago: free again
            if (s2 < n_steps) issue(s2, ld_far, sb_far, db_far);
            load_a(tb + 16 * WROW, ao);
            load_f(buf, 0, fe);
            mfma_a(ae, acc0);  // block 0
        } else if (s2 < n_steps) {
            issue(s2, ld_far, sb_far, db_far);
        }
        if (dflag != 0u) drain();  // uniform over the workgroup
        if (active) {
            // The step as a pipeline of NT blocks, block t = "MFMAs of tile t, with the survivor test of tile t - 1 in their shadow",
            // each block straight-line up to its rarely-taken re-test branch; block t also requests tile t + 1's A fragments and tile
            // t's factors from LDS, so every block finds its operands in registers.  (Until round 6 the loop body was [MFMA t + 1 |
            // test t] [MFMA t + 2, conditional] [test t + 1]: every second test ran as a block of its own -- vector work with the
            // matrix pipe idle, in both waves of a SIMD at once, the barrier keeps them in step -- and every tile's MFMAs began with
            // a wait for their own ds_reads.)
#if !defined(PB_MQ_ABL) || (PB_MQ_ABL != 1 && PB_MQ_ABL != 3)  // ablation 1 / 3: no survivor tests (timing only: nothing is collected)
#define PB_MQ_TEST(ACC, TL, F) test_tile(ACC, buf, TL, stp, F)
#else
#define PB_MQ_TEST(ACC, TL, F) \
    if ((ACC[0][0] ^ ACC[1][1] ^ ACC[2][2] ^ ACC[3][3]) == 0x7fffffff) s_qflag = 1
#endif
            // two blocks per trip and NOT unrolled further: with all NT tiles (and their rarely-taken re-test
            // blocks) unrolled the loop body outgrows the instruction cache and every step refetches it
#pragma nounroll
            for (int tp = 0; tp < NT / 2 - 1; ++tp) {
                load_a(tb + (16 * (2 * tp + 2)) * WROW, ae);
                load_f(buf, 2 * tp + 1, fo);
                mfma_a(ao, acc1);  // block 2 tp + 1
                PB_MQ_TEST(acc0, 2 * tp, fe);
                load_a(tb + (16 * (2 * tp + 3)) * WROW, ao);
                load_f(buf, 2 * tp + 2, fe);
                mfma_a(ae, acc0);  // block 2 tp + 2
                PB_MQ_TEST(acc1, 2 * tp + 1, fo);
            }
            load_f(buf, NT - 1, fo);
            mfma_a(ao, acc1);  // block NT - 1
            PB_MQ_TEST(acc0, NT - 2, fe);
            PB_MQ_TEST(acc1, NT - 1, fo);
#undef PB_MQ_TEST
        }
#ifdef PB_MQ_STAMP
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
        if (s1 < n_steps) stage_tile(buf ^ 1, ld_near);
#ifdef PB_MQ_STAMP
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef PB_MQ_STAMP
        const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
        st_c += ts1 - ts0; st_s += ts2 - ts1; st_b += ts3 - ts2; st_n += 1;
#endif
    };
    uint64_t st = blockIdx.x;
    if (st < n_steps) {
        issue(st, ld[0], ld_sb[0], ld_db[0]);
        stage(st, 0, ld[0], ld_sb[0], ld_db[0]);
        if (st + gridDim.x < n_steps) issue(st + gridDim.x, ld[1], ld_sb[1], ld_db[1]);
    }
    __syncthreads();
    // step i computes from buffer i & 1; its "near" registers (step i + 1) are slot (i + 1) & 1, "far" slot i & 1
    while (st < n_steps) {
        step(st, 0, ld[0], ld_sb[0], ld_db[0], ld[1], ld_sb[1], ld_db[1]);
        st += gridDim.x;
        if (st >= n_steps) break;
        step(st, 1, ld[1], ld_sb[1], ld_db[1], ld[0], ld_sb[0], ld_db[0]);
        st += gridDim.x;
    }
    drain();
#ifdef PB_MQ_STAMP
    if (lane == 0 && active) {
        atomicAdd(&g_mq_stamp[0], st_c); atomicAdd(&g_mq_stamp[1], st_s); atomicAdd(&g_mq_stamp[2], st_b); atomicAdd(&g_mq_stamp[3], st_n);
    }
#endif
}

// per query: tau = lower edge of the highest histogram bin at which the sampled count reaches `target_sample`
// (never below thr0); one wave per query.
// A bin that holds far more than the target (a table full of near-duplicates of the query: all of them land in one bin)
// would put more rows above tau than a candidate list holds, and the collect pass would spend its time queueing
// survivors of lists that overflow anyway (23 ms instead of 1.5 ms per 512 queries on the end-to-end table).  Such a
// query is GATED: tau = 2 + h, unreachable for the collect pass (cosines are <= 1), where h in [0, 1) hands the second
// chance its starting point -- h = (lower edge of the bin at which the SAMPLED count reaches k) - m.  The k-th largest
// cos_filter of a subset is <= the k-th largest of the whole table, so at least k rows have an exact cosine >= h: h is a
// lower bound of the true k-th exact cosine, which is all the second chance needs (pb_scan_kernels.h (4)); h = 0: none.
__global__ void k_mq_pick_tau(const uint32_t *__restrict__ ghist, const QParams *__restrict__ qp, int n_q,
                              uint32_t target_sample, float *__restrict__ tau) {
    const int q = blockIdx.x;
    if (q >= n_q) return;
    const int lane = lane_id();
    const uint32_t k = qp[q].k;
    uint32_t acc = 0, cum = 0;
    int found = -1, found_k = -1;
    for (int chunk = MQ_BINS / WAVE - 1; chunk >= 0 && (found < 0 || found_k < 0); --chunk) {
        const int bin = chunk * WAVE + (WAVE - 1 - lane);
        uint32_t incl = ghist[(size_t)q * MQ_BINS + bin];
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (found < 0) {
            const uint64_t hit = __ballot(acc + incl >= target_sample);
            if (hit) {
                const int first = __ffsll((unsigned long long)hit) - 1;
                found = chunk * WAVE + (WAVE - 1 - first);
                cum = acc + __shfl(incl, first);  // sampled rows at or above the chosen edge
            }
        }
        if (found_k < 0) {
            const uint64_t hit = __ballot(acc + incl >= k);
            if (hit) found_k = chunk * WAVE + (WAVE - 1 - (__ffsll((unsigned long long)hit) - 1));
        }
        acc += __shfl(incl, WAVE - 1);
    }
    if (lane == 0) {
        float t = found >= 0 ? mq_bin_edge(found) - 2e-6f : 0.0f;
        // never <= 0: the collect pass's group bound (4 max acc' + max cr) / min W bounds a row's value only while that
        // numerator is non-negative, i.e. for thresholds above zero (thr0 is >= 2 M + 2e-6 by construction, make_qparams;
        // the clamp states it here: a query that needs rows with a cosine below it ends in the exhaustive pass through
        // its failed certificate)
        t = fmaxf(t, fmaxf(qp[q].thr0, 1e-6f));
        if ((uint64_t)cum * MQ_SAMPLE > (uint64_t)MQ_CAP) {
            float h = found_k >= 0 ? mq_bin_edge(found_k) - 4e-6f - qp[q].m : 0.0f;
            h = h > 0.0f && h < 0.999f ? h : 0.0f;
            t = 2.0f + h;
        }
        tau[q] = t;
    }
}

// Seeds of the per-query starting threshold of the looped filter launch (round 6).  k_scan_filter starts every wave at thr0 -- the
// max_dist floor, which half of a uniform table passes -- and raises it by pruning its buffer (wave_keep_smallest: ~2 us of vector
// work in an HBM-bound loop, ~6 times per wave and pass on uniform data, more on clustered data: the embedding-like table streamed
// 6 % slower for it).  The k-th largest cosine of ANY subset of the table is a lower bound on the k-th largest of the whole, so the
// 1/32 sample's histogram (one shared read of the sample for all queries of the launch: k_scan_multi<.., HIST>) gives every query a
// threshold only ~k * 32 rows of the whole table pass: bin edge of the sample's k-th largest value, less `slack` (the sample pass
// forms its cosine with two reciprocal square roots instead of one: a few ulp) and four error margins.  Efficiency only: the seed
// becomes the query's thr0, which the certificate of k_select_rescore counts among the bounds on unexamined rows (o_max), so a
// seed that were too high could only fail the certificate, never pass a wrong list.  A sample with fewer than k rows above the
// old thr0 leaves it alone.
__global__ void k_seed_thr(const uint32_t *__restrict__ ghist, QParams *__restrict__ qp, int n_q, float slack) {
    const int q = blockIdx.x;
    if (q >= n_q) return;
    const int lane = lane_id();
    const uint32_t k = qp[q].k;
    uint32_t acc = 0;
    int found = -1;
    for (int chunk = MQ_BINS / WAVE - 1; chunk >= 0 && found < 0; --chunk) {
        const int bin = chunk * WAVE + (WAVE - 1 - lane);
        uint32_t incl = ghist[(size_t)q * MQ_BINS + bin];
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        const uint64_t hit = __ballot(acc + incl >= k);
        if (hit) found = chunk * WAVE + (WAVE - 1 - (__ffsll((unsigned long long)hit) - 1));
        acc += __shfl(incl, WAVE - 1);
    }
    if (lane == 0 && found > 0) {
        const float t = mq_bin_edge(found) - slack - 4.0f * qp[q].m;
        if (t > qp[q].thr0 && t < 0.9999f) qp[q].thr0 = t;
    }
}

// exact re-scoring of one query's candidate list (<= MQ_CAP), sort, top-k, certificate.  One block per query.
__global__ __launch_bounds__(1024) void k_mq_rescore(
    const uint8_t *__restrict__ rows, const int64_t *__restrict__ ids, const float *__restrict__ norms, int d,
    const uint8_t *__restrict__ queries, const QParams *__restrict__ qp, const float *__restrict__ lut,
    const float *__restrict__ tau, const uint64_t *__restrict__ cand, const uint32_t *__restrict__ cand_cnt,
    int64_t *__restrict__ out_ids, float *__restrict__ out_dist, ResultHdr *__restrict__ out_hdr, uint32_t out_stride) {
    constexpr int PER = MQ_CAP / 1024;
    __shared__ float s_lut[256];
    __shared__ float s_qf[256];
    __shared__ uint64_t s_sort[MQ_CAP];
    __shared__ float s_red[16];
    __shared__ float s_red2[16];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const QParams P = qp[q];
    for (int i = tid; i < 256; i += 1024) s_lut[i] = lut[i];
    __syncthreads();
    for (int i = tid; i < d; i += 1024) s_qf[i] = s_lut[queries[(size_t)q * d + i]];
    const uint32_t raw = cand_cnt[q];  // bit 31: the collect pass dropped candidates (survivor queue overflow)
    const uint32_t listed = raw & 0x7FFFFFFFu;
    const int cnt = listed < (uint32_t)MQ_CAP ? (int)listed : MQ_CAP;
    int nsort = 64;
    while (nsort < cnt) nsort <<= 1;
    __syncthreads();
    uint64_t xkey[PER];
    float xcs[PER];
    float cfilt = -2.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + j * 1024;
        xkey[j] = ~0ull;
        xcs[j] = 3.0f;
        if (i < cnt) {
            const uint32_t r = (uint32_t)cand[(size_t)q * MQ_CAP + i];
            const float dot = ref_fold_dot_any(rows + (uint64_t)r * d, s_qf, s_lut, d);
            float cs;
            const float dist = ref_distance(dot, P.sqrt_sa, norms[r], &cs);
            xcs[j] = cs;
            if ((double)dist < P.max_dist) xkey[j] = ((uint64_t)sortable_f32(dist) << 32) | r;
            else cfilt = fmaxf(cfilt, cs);
        }
        if (i < nsort) s_sort[i] = xkey[j];
    }
    block_bitonic_sort(s_sort, nsort);
    // number of valid keys = position of the first ~0 (sorted): count by ballot
    __shared__ uint32_t s_nvalid;
    if (tid == 0) s_nvalid = 0;
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + j * 1024;
        mine += (i < nsort && s_sort[i] != ~0ull) ? 1u : 0u;
    }
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor((int)mine, off);
    if ((tid & 63) == 0 && mine) atomicAdd(&s_nvalid, mine);
    __syncthreads();
    const uint32_t n_valid = s_nvalid;
    const uint32_t n_out = n_valid < P.k ? n_valid : P.k;
    if (tid < (int)n_out) {
        const uint64_t key = s_sort[tid];
        out_ids[(size_t)q * out_stride + tid] = ids[(uint32_t)key];
        out_dist[(size_t)q * out_stride + tid] = unsortable_f32((uint32_t)(key >> 32));
    }
    // smallest exact cosine among the selected, largest among the rejected-by-filter
    const uint64_t kth_key = n_out ? s_sort[n_out - 1] : 0ull;
    float ck = 3.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (n_out && xkey[j] <= kth_key) ck = fminf(ck, xcs[j]);
    for (int off = 32; off >= 1; off >>= 1) {
        ck = fminf(ck, __shfl_xor(ck, off));
        cfilt = fmaxf(cfilt, __shfl_xor(cfilt, off));
    }
    if ((tid & 63) == 0) {
        s_red[tid >> 6] = ck;
        s_red2[tid >> 6] = cfilt;
    }
    __syncthreads();
    if (tid == 0) {
        ck = 3.0f;
        cfilt = -2.0f;
        for (int w = 0; w < 16; ++w) {
            ck = fminf(ck, s_red[w]);
            cfilt = fmaxf(cfilt, s_red2[w]);
        }
        const float o_max = fmaxf(tau[q], P.thr0) + P.m;  // no unlisted row's exact cosine reaches this
        bool ok = raw <= (uint32_t)MQ_CAP;
        if (n_out == P.k) ok = ok && (o_max <= ck * (1.0f - 1e-6f));
        else ok = ok && ((P.floor_is_filter && o_max <= P.c_floor) || (o_max <= cfilt));
        ResultHdr h;
        h.count = n_out;
        h.status = ok ? 0u : 1u;
        h.n_cand = raw;
        h.o_max = o_max;
        h.ck = n_out == P.k ? ck : -1.0f;
        if (tau[q] >= 2.0f) {  // gated by k_mq_pick_tau: nothing was collected; hand on its lower bound of the k-th cosine
            const float hint = tau[q] - 2.0f - 1e-6f;
            h.status = 1u;
            h.ck = hint > 0.0f ? hint : -1.0f;
        }
        out_hdr[q] = h;
    }
}

// ------------------------------------------------------------------------------------------------
// (4) second chance for a query whose first attempt FOUND k results but could not certify them (clustered
// tables: more rows within the margin of the k-th cosine than the candidate lists hold).  Let ck1 be the smallest
// exact cosine among those k results.  The true k-th cosine is >= ck1, so every row of the true top-k has
// cos_ref >= ck1, hence cos_filter >= ck1 - m.  The collect pass (k_scan_multi, HIST = false) is re-run with
// tau2 = ck1 (1 - 1e-6) - 1.01 m and a 64 Ki-entry list per query, and k_mq_rescore_big scores EVERY listed row with
// the reference arithmetic and selects the k smallest (dist, id): every unlisted row has
// cos_ref < tau2 + m < ck1 (1 - 1e-6) <= c_k (1 - 1e-6), i.e. a strictly larger distance than the k-th result (same
// argument as the certificate) -- the result is exact without a certificate, provided the list did
// not overflow (then: exhaustive pass).  Cost: one sweep of the table shared by up to 64 such queries plus the
// re-scoring of the listed rows, instead of an exhaustive exact sweep per query.
constexpr int MQ_CAP2 = 65536;

// queries[sel[i]], qp[sel[i]] -> contiguous copies (the collect kernel wants its queries in one block)
__global__ void k_gather_queries(const uint8_t *__restrict__ queries, const QParams *__restrict__ qp,
                                 const uint32_t *__restrict__ sel, int d, uint8_t *__restrict__ out_q,
                                 QParams *__restrict__ out_qp) {
    const int i = blockIdx.x;
    const uint32_t s = sel[i];
    for (int j = threadIdx.x; j < d; j += blockDim.x) out_q[(size_t)i * d + j] = queries[(size_t)s * d + j];
    if (threadIdx.x == 0) out_qp[i] = qp[s];
}

// second-chance gate: queries whose sampled count of rows above tau2 (bin 0) predicts a list overflow get a
// threshold no row reaches
__global__ void k_sc_gate(const uint32_t *__restrict__ ghist, int n_q, uint32_t max_sampled, float *__restrict__ tau2) {
    for (int q = threadIdx.x; q < n_q; q += blockDim.x)
        if (ghist[(size_t)q * MQ_BINS] > max_sampled) tau2[q] = 2.0f;
}

// one block per (compacted) query i; results go to slot sel[i]
__global__ __launch_bounds__(1024) void k_mq_rescore_big(
    const uint8_t *__restrict__ rows, const int64_t *__restrict__ ids, const float *__restrict__ norms, int d,
    const uint8_t *__restrict__ queries, const QParams *__restrict__ qp, const float *__restrict__ lut,
    const uint64_t *__restrict__ cand, const uint32_t *__restrict__ cand_cnt, uint32_t cap,
    const uint32_t *__restrict__ sel, int64_t *__restrict__ out_ids, float *__restrict__ out_dist,
    ResultHdr *__restrict__ out_hdr, uint32_t out_stride) {
    constexpr int CH = 4096, PER = CH / 1024;
    __shared__ float s_lut[256];
    __shared__ float s_qf[256];  // dim 256 only (as the collect pass)
    __shared__ uint64_t s_sort[CH];
    __shared__ uint64_t s_best[2 * (int)PB_MAX_K];
    __shared__ uint32_t s_nvalid;
    const int q = blockIdx.x;
    const uint32_t oq = sel[q];
    const int tid = threadIdx.x;
    const QParams P = qp[q];
    for (int i = tid; i < 256; i += 1024) s_lut[i] = lut[i];
    for (int i = tid; i < 2 * (int)PB_MAX_K; i += 1024) s_best[i] = ~0ull;
    if (tid == 0) s_nvalid = 0;
    __syncthreads();
    for (int i = tid; i < d; i += 1024) s_qf[i] = s_lut[queries[(size_t)q * d + i]];
    const uint32_t raw = cand_cnt[q];
    if (raw > cap) {  // list overflow: nothing to gain from scoring a truncated list (uniform branch)
        if (tid == 0) {
            ResultHdr h;
            h.count = 0;
            h.status = 1u;
            h.n_cand = raw;
            h.o_max = 0.0f;
            h.ck = -1.0f;
            out_hdr[oq] = h;
        }
        return;
    }
    const uint32_t cnt = raw;
    __syncthreads();
    for (uint32_t base = 0; base < cnt; base += CH) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t i = tid + j * 1024;
            uint64_t key = ~0ull;
            if (base + i < cnt) {
                const uint32_t r = (uint32_t)cand[(size_t)q * cap + base + i];
                const float dot = ref_fold_dot_any(rows + (uint64_t)r * d, s_qf, s_lut, d);
                float cs;
                const float dist = ref_distance(dot, P.sqrt_sa, norms[r], &cs);
                if ((double)dist < P.max_dist) key = ((uint64_t)sortable_f32(dist) << 32) | r;
            }
            s_sort[i] = key;
        }
        block_bitonic_sort(s_sort, CH);
        // the chunk's best PB_MAX_K join the running best PB_MAX_K
        if (tid < (int)PB_MAX_K) s_best[PB_MAX_K + tid] = s_sort[tid];
        block_bitonic_sort(s_best, 2 * PB_MAX_K);
    }
    if (tid < (int)PB_MAX_K && s_best[tid] != ~0ull) atomicAdd(&s_nvalid, 1u);
    __syncthreads();
    const uint32_t n_valid = s_nvalid;
    const uint32_t n_out = n_valid < P.k ? n_valid : P.k;
    if (tid < (int)n_out) {
        const uint64_t key = s_best[tid];
        out_ids[(size_t)oq * out_stride + tid] = ids[(uint32_t)key];
        out_dist[(size_t)oq * out_stride + tid] = unsortable_f32((uint32_t)(key >> 32));
    }
    if (tid == 0) {
        ResultHdr h;
        h.count = n_out;
        h.status = (raw <= cap && n_out == P.k) ? 0u : 1u;  // fewer than k: leave it to the exhaustive pass
        h.n_cand = raw;
        h.o_max = 0.0f;
        h.ck = -1.0f;
        out_hdr[oq] = h;
    }
}

// single-query calls: the query and its constants arrive as kernel arguments and are put where every kernel of
// the search path expects them (slot 0 of the staged query / parameter arrays)
__global__ void k_stage_query(const QArg a, uint8_t *__restrict__ d_queries, QParams *__restrict__ d_qp) {
    for (uint32_t i = threadIdx.x; i < a.dim; i += blockDim.x) d_queries[i] = a.q[i];
    if (threadIdx.x == 0) d_qp[0] = a.p;
}

// results -> the all-gather message: packed[q][0..k) = ids, [k..2k) = dist bits (zero-extended), [2k] = count
__global__ void k_pack_results(const int64_t *__restrict__ ids, const float *__restrict__ dist,
                               const ResultHdr *__restrict__ hdr, uint32_t res_stride, uint32_t k,
                               int64_t *__restrict__ packed) {
    const uint32_t q = blockIdx.x;
    int64_t *row = packed + (size_t)q * (2 * k + 1);
    const uint32_t c = hdr ? hdr[q].count : 0u;  // hdr == nullptr: an empty shard's message
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        const bool v = i < c;
        row[i] = v ? ids[(size_t)q * res_stride + i] : INT64_MAX;
        row[k + i] = v ? (int64_t)__float_as_uint(dist[(size_t)q * res_stride + i]) : (int64_t)0x7F800000u;
    }
    if (threadIdx.x == 0) row[2 * k] = (int64_t)c;
}
// results -> plain device arrays (pb_index_search_device): ids[q][k], dist[q][k], count[q]; unused slots hold
// id = INT64_MAX, dist = +inf
__global__ void k_export_results(const int64_t *__restrict__ ids, const float *__restrict__ dist,
                                 const ResultHdr *__restrict__ hdr, uint32_t res_stride, uint32_t k,
                                 int64_t *__restrict__ out_ids, float *__restrict__ out_dist, uint32_t *__restrict__ out_count) {
    const uint32_t q = blockIdx.x;
    const uint32_t c = hdr ? hdr[q].count : 0u;
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        const bool v = i < c;
        out_ids[(size_t)q * k + i] = v ? ids[(size_t)q * res_stride + i] : INT64_MAX;
        out_dist[(size_t)q * k + i] = v ? dist[(size_t)q * res_stride + i] : __uint_as_float(0x7F800000u);
    }
    if (threadIdx.x == 0) out_count[q] = c;
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t w) {
    uint64_t z = seed + (w + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// out[0..n_words) = words [first_word, ..) of stream `seed` (8 bytes per thread-iteration)
__global__ void k_fill_synth(uint64_t seed, uint64_t first_word, uint64_t n_words, uint64_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
         i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = splitmix64_at(seed, first_word + i);
}
// synthetic RGB8 images (pixelbox_amd/synth.py:synthetic_images): word i of the output holds bytes 8i..8i+7 of the
// pixel stream starting at image `start`; per = h*w*3 bytes per image
__global__ void k_fill_synth_images(uint64_t seed, uint64_t start, uint64_t per, uint64_t n_words, uint64_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t g0 = start * per + 8 * i;  // global byte index of this word's first byte
        const uint64_t noise = splitmix64_at(seed, g0 >> 3);
        uint64_t o = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t g = g0 + b;
            const uint64_t img = g / per, c = g % 3;  // per % 3 == 0: channel = byte index mod 3
            const uint64_t z = splitmix64_at(seed ^ 0xC0FFEEull, img * 3 + c);
            const uint32_t lo = (uint32_t)(z & 0xFF) * 3 / 4;
            const uint32_t span = (uint32_t)((z >> 8) & 0xFF) / 4 + 1;
            const uint32_t nz = (uint32_t)(noise >> (8 * b)) & 0xFF;
            uint32_t px = lo + ((nz * span) >> 8);
            px = px > 255 ? 255 : px;
            o |= (uint64_t)px << (8 * b);
        }
        out[i] = o;
    }
}
// structured synthetic RGB8 images (pixelbox_amd/synth.py:synthetic_scenes): the same noise stream, a brightness window per
// (image, cell of a grid x grid partition, channel); w3 = w * 3 bytes per image row, ch / cw = cell height / width in pixels
__global__ void k_fill_synth_scenes(uint64_t seed, uint64_t start, uint64_t per, uint64_t n_words, uint32_t w3, uint32_t ch, uint32_t cw,
                                    uint32_t grid, uint64_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t g0 = start * per + 8 * i;
        const uint64_t noise = splitmix64_at(seed, g0 >> 3);
        uint64_t o = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t g = g0 + b;
            const uint64_t img = g / per;
            const uint32_t r = (uint32_t)(g - img * per);  // byte inside the image
            const uint32_t y = r / w3, xc = r - y * w3, x = xc / 3, c = xc - 3 * x;
            const uint64_t cell = ((img * grid + y / ch) * grid + x / cw) * 3 + c;
            const uint64_t z = splitmix64_at(seed ^ 0xC0FFEEull, cell);
            const uint32_t lo = (uint32_t)(z & 0xFF) * 3 / 4;
            const uint32_t span = (uint32_t)((z >> 8) & 0xFF) / 4 + 1;
            const uint32_t nz = (uint32_t)(noise >> (8 * b)) & 0xFF;
            uint32_t px = lo + ((nz * span) >> 8);
            px = px > 255 ? 255 : px;
            o |= (uint64_t)px << (8 * b);
        }
        out[i] = o;
    }
}
// out[i] = src[perm[i]] for i < n: the gather step of an out-of-order insert (rows of `elt` bytes, 16-byte pieces when
// elt % 16 == 0, else 4-byte or single bytes); perm holds positions relative to `src`
__global__ void k_gather_elts(const uint8_t *__restrict__ src, const uint32_t *__restrict__ perm, uint64_t n, uint32_t elt,
                              uint8_t *__restrict__ dst) {
    const uint32_t piece = (elt % 16 == 0) ? 16u : ((elt % 4 == 0) ? 4u : 1u);
    const uint32_t per = elt / piece;
    const uint64_t total = n * per;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / per;
        const uint32_t c = (uint32_t)(i % per);
        const uint8_t *s = src + (uint64_t)perm[r] * elt + (uint64_t)c * piece;
        uint8_t *t = dst + r * elt + (uint64_t)c * piece;
        if (piece == 16) *reinterpret_cast<uint4 *>(t) = *reinterpret_cast<const uint4 *>(s);
        else if (piece == 4) *reinterpret_cast<uint32_t *>(t) = *reinterpret_cast<const uint32_t *>(s);
        else *t = *s;
    }
}
__global__ void k_iota_ids(int64_t first_id, uint64_t n, int64_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = first_id + (int64_t)i;
}

}  // namespace pbk
