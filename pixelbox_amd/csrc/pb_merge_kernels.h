// pb_merge_kernels.h -- device-side merge of the all-gathered per-shard top-k messages (gfx950).
//
// Input: gathered[g][q][2k+1] int64 as written by pb_index_search_packed on every shard: [0..k) image_ids, [k..2k) the f32
// distance bits (zero-extended), [2k] the count; each list sorted by (dist, image_id) ascending.  Output: the first k of
// the merged order per query -- the `ORDER BY dist ASC LIMIT k` of engine.rs:375-381 over the union of the shards, ties
// by image_id (the rowid order SQLite's scan emits).  Image ids are unique across shards (INSERT OR IGNORE is enforced
// over all shards), so the order is strict.
//
// One workgroup per query.  Every listed entry computes its final rank directly: its position in its own list plus,
// for every other list, the number of entries that precede it (a binary search: the lists are sorted) -- no
// compare-exchange network, no atomics; entries with rank < k store themselves.  Latency-bound and tiny: 8 lists x 100
// entries = 800 ranks of 7 x 7 probes from LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pbm {

constexpr int MERGE_BLOCK = 256;
constexpr int MERGE_LDS_ENTRIES = 2048;  // lists x k staged in LDS (8 shards x k = 256); larger unions probe global memory

__device__ __forceinline__ bool before(float da, int64_t ia, float db, int64_t ib) { return da < db || (da == db && ia < ib); }

__global__ __launch_bounds__(MERGE_BLOCK) void k_merge_packed(const int64_t *__restrict__ gathered, uint32_t n_lists, uint32_t nq,
                                                             uint32_t k, int64_t *__restrict__ out_ids, float *__restrict__ out_dist,
                                                             uint32_t *__restrict__ out_count) {
    __shared__ int64_t s_id[MERGE_LDS_ENTRIES];
    __shared__ float s_d[MERGE_LDS_ENTRIES];
    __shared__ uint32_t s_cnt[64];
    const uint32_t q = blockIdx.x;
    const size_t row = 2 * (size_t)k + 1;
    const bool in_lds = n_lists * k <= (uint32_t)MERGE_LDS_ENTRIES;
    for (uint32_t g = threadIdx.x; g < n_lists; g += MERGE_BLOCK) {
        const uint32_t c = (uint32_t)gathered[((size_t)g * nq + q) * row + 2 * k];
        s_cnt[g] = c < k ? c : k;
    }
    if (in_lds)
        for (uint32_t e = threadIdx.x; e < n_lists * k; e += MERGE_BLOCK) {
            const uint32_t g = e / k, i = e % k;
            const int64_t *p = gathered + ((size_t)g * nq + q) * row;
            s_id[e] = p[i];
            s_d[e] = __uint_as_float((uint32_t)p[k + i]);
        }
    __syncthreads();
    auto id_at = [&](uint32_t g, uint32_t i) -> int64_t { return in_lds ? s_id[g * k + i] : gathered[((size_t)g * nq + q) * row + i]; };
    auto d_at = [&](uint32_t g, uint32_t i) -> float {
        return in_lds ? s_d[g * k + i] : __uint_as_float((uint32_t)gathered[((size_t)g * nq + q) * row + k + i]);
    };
    uint32_t total = 0;
    for (uint32_t g = 0; g < n_lists; ++g) total += s_cnt[g];
    for (uint32_t e = threadIdx.x; e < n_lists * k; e += MERGE_BLOCK) {
        const uint32_t g = e / k, i = e % k;
        if (i >= s_cnt[g]) continue;
        const float de = d_at(g, i);
        const int64_t ie = id_at(g, i);
        uint32_t rank = i;
        for (uint32_t h = 0; h < n_lists; ++h) {
            if (h == g) continue;
            // entries of list h that come before e; an exact tie (same distance AND id: cannot happen with unique ids)
            // is resolved by list order so that ranks stay distinct
            uint32_t lo = 0, hi = s_cnt[h];
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const float dm = d_at(h, mid);
                const int64_t im = id_at(h, mid);
                const bool prec = before(dm, im, de, ie) || (dm == de && im == ie && h < g);
                if (prec) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_ids[(size_t)q * k + rank] = ie;
            out_dist[(size_t)q * k + rank] = de;
        }
    }
    if (threadIdx.x == 0) out_count[q] = total < k ? total : k;
}

}  // namespace pbm
