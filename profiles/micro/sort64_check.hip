// sort64_check.hip -- the wave-level bitonic networks of pb_scan_kernels.h (wave_sort64, wave_finish_list,
// wave_merge_lists) against std::sort on random keys with ties in the score word, sentinels and short lists.
// hipcc --offload-arch=gfx950 -O3 -I include -I pixelbox_amd/csrc profiles/micro/sort64_check.hip -o /tmp/sort64_check && /tmp/sort64_check
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "pixelbox_hip.h"
#include "pb_scan_kernels.h"
using namespace pbk;

__global__ void k_sort(const uint64_t *in, uint64_t *out) {
    const uint64_t k = in[blockIdx.x * 64 + threadIdx.x];
    out[blockIdx.x * 64 + threadIdx.x] = wave_sort64(k);
}
// 8 waves, wave w has cnt[w] keys at in[(blk * 8 + w) * 64 ..): finish each list, then wave 0 merges
__global__ void k_merge(const uint64_t *in, const int *cnts, uint64_t *out, int *n_out, uint64_t *first_out) {
    __shared__ uint64_t s_buf[8][F_CAPW];
    __shared__ uint64_t s_first[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int cnt = cnts[blockIdx.x * 8 + wave];
    if (lane < cnt) s_buf[wave][lane] = in[(blockIdx.x * 8 + wave) * 64 + lane];
    const uint64_t fo = wave_finish_list(s_buf[wave], cnt);
    if (lane == 0) s_first[wave] = fo;
    __syncthreads();
    if (wave == 0) {
        int total;
        uint64_t f2;
        const uint64_t key = wave_merge_lists<8>(&s_buf[0][0], F_CAPW, &total, &f2);
        for (int w = 0; w < 8; ++w) f2 = s_first[w] < f2 ? s_first[w] : f2;
        if (lane < 32) out[blockIdx.x * 32 + lane] = key;
        if (lane == 0) {
            n_out[blockIdx.x] = total;
            first_out[blockIdx.x] = f2;
        }
    }
}

int main() {
    std::mt19937_64 rng(12345);
    const int NB = 4096;
    std::vector<uint64_t> h(NB * 64), r(NB * 64);
    for (int b = 0; b < NB; ++b)
        for (int i = 0; i < 64; ++i) {
            uint64_t hi = (b % 3 == 0) ? (rng() % 4) : (uint32_t)rng();  // many ties in the score word
            uint64_t k = (hi << 32) | (uint32_t)(b * 64 + i);
            if (b % 5 == 1 && i >= (int)(rng() % 65)) k = ~0ull;
            h[b * 64 + i] = k;
        }
    uint64_t *d_in, *d_out;
    hipMalloc(&d_in, h.size() * 8);
    hipMalloc(&d_out, h.size() * 8);
    hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    k_sort<<<NB, 64>>>(d_in, d_out);
    hipMemcpy(r.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < NB; ++b) {
        std::vector<uint64_t> e(h.begin() + b * 64, h.begin() + b * 64 + 64);
        std::sort(e.begin(), e.end());
        if (!std::equal(e.begin(), e.end(), r.begin() + b * 64)) ++bad;
    }
    std::printf("wave_sort64: %d of %d blocks wrong\n", bad, NB);

    // merge: 8 lists of 0..64 keys
    std::vector<uint64_t> m(NB * 8 * 64);
    std::vector<int> cn(NB * 8);
    for (int b = 0; b < NB; ++b)
        for (int w = 0; w < 8; ++w) {
            int c = (b % 7 == 0) ? (int)(rng() % 5) : (int)(rng() % 65);
            if (b % 11 == 3) c = 64;
            cn[b * 8 + w] = c;
            for (int i = 0; i < 64; ++i) {
                uint64_t hi = (b % 3 == 0) ? (rng() % 4) : (uint32_t)rng();
                m[(b * 8 + w) * 64 + i] = (hi << 32) | (uint32_t)((b * 8 + w) * 64 + i);
            }
        }
    uint64_t *d_m, *d_mo, *d_fo;
    int *d_c, *d_n;
    hipMalloc(&d_m, m.size() * 8);
    hipMalloc(&d_mo, NB * 32 * 8);
    hipMalloc(&d_fo, NB * 8);
    hipMalloc(&d_c, cn.size() * 4);
    hipMalloc(&d_n, NB * 4);
    hipMemcpy(d_m, m.data(), m.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_c, cn.data(), cn.size() * 4, hipMemcpyHostToDevice);
    k_merge<<<NB, 512>>>(d_m, d_c, d_mo, d_n, d_fo);
    std::vector<uint64_t> mo(NB * 32), fo(NB);
    std::vector<int> no(NB);
    hipMemcpy(mo.data(), d_mo, mo.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(fo.data(), d_fo, fo.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(no.data(), d_n, no.size() * 4, hipMemcpyDeviceToHost);
    bad = 0;
    for (int b = 0; b < NB; ++b) {
        std::vector<uint64_t> all;
        for (int w = 0; w < 8; ++w)
            for (int i = 0; i < cn[b * 8 + w]; ++i) all.push_back(m[(b * 8 + w) * 64 + i]);
        std::sort(all.begin(), all.end());
        const int n = (int)std::min<size_t>(32, all.size());
        bool ok = no[b] == n;
        for (int i = 0; i < n && ok; ++i) ok = mo[b * 32 + i] == all[i];
        const uint64_t ef = all.size() > 32 ? all[32] : ~0ull;
        ok = ok && fo[b] == ef;
        if (!ok) ++bad;
    }
    std::printf("finish + merge of 8 lists: %d of %d workgroups wrong\n", bad, NB);
    return bad ? 1 : 0;
}
