// pb_se_project.h -- squeeze-excite + project 1x1 of one MBConv block in ONE launch, for a handful of images (the query-time
// `mlhash` of the reference: batch 1, src/image_hashes/efficientnet.rs:31-42; src/engine.rs:352-361 prints its latency).
// Included by pb_embed.hip after pb_embed_kernels.h and pb_gemm_p3.h.
//
// At batch 1 a forward pass is a chain of ~50 dependent launches of 5-11 us each (three dependent memory round trips per
// kernel: profiles/r03_embed_layers_batch1.txt); sixteen of them are k_se, whose gate the project GEMM behind it needs before
// its first multiply.  Here a workgroup first computes the gate of ITS image -- se_body, the body of k_se: same arithmetic in
// the same order -- into LDS, then each of its waves runs one 16-row x 16-channel tile of the project GEMM with the gate read
// from LDS: the one-wave forms of the tiled kernels (k_gemm_thin for an f32-chain layer, the DIRECT form of k_gemm_p3 for a
// P3 layer) with the same operand maps, the same k order and the same epilogue -- the block's output is bit-identical to
// k_se + any project form.  Workgroups of one image recompute its gate (a few hundred KB of excite weights from L2 each).
// grid = (ceil(tiles per image / waves per workgroup), images); block = 64 x waves, >= the threads se_body needs (QP x groups).
// ARCHIVED EXPERIMENT (round 4, not part of the library): needs k_se's body as a device function `se_body<SP, IMG>(part, n_tiles,
// E, inv_hw, w1, b1, w2t, b2, gate, gate_img0, QP, n_img, b0)` whose threads beyond QP x groups only take part in the barriers.
// Result: profiles/r04_se_project_experiment.txt (slower than two launches at every layer).
#pragma once

namespace pbe {

struct SeProjArgs {
    // squeeze-excite (as k_se)
    const long long *part;
    int n_tiles;
    float inv_hw;
    const float *w1, *b1, *w2t, *b2;
    int QP;
    // project GEMM: out[m][n] = sum_k act[m][k] gate[k] w[k][n] + bias[n] (+ resid[m][n])
    const float *act;   // [images][hw][K]
    int hw, K;
    const float *wt;    // f32-chain layer: [Kpad][Npad]
    int Kpad, Npad;
    const void *wt3;    // P3 layer: fragment order (pb_gemm_p3.h)
    const float *bias;
    int N;
    const float *resid;
    float *out;
    int n_img;
};

template <int SP, bool P3>
__global__ __launch_bounds__(1024) void k_se_project(SeProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_gate[];  // [K]
    const int img = blockIdx.y;
    se_body<SP, 1>(a.part, a.n_tiles, a.K, a.inv_hw, a.w1, a.b1, a.w2t, a.b2, s_gate, 0, a.QP, a.n_img, img);
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = blockDim.x >> 6;
    const int ct_n = a.Npad >> 4, rt_n = (a.hw + 15) >> 4;
    const int t = blockIdx.x * n_waves + wave;  // this wave's tile of the image: row tile t / ct_n, column tile t % ct_n
    if (t >= rt_n * ct_n) return;
    const int rt = t / ct_n, ct = t - rt * ct_n;
    const int K = a.K;
    const int prow = rt * 16 + li;
    const bool mval = prow < a.hw;
    const long mrow = (long)img * a.hw + (mval ? prow : a.hw - 1);
    const float *arow = a.act + mrow * K;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!P3) {
        // ---- k_gemm_thin's loop: k = 16 s + 4 kk + e, weights straight from memory, PD steps in flight (a 1024-thread workgroup
        // leaves a lane 128 registers: k_gemm_thin's ring of 16 steps spilled to scratch and made this kernel 3x slower than
        // the two launches it replaces; the other waves of the workgroup cover the latency instead)
        constexpr int PD = 4;
        const float *wcol = a.wt + ct * 16 + li;
        const int n_steps = a.Kpad / 16;
        f32x4 ar[PD];
        float wr[PD][4];
        auto load_step = [&](int s, int slot) __attribute__((always_inline)) {
            const int kbase = s * 16 + 4 * kk;
            const int kb = kbase < K ? kbase : 0;
            ar[slot] = *reinterpret_cast<const f32x4 *>(arow + kb);
#pragma unroll
            for (int e = 0; e < 4; ++e) wr[slot][e] = wcol[(size_t)(kbase + e) * a.Npad];
        };
        auto k_step = [&](int s, int slot) __attribute__((always_inline)) {
            const int kbase = s * 16 + 4 * kk;
            f32x4 v = ar[slot];
            const f32x4 g = *reinterpret_cast<const f32x4 *>(s_gate + (kbase < K ? kbase : 0));
            v.x = v.x * g.x; v.y = v.y * g.y; v.z = v.z * g.z; v.w = v.w * g.w;
            if (!(mval && kbase < K)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            float w4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w4[e] = wr[slot][e];
            load_step((s + PD < n_steps) ? (s + PD) : (n_steps - 1), slot);  // unconditional: see k_gemm1x1
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[0], v.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[1], v.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[2], v.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[3], v.w, acc, 0, 0, 0);
        };
#pragma unroll
        for (int s = 0; s < PD; ++s) load_step(s < n_steps ? s : n_steps - 1, s);
        int s0 = 0;
        for (; s0 + PD <= n_steps; s0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) k_step(s0 + u, u);
        }
#pragma unroll
        for (int u = 0; u < PD; ++u)
            if (s0 + u < n_steps) k_step(s0 + u, u);
    } else {
        // ---- the one-wave (DIRECT) form of k_gemm_p3: k = 32 s + 8 kk + j, three weight planes per step straight from memory
        constexpr int PD = 2;
        const int n_steps = (K + 31) >> 5;
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(a.wt3) + (size_t)ct * 192 + lane;  // + step * ct_n * 192 + plane * 64
        f32x4 ar[PD][2];
        u32x4 wd[PD][3];
        auto request = [&](int s, int slot) __attribute__((always_inline)) {
            const int sc = s < n_steps ? s : n_steps - 1;
            int kb = 32 * sc + 8 * kk;
            kb = kb < K ? kb : 0;
            ar[slot][0] = *reinterpret_cast<const f32x4 *>(arow + kb);
            ar[slot][1] = *reinterpret_cast<const f32x4 *>(arow + kb + 4);
            const u32x4 *src = wsrc + (size_t)sc * ct_n * 192;
#pragma unroll
            for (int p = 0; p < 3; ++p) wd[slot][p] = src[p * 64];
        };
        auto k_step = [&](int s, int slot) __attribute__((always_inline)) {
            const int kb = 32 * s + 8 * kk;
            f32x4 v0 = ar[slot][0], v1 = ar[slot][1];
            const int kg = kb < K ? kb : 0;
            const f32x4 g0 = *reinterpret_cast<const f32x4 *>(s_gate + kg), g1 = *reinterpret_cast<const f32x4 *>(s_gate + kg + 4);
            v0.x = v0.x * g0.x; v0.y = v0.y * g0.y; v0.z = v0.z * g0.z; v0.w = v0.w * g0.w;
            v1.x = v1.x * g1.x; v1.y = v1.y * g1.y; v1.z = v1.z * g1.z; v1.w = v1.w * g1.w;
            if (kb >= K) {
                v0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                v1 = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const u32x4 wh = wd[slot][0], wm = wd[slot][1], wl = wd[slot][2];
            request(s + PD, slot);
            const P3Act pa = p3_split8(v0, v1);
            p3_step(acc, wh, wm, wl, pa);
        };
#pragma unroll
        for (int s = 0; s < PD; ++s) request(s, s);
        int s0 = 0;
        for (; s0 + PD <= n_steps; s0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) k_step(s0 + u, u);
        }
#pragma unroll
        for (int u = 0; u < PD; ++u)
            if (s0 + u < n_steps) k_step(s0 + u, u);
    }
    const int n = ct * 16 + kk * 4;
    if (!mval || n >= a.N) return;  // N % 4 == 0
    const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + n);
    f32x4 v = acc;
    v.x = v.x + b.x; v.y = v.y + b.y; v.z = v.z + b.z; v.w = v.w + b.w;
    const long orow = (long)img * a.hw + prow;
    if (a.resid) {
        const f32x4 rv = *reinterpret_cast<const f32x4 *>(a.resid + orow * a.N + n);
        v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w;
    }
    *reinterpret_cast<f32x4 *>(a.out + orow * a.N + n) = v;
}

}  // namespace pbe
