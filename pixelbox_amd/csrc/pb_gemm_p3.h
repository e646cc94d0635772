// pb_gemm_p3.h -- the dense contractions of the late layers on the bf16 matrix cores, with f32-grade products
// ("piece arithmetic", P3).  Included by pb_embed.hip after pb_embed_kernels.h.
//
// Why: the f32-input MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate and shares the SIMD's vector issue, and the
// MFMA-bound kernels of the embed half (project / head / FC GEMMs, the whole-block kernel of the 4 x 4 maps) sat at 0.50-0.58 of
// its peak (profiles/r03_embed_layers.txt).  Every f32 value is EXACTLY hi + mid + lo with three bf16 (8 + 8 + 8 significand
// bits, split by truncation), a product a * w is the sum of nine piece products, each exact in f32, and the six leading ones
// are accumulated by v_mfma_f32_16x16x32_bf16; the three dropped ones are <= 2^-23 |a w|.  Measured (profiles/micro/
// mfma_b3_rate.hip -> profiles/r04_mfma_b3_rate.txt): six such MFMAs per 32 k of a 16 x 16 tile take 6 x 19.5 clocks against
// 8 x 34.5 for the f32 form (2.36 x), and the sum's error against the exact value is the f32 chain's own (rms 3.8e-8 vs 4.3e-8
// of sum |a w| at K = 1152) -- the matrix pipe accumulates in f32.
//
// THE ARITHMETIC OF A P3 LAYER (one definition; every kernel form that computes such a layer uses p3_step, so all forms give
// the same bits -- tests/test_embed_gpu.py::test_every_kernel_form_gives_the_same_bits):
//   out[m][n] = epi( S + bias[n] ),  S = the value of an f32 accumulator that starts at 0 and receives, for the k-steps
//   s = 0, 1, ... (32 consecutive k each, zero beyond K) in ascending order, SIX v_mfma_f32_16x16x32_bf16 in this order:
//       (w_lo, a_hi) (w_mid, a_mid) (w_mid, a_hi) (w_hi, a_lo) (w_hi, a_mid) (w_hi, a_hi)
//   where a = fl(act[m][k] * gate[img][k]) (one rounding, as in the f32 forms) or act[m][k], split hi = top 16 bits of a,
//   mid = top 16 bits of fl(a - hi), lo = top 16 bits of fl(a - hi - mid) (both subtractions exact), w split the same way
//   on the host, and lane (li, kk) of the MFMA holds k = 32 s + 8 kk .. + 7 of its row / column.
// The order keeps each weight plane's fragments live for one stretch (lo: 1 pass, mid: 2, hi: 3).
//
// Weights arrive in fragment order: wt3[k-step][16-column tile][plane hi / mid / lo][lane][8 bf16] -- a lane's operand of one
// MFMA is 16 contiguous bytes, a wave's 1 KB; zero beyond K and N.
#pragma once
#include "pb_p3_common.h"

#ifndef PB_P3_ABL
#define PB_P3_ABL 0  // timing experiments only (results invalid): 1 no operand split, 2 no gate, 4 weights never restaged (no staging loads /
#endif               // stores / barrier after the first step), 8 activations never re-requested

namespace pbe {

// ------------------------------------------------------------------------------------------------
// k_gemm_p3: the tiled GEMM of a P3 layer.  Orientation, row clamping, epilogues and grid as k_gemm_t: the MFMA's "A" operand
// is the weight (row = output channel), its "B" operand the activation (column = pixel), a wave owns MR tiles of 16 pixel
// rows x NR tiles of 16 channels, a lane ends with 4 consecutive channels of one pixel.
//  * LDS form (DIRECT = false): the NR x 3 KB of weight fragments of a k-step are shared by the workgroup's NW waves through a
//    double-buffered LDS image in fragment order (staging = straight 16-byte copies; a wave's fragment read is one conflict-free
//    ds_read_b128 per tile and plane), one barrier per k-step (6 NR MR MFMAs = 120 NR MR clocks apart);
//  * DIRECT = true (a few pixel rows: small batches): no LDS, no barrier -- a wave requests its fragments straight from
//    memory PD steps ahead, like k_gemm_thin.
// The activation operand comes from global memory as two float4 per lane and k-step (the 4 kk-lanes of a pixel cover 128
// contiguous bytes) through a register ring PD steps deep; the split (44 vector instructions per 8 values: and / sub / and /
// sub per value + 3 v_perm per pair) and the gate multiplies run between the MFMAs of the same wave: two vector instructions
// per bf16-MFMA slot are hidden (r04_mfma_b3_rate.txt (b)), i.e. from NR MR >= 5 on the split costs nothing.
// KT: K % 32 != 0 (the last step is half empty: K = 240): lanes beyond K read a clamped address and are zeroed.
// EPI as k_gemm_t: 0 bias (+ SiLU) (+ residual); 1 head conv of a 4 x 4 map: bias + SiLU + average pool of the wave's 16 rows;
// 2 final Linear: bias + tanh + u8 quantiser.
// grid = (ceil(M / (16 NW MR)), tiles16 / NR); block = 64 NW.
template <int NR, int MR, bool GATE, int NW, int EPI, bool DIRECT, bool KT>
__global__ __launch_bounds__(64 * NW) void k_gemm_p3(const float *__restrict__ act, int M, int K, const u32x4 *__restrict__ wt3,
                                                    int tiles16, const float *__restrict__ bias, int N,
                                                    const float *__restrict__ gate, int hw, const float *__restrict__ resid,
                                                    int do_silu, float *__restrict__ out, float scale, uint8_t *__restrict__ out_u8) {
    constexpr int NTHR = 64 * NW;
    // k-steps in flight (even: a step's parity is its ring slot's).  The one-wave form is a chain of K / 32 dependent steps fed
    // straight from memory, once per forward at batch 1: with 4 steps in flight a K = 1152 layer took 36 / 4 round trips = 11 us;
    // a lone wave has the whole register file, so the single-tile shape keeps 8 (28 registers per step)
#ifndef PB_P3_DIRECT_PD
#define PB_P3_DIRECT_PD 8  // (round 6: 16 tried -- 28 registers per step in flight: past 256 architectural VGPRs hipcc spills 150-170 registers to scratch)
#endif
    constexpr int PD = DIRECT ? (NR == 1 ? PB_P3_DIRECT_PD : 4) : (NR * MR >= 4 ? 4 : 6);
    static_assert(PD % 2 == 0, "parity of a step = parity of its slot");
    constexpr int FR = NR * 192;                                               // 16-byte pieces per k-step of this block's tiles
    constexpr int WREGS = DIRECT ? 1 : (FR + NTHR - 1) / NTHR;
    constexpr int FRP = WREGS * NTHR;  // LDS buffer pitch: every thread stages WREGS pieces unconditionally (pieces >= FR: copies of the last one, never read)
#ifndef PB_P3_DMA
#define PB_P3_DMA 0  // 1 (round 6 experiment, -DPB_P3_DMA=1): the weight fragments go global -> LDS directly (global_load_lds_dwordx4: no staging
#endif               // registers, no ds_write), two k-steps ahead through THREE LDS buffers.  Same bits, and SLOWER: project b6 22.3 -> 27.7 us,
                     // b9 33.7 -> 42.5, head 49.2 -> 52.6 (profiles/r06_p3_dma.txt) -- a 1-KiB LDS-DMA piece costs the issuing wave 60-185
                     // clocks among MFMAs (MI355X guide), more than two loads + two ds_write_b128.  0: round 4's form (two register sets, two buffers)
    constexpr bool DMA = !DIRECT && PB_P3_DMA != 0;
    constexpr int NBUF = DMA ? 3 : 2;
    __shared__ __attribute__((aligned(16))) u32x4 s_w[DIRECT ? 1 : NBUF * FRP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int c0 = blockIdx.y * NR;
    const int n_steps = (K + 31) >> 5;
    long mrow[MR];
    bool mval[MR];
    const float *ap[MR], *gp[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        mrow[r] = (long)blockIdx.x * (16 * NW * MR) + (wave * MR + r) * 16 + li;
        mval[r] = mrow[r] < M;
        const long mc = mval[r] ? mrow[r] : (long)M - 1;
        ap[r] = act + mc * K;
        gp[r] = GATE ? gate + (mc / hw) * K : nullptr;
    }
    const int kl = 8 * kk;  // this lane's k offset inside a step
    f32x4 acc[MR][NR];
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int c = 0; c < NR; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ar[PD][MR][2], gr[GATE ? PD : 1][GATE ? MR : 1][2];
    u32x4 wd[DIRECT ? PD : 1][DIRECT ? NR : 1][3];
    const u32x4 *wsrc = wt3 + (size_t)c0 * 192;  // + step * tiles16 * 192
    auto request = [&](int t, auto slotc) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slotc)::value;
        const int tc = t < n_steps ? t : n_steps - 1;  // past the end: the last step again, never used
        int kb = 32 * tc + kl;
        if constexpr (KT) kb = kb < K ? kb : 0;
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            ar[SLOT][r][0] = *reinterpret_cast<const f32x4 *>(ap[r] + kb);
            ar[SLOT][r][1] = *reinterpret_cast<const f32x4 *>(ap[r] + kb + 4);
            if constexpr (GATE && (PB_P3_ABL & 2) == 0) {
                gr[SLOT][r][0] = *reinterpret_cast<const f32x4 *>(gp[r] + kb);
                gr[SLOT][r][1] = *reinterpret_cast<const f32x4 *>(gp[r] + kb + 4);
            }
        }
        if constexpr (DIRECT) {
            const u32x4 *src = wsrc + (size_t)tc * tiles16 * 192 + lane;
#pragma unroll
            for (int c = 0; c < NR; ++c)
#pragma unroll
                for (int p = 0; p < 3; ++p) wd[SLOT][c][p] = src[(c * 3 + p) * 64];
        }
    };
    // Weight staging runs TWO steps ahead through two register sets: the fragments of step t + 2 are requested at the start of
    // step t and stored into LDS at the end of step t + 1.  Vector-memory loads complete in order, so the wait in front of a
    // staging store also waits for every older load; with the fetch only one step ahead that wait drained the activation ring
    // down to the requests of the current step (seen in the ISA: vmcnt(4 MR) at every step's end), i.e. the ring hid one
    // step of latency whatever its depth.
    u32x4 wreg[2][WREGS];
    auto load_w = [&](int t, auto setc) __attribute__((always_inline)) {
        if constexpr (!DIRECT) {
            constexpr int SET = decltype(setc)::value;
            const int tc = t < n_steps ? t : n_steps - 1;
            const u32x4 *src = wsrc + (size_t)tc * tiles16 * 192;
#pragma unroll
            for (int j = 0; j < WREGS; ++j) {
                const int i = threadIdx.x + j * NTHR;
                wreg[SET][j] = src[i < FR ? i : FR - 1];
            }
        }
    };
    auto store_w = [&](int buf, auto setc) __attribute__((always_inline)) {
        if constexpr (!DIRECT) {
            constexpr int SET = decltype(setc)::value;
#pragma unroll
            for (int j = 0; j < WREGS; ++j) {
                // unconditional: a store under `if (i < FR)` makes hipcc drain every outstanding load (vmcnt(0)) around the branch
                s_w[buf * FRP + threadIdx.x + j * NTHR] = wreg[SET][j];
            }
        }
    };
    // DMA form: the fragments of step t straight into LDS buffer t % 3 -- a wave's chunk j is 64 consecutive 16-byte pieces (its lanes'
    // pieces are consecutive), so the destination is wave-uniform + lane * 16, which is what the instruction writes
    auto dma_w = [&](int t) __attribute__((always_inline)) {
        if constexpr (DMA) {
            const int tc = t < n_steps ? t : n_steps - 1;
            const u32x4 *src = wsrc + (size_t)tc * tiles16 * 192;
            const int b3 = t % 3;
#pragma unroll
            for (int j = 0; j < WREGS; ++j) {
                const int i = threadIdx.x + j * NTHR;
#ifdef __HIP_DEVICE_COMPILE__  // (the host pass has no such builtin: with the call in sight it silently emits no launch stub for the kernel)
                __builtin_amdgcn_global_load_lds(src + (i < FR ? i : FR - 1), &s_w[b3 * FRP + j * NTHR + wave * 64], 16, 0, 0);
#else
                (void)src; (void)b3; (void)i;
#endif
            }
        }
    };
    // the loads a wave issues per k-step besides the DMA: the activation (and gate) requests of one ring slot
    constexpr int ACT_LOADS = MR * 2 * (GATE ? 2 : 1);
    // one k-step from ring slot SLOT / LDS buffer buf
    auto k_step = [&](int t, int buf, auto slotc) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slotc)::value;
        // nothing of this step may move above this point (hipcc otherwise hoists the gate multiplies of the NEXT steps of the
        // unrolled loop to the top of the loop body, and with them the waits for loads issued a moment ago: vmcnt(0) once per turn)
        __builtin_amdgcn_sched_barrier(0);
        // (1) this step's weight fragments, all three planes, requested in the order the passes use them (lo, mid, hi): LDS
        //     answers in order, so pass 1 starts after the first NR reads while the rest are still arriving.  Issued as one burst
        //     at the top: left to itself hipcc sinks each read next to its MFMA (read -> lgkmcnt(0) -> MFMA, the LDS latency
        //     exposed once per MFMA: the first version of this kernel ran at 40 % of the matrix pipe's rate)
        u32x4 wq[NR][3];
        if constexpr (DIRECT) {
#pragma unroll
            for (int p = 2; p >= 0; --p)
#pragma unroll
                for (int c = 0; c < NR; ++c) wq[c][p] = wd[SLOT][c][p];
        } else {
            const u32x4 *sw = s_w + (DMA ? t % 3 : buf) * FRP + lane;
#pragma unroll
            for (int p = 2; p >= 0; --p)
#pragma unroll
                for (int c = 0; c < NR; ++c) wq[c][p] = sw[(c * 3 + p) * 64];
        }
        // (2) this step's activations out of the ring (gate multiplied in: one rounding, as in the f32 forms)
        f32x4 a[MR][2];
#pragma unroll
        for (int r = 0; r < MR; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                a[r][h] = ar[SLOT][r][h];
                if constexpr (GATE && (PB_P3_ABL & 2) == 0) {
                    const f32x4 g = gr[SLOT][r][h];
                    a[r][h].x = a[r][h].x * g.x; a[r][h].y = a[r][h].y * g.y; a[r][h].z = a[r][h].z * g.z; a[r][h].w = a[r][h].w * g.w;
                }
                if constexpr (KT) {
                    if (32 * t + kl >= K) a[r][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        // (3) the weight fetch of step t + 2, then the activation request of step t + PD into the slot just read.  The weight
        //     fetch comes FIRST: loads return in order, and the staging store at the end of step t + 1 must not have to wait for
        //     the (younger) ring loads
        if constexpr (DMA) dma_w(t + 2);  // into the buffer step t - 1 read (every wave is past the barrier that ended it)
        else if constexpr ((PB_P3_ABL & (4 | 32)) == 0) load_w(t + 2, std::integral_constant<int, SLOT & 1>{});
        if constexpr ((PB_P3_ABL & 8) == 0) request(t + PD, slotc);
        __builtin_amdgcn_sched_barrier(0);
        P3Act pa[MR];
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            if constexpr ((PB_P3_ABL & 1) != 0) {
                pa[r].h = pa[r].m = pa[r].l = (u32x4){__float_as_uint(a[r][0].x), __float_as_uint(a[r][0].z), __float_as_uint(a[r][1].x), __float_as_uint(a[r][1].z)};
            } else {
                pa[r] = p3_split8(a[r][0], a[r][1]);
            }
        }
        // the six passes, pass outermost: consecutive MFMAs write different accumulators; every accumulator receives its six
        // products in the order of p3_step
#define PB_P3_PASS(WP, AP)                                                                       \
    _Pragma("unroll") for (int r = 0; r < MR; ++r) _Pragma("unroll") for (int c = 0; c < NR; ++c) \
        acc[r][c] = p3_mfma(wq[c][WP], pa[r].AP, acc[r][c]);
        PB_P3_PASS(2, h)
        PB_P3_PASS(1, m)
        PB_P3_PASS(1, h)
        PB_P3_PASS(0, l)
        PB_P3_PASS(0, m)
        PB_P3_PASS(0, h)
#undef PB_P3_PASS
        if constexpr (DMA) {
            // step t + 1's fragments (requested at the top of step t - 1) must have landed before the barrier that lets every wave
            // read them: everything issued since then may stay in flight -- the activation requests of steps t - 1 and t and step
            // t + 2's DMA (vector-memory operations complete in order; the compiler does not track LDS-DMA data: the wait is explicit)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ACT_LOADS + WREGS) : "memory");
            __syncthreads();
        } else if constexpr (!DIRECT && (PB_P3_ABL & 4) == 0) {
            if constexpr ((PB_P3_ABL & 32) == 0) store_w(buf ^ 1, std::integral_constant<int, (SLOT & 1) ^ 1>{});
            if constexpr ((PB_P3_ABL & 16) == 0) __syncthreads();
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    p3_static_for<0, PD>([&](auto qc) __attribute__((always_inline)) { request(decltype(qc)::value, qc); });
    if constexpr (DMA) {
        dma_w(0);
        dma_w(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both steps' fragments are in (once per kernel; the counted waits below assume it)
        __syncthreads();
    } else {
        load_w(0, I0{});
        store_w(0, I0{});
        load_w(1, std::integral_constant<int, 1>{});
        if constexpr (!DIRECT) __syncthreads();
    }
    int t = 0;
    for (; t + PD <= n_steps; t += PD)
        p3_static_for<0, PD>([&](auto qc) __attribute__((always_inline)) { k_step(t + decltype(qc)::value, (t + decltype(qc)::value) & 1, qc); });
    {   // the steps left over (fewer than PD, once per kernel), each on the slot it was requested into
        const int rem = n_steps - t;
        p3_static_for<0, PD - 1>([&](auto qc) __attribute__((always_inline)) {
            constexpr int Q = decltype(qc)::value;
            if (rem > Q) k_step(t + Q, (t + Q) & 1, qc);
        });
    }
    // ---- epilogues (k_gemm_t's, per row tile)
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        if constexpr (EPI == 1) {
            // M is a multiple of 16 (whole 4 x 4 images): a tile is valid or not as a whole, no lane leaves before the shifts
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const int n = (c0 + c) * 16 + kk * 4;
                const int nc = n < N ? n : 0;
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + nc);
                f32x4 v = acc[r][c];
                v.x = silu_f(v.x + b.x); v.y = silu_f(v.y + b.y); v.z = silu_f(v.z + b.z); v.w = silu_f(v.w + b.w);
                // k_avgpool's order (t = 0; t = t + v[p] over the 16 pixels) as 15 unconditional steps t = t[lane - 1] + v: lane j
                // holds its final prefix sum after step j and every later step recomputes the same value from the (final) lane
                // below it, lane 0 keeps 0 + v[0] -- so no lane needs a predicate and a step is one v_add_f32 with a DPP source
                // per value (the predicated form: a move, an add and a select)
                f32x4 s = {0.0f + v.x, 0.0f + v.y, 0.0f + v.z, 0.0f + v.w};
#pragma unroll
                for (int j = 1; j < 16; ++j) {
                    s.x = dpp_shr1(s.x) + v.x; s.y = dpp_shr1(s.y) + v.y; s.z = dpp_shr1(s.z) + v.z; s.w = dpp_shr1(s.w) + v.w;
                }
                if (mval[r] && li == 15 && n < N) {
                    const f32x4 o = {s.x * scale, s.y * scale, s.z * scale, s.w * scale};
                    *reinterpret_cast<f32x4 *>(out + (mrow[r] >> 4) * N + n) = o;
                }
            }
        } else {
            if (!mval[r]) continue;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const int n = (c0 + c) * 16 + kk * 4;
                if (n >= N) continue;  // N % 4 == 0
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + n);
                f32x4 v = acc[r][c];
                v.x = v.x + b.x; v.y = v.y + b.y; v.z = v.z + b.z; v.w = v.w + b.w;
                if constexpr (EPI == 2) {
                    const f32x4 y = {tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w)};
                    if (out) *reinterpret_cast<f32x4 *>(out + mrow[r] * N + n) = y;
                    const uint32_t pk = (uint32_t)quantize_u8(y.x) | ((uint32_t)quantize_u8(y.y) << 8) | ((uint32_t)quantize_u8(y.z) << 16) |
                                        ((uint32_t)quantize_u8(y.w) << 24);
                    *reinterpret_cast<uint32_t *>(out_u8 + mrow[r] * N + n) = pk;
                    continue;
                }
                if (do_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                if (resid) {
                    const f32x4 rv = *reinterpret_cast<const f32x4 *>(resid + mrow[r] * N + n);
                    v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w;
                }
                *reinterpret_cast<f32x4 *>(out + mrow[r] * N + n) = v;
            }
        }
    }
}

}  // namespace pbe
