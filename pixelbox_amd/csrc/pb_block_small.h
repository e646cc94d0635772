// A WHOLE MBConv block of a small map in one kernel (included by pb_embed.hip after pb_embed_kernels.h).
//
// For the 4 x 4 maps of EfficientNet-B0's last stage (blocks 12-15: 192 -> 1152 -> 192 / 320 channels) the 6x-expanded
// activation of two images is 32 pixel rows x 1152 channels x 4 B = 144 KB: it fits the 160 KB of LDS of one CU.  One
// workgroup of eight waves therefore runs, for its G images,
//   1. expand 1x1 + bias + SiLU on the f32 MFMA, one group of 16 GC channels at a time, into a small LDS window,
//   2. the depthwise KS x KS filter + bias + SiLU from that window into the LDS-resident depthwise output `dwo`,
//   3. squeeze-excite: pooled means (the same 2^-24 fixed-point sums as se_acc / k_se), FC1 + SiLU, FC2 + sigmoid, and the
//      gate multiplied into `dwo` in place,
//   4. project 1x1 + bias (+ residual) on the f32 MFMA with `dwo` as the activation operand,
// and only the block's [P][COUT] output goes back to memory.  Against k_mbconv_small + k_se + k_gemm_t this removes the
// depthwise output's HBM/L2 round trip (2 x 37.7 MB per 512 images and layer), the pooled-sum and gate round trips and two
// dependent launches per block.  Every value is produced by the SAME operations in the SAME order as the unfused kernels
// (MFMA operand maps and k order of k_gemm1x1 / k_gemm_t, tap order of k_dwconv, k_se's reduction order; gate * activation is
// one rounding whether it happens in the project GEMM's operand fetch or in place here), so the block's output is
// bit-identical -- tests/test_embed_gpu.py compares the kernel forms.
//
// Rows: R = G * P pixel rows (P = HW * HW), PT = R / 16 row tiles; a channel group has GC = 8 / PT 16-channel tiles.
//  * expand: waves 0-3 own one 16-channel tile of the group for two row tiles each (two accumulator chains per weight
//    fragment); a lane's fragments come straight from memory into registers (Gemm::wt4: a k-step's 64 lanes contiguous,
//    one KB per wave request) and each register is re-requested for the NEXT group right after its last MFMA; the block
//    input x stays in registers for the whole kernel;
//  * depthwise: wave = (image, output row), lane = channel of the group; a lane keeps ITS channel's taps in registers and
//    reads single floats of the window; rows / columns of the filter that fall outside the map are skipped (see below);
//  * squeeze-excite: wave w owns the units w, w + 8, ...; FC1's butterflies are formed transposed; FC2 is a quad per thread;
//  * project: waves 0-3 take ALL the row tiles and a quarter of the output tiles each (a weight fragment is requested once).
// Both MFMA phases run as ONE instruction stream per SIMD (waves 0-3 sit on the four SIMDs): two waves taking turns on a
// SIMD's matrix pipe ran the project phase at 75 % of the bare MFMA rate, one wave with 6 or 10 accumulator chains at 96 %.
// What else sets the time (profiles/micro/block_small_bench.hip, in-kernel stamps): the vector-memory instructions -- a
// 16-byte-per-lane load occupies the CU's address path for 16 cycles whatever it hits, and at 32 pixel rows per workgroup
// every MFMA operand fragment is used for only two row tiles -- and the phases without MFMA work (filter, squeeze-excite:
// one workgroup per CU, nothing to hide them under).  Per 512 images: 94 us (5 x 5 blocks; front + k_se + project GEMM: 128)
// and 113 us (the 3 x 3 / 320-column block; 139).  At small batches the unfused kernels win (a workgroup here takes ~95 us
// however few images there are), so the host times both forms per (block, batch bucket).
// LDS (floats): dwo [R][E + 8] (pitch = 8 mod 64 dwords: the project phase's 16-byte fragment reads are conflict-free in
// ds_read_b128's lane groups) | the expand window [R][16 GC + 4], later the pooled means [G][E], FC1 partial sums
// [G][NB][SP] and squeezed units [G][SP].
#pragma once

namespace pbe {

struct BlockW {
    const float *we2;   // expand weights, fragments by k-step (Gemm::wt4): [CIN / 16][E / 16][64 lanes][4 e]
    const void *we3;    // P3 expand layers (pb_gemm_p3.h): the three bf16 planes in fragment order, [CIN / 32][E / 16][3][64 lanes] x 16 B; else null
    const float *be;    // [E] expand bias
    const float *dwc;   // [E][KS * KS rounded up to 4] depthwise taps per channel
    const float *bd;    // [E] depthwise bias
    const float *w1, *b1, *w2t, *b2;  // squeeze-excite, as k_se: [SP][E], [SP], [SP][E], [E]
    const float *wp2;   // project weights, fragments by k-step: [E / 16][NT16][64 lanes][4 e]
    const void *wp3;    // P3 project layers (pb_gemm_p3.h): the three bf16 planes in fragment order, [E / 32][NT16][3][64 lanes] x 16 B; else null
    const float *bp;    // [16 NT16] project bias
    int nt16;           // 16-column tiles of the project weights' padded width
    const long long *range_slot;  // se_range_check: the word that holds the address of the embedder's range flag (buf_part - 1)
    unsigned long long *dbg;  // ABL 32 (stamped diagnostic build): [workgroup][wave][16] cycle sums per phase; else unused
};

// P3: the project phase runs on the bf16 matrix cores from three pieces per operand (the layer's arithmetic when the tiled
// GEMM computes it: pb_gemm_p3.h); a lane then reads 8 consecutive k of its pixel row, and the dwo pitch is 4 mod 64 dwords
// (16 lanes x 16 bytes of one k-slot land on 16 different bank quads) instead of 8 mod 64.
template <int KS, int CIN, int E, int COUT, int HW, int G, int SP, bool P3 = false>
struct BlockGeom {
    static constexpr int P = HW * HW, R = G * P, PT = R / 16, GC = 8 / PT, QG = 4 * GC, NG = E / (16 * GC);
    static constexpr int KC = CIN / 16, NT = COUT / 16, NGR = 8 / PT, NRP = (NT + NGR - 1) / NGR;
    static constexpr int WP = 16 * GC + 4, DP = E + (P3 ? 4 : 8), KK = KS * KS, KKP = (KK + 3) / 4 * 4, PAD = (KS - 1) / 2;
    static constexpr int NQ = E / 4, NB = (NQ + 63) / 64, KSP = E / 16;
    static constexpr int WIN = R * WP, SEF = G * E + G * NB * SP + G * SP, SCR = (WIN > SEF ? WIN : SEF);
    static constexpr int LDS_FLOATS = R * DP + SCR;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * 4;
    static_assert(R % 16 == 0 && (PT == 2 || PT == 4), "two or four row tiles");
    static_assert(E % (16 * GC) == 0 && CIN % 16 == 0 && COUT % 16 == 0, "whole tiles");
    static_assert(G * HW == 8 && (HW == 4 || HW == 8), "a wave per (image, output row)");
};

// ABL (timing experiments only, results invalid): 1 no expand MFMAs, 2 no filter taps, 4 no squeeze-excite, 8 no project MFMAs,
// 16 the expand / project weights of the first group / chunk re-read throughout (no L2 traffic for them)
// P3E (round 6): the EXPAND layer is a P3 layer too (blocks 12-15: K = 192).  The block input is split ONCE, in the prologue, into the
// three bf16 planes of its 6 k-steps (a lane keeps 2 row tiles x 6 steps x 12 registers: the split costs nothing per channel group --
// the consumer-side split the producers' pre-split planes of round 5's experiment could not give), a group's expand is 72 bf16 MFMAs
// per SIMD (1 152 clocks) where the f32 chain took 96 f32 MFMAs (3 072), and the weight planes come straight from memory through a
// three-step register ring.  Arithmetic: p3_step per k-step, steps ascending -- the same bits as k_gemm_p3 on this layer.
template <int KS, int CIN, int E, int COUT, int HW, int G, int SP, bool RESID, int ABL = 0, bool P3 = false, bool P3E = false>
__global__ __launch_bounds__(512) void k_block_small(const float *__restrict__ x, BlockW w, float *__restrict__ out, int n_img) {
    using GEO = BlockGeom<KS, CIN, E, COUT, HW, G, SP, P3>;
    constexpr int P = GEO::P, R = GEO::R, PT = GEO::PT, GC = GEO::GC, QG = GEO::QG, NG = GEO::NG, KC = GEO::KC, NT = GEO::NT;
    constexpr int NRP = GEO::NRP, WP = GEO::WP, DP = GEO::DP, KK = GEO::KK, PAD = GEO::PAD, NQ = GEO::NQ, NB = GEO::NB;
    constexpr int KSP = GEO::KSP, ET = E / 16;
    extern __shared__ __attribute__((aligned(16))) float s_blk[];
    float *s_dwo = s_blk;                    // [R][DP]
    float *s_scr = s_dwo + R * DP;           // window [R][WP], later means [G][E] | FC1 partial sums | squeezed units
    float *s_p = s_scr + G * E;              // [G][NB][SP]
    float *s_s = s_p + G * NB * SP;          // [G][SP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int b0 = blockIdx.x * G;
    unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr ((ABL & 32) != 0) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (i >= 0) st_[i] += t - st_t;
            st_t = t;
        }
    };
    stamp(-1);
    // ---- expand role: waves 0-3 (one per SIMD: a single MFMA stream per SIMD runs closer to the pipe's rate than two waves
    // taking turns) own channel tile ect of the group for PTW = 2 row tiles each -- two accumulator chains per weight fragment,
    // so a fragment is requested once; waves 4-7 have no MFMA work in this phase.  A fragment register is re-requested for the
    // NEXT group right after its last MFMA of this group (in place: the requests spread over the MFMA phase and have the
    // epilogue, both barriers and the filter phase to land).
    constexpr int PTW = PT * GC / 4;
    constexpr int KS32 = CIN / 32;  // P3E: k-steps of 32
    static_assert(!P3E || (CIN % 32 == 0 && KS32 % 3 == 0), "P3E: whole k-steps, whole turns of the three-slot weight ring");
    const bool ew = wave < 4;
    const int ect = wave % GC, ept0 = (wave % 4) / GC * PTW;
    f32x4 xb[P3E ? 1 : PTW][P3E ? 1 : KC];
    P3Act xp[P3E ? PTW : 1][P3E ? KS32 : 1];
    if constexpr (P3E) {
        if (ew) {
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt) {
                const int erow = 16 * (ept0 + pt) + li;
                int eimg = b0 + erow / P;
                if (eimg >= n_img) eimg = n_img - 1;  // a padded slot repeats the last image, stores nothing
                const float *xrow = x + ((size_t)eimg * P + erow % P) * CIN + 8 * kk;
                f32x4 xr[KS32][2];
#pragma unroll
                for (int s = 0; s < KS32; ++s) {
                    xr[s][0] = *reinterpret_cast<const f32x4 *>(xrow + 32 * s);
                    xr[s][1] = *reinterpret_cast<const f32x4 *>(xrow + 32 * s + 4);
                }
#pragma unroll
                for (int s = 0; s < KS32; ++s) xp[pt][s] = p3_split8(xr[s][0], xr[s][1]);
            }
        }
    } else {
#pragma unroll
        for (int pt = 0; pt < PTW; ++pt) {
            const int erow = 16 * (ept0 + pt) + li;
            int eimg = b0 + erow / P;
            if (eimg >= n_img) eimg = n_img - 1;  // a padded slot repeats the last image, stores nothing
            const float *xrow = x + ((size_t)eimg * P + erow % P) * CIN + 4 * kk;
#pragma unroll
            for (int s = 0; s < KC; ++s) xb[pt][s] = *reinterpret_cast<const f32x4 *>(xrow + 16 * s);
        }
    }
    f32x4 aw[P3E ? 1 : KC], bev;
    const float *awp = w.we2 + ((size_t)ect * 64 + lane) * 4;  // + (s * ET + g * GC) * 256
    // P3E: plane p of (k-step s, tile t) is fragment ((s * ET + t) * 3 + p) of 64 lanes x 16 bytes; the ring holds steps s, s + 1, s + 2
    // (slot = s % 3); steps 0 and 1 of the first group are requested here, every later step two steps ahead of its MFMAs
    u32x4 wr[P3E ? 3 : 1][3];
    const u32x4 *wq3 = reinterpret_cast<const u32x4 *>(w.we3) + lane;
    auto load_w3 = [&](int s, int tile, auto slotc) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slotc)::value;
        const u32x4 *src = wq3 + ((size_t)s * ET + tile) * 192;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wr[SLOT][pl] = src[pl * 64];
    };
    if (ew) {
        if constexpr (P3E) {
            load_w3(0, ect, std::integral_constant<int, 0>{});
            load_w3(1, ect, std::integral_constant<int, 1>{});
        } else {
#pragma unroll
            for (int s = 0; s < KC; ++s) aw[s] = *reinterpret_cast<const f32x4 *>(awp + (size_t)s * ET * 256);
        }
        bev = *reinterpret_cast<const f32x4 *>(w.be + 16 * ect + 4 * kk);
    }
    // ---- depthwise role: wave = (image dimg, output row dpy), lane = (channel dc of the group, strip dh of 4 output pixels).
    // A lane holds ITS channel's taps in registers (KKP floats, 16-byte loads of the [E][KKP] table) and reads single
    // floats of the window (64 lanes = consecutive channels: conflict-free).  Filter rows that fall outside the map are
    // skipped by a wave-uniform branch and, on the 4-wide maps, columns outside it at compile time: on a 4 x 4 map 51 %
    // of a 5 x 5 filter's taps are padding.  (A skipped tap is fma(0, w, o) = o in the unfused kernels.)
    constexpr int CH = 16 * GC, KKP = GEO::KKP, NIN = 4 + 2 * PAD;
    const int dimg = wave / HW, dpy = wave % HW;
    const int dc = lane % CH, dh = lane / CH;
    const float *dwin = s_scr + (dimg * P) * WP + dc;  // + (iy * HW + ix) * WP
    float *ddst = s_dwo + (dimg * P + dpy * HW + 4 * dh) * DP + dc;  // + px * DP + CH * g
    float *ewin = s_scr + (16 * ept0 + li) * WP + 16 * ect + 4 * kk;  // + 16 pt * WP
    stamp(0);
    if constexpr (P3E) {
        // ---- the group loop as a two-stage pipeline (round 6): in turn `it` waves 0-3 EXPAND group it while waves 4-7 FILTER group it - 1;
        // one barrier per turn.  (The serial form -- expand by four waves, barrier, filter by eight, barrier -- left the matrix pipe idle
        // during every filter and waves 4-7 idle during every expand: 4.7 k clocks per group of which 2.3 k expand; profiles/r06_block_small.txt.)
        // No second window buffer is needed: group g's window lives in dwo's columns of group g + 1 (pitch DP; free until filter g + 1
        // writes them, by which time filter g has read it), the last group's in the scratch window (pitch WP).  Turn `it` then touches:
        // expand -> columns of it + 1; filter -> reads columns of it, writes columns of it - 1: disjoint.
        // Each role has a loop of its own (NG + 1 turns and barriers each): in one loop with a branch per turn the expand role's 180
        // resident registers (input planes, weight ring) counted as live through the filter role's code too -- 61 spills.
        if (ew) {
            for (int it = 0; it <= NG; ++it) {
                if (it < NG) {
                    const int g = it;
                    const int gn = ((ABL & 16) ? 0 : (g + 1 < NG ? g + 1 : g)) * GC;  // ABL 16: every group re-reads the first group's weights
                    float *const wbase = g + 1 < NG ? s_dwo + CH * (g + 1) : s_scr;
                    const int wpitch = g + 1 < NG ? DP : WP;
                    f32x4 acc[PTW];
#pragma unroll
                    for (int pt = 0; pt < PTW; ++pt) acc[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    auto step3 = [&](auto sc) __attribute__((always_inline)) {
                        constexpr int S = decltype(sc)::value, SLOT = S % 3, SN = S + 2;
                        // the step two ahead (of the next group past this group's last) into the slot step S - 1 has just left
                        if constexpr (SN < KS32) load_w3(SN, g * GC + ect, std::integral_constant<int, SN % 3>{});
                        else load_w3(SN - KS32, gn + ect, std::integral_constant<int, SN % 3>{});
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (!(ABL & 1) || S == 0) {
#define PB_BLK_EPASS(WP_, AP_) \
    _Pragma("unroll") for (int pt = 0; pt < PTW; ++pt) acc[pt] = p3_mfma(wr[SLOT][WP_], xp[pt][S].AP_, acc[pt]);
                            PB_BLK_EPASS(2, h)
                            PB_BLK_EPASS(1, m)
                            PB_BLK_EPASS(1, h)
                            PB_BLK_EPASS(0, l)
                            PB_BLK_EPASS(0, m)
                            PB_BLK_EPASS(0, h)
#undef PB_BLK_EPASS
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    static_assert(KS32 == 6, "P3E: six k-steps (CIN = 192)");
                    step3(std::integral_constant<int, 0>{});
                    step3(std::integral_constant<int, 1>{});
                    step3(std::integral_constant<int, 2>{});
                    step3(std::integral_constant<int, 3>{});
                    step3(std::integral_constant<int, 4>{});
                    step3(std::integral_constant<int, 5>{});
                    stamp(2);
                    const f32x4 bq = bev;
                    bev = *reinterpret_cast<const f32x4 *>(w.be + 16 * (gn + ect) + 4 * kk);
                    float *ew2 = wbase + (16 * ept0 + li) * wpitch + 16 * ect + 4 * kk;
#pragma unroll
                    for (int pt = 0; pt < PTW; ++pt) {
                        f32x4 v = acc[pt];
                        v.x = silu_f(v.x + bq.x); v.y = silu_f(v.y + bq.y); v.z = silu_f(v.z + bq.z); v.w = silu_f(v.w + bq.w);
                        *reinterpret_cast<f32x4 *>(ew2 + 16 * pt * wpitch) = v;
                    }
                    stamp(3);
                }
                __syncthreads();
                stamp(6);
            }
        } else {
            // taps of the group this wave filters NEXT turn are requested a whole turn ahead into the OTHER of two register sets (the
            // turns alternate sets; a copy at the end of the turn would wait for the loads there -- and behind the expand waves' 18
            // weight requests per group on the same address path they take most of a turn)
            f32x4 tqs[2][KKP / 4];
            float bds[2] = {0.f, 0.f};
#ifndef PB_BLK_NO_PRIO
            __builtin_amdgcn_s_setprio(3);  // this role is the longer stage and shares its SIMD's issue with an MFMA stream (see below)
#endif
            auto turn = [&](int it, auto parc) __attribute__((always_inline)) {
                constexpr int CUR = decltype(parc)::value, NXT = CUR ^ 1;
                f32x4 (&tq)[KKP / 4] = tqs[CUR];
                const float bdv = bds[CUR];
                if (it < NG && (!(ABL & 64) || it < 2)) {  // ABL 64: the taps of the first two groups for every group (no tap requests later)
                    const f32x4 *tp = reinterpret_cast<const f32x4 *>(w.dwc + (size_t)(it * CH + dc) * KKP);
#pragma unroll
                    for (int t = 0; t < KKP / 4; ++t) tqs[NXT][t] = tp[t];
                    bds[NXT] = w.bd[it * CH + dc];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (it >= 1) {
                    const int g = it - 1;
                    const float *const wbase = g + 1 < NG ? s_dwo + CH * (g + 1) : s_scr;
                    const int wpitch = g + 1 < NG ? DP : WP;
                    static_assert(HW == 4 && G == 2, "P3E: one image's two output rows per filter wave");
                    // wave 4 + v: image v / 2, output rows 2 (v % 2) and 2 (v % 2) + 1 -- the image's 16 window values are read once
                    // (one burst of LDS reads, one latency) and serve both rows
                    const int fv = wave - 4, fimg = fv >> 1;
                    const float *fwin = wbase + (fimg * P) * wpitch + dc;
                    float in[HW][HW];
#pragma unroll
                    for (int iy = 0; iy < HW; ++iy)
#pragma unroll
                        for (int ix = 0; ix < HW; ++ix) in[iy][ix] = fwin[(iy * HW + ix) * wpitch];
                    float o[2][4];
                    // the output rows are compile-time constants in each of the two copies: every tap that falls outside the map is gone
                    // at compile time (on a 4 x 4 map 51 % of a 5 x 5 filter's taps are padding), no branch per filter row
                    auto rows = [&](auto fpyc) __attribute__((always_inline)) {
                        constexpr int FPY0 = decltype(fpyc)::value;
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                            for (int px = 0; px < 4; ++px) o[h2][px] = bdv;
#pragma unroll
                            for (int ky = 0; ky < ((ABL & 2) ? 1 : KS); ++ky) {
                                const int iy = FPY0 + h2 + ky - PAD;
                                if (iy < 0 || iy >= HW) continue;
#pragma unroll
                                for (int px = 0; px < 4; ++px)
#pragma unroll
                                    for (int kx = 0; kx < KS; ++kx) {
                                        const int ix = px + kx - PAD;
                                        if (ix < 0 || ix >= HW) continue;
                                        const int t = ky * KS + kx;
                                        const f32x4 qv = tq[t >> 2];
                                        const float wv = (t & 3) == 0 ? qv.x : ((t & 3) == 1 ? qv.y : ((t & 3) == 2 ? qv.z : qv.w));
                                        o[h2][px] = __builtin_fmaf(in[iy][ix], wv, o[h2][px]);
                                    }
                            }
                        }
                    };
                    if (fv & 1) rows(std::integral_constant<int, 2>{});
                    else rows(std::integral_constant<int, 0>{});
                    const int fpy0 = 2 * (fv & 1);
                    // (the window of group g is read; its outputs go to the columns filter g - 1's window occupied)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        float *fdst = s_dwo + (fimg * P + (fpy0 + h2) * HW) * DP + dc + CH * g;
#pragma unroll
                        for (int px = 0; px < 4; ++px) fdst[px * DP] = silu_f(o[h2][px]);
                    }
                }
                stamp(5);
                __syncthreads();
                stamp(6);
            };
            static_assert(NG % 2 == 0, "NG + 1 turns: pairs, then one");
            for (int it = 0; it < NG; it += 2) {
                turn(it, std::integral_constant<int, 1>{});      // turn 0 filters nothing and requests group 0's taps into set 0
                turn(it + 1, std::integral_constant<int, 0>{});
            }
            turn(NG, std::integral_constant<int, 1>{});
            __builtin_amdgcn_s_setprio(0);
        }
    } else {
    for (int g = 0; g < NG; ++g) {
        f32x4 tq[KKP / 4];
        float bdv;
        {
            const f32x4 *tp = reinterpret_cast<const f32x4 *>(w.dwc + (size_t)(g * CH + dc) * KKP);
#pragma unroll
            for (int t = 0; t < KKP / 4; ++t) tq[t] = tp[t];
            bdv = w.bd[g * CH + dc];
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(1);
        if (ew) {
            const int gn = ((ABL & 16) ? 0 : (g + 1 < NG ? g + 1 : g)) * GC;  // ABL 16: every group re-reads the first group's weights
            f32x4 acc[PTW];
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt) acc[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < ((ABL & 1) ? 1 : KC); ++s) {
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const float av = e2 == 0 ? aw[s].x : (e2 == 1 ? aw[s].y : (e2 == 2 ? aw[s].z : aw[s].w));
#pragma unroll
                    for (int pt = 0; pt < PTW; ++pt) {
                        const f32x4 xq = xb[pt][s];
                        const float xv = e2 == 0 ? xq.x : (e2 == 1 ? xq.y : (e2 == 2 ? xq.z : xq.w));
                        acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xv, acc[pt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                aw[s] = *reinterpret_cast<const f32x4 *>(awp + ((size_t)s * ET + gn) * 256);
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp(2);
            const f32x4 bq = bev;
            bev = *reinterpret_cast<const f32x4 *>(w.be + 16 * (gn + ect) + 4 * kk);
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt) {
                f32x4 v = acc[pt];
                v.x = silu_f(v.x + bq.x); v.y = silu_f(v.y + bq.y); v.z = silu_f(v.z + bq.z); v.w = silu_f(v.w + bq.w);
                *reinterpret_cast<f32x4 *>(ewin + 16 * pt * WP) = v;
            }
        }
        stamp(3);
        __syncthreads();
        stamp(4);
        float o[4] = {bdv, bdv, bdv, bdv};
#pragma unroll
        for (int ky = 0; ky < ((ABL & 2) ? 1 : KS); ++ky) {
            const int iy = dpy + ky - PAD;
            if (iy >= 0 && iy < HW) {  // wave-uniform
                float in[NIN];
#pragma unroll
                for (int j = 0; j < NIN; ++j) {
                    if constexpr (HW == 4) {
                        const int ix = j - PAD;  // dh = 0
                        in[j] = (ix >= 0 && ix < HW) ? dwin[(iy * HW + ix) * WP] : 0.f;
                    } else {
                        const int ix = 4 * dh + j - PAD;
                        const bool ok = ix >= 0 && ix < HW;
                        const float v = dwin[(iy * HW + (ok ? ix : 0)) * WP];
                        in[j] = ok ? v : 0.f;
                    }
                }
#pragma unroll
                for (int px = 0; px < 4; ++px)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        if (HW == 4 && (px + kx - PAD < 0 || px + kx - PAD >= HW)) continue;
                        const int t = ky * KS + kx;
                        const f32x4 q = tq[t >> 2];
                        const float wv = (t & 3) == 0 ? q.x : ((t & 3) == 1 ? q.y : ((t & 3) == 2 ? q.z : q.w));
                        o[px] = __builtin_fmaf(in[px + kx], wv, o[px]);
                    }
            }
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) ddst[px * DP + CH * g] = silu_f(o[px]);
        stamp(5);
        __syncthreads();
        stamp(6);
    }
    }
    // ---- squeeze-excite (k_se's arithmetic and order).  Wave w owns the units jj = w, w + 8, ...: their FC1 weights for every
    // block of 64 quads are requested before the pooled means are formed (one L2 round trip under that pass).
    if constexpr (!(ABL & 4)) {
        float *s_m = s_scr;  // [G][E] pooled means
        constexpr int JW = (SP + 7) / 8;  // units per wave
        f32x4 w1v[NB][JW];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            const int cq = blk * 64 + lane;
            const int c = cq < NQ ? 4 * cq : 0;
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                const int jj = wave + 8 * j;
                w1v[blk][j] = *reinterpret_cast<const f32x4 *>(w.w1 + (size_t)(jj < SP ? jj : 0) * E + c);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const double sc = (1.0 / 16777216.0) * (double)(1.0f / (float)P);  // k_se: 2^-24 * (double)inv_hw
        for (int it = tid; it < G * NQ; it += 512) {
            const int im = it / NQ, cq = it - im * NQ;
            ll4 t = {0, 0, 0, 0};
            int qmax = 0;  // largest converted output seen (se_range_check)
#pragma unroll 8
            for (int p = 0; p < P; ++p) se_acc(t, qmax, *reinterpret_cast<const f32x4 *>(s_dwo + (im * P + p) * DP + 4 * cq));
            se_range_check(qmax, w.range_slot);
            const f32x4 m = {(float)((double)t.x * sc), (float)((double)t.y * sc), (float)((double)t.z * sc), (float)((double)t.w * sc)};
            *reinterpret_cast<f32x4 *>(s_m + im * E + 4 * cq) = m;
        }
        __syncthreads();
        stamp(7);
        // FC1: per unit jj the products of a quad summed x, y, z, w, then the 64 quads of a block by the xor butterfly
        // (a + shfl_xor(a, 32), then 16, 8, 4, 2, 1 -- k_se's order).  The wave has NV = NB * JW * G such sums to form; run one
        // by one that is 6 NV cross-lane steps.  They are formed TRANSPOSED instead: at the level of offset `off`, sums k and
        // k + off share a register -- the lanes with bit `off` clear keep their element of sum k and hand over their element of
        // sum k + off, the other half the reverse -- so each level halves the registers and all NV sums take 63 steps; every
        // addition has the same two operands in the same order (own element + partner's) as in the one-by-one butterfly, and at
        // the end lane L holds sum number L.
        {
            constexpr int NV = NB * JW * G;
            static_assert(NV <= 64, "one finished sum per lane");
            float v[64];
#pragma unroll
            for (int k = NV; k < 64; ++k) v[k] = 0.f;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int cq = blk * 64 + lane;
                const bool on = cq < NQ;
                const int c = on ? 4 * cq : 0;
#pragma unroll
                for (int im = 0; im < G; ++im) {
                    f32x4 m = *reinterpret_cast<const f32x4 *>(s_m + im * E + c);
                    if (!on) m = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < JW; ++j) {
                        const f32x4 wv = w1v[blk][j];
                        float t = m.x * wv.x;
                        t = t + m.y * wv.y; t = t + m.z * wv.z; t = t + m.w * wv.w;
                        v[(blk * JW + j) * G + im] = t;
                    }
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const bool hi = (lane & off) != 0;
                const int src = (lane ^ off) << 2;
#pragma unroll
                for (int k = 0; k < off; ++k) {
                    const float keep = hi ? v[k + off] : v[k];
                    const float send = hi ? v[k] : v[k + off];
                    v[k] = keep + __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(send)));
                }
            }
            if (lane < NV) {
                const int im = lane % G, j = (lane / G) % JW, blk = lane / (G * JW);
                const int jj = wave + 8 * j;
                if (jj < SP) s_p[(im * NB + blk) * SP + jj] = v[0];
            }
        }
        stamp(8);
        // FC2 weights of the first 16 units: requested before the squeezed units exist
        constexpr int JG = SP < 16 ? SP : 16, GG = SP / JG;
        static_assert(NQ <= 512, "one quad per thread in FC2");
        const int cq2 = tid < NQ ? tid : 0, c2 = 4 * cq2;
        f32x4 w2v[JG];
#pragma unroll
        for (int j = 0; j < JG; ++j) w2v[j] = *reinterpret_cast<const f32x4 *>(w.w2t + (size_t)j * E + c2);
        __syncthreads();
        if (tid < G * SP) {
            const int im = tid / SP, jj = tid - im * SP;
            float v = s_p[(im * NB) * SP + jj];
            for (int blk = 1; blk < NB; ++blk) v = v + s_p[(im * NB + blk) * SP + jj];
            s_s[im * SP + jj] = silu_f(v + w.b1[jj]);
        }
        __syncthreads();
        stamp(9);
        // FC2 + sigmoid, then the gate multiplied into the quad's column of dwo
        if (tid < NQ) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(w.b2 + c2);
            f32x4 v[G];
#pragma unroll
            for (int im = 0; im < G; ++im) v[im] = bv;
#pragma unroll
            for (int gg = 0; gg < GG; ++gg) {
                f32x4 wn[JG];
                if (gg + 1 < GG) {
#pragma unroll
                    for (int j = 0; j < JG; ++j) wn[j] = *reinterpret_cast<const f32x4 *>(w.w2t + (size_t)((gg + 1) * JG + j) * E + c2);
                }
                f32x4 a2[G];
#pragma unroll
                for (int im = 0; im < G; ++im) a2[im] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < JG; ++j) {
                    const f32x4 wv = w2v[j];
#pragma unroll
                    for (int im = 0; im < G; ++im) {
                        const float sj = s_s[im * SP + gg * JG + j];
                        a2[im].x = a2[im].x + sj * wv.x; a2[im].y = a2[im].y + sj * wv.y;
                        a2[im].z = a2[im].z + sj * wv.z; a2[im].w = a2[im].w + sj * wv.w;
                    }
                }
#pragma unroll
                for (int im = 0; im < G; ++im) {
                    v[im].x = v[im].x + a2[im].x; v[im].y = v[im].y + a2[im].y;
                    v[im].z = v[im].z + a2[im].z; v[im].w = v[im].w + a2[im].w;
                }
                if (gg + 1 < GG) {
#pragma unroll
                    for (int j = 0; j < JG; ++j) w2v[j] = wn[j];
                }
            }
#pragma unroll
            for (int im = 0; im < G; ++im) {
                const f32x4 gt = {sigmoid_f(v[im].x), sigmoid_f(v[im].y), sigmoid_f(v[im].z), sigmoid_f(v[im].w)};
#pragma unroll 8
                for (int p = 0; p < P; ++p) {
                    f32x4 *d = reinterpret_cast<f32x4 *>(s_dwo + (im * P + p) * DP + c2);
                    f32x4 a = *d;
                    a.x = a.x * gt.x; a.y = a.y * gt.y; a.z = a.z * gt.z; a.w = a.w * gt.w;
                    *d = a;
                }
            }
        }
        __syncthreads();
    }
    stamp(10);
    // ---- project: a wave takes ALL the row tiles and a share of the 16-column output tiles, so that a weight fragment is
    // requested by exactly one wave (with a wave per row tile the PT waves of a column group each requested it: twice the
    // vector-memory instructions and L1 fills).  Waves 0-3 take TA = NT / 4 tiles each and waves 4-7 none: ONE MFMA stream per
    // SIMD (2 TA accumulator chains) runs at 96 % of the bare MFMA rate where waves w and w + 4 sharing a SIMD's pipe with
    // TA + TB tiles between them ran at 75 % (-DPB_BLK_SPLIT8 restores that split).
    {
#ifdef PB_BLK_SPLIT8  // comparison build: the output tiles over all eight waves (two waves per SIMD)
        constexpr int TA = (NT / 4 + 1) / 2, TB = NT / 4 - TA;
#else
        constexpr int TA = NT / 4, TB = 0;
#endif
        static_assert(NT % 4 == 0 && TB >= 0, "output tiles split over four SIMDs");
        auto project = [&](auto cntc, int tile0) __attribute__((always_inline)) {
            constexpr int CNT = decltype(cntc)::value;
            if constexpr (CNT > 0) {
                constexpr int CS = (KSP % 8 == 0 && CNT * PT <= 4) ? 4 : 2;  // k-steps per weight request
                constexpr int NCH = KSP / CS;
                static_assert(KSP % CS == 0 && NCH % 3 == 0, "whole rounds of three weight requests");
                const float *brow = s_dwo + li * DP + 4 * kk;  // + 16 pt * DP + 16 s
                const float *wl = w.wp2 + ((size_t)tile0 * 64 + lane) * 4;
                f32x4 acc[PT][CNT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int c = 0; c < CNT; ++c) acc[pt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // three weight sets: the requests run two chunks ahead of the MFMAs
                f32x4 wa[3][CS][CNT];
                auto load_w = [&](int s0, auto setc) __attribute__((always_inline)) {
                    constexpr int SET = decltype(setc)::value;
                    s0 = s0 < KSP ? s0 : KSP - CS;  // past the end: the last chunk again, never used
                    if (ABL & 16) s0 = 0;
#pragma unroll
                    for (int j = 0; j < CS; ++j)
#pragma unroll
                        for (int c = 0; c < CNT; ++c)
                            wa[SET][j][c] = *reinterpret_cast<const f32x4 *>(wl + ((size_t)(s0 + j) * w.nt16 + c) * 256);
                };
                f32x4 b[CS][PT], bn[CS][PT];
                auto comp = [&](int s0, auto setc) __attribute__((always_inline)) {
                    constexpr int SET = decltype(setc)::value;
                    const int sn = s0 + CS < KSP ? s0 + CS : s0;
#pragma unroll
                    for (int j = 0; j < CS; ++j)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt) bn[j][pt] = *reinterpret_cast<const f32x4 *>(brow + 16 * pt * DP + 16 * (sn + j));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < CS; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int c = 0; c < CNT; ++c) {
                                const f32x4 wq = wa[SET][j][c];
                                const float wv = e == 0 ? wq.x : (e == 1 ? wq.y : (e == 2 ? wq.z : wq.w));
#pragma unroll
                                for (int pt = 0; pt < PT; ++pt) {
                                    const f32x4 bq = b[j][pt];
                                    const float bv = e == 0 ? bq.x : (e == 1 ? bq.y : (e == 2 ? bq.z : bq.w));
                                    acc[pt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, bv, acc[pt][c], 0, 0, 0);
                                }
                            }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < CS; ++j)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt) b[j][pt] = bn[j][pt];
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>;
                load_w(0, I0{});
                load_w(CS, I1{});
#pragma unroll
                for (int j = 0; j < CS; ++j)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) b[j][pt] = *reinterpret_cast<const f32x4 *>(brow + 16 * pt * DP + 16 * j);
                for (int s0 = 0; s0 < ((ABL & 8) ? 3 * CS : KSP); s0 += 3 * CS) {
                    load_w(s0 + 2 * CS, I2{});
                    comp(s0, I0{});
                    load_w(s0 + 3 * CS, I0{});
                    comp(s0 + CS, I1{});
                    load_w(s0 + 4 * CS, I1{});
                    comp(s0 + 2 * CS, I2{});
                }
                stamp(11);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const int prow = 16 * pt + li;
                    const int pimg = b0 + prow / P;
                    if (pimg >= n_img) continue;
                    const size_t orow = (size_t)pimg * P + prow % P;
#pragma unroll
                    for (int c = 0; c < CNT; ++c) {
                        const int n = 16 * (tile0 + c) + 4 * kk;
                        const f32x4 bq = *reinterpret_cast<const f32x4 *>(w.bp + n);
                        f32x4 v = acc[pt][c];
                        v.x = v.x + bq.x; v.y = v.y + bq.y; v.z = v.z + bq.z; v.w = v.w + bq.w;
                        if constexpr (RESID) {
                            const f32x4 rv = *reinterpret_cast<const f32x4 *>(x + orow * CIN + n);
                            v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w;
                        }
                        *reinterpret_cast<f32x4 *>(out + orow * COUT + n) = v;
                    }
                }
                stamp(12);
            }
        };
        // The same phase for a P3 project layer: per k-step of 32 a lane reads its 8 values of each row tile from dwo (two
        // ds_read_b128), splits them into the three bf16 planes (p3_split8 -- the gate is already multiplied in: one rounding,
        // as in the tiled GEMM's operand fetch) and runs the six passes of p3_step on CNT x PT accumulators; the weight
        // fragments (three planes per tile, 16 bytes per lane each) come straight from memory three k-steps deep.  The split
        // of step s + 1 is issued among the MFMAs of step s (one wave per SIMD: nothing else would fill the gaps).
        auto project3 = [&](auto cntc, int tile0) __attribute__((always_inline)) {
            constexpr int CNT = decltype(cntc)::value;
            if constexpr (CNT > 0 && P3) {
                constexpr int KS32 = E / 32;
                static_assert(E % 32 == 0, "whole k-steps");
                const float *brow = s_dwo + li * DP + 8 * kk;  // + 16 pt * DP + 32 s
                const u32x4 *wl = reinterpret_cast<const u32x4 *>(w.wp3) + (size_t)tile0 * 192 + lane;
                f32x4 acc[PT][CNT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int c = 0; c < CNT; ++c) acc[pt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                constexpr int RING = CNT * PT > 6 ? 2 : 3;  // weight sets in flight (12 CNT registers each: five tiles x three sets spill)
                static_assert(KS32 % RING == 0, "whole rounds of weight requests");
                u32x4 wa[RING][CNT][3];
                auto load_w = [&](int s0, auto setc) __attribute__((always_inline)) {
                    constexpr int SET = decltype(setc)::value;
                    s0 = s0 < KS32 ? s0 : KS32 - 1;  // past the end: the last step again, never used
                    if (ABL & 16) s0 = 0;
                    const u32x4 *src = wl + (size_t)s0 * w.nt16 * 192;
#pragma unroll
                    for (int c = 0; c < CNT; ++c)
#pragma unroll
                        for (int p = 0; p < 3; ++p) wa[SET][c][p] = src[(c * 3 + p) * 64];
                };
                P3Act pb[PT];
                auto read_b = [&](int s, f32x4 (&rb)[PT][2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        rb[pt][0] = *reinterpret_cast<const f32x4 *>(brow + 16 * pt * DP + 32 * s);
                        rb[pt][1] = *reinterpret_cast<const f32x4 *>(brow + 16 * pt * DP + 32 * s + 4);
                    }
                };
                auto comp = [&](int s, auto setc) __attribute__((always_inline)) {
                    constexpr int SET = decltype(setc)::value;
                    f32x4 rb[PT][2];
                    read_b(s + 1 < KS32 ? s + 1 : s, rb);
                    __builtin_amdgcn_sched_barrier(0);
#define PB_BLK_PASS(WP_, AP_)                                                                        \
    _Pragma("unroll") for (int c = 0; c < CNT; ++c) _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) \
        acc[pt][c] = p3_mfma(wa[SET][c][WP_], pb[pt].AP_, acc[pt][c]);
                    PB_BLK_PASS(2, h)
                    PB_BLK_PASS(1, m)
                    PB_BLK_PASS(1, h)
                    PB_BLK_PASS(0, l)
                    PB_BLK_PASS(0, m)
                    PB_BLK_PASS(0, h)
#undef PB_BLK_PASS
                    P3Act pn[PT];
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) pn[pt] = p3_split8(rb[pt][0], rb[pt][1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) pb[pt] = pn[pt];
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>;
                load_w(0, I0{});
                load_w(1, I1{});
                {
                    f32x4 rb[PT][2];
                    read_b(0, rb);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) pb[pt] = p3_split8(rb[pt][0], rb[pt][1]);
                }
                if constexpr (RING == 3) {
                    for (int s0 = 0; s0 < ((ABL & 8) ? 3 : KS32); s0 += 3) {
                        load_w(s0 + 2, I2{});
                        comp(s0, I0{});
                        load_w(s0 + 3, I0{});
                        comp(s0 + 1, I1{});
                        load_w(s0 + 4, I1{});
                        comp(s0 + 2, I2{});
                    }
                } else {
                    for (int s0 = 0; s0 < ((ABL & 8) ? 2 : KS32); s0 += 2) {
                        comp(s0, I0{});
                        load_w(s0 + 2, I0{});
                        comp(s0 + 1, I1{});
                        load_w(s0 + 3, I1{});
                    }
                }
                stamp(11);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const int prow = 16 * pt + li;
                    const int pimg = b0 + prow / P;
                    if (pimg >= n_img) continue;
                    const size_t orow = (size_t)pimg * P + prow % P;
#pragma unroll
                    for (int c = 0; c < CNT; ++c) {
                        const int n = 16 * (tile0 + c) + 4 * kk;
                        const f32x4 bq = *reinterpret_cast<const f32x4 *>(w.bp + n);
                        f32x4 v = acc[pt][c];
                        v.x = v.x + bq.x; v.y = v.y + bq.y; v.z = v.z + bq.z; v.w = v.w + bq.w;
                        if constexpr (RESID) {
                            const f32x4 rv = *reinterpret_cast<const f32x4 *>(x + orow * CIN + n);
                            v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w;
                        }
                        *reinterpret_cast<f32x4 *>(out + orow * COUT + n) = v;
                    }
                }
                stamp(12);
            }
        };
        if constexpr (P3) {
            if (wave < 4) project3(std::integral_constant<int, TA>{}, wave * TA);
        } else {
            if (wave < 4) project(std::integral_constant<int, TA>{}, wave * TA);
            else project(std::integral_constant<int, TB>{}, 4 * TA + (wave - 4) * TB);
        }
    }
    if constexpr ((ABL & 32) != 0) {
        if (lane == 0) {
            unsigned long long *d = w.dbg + ((size_t)blockIdx.x * 8 + wave) * 16;
            for (int i = 0; i < 16; ++i) d[i] = st_[i];
        }
    }
}

}  // namespace pbe
