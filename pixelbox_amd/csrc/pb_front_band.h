// pb_front_band.h -- fused MBConv front for the early blocks (maps 64 / 32 / 16 pixels wide): expand 1x1 (+bias+SiLU)
// -> depthwise KSxKS stride S (+bias+SiLU) -> SE partial sums, with the 6x-expanded activation held in an LDS ring of
// image rows.  Replaces, for those blocks, k_front_roll (register ring per wave: 16-column strips whose x halo is
// recomputed and whose output lanes are 7 / 6 of 16 at stride 2) and k_mbconv_small<MR = 4>.
// Reference: the layers `MODEL.run` evaluates (src/image_hashes/efficientnet.rs:34; architecture resources/train.py:30-46).
//
// Shape of the work.  A workgroup = 4 waves owns (image, tile of 16 expanded channels, band of output rows) and walks
// down the band in STEPS of RPS output rows, RPS * Wo = 64 outputs per channel quad.  Per step:
//   expand phase   the S * RPS new input rows are expanded on the f32 matrix cores: a (row, 16-pixel tile, 16-channel
//                  tile) is KC * 4 MFMAs with the operand maps and the k order of k_gemm1x1 (weights of the wave's channel
//                  tile resident in registers), + bias, SiLU, and ONE ds_write_b128 per lane into the ring;
//   barrier
//   filter phase   wave w is channel quad q = w of the tile for the whole kernel; its 64 lanes are the step's 64 output pixels.  Per tap
//                  one conflict-free ds_read_b128 (consecutive lanes read consecutive 16-byte slots) and four fused
//                  multiply-adds whose tap operand is WAVE-UNIFORM: the 36 tap values of a 3x3 quad live in scalar
//                  registers for the whole kernel, those of a 5x5 quad are re-read per filter row -- no vector register and
//                  no LDS bandwidth goes to filter taps -- then bias (the accumulator's start value), SiLU, a 16-byte
//                  store and the fixed-point SE sum;
//   barrier        (the next expand phase overwrites ring rows this phase read).
// Every lane of every phase produces a needed value: no x halo, no masked output lanes; the y halo (KS - S rows) is
// recomputed only where a band starts.
//
// LDS image: ring[RR rows][4 quads][PLS slots] of float4, i.e. one PLANE per (row, channel quad) with the pixels of the
// row (zero columns left and right: the depthwise padding) in consecutive 16-byte slots; for stride 2 a plane is split in
// an even-pixel half and an odd-pixel half, so that output lane ox reads slot (kx & 1) * HALF + ox + (kx >> 1) -- again
// consecutive.  The MFMA layout (lane = pixel li of the tile x channel quad kq) stores 8 consecutive slots per 8-lane
// store group (stride 2: two runs of four).  PLS is a multiple of 4, so ring rows differ by multiples of 16 slots and the
// rows a 16-lane read group straddles (maps 16 pixels wide) keep their slots distinct modulo the 16 slots of a bank row.
//
// Output stores (round 5): the filter phase leaves a wave with ONE channel quad of 64 pixels -- stored from there, a store instruction
// touches 64 separate 16-byte pieces 4 E bytes apart, and blocks 1 and 2 (201 / 302 MB of output) ran at the rate of those stores, not
// of their arithmetic (ablation builds, profiles/r05_front_band_ablation.txt: everything but the stores removed: 162 -> 152 us; the
// stores removed too: 65).  The step's 64 x 16 results are now staged in LDS and stored after the step's closing barrier, four
// lanes per pixel: a pixel's 64 bytes leave in one piece, and the stores issue under the next step's expand phase.
//
// Same arithmetic as the kernels it replaces (k order of the expand chain, (ky, kx) tap order with one fused
// multiply-add per tap, 2^-24 fixed-point SE sums): bit-identical outputs, whatever the band count.
#pragma once
#include "pb_embed_kernels.h"

#ifndef PB_BAND_WAVES
#define PB_BAND_WAVES 4  // waves per SIMD the register allocation may assume (128 VGPRs: none of the shapes spills; they use 78-102)
#endif

namespace pbe {

template <int KS, int S, int WT, int RPS>
struct FrontBandGeom {
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int W = 16 * WT;           // input map width (and height: the maps are square)
    static constexpr int Wo = W / S;            // output width
    static constexpr int PLW = W + 2 * PAD;     // padded row length in pixels
    // stride 2: slots per parity half (even: so that PLS is a multiple of 4); 36 = 4 (mod 8) keeps the two 4-slot runs
    // of an 8-lane store group on different banks where the LDS budget allows it (W = 64)
    static constexpr int HALF = S == 2 ? (W == 64 ? 36 : ((PLW + 1) / 2 + 1) / 2 * 2) : 0;
    static constexpr int PLS = S == 2 ? 2 * HALF : (PLW + 3) / 4 * 4;  // slots per plane
    static constexpr int RN = S * RPS;          // new input rows per step
    static constexpr int R0 = KS - S;           // rows carried from the step before (primed at the band's start)
    static constexpr int RR = RN + R0;          // ring rows
    static constexpr int NP = RN * WT;          // (row, pixel tile) pairs expanded per step, per channel tile
    static constexpr int TPW = NP / 4;          // pairs per wave (4 waves per channel tile)
    static constexpr int NP0 = R0 * WT;         // pairs of the priming rows
    static constexpr int TPW0 = (NP0 + 3) / 4;
    static constexpr int NQ = 4;                // channel quads per workgroup: one 16-channel MFMA tile
    static constexpr int STG_PITCH = 5;         // float4 per staged pixel: 16 channels + one float4 of padding (conflict-free both ways)
    static constexpr size_t RING_BYTES = (size_t)RR * NQ * PLS * 16;
    static constexpr size_t LDS_BYTES = RING_BYTES + (size_t)64 * STG_PITCH * 16;  // + the step's 64 x 16 outputs staged for the stores
    static_assert(RPS * Wo == 64, "a step is 64 output pixels per channel quad");
    static_assert(NP % 4 == 0, "pairs of a step divide over the 4 waves of a channel tile");
    static_assert(W % S == 0 && PLS % 4 == 0, "geometry");
    // slot of padded pixel px' (0 .. PLW-1) inside a plane
    __host__ __device__ static constexpr int slot(int pxp) { return S == 2 ? (pxp & 1) * HALF + (pxp >> 1) : pxp; }
};

// dw_wq: the depthwise taps regrouped per channel quad, [E / 4][KS * KS] float4 (tap t of channels 4 cq .. 4 cq + 3), so that
// the taps of a quad are contiguous for the scalar loads of the filter phase.
// CIN = input channels (compile time: 16 / 24 / 40); the expand weights wt[Kpad][Epad] are zero beyond row CIN, so the
// activation lanes whose k slot lies beyond CIN load a valid dummy address and contribute exact zeros.
// se: squeeze-excite tail (SeTail; sp = 0: none).
// grid = round_up(B * n_bands, 8) * E / 16 (1-D, see the decode); n_items = B * n_bands; block = 256; dynamic LDS = FrontBandGeom::LDS_BYTES.
template <int KS, int S, int CIN, int WT, int RPS>
__global__ __launch_bounds__(256, PB_BAND_WAVES) void k_front_band(
    const float *__restrict__ x, const float *__restrict__ wt, int Epad, const float *__restrict__ bias_e,
    const f32x4 *__restrict__ dw_wq, const float *__restrict__ dw_b, int E, float *__restrict__ out,
    long long *__restrict__ part, int n_bands, int rows_per_band, unsigned n_items, SeTail se, int ipw = 1) {
    using G = FrontBandGeom<KS, S, WT, RPS>;
    constexpr int KC = (CIN + 15) / 16;
    constexpr int PAD = G::PAD, W = G::W, H = G::W, Wo = G::Wo, Ho = G::Wo, RR = G::RR, RN = G::RN, R0 = G::R0, PLS = G::PLS;
    constexpr int TPW = G::TPW, NQ = G::NQ;
    constexpr int PF = TPW * KC <= 4 ? TPW : (TPW < 2 ? TPW : 2);  // pairs per group: their operands are in flight ahead of use
    static_assert(TPW % PF == 0, "groups of PF pairs");
    extern __shared__ __attribute__((aligned(16))) f32x4 s_ring[];  // [RR][NQ][PLS], then the store staging [64 pixels][STG_PITCH]
    f32x4 *s_stage = s_ring + RR * NQ * PLS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what it indexes stays in SGPRs
    const int li = lane & 15, kq = lane >> 4;
    // 1-D grid, decoded so that the E / 16 channel tiles of one (image, band) are the blocks i, i + 8, i + 16, ...: blocks
    // are dealt round-robin over the 8 XCDs, so those tiles run at about the same time on ONE XCD and share the image's input
    // rows through its L2 (speed only: any placement gives the same results).  With (band, image, tile) as a 3-D grid the
    // input was streamed from beyond L2 once per channel tile -- 6 to 15 times.
    const unsigned nz = (unsigned)E >> 4;
#if defined(PB_BAND_ABL) && (PB_BAND_ABL & 1)
    const unsigned items_pad = (n_items + 7u) / 8u * 8u;  // ablation: channel tile slowest (every tile of the batch before the next)
    const unsigned item = blockIdx.x % items_pad;
    const int e0 = (int)(blockIdx.x / items_pad) * 16;
#else
    const unsigned item = (blockIdx.x >> 3) / nz * 8u + (blockIdx.x & 7u);  // (image, band) pair
    const int e0 = (int)((blockIdx.x >> 3) % nz) * 16;
#endif
    // ipw (round 6): a workgroup walks `ipw` consecutive (image, band) items of its channel tile one after the other -- weights, taps
    // and biases are loaded once, and the grid has ipw times fewer workgroups.  Where a band is ONE step (block 5 at 128 x 128: the
    // whole 16 x 16 -> 8 x 8 image; 7 680 workgroups of 2.3 us each) the launch was bound by the rate at which workgroups can be
    // dispatched -- one wave per SIMD resident on average (profiles/r06_embed_issue_floor_before.txt); ipw is timed per (block, batch).
    const unsigned item0 = item * (unsigned)ipw;
    if (item0 >= n_items) return;
    const int n_steps = rows_per_band / RPS;

    // ---- zero columns of every plane (left / right depthwise padding): written once, never overwritten
    for (int i = tid; i < RR * NQ * 2 * PAD; i += 256) {
        const int j = i % (2 * PAD), pl = i / (2 * PAD);
        const int pxp = j < PAD ? j : W + j;
        s_ring[pl * PLS + G::slot(pxp)] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // ---- expand role: wave g takes the pairs g, g + 4, ... of the row range
    const int g = wave;
    float wreg[KC][4];  // A operand: lane (li, kq) holds wt[k = 16 s + 4 kq + e][e0 + 16 c + li]
#pragma unroll
    for (int s2 = 0; s2 < KC; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e) wreg[s2][e] = wt[(size_t)(16 * s2 + 4 * kq + e) * Epad + e0 + li];
    const f32x4 bev = *reinterpret_cast<const f32x4 *>(bias_e + e0 + 4 * kq);
    const float *xb = x + 4 * kq;  // + the item's image (set per item below)
    // this lane's slot inside a ring row, for pixel tile 0 (a tile further on is + 16 slots, stride 2: + 8)
    const unsigned wr_slot = (unsigned)(kq * PLS + G::slot(li + PAD));
    constexpr unsigned WR_TILE = S == 2 ? 8u : 16u;
    static_assert(S == 1 || PAD % 2 == 1 || true, "tile offsets keep the parity of li + PAD");

    // activation operands of pair p of the row range starting at input row iy_first: lane (li, kq) loads
    // x[row][16 pt + li][16 s + 4 kq .. + 3]; rows outside the image read row 0 (never used: zero rows are written instead),
    // k slots beyond CIN read slot 0 (multiplied by zero weights)
    auto load_pair = [&](int iy_first, int p, f32x4 (&xv)[KC]) __attribute__((always_inline)) {
        const int r = p / WT, pt = p % WT;
        const int iy = iy_first + r;
        const int iyc = (iy >= 0 && iy < H) ? iy : 0;
        const float *ptr = xb + (unsigned)((iyc * W + 16 * pt + li) * CIN);
#pragma unroll
        for (int s2 = 0; s2 < KC; ++s2)
            xv[s2] = *reinterpret_cast<const f32x4 *>(ptr + ((16 * s2 + 16 <= CIN || (16 * s2 + 4 * kq) < CIN) ? 16 * s2 : 0));
    };
    // MFMA chains (interleaved over the group's pairs) + bias + SiLU, one 16-byte store per lane and pair into the ring.
    // The operands of the NEXT group (`next_first`, pairs g + 4 (jn + jj)) are requested right after this group's last MFMA
    // has been issued, into the registers the MFMAs have just read: they are in flight under the epilogue and, for a step's
    // last group, under the whole filter phase.  (Requested BEFORE the MFMAs, as a software pipeline would have it, the new
    // values need registers of their own and hipcc spills them -- behind a wait for the loads.)
    auto expand_group = [&](int iy_first, int rel_first, int j0, f32x4 (&xg)[PF][KC], int next_first, int jn, auto check)
                            __attribute__((always_inline)) {
        constexpr bool CHECK = decltype(check)::value;
        f32x4 acc[PF];
#pragma unroll
        for (int jj = 0; jj < PF; ++jj) acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if defined(PB_BAND_ABL) && (PB_BAND_ABL & 8)
#pragma unroll
        for (int jj = 0; jj < PF; ++jj) acc[jj] = xg[jj][0];  // ablation: no expand MFMAs (the operands are still loaded and consumed)
#else
#pragma unroll
        for (int s2 = 0; s2 < KC; ++s2)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int jj = 0; jj < PF; ++jj) {
                    const f32x4 a = xg[jj][s2];
                    const float av = e == 0 ? a.x : (e == 1 ? a.y : (e == 2 ? a.z : a.w));
                    acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s2][e], av, acc[jj], 0, 0, 0);
                }
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < PF; ++jj) load_pair(next_first, g + 4 * (jn + jj), xg[jj]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < PF; ++jj) {
            const int p = g + 4 * (j0 + jj);
            const int r = p / WT, pt = p % WT;
            const unsigned rs = (unsigned)(rel_first + r) % (unsigned)RR;
            f32x4 v;
#if defined(PB_BAND_ABL) && (PB_BAND_ABL & 32)
            v.x = acc[jj].x + bev.x; v.y = acc[jj].y + bev.y; v.z = acc[jj].z + bev.z; v.w = acc[jj].w + bev.w;  // ablation: no SiLU
#else
            v.x = silu_f(acc[jj].x + bev.x); v.y = silu_f(acc[jj].y + bev.y); v.z = silu_f(acc[jj].z + bev.z); v.w = silu_f(acc[jj].w + bev.w);
#endif
            s_ring[rs * (unsigned)(NQ * PLS) + wr_slot + (unsigned)pt * WR_TILE] = v;
        }
        if constexpr (CHECK) {
            // rows outside the image (the first / last step of the image only): the depthwise padding is zeros of the
            // EXPANDED activation -- the same lanes overwrite what they have just stored (LDS operations of a wave complete in order)
#pragma unroll
            for (int jj = 0; jj < PF; ++jj) {
                const int p = g + 4 * (j0 + jj);
                const int r = p / WT, pt = p % WT;
                const int iy = iy_first + r;
                if (!(iy >= 0 && iy < H))  // wave-uniform
                    s_ring[(unsigned)(rel_first + r) % (unsigned)RR * (unsigned)(NQ * PLS) + wr_slot + (unsigned)pt * WR_TILE] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // ---- filter role: channel quad q = wave, lane = output pixel (oyl, ox) of the step
    const int q = wave;
    const int oyl = lane / Wo, ox = lane % Wo;
    const f32x4 *tq = dw_wq + (size_t)(e0 / 4 + q) * (KS * KS);
    const f32x4 dbv = *reinterpret_cast<const f32x4 *>(dw_b + e0 + 4 * q);  // wave-uniform
    // store role (after the step's closing barrier): lane = (pixel 16 wave + lane / 4 of the step, channel quad lane % 4)
    const int sp_px = 16 * wave + (lane >> 2), sp_q = lane & 3;
    float *op = out;  // set per item below
    const f32x4 *stg_rd = s_stage + sp_px * G::STG_PITCH + sp_q;
    f32x4 *stg_wr = s_stage + lane * G::STG_PITCH + q;
    auto flush_stage = [&]() __attribute__((always_inline)) {
#if defined(PB_BAND_ABL) && (PB_BAND_ABL & 2)
        const f32x4 r4 = *stg_rd;
        if (r4.x == 12345.678f) *reinterpret_cast<f32x4 *>(op) = r4;  // ablation: no output stores
#else
        *reinterpret_cast<f32x4 *>(op) = *stg_rd;
#endif
        op += (size_t)RPS * Wo * E;
    };
    const unsigned rd_col = (unsigned)(q * PLS + ox);  // + compile-time slot offset of kx
    int qmax = 0;  // largest converted output seen (se_range_check)
  for (int ii = 0; ii < ipw; ++ii) {
    const unsigned item_i = item0 + (unsigned)ii;
    if (item_i >= n_items) break;  // (uniform over the workgroup)
    const int band = (int)(item_i % (unsigned)n_bands), b = (int)(item_i / (unsigned)n_bands);
    const int oy_b = band * rows_per_band;
    const int iy_origin = oy_b * S - PAD;  // input row of ring position 0
    xb = x + (size_t)b * H * W * CIN + 4 * kq;
    op = out + ((size_t)(b * Ho + oy_b + sp_px / Wo) * Wo + sp_px % Wo) * E + e0 + 4 * sp_q;
    ll4 psum = {0, 0, 0, 0};

    // ---- priming rows (ring positions 0 .. R0-1) of the band
    if constexpr (G::NP0 > 0) {
#pragma unroll
        for (int j = 0; j < G::TPW0; ++j) {
            const int p = g + 4 * j;
            if (p < G::NP0) {
                f32x4 xv[KC];
                load_pair(iy_origin, p, xv);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < KC; ++s2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float av = e == 0 ? xv[s2].x : (e == 1 ? xv[s2].y : (e == 2 ? xv[s2].z : xv[s2].w));
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s2][e], av, acc, 0, 0, 0);
                    }
                const int r = p / WT, pt = p % WT;
                const int iy = iy_origin + r;
                f32x4 v = {silu_f(acc.x + bev.x), silu_f(acc.y + bev.y), silu_f(acc.z + bev.z), silu_f(acc.w + bev.w)};
                if (!(iy >= 0 && iy < H)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                s_ring[(unsigned)r * (unsigned)(NQ * PLS) + wr_slot + (unsigned)pt * WR_TILE] = v;
            }
        }
    }
    f32x4 xq[PF][KC];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_pair(iy_origin + R0, g + 4 * j, xq[j]);

    int sbase = 0;  // (t * RN) % RR: ring position of the first input row the step's first output row reads
    for (int t = 0; t < n_steps; ++t) {
        const int rel_first = R0 + t * RN;
        const int iy_first = iy_origin + rel_first;
        // ---- expand phase: groups of PF pairs; each group requests the operands of the one after it (the step's last
        // group those of the next step's first: after the last step they re-read valid rows and are never used)
#pragma unroll
        for (int j0 = 0; j0 < TPW; j0 += PF) {
            const bool last = j0 + PF >= TPW;
            const int next_first = !last ? iy_first : ((t + 1 < n_steps) ? iy_first + RN : iy_first);
            const int jn = last ? 0 : j0 + PF;
            expand_group(iy_first, rel_first, j0, xq, next_first, jn, std::true_type{});
        }
#ifndef PB_BAND_DIRECT_STORE
        // the PREVIOUS step's outputs leave now, four lanes per pixel, behind this step's operand requests: nothing this wave waits
        // for before the next step's expand phase was issued after them (the staging block is rewritten by the filter phase below,
        // i.e. behind the barrier this wave reaches after these reads)
        if (t > 0) flush_stage();
#endif
        __syncthreads();
        // ---- filter phase
        {
            unsigned rb = (unsigned)(sbase + oyl * S);
            rb = rb >= (unsigned)RR ? rb - RR : rb;
            f32x4 acc = dbv;
#if defined(PB_BAND_ABL) && (PB_BAND_ABL & 16)
            constexpr bool NO_TAPS = true;
#else
            constexpr bool NO_TAPS = false;
#endif
            if constexpr (NO_TAPS) {  // ablation: one window read, no taps
                const f32x4 v0 = s_ring[rb * (unsigned)(NQ * PLS) + rd_col];
                acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
            } else if constexpr (KS * KS <= 9) {
                // 3x3: the 36 tap values stay in scalar registers for the whole kernel (loop-invariant scalar loads)
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    unsigned rk = rb + ky;
                    rk = rk >= (unsigned)RR ? rk - RR : rk;
                    const unsigned ro = rk * (unsigned)(NQ * PLS) + rd_col;
                    f32x4 v[KS];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) v[kx] = s_ring[ro + (S == 2 ? (kx & 1) * G::HALF + (kx >> 1) : kx)];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) dw_tap(acc, v[kx], tq[ky * KS + kx]);
                }
            } else {
                // 5x5: 100 tap values do not fit the scalar register file beside everything else, so the taps of filter row
                // ky + 1 are re-read (scalar loads: the offset is made opaque per row, which keeps hipcc from hoisting all 25
                // loads out of the step loop and spilling them to vector lanes) while row ky is applied
                f32x4 wn[KS];
                {
                    int zo = 0;
                    asm volatile("" : "+s"(zo));
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) wn[kx] = tq[zo + kx];
                }
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    unsigned rk = rb + ky;
                    rk = rk >= (unsigned)RR ? rk - RR : rk;
                    const unsigned ro = rk * (unsigned)(NQ * PLS) + rd_col;
                    f32x4 wv[KS];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) wv[kx] = wn[kx];
                    if (ky + 1 < KS) {
                        int zo = (ky + 1) * KS;
                        asm volatile("" : "+s"(zo));
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) wn[kx] = tq[zo + kx];
                    }
                    f32x4 v[KS];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) v[kx] = s_ring[ro + (S == 2 ? (kx & 1) * G::HALF + (kx >> 1) : kx)];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) dw_tap(acc, v[kx], wv[kx]);
                    asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));
                }
            }
#if defined(PB_BAND_ABL) && (PB_BAND_ABL & 32)
            const f32x4 r4 = acc;
#else
            const f32x4 r4 = {silu_f(acc.x), silu_f(acc.y), silu_f(acc.z), silu_f(acc.w)};
#endif
#ifdef PB_BAND_DIRECT_STORE  // comparison build: rounds 3-4's stores, a channel quad of 64 pixels per wave
            *reinterpret_cast<f32x4 *>(out + ((size_t)(b * Ho + oy_b + t * RPS + oyl) * Wo + ox) * E + e0 + 4 * q) = r4;
#else
            *stg_wr = r4;  // stored after the closing barrier (flush_stage)
#endif
#if defined(PB_BAND_ABL) && (PB_BAND_ABL & 4)
            psum.x += __float_as_int(r4.x) ^ __float_as_int(r4.y) ^ __float_as_int(r4.z) ^ __float_as_int(r4.w);  // ablation: no fixed-point SE sums
#else
            se_acc(psum, qmax, r4);
#endif
        }
        sbase += RN;
        sbase = sbase >= RR ? sbase - RR : sbase;
        __syncthreads();
    }
#ifndef PB_BAND_DIRECT_STORE
    if (n_steps > 0) flush_stage();  // the last step's
#endif
    // ---- SE partial of this (image, band, quad): exact integer sum over the 64 lanes
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        psum.x += __shfl_xor(psum.x, m, 64);
        psum.y += __shfl_xor(psum.y, m, 64);
        psum.z += __shfl_xor(psum.z, m, 64);
        psum.w += __shfl_xor(psum.w, m, 64);
    }
    if (lane == 0) se_part_store(part + ((size_t)b * n_bands + band) * E + e0 + 4 * q, psum);
    // ---- the workgroup that completes the image computes its squeeze-excite gate (se_gate_image; the ring is free by now; the host
    // launches ipw = 1 with a squeeze-excite tail)
    if (se.sp) {
        if (se_arrive(se.cnt + b, nz * (unsigned)n_bands, reinterpret_cast<unsigned *>(s_ring)))
            se_gate_image_sp<256>(part + (size_t)b * n_bands * E, n_bands, E, se, se.gate + (size_t)b * E, reinterpret_cast<float *>(s_ring + 1));
    }
  }
    se_range_check(qmax, part - 1);
}

}  // namespace pbe
