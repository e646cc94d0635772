// pb_gemm_p3.hip -- instantiations and dispatch of k_gemm_p3 (pb_gemm_p3.h): the dense contractions of the late
// EfficientNet-B0 layers (reference: `MODEL.run`, src/image_hashes/efficientnet.rs:34; architecture resources/train.py:30-46)
// from three bf16 pieces per f32 operand on the bf16 matrix cores.  gfx950 only.
#include "pb_embed_common.h"
#include "pb_gemm_p3.h"
#include "pb_gemm_p3_launch.h"

namespace pbe {

namespace {

template <int NR, int MR, bool GATE, int NW, int EPI, bool DIRECT, bool KT>
void launch_t(hipStream_t st, const P3Args &a) {
    const long rows = 16L * NW * MR;
    const dim3 grid((unsigned)((a.M + rows - 1) / rows), (unsigned)(a.tiles16 / NR));
    hipLaunchKernelGGL((k_gemm_p3<NR, MR, GATE, NW, EPI, DIRECT, KT>), grid, dim3(64 * NW), 0, st, a.act, (int)a.M, a.K,
                       reinterpret_cast<const u32x4 *>(a.wt3), a.tiles16, a.bias, a.N, a.gate, a.hw, a.resid, a.do_silu, a.out, a.scale,
                       a.out_u8);
}

// shapes of the store epilogue: (NR, MR) for the LDS forms, NR for the one-wave form
#define PB_P3_LDS_SHAPES(X) X(2, 1) X(3, 1) X(4, 1) X(5, 1) X(6, 1) X(7, 1) X(8, 1) X(2, 2) X(3, 2) X(4, 2)
#define PB_P3_DIRECT_SHAPES(X) X(1) X(2) X(4)

template <bool GATE, bool KT>
bool launch_epi0(int nr, int mr, int nw, hipStream_t st, const P3Args &a, bool probe) {
#define X(NRV, MRV)                                                             \
    if (nr == NRV && mr == MRV && (nw == 4 || nw == 8)) {                       \
        if (!probe) {                                                           \
            if (nw == 8) launch_t<NRV, MRV, GATE, 8, 0, false, KT>(st, a);      \
            else launch_t<NRV, MRV, GATE, 4, 0, false, KT>(st, a);              \
        }                                                                       \
        return true;                                                            \
    }
    PB_P3_LDS_SHAPES(X)
#undef X
#define X(NRV)                                                       \
    if (nr == NRV && mr == 1 && nw == 1) {                           \
        if (!probe) launch_t<NRV, 1, GATE, 1, 0, true, KT>(st, a);   \
        return true;                                                 \
    }
    PB_P3_DIRECT_SHAPES(X)
#undef X
    return false;
}

bool dispatch(int nr, int mr, int nw, int epi, bool gate, bool kt, hipStream_t st, const P3Args *a, bool probe) {
    static const P3Args none{};
    const P3Args &aa = a ? *a : none;
    if (epi == 0) {
        if (gate) return kt ? launch_epi0<true, true>(nr, mr, nw, st, aa, probe) : launch_epi0<true, false>(nr, mr, nw, st, aa, probe);
        return kt ? launch_epi0<false, true>(nr, mr, nw, st, aa, probe) : launch_epi0<false, false>(nr, mr, nw, st, aa, probe);  // (un-gated with a half-empty last step: the K = 80 / 112 expand layers, round 6)
    }
    if (gate || kt) return false;
    if (epi == 1) {
#define X(NRV, MRV)                                                                  \
    if (nr == NRV && mr == MRV && (nw == 4 || nw == 8)) {                            \
        if (!probe) {                                                                \
            if (nw == 8) launch_t<NRV, MRV, false, 8, 1, false, false>(st, aa);      \
            else launch_t<NRV, MRV, false, 4, 1, false, false>(st, aa);              \
        }                                                                            \
        return true;                                                                 \
    }
        X(4, 1) X(5, 1) X(8, 1) X(4, 2)
#undef X
#define X(NRV)                                                              \
    if (nr == NRV && mr == 1 && nw == 1) {                                  \
        if (!probe) launch_t<NRV, 1, false, 1, 1, true, false>(st, aa);     \
        return true;                                                        \
    }
        X(1) X(2) X(4)
#undef X
        return false;
    }
    if (epi == 2) {
#define X(NRV)                                                                  \
    if (nr == NRV && mr == 1 && nw == 4) {                                      \
        if (!probe) launch_t<NRV, 1, false, 4, 2, false, false>(st, aa);        \
        return true;                                                            \
    }
        X(2) X(4) X(8)
#undef X
#define X(NRV)                                                              \
    if (nr == NRV && mr == 1 && nw == 1) {                                  \
        if (!probe) launch_t<NRV, 1, false, 1, 2, true, false>(st, aa);     \
        return true;                                                        \
    }
        X(1) X(2)
#undef X
        return false;
    }
    return false;
}

}  // namespace

bool p3_has(int nr, int mr, int nw, int epi, bool gate, bool ktail) { return dispatch(nr, mr, nw, epi, gate, ktail, nullptr, nullptr, true); }

bool p3_launch(int nr, int mr, int nw, int epi, hipStream_t st, const P3Args &a) {
    return dispatch(nr, mr, nw, epi, a.gate != nullptr, (a.K & 31) != 0, st, &a, false);
}

}  // namespace pbe
