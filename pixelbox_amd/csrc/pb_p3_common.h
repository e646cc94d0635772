// pb_p3_common.h -- the operand split and the k-step of the piece arithmetic (P3; definition and measurements: pb_gemm_p3.h), shared
// by every kernel form that computes a P3 layer (k_gemm_p3, k_block_small, k_mbconv_small).
#pragma once
#include <type_traits>
#include "pb_embed_common.h"

namespace pbe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the three piece planes of 8 consecutive k of one row, packed bf16 pairs (element j = half j & 1 of dword j >> 1)
struct P3Act {
    u32x4 h, m, l;
};

__device__ __forceinline__ P3Act p3_split8(const f32x4 &a0, const f32x4 &a1) {
    const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    uint32_t hb[8], mb[8], lb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t xb = __float_as_uint(x[e]);
        hb[e] = xb;
        const float r1 = x[e] - __uint_as_float(xb & 0xFFFF0000u);  // exact
        const uint32_t rb = __float_as_uint(r1);
        mb[e] = rb;
        lb[e] = __float_as_uint(r1 - __uint_as_float(rb & 0xFFFF0000u));  // exact
    }
    P3Act p;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p.h[j] = __builtin_amdgcn_perm(hb[2 * j + 1], hb[2 * j], 0x07060302u);  // the top halves of two floats
        p.m[j] = __builtin_amdgcn_perm(mb[2 * j + 1], mb[2 * j], 0x07060302u);
        p.l[j] = __builtin_amdgcn_perm(lb[2 * j + 1], lb[2 * j], 0x07060302u);
    }
    return p;
}

template <int I, int N, class F>
__device__ __forceinline__ void p3_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        p3_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ f32x4 p3_mfma(const u32x4 &w, const u32x4 &a, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}

// One k-step on one accumulator with all three weight planes at hand (the forms that hold a step's fragments in registers).
__device__ __forceinline__ void p3_step(f32x4 &acc, const u32x4 &wh, const u32x4 &wm, const u32x4 &wl, const P3Act &a) {
    acc = p3_mfma(wl, a.h, acc);
    acc = p3_mfma(wm, a.m, acc);
    acc = p3_mfma(wm, a.h, acc);
    acc = p3_mfma(wh, a.l, acc);
    acc = p3_mfma(wh, a.m, acc);
    acc = p3_mfma(wh, a.h, acc);
}

}  // namespace pbe
