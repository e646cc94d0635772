// pb_embed_common.h -- the small device helpers every embed translation unit shares (SiLU / sigmoid, the fixed-point
// squeeze-excite sums, the depthwise tap, the u8 quantiser of efficientnet.rs:39).  Everything here is inline.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pbe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// SiLU / sigmoid in 5 VALU instructions: e^-x = v_exp_f32(x * -log2 e) (1 ulp on the exponential, plus
// |x| * 7e-8 relative from the rounded argument), reciprocal through v_rcp_f32 (1 ulp).  Absolute error on
// SiLU ~1e-7 * |x|, the same scale as the f32 rounding of the convolution sum feeding it; ocml expf plus a
// correctly rounded divide costs ~25 instructions per element and made the epilogues VALU-bound.  The
// embedding floats are compared at 1e-5 (tests/embed_tol.py).
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// the value held by the previous lane of the 16-lane row (lane 0 of a row: 0)
__device__ __forceinline__ float dpp_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /* row_shr:1 */, 0xF, 0xF, true));
}

// Squeeze-excite pooling sums are accumulated in 64-bit fixed point (2^-24 resolution): integer addition is
// associative, so the pooled mean -- and with it the whole embedding -- is bit-identical whatever the batch size,
// tiling or kernel form that produced the partial sums.  (f32 partial sums made an image's hash depend, in the
// last bit, on how many images shared its batch.)
// The conversion is ONE instruction per value: q = v_cvt_rpi_i32_f32(o * 2^24) = floor(o * 2^24 + 0.5) as an i32
// (round 5; rounds 1-4: rint() through v_rndne + v_cvt + a magnitude test per quad with an i64 conversion behind it --
// 7.5 vector instructions per element where the depthwise phases are issue-bound, now 4).  The 2^-24 grid is this
// library's own definition (the oracle sums f32), so which way a tie rounds is immaterial; what matters is that every
// form goes through this one function.  Domain: |o| < 128 (2^31 / 2^24).  Beyond it the instruction saturates, the pooled
// mean would be wrong and nothing downstream could tell, so se_acc also keeps the largest converted value it has seen (v_max3_i32:
// half an instruction per element) and the accumulating kernel ends with se_range_check: a saturated conversion raises a
// word in pinned host memory that the entry point reads after its stream wait -> PB_ERR_RANGE, never a silent answer.
// (No model seen comes near: these are post-SiLU activations of a batch-normalised network, O(1-10).)
struct ll4 {
    long long x, y, z, w;
};
__device__ __forceinline__ int se_fix(float o) {
    int q;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(o * 16777216.0f));
    return q;
}
__device__ __forceinline__ void se_acc(ll4 &s, int &qmax, const f32x4 &o) {
    const int qx = se_fix(o.x), qy = se_fix(o.y), qz = se_fix(o.z), qw = se_fix(o.w);
    s.x += (long long)qx;
    s.y += (long long)qy;
    s.z += (long long)qz;
    s.w += (long long)qw;
#ifndef PB_NO_SE_RANGE  // (comparison build: what the tracking costs -- nothing measurable, two v_max3_i32 per quad)
    qmax = max(max(qmax, qx), qy);
    qmax = max(max(qmax, qz), qw);
#endif
}
// `flag_slot` holds the DEVICE-VISIBLE ADDRESS of the embedder's range word (pinned host memory) as an integer: for the kernels
// that write pooled partial sums it is part[-1] (the buffer has a 16-byte header, pb_embed_create), for k_block_small a field.
// A conversion that saturated is INT_MAX, which no in-range value reaches (the largest f32 below 2^31 is 2^31 - 128); the inputs
// are post-SiLU (>= -0.28), so the negative end cannot be reached; +inf saturates too, a NaN converts to 0 and is not seen here.
__device__ __forceinline__ void se_range_check(int qmax, const long long *flag_slot) {
    if (qmax == 0x7fffffff) *reinterpret_cast<unsigned *>(static_cast<uintptr_t>(*flag_slot)) = 1u;
}
__device__ __forceinline__ void se_add(ll4 &s, const ll4 &o) {
    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
}

// One depthwise filter tap on a channel quad: acc += v * w, as ONE fused multiply-add per channel (v_fma_f32 /
// v_pk_fma_f32 / v_fmac_f32_dpp) in EVERY depthwise form (strip, rolling, LDS, fused fronts, stem), so that all forms
// still give identical bits.  The unfused mul + add of the CPU restatement costs twice the VALU issue slots, and the
// depthwise phases are VALU-issue-bound; the fused form differs from it by at most half an ulp of the sum per tap
// (it is the more accurate of the two) -- far inside the 1e-5 bar the parity tests hold the embedding to.
// -DPB_DW_UNFUSED restores the two-instruction form for comparison.
__device__ __forceinline__ void dw_tap(f32x4 &acc, const f32x4 &v, const f32x4 &w) {
#ifdef PB_DW_UNFUSED
    const float p0 = v.x * w.x, p1 = v.y * w.y, p2 = v.z * w.z, p3 = v.w * w.w;
    acc.x = acc.x + p0; acc.y = acc.y + p1; acc.z = acc.z + p2; acc.w = acc.w + p3;
#else
    acc.x = __builtin_fmaf(v.x, w.x, acc.x); acc.y = __builtin_fmaf(v.y, w.y, acc.y);
    acc.z = __builtin_fmaf(v.z, w.z, acc.z); acc.w = __builtin_fmaf(v.w, w.w, acc.w);
#endif
}

// efficientnet.rs:39 -- 128u8.saturating_add_signed((f*128).max(-128).min(128) as i8), bit-exact
__device__ __forceinline__ uint8_t quantize_u8(float f) {
    float t = f * 128.0f;
    t = (t != t) ? -128.0f : (t > -128.0f ? t : -128.0f);  // f32::max(NaN, x) = x
    t = t < 128.0f ? t : 128.0f;
    int i;
    if (t >= 127.0f) i = 127;        // `as i8` saturates
    else if (t <= -128.0f) i = -128;
    else i = (int)t;                 // truncation toward zero
    int u = 128 + i;
    u = u < 0 ? 0 : (u > 255 ? 255 : u);
    return (uint8_t)u;
}

}  // namespace pbe
