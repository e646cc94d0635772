// pb_embed_kernels.h -- gfx950 device code for the embed half: EfficientNet-B0 features -> avgpool ->
// Linear(1280, D) -> tanh -> u8 quantiser, the network `MODEL.run` evaluates in the reference
// (src/image_hashes/efficientnet.rs:10-14,34; architecture: resources/train.py:30-46; quantiser:
// efficientnet.rs:39; pre-processing px/255: efficientnet.rs:19-29).
//
// Activations are NHWC f32 ([B*H*W][C] matrices), so every 1x1 convolution (87.5 % of the MACs) is a
// GEMM on the f32-input matrix cores: v_mfma_f32_16x16x4_f32 -- exact f32 products, f32 accumulate
// (bit-for-bit an fmaf chain), the only MFMA dtype that keeps the embedding within 1e-5 of an f32 CPU
// implementation.  Depthwise convolutions, squeeze-excite and the head are HBM/latency-bound VALU kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pbe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

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

// ------------------------------------------------------------------------------------------------
// stem: u8 NHWC [B,H,W,3] -> f32 NHWC [B,H/2,W/2,32]; 3x3 stride 2 pad 1, + bias, SiLU.
// The px/255 conversion of efficientnet.rs:27 is fused here (correctly rounded divide).
// One thread = one output pixel x 8 channels (4 threads per pixel).  w: [27][32] (tap-major: ky,kx,ci).
__global__ __launch_bounds__(256) void k_stem(const uint8_t *__restrict__ img, int B, int H, int W,
                                              const float *__restrict__ w, const float *__restrict__ bias,
                                              float *__restrict__ out) {
    __shared__ float s_w[27 * 32];
    __shared__ float s_b[32];
    for (int i = threadIdx.x; i < 27 * 32; i += blockDim.x) s_w[i] = w[i];
    if (threadIdx.x < 32) s_b[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo * 4;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(t & 3);
        const long p = t >> 2;
        const int x = (int)(p % Wo);
        const int y = (int)((p / Wo) % Ho);
        const int b = (int)(p / ((long)Wo * Ho));
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = s_b[cg * 8 + c];
        const uint8_t *ib = img + (size_t)b * H * W * 3;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y * 2 + ky - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = x * 2 + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const uint8_t *px = ib + ((size_t)iy * W + ix) * 3;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float a = (float)px[ci] / 255.0f;
                    const float *wr = s_w + ((ky * 3 + kx) * 3 + ci) * 32 + cg * 8;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float pr = a * wr[c];
                        acc[c] = acc[c] + pr;
                    }
                }
            }
        }
        float *o = out + p * 32 + cg * 8;
        f32x4 v0 = {silu_f(acc[0]), silu_f(acc[1]), silu_f(acc[2]), silu_f(acc[3])};
        f32x4 v1 = {silu_f(acc[4]), silu_f(acc[5]), silu_f(acc[6]), silu_f(acc[7])};
        *reinterpret_cast<f32x4 *>(o) = v0;
        *reinterpret_cast<f32x4 *>(o + 4) = v1;
    }
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution as GEMM on f32 MFMA:  out[m][n] = epi( sum_k act'[m][k] * wt[k][n] + bias[n] )
//   act'[m][k] = act[m][k] * gate[m / hw][k]   (squeeze-excite scale fused on the operand; gate may be null)
//   epi: optional SiLU, optional residual add.
// Orientation: the MFMA's "A" operand is the weight (row index = output channel n), its "B" operand the
// activation (column index = pixel m), so each lane ends up with 4 consecutive channels of one pixel:
// bias / residual / store are float4 accesses of the NHWC row.
// A lane owns k-slot kk = lane>>4 and loads act[m][16s + 4kk .. +3] as one float4 (16 B, the 4 kk-lanes of
// a pixel cover 64 contiguous bytes); MFMA number e of a k-step uses element e, i.e. k = 16s + 4kk + e,
// and the weight operand is read from LDS at that same k (the k labels only have to agree between the
// two operands).  Weights: wt[Kpad][Npad] k-major, zero padded (Kpad % 16 == 0, Npad % 16 == 0).
// Block = 4 waves; wave w owns MR pixel tiles of 16 rows; all waves share the weight tile in LDS.
constexpr int G_KC = 32;  // K chunk staged in LDS per barrier pair

template <int MR, int NR>
__global__ __launch_bounds__(256) void k_gemm1x1(const float *__restrict__ act, int M, int K,
                                                 const float *__restrict__ wt, int Kpad, int Npad,
                                                 const float *__restrict__ bias, int N,
                                                 const float *__restrict__ gate, int hw,
                                                 const float *__restrict__ resid, int do_silu,
                                                 float *__restrict__ out) {
    constexpr int NT = 16 * NR;
    constexpr int LDW = NT + 4;  // +4: rows k and k+4 land 16 banks apart (conflict-free ds_read_b32)
    __shared__ __attribute__((aligned(16))) float s_w[G_KC * LDW];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15;   // pixel within a tile (activation operand) / channel within a tile (weight operand)
    const int kk = lane >> 4;   // k slot
    const int n0 = blockIdx.y * NT;
    const long m_block = (long)blockIdx.x * (64 * MR);
    long mrow[MR];
    bool mval[MR];
    const float *arow[MR];
    const float *grow[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        mrow[r] = m_block + (long)(wave * MR + r) * 16 + li;
        mval[r] = mrow[r] < M;
        const long mc = mval[r] ? mrow[r] : 0;
        arow[r] = act + mc * K;
        grow[r] = gate ? gate + (mc / hw) * K : nullptr;
    }
    f32x4 acc[MR][NR];
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int c = 0; c < NR; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < Kpad; k0 += G_KC) {
        const int kc = (Kpad - k0) < G_KC ? (Kpad - k0) : G_KC;
        __syncthreads();
        // stage wt[k0 .. k0+kc)[n0 .. n0+NT) -> LDS (float4 along n)
        for (int i = threadIdx.x; i < kc * (NT / 4); i += 256) {
            const int kr = i / (NT / 4), c4 = i % (NT / 4);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(wt + (size_t)(k0 + kr) * Npad + n0 + c4 * 4);
            *reinterpret_cast<f32x4 *>(s_w + kr * LDW + c4 * 4) = v;
        }
        __syncthreads();
        for (int s = 0; s < kc; s += 16) {
            const int kbase = k0 + s + 4 * kk;  // this lane's 4 consecutive k
            f32x4 a[MR];
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                a[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (mval[r] && kbase < K) {
                    a[r] = *reinterpret_cast<const f32x4 *>(arow[r] + kbase);
                    if (grow[r]) {
                        const f32x4 g = *reinterpret_cast<const f32x4 *>(grow[r] + kbase);
                        a[r].x = a[r].x * g.x; a[r].y = a[r].y * g.y; a[r].z = a[r].z * g.z; a[r].w = a[r].w * g.w;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float *wrow = s_w + (s + 4 * kk + e) * LDW + li;
                float wv[NR];
#pragma unroll
                for (int c = 0; c < NR; ++c) wv[c] = wrow[c * 16];
#pragma unroll
                for (int r = 0; r < MR; ++r) {
                    const float av = e == 0 ? a[r].x : (e == 1 ? a[r].y : (e == 2 ? a[r].z : a[r].w));
#pragma unroll
                    for (int c = 0; c < NR; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c], av, acc[r][c], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: lane holds channels n0 + 16c + 4kk .. +3 of pixel mrow[r]
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        if (!mval[r]) continue;
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const int n = n0 + c * 16 + kk * 4;
            if (n >= N) continue;  // N % 4 == 0
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + n);
            f32x4 v = acc[r][c];
            v.x = v.x + b.x; v.y = v.y + b.y; v.z = v.z + b.z; v.w = v.w + b.w;
            if (do_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            if (resid) {
                const f32x4 rv = *reinterpret_cast<const f32x4 *>(resid + mrow[r] * N + n);
                v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w;
            }
            *reinterpret_cast<f32x4 *>(out + mrow[r] * N + n) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// depthwise KSxKS conv, stride S, pad (KS-1)/2, + bias + SiLU, NHWC, with the squeeze-excite pooling
// partial sums fused: part[b][tile][c] = sum of the outputs of this block's pixels (fixed order ->
// deterministic).  Thread = (pixel slot, channel quad); block = 256 threads = PX pixel slots x CQ quads.
// w: [KS*KS][C] tap-major.  grid = (tiles_per_image, B, zsplit); blockDim = cq_per_block * px_slots with
// cq_per_block = (C/4) / zsplit channel quads per block (exact), px_slots = 256 / cq_per_block.
template <int KS, int S>
__global__ __launch_bounds__(256) void k_dwconv(const float *__restrict__ in, int H, int W, int C,
                                                const float *__restrict__ w, const float *__restrict__ bias,
                                                float *__restrict__ out, int Ho, int Wo, int px_per_tile,
                                                float *__restrict__ part, int n_tiles, int cq_per_block) {
    constexpr int PAD = (KS - 1) / 2;
    __shared__ f32x4 s_red[256];
    const int px_slots = blockDim.x / cq_per_block;
    const int cq_l = threadIdx.x % cq_per_block;
    const int slot = threadIdx.x / cq_per_block;
    const int cq = blockIdx.z * cq_per_block + cq_l;
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    const bool cvalid = cq * 4 < C;
    const int c0 = cvalid ? cq * 4 : 0;
    f32x4 wreg[KS * KS];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) wreg[t] = *reinterpret_cast<const f32x4 *>(w + (size_t)t * C + c0);
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + c0);
    const float *ib = in + (size_t)b * H * W * C;
    float *ob = out + (size_t)b * Ho * Wo * C;
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
    const int p_begin = tile * px_per_tile;
    const int p_end = (p_begin + px_per_tile) < Ho * Wo ? (p_begin + px_per_tile) : Ho * Wo;
    for (int p = p_begin + slot; p < p_end; p += px_slots) {
        const int y = p / Wo, x = p % Wo;
        f32x4 acc = bv;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = y * S + ky - PAD;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int ix = x * S + kx - PAD;
                if (ix < 0 || ix >= W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(ib + ((size_t)iy * W + ix) * C + c0);
                const f32x4 wv = wreg[ky * KS + kx];
                const float p0 = v.x * wv.x, p1 = v.y * wv.y, p2 = v.z * wv.z, p3 = v.w * wv.w;
                acc.x = acc.x + p0; acc.y = acc.y + p1; acc.z = acc.z + p2; acc.w = acc.w + p3;
            }
        }
        acc.x = silu_f(acc.x); acc.y = silu_f(acc.y); acc.z = silu_f(acc.z); acc.w = silu_f(acc.w);
        if (cvalid) *reinterpret_cast<f32x4 *>(ob + (size_t)p * C + c0) = acc;
        psum.x = psum.x + acc.x; psum.y = psum.y + acc.y; psum.z = psum.z + acc.z; psum.w = psum.w + acc.w;
    }
    // reduce the pixel slots in fixed order
    s_red[threadIdx.x] = psum;
    __syncthreads();
    if (slot == 0 && cvalid) {
        f32x4 t = s_red[cq_l];
        for (int sl = 1; sl < px_slots; ++sl) {
            const f32x4 o = s_red[sl * cq_per_block + cq_l];
            t.x = t.x + o.x; t.y = t.y + o.y; t.z = t.z + o.z; t.w = t.w + o.w;
        }
        *reinterpret_cast<f32x4 *>(part + ((size_t)b * n_tiles + tile) * C + c0) = t;
    }
}

// ------------------------------------------------------------------------------------------------
// squeeze-excite gates for one image per block: mean over pixels (sum of the tile partials in order),
// FC(E->S)+SiLU, FC(S->E)+sigmoid.  w1: [S][E]; w2t: [S][E] (transposed se_expand); gate: [B][E].
__global__ __launch_bounds__(256) void k_se(const float *__restrict__ part, int n_tiles, int E, int S, float inv_hw,
                                            const float *__restrict__ w1, const float *__restrict__ b1,
                                            const float *__restrict__ w2t, const float *__restrict__ b2,
                                            float *__restrict__ gate) {
    __shared__ float s_mean[1152];
    __shared__ float s_s[64];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < E; c += 256) {
        float t = 0.0f;
        for (int tl = 0; tl < n_tiles; ++tl) t = t + part[((size_t)b * n_tiles + tl) * E + c];
        s_mean[c] = t * inv_hw;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < S; j += 4) {
        float t = 0.0f;
        for (int c = lane; c < E; c += 64) t = t + s_mean[c] * w1[(size_t)j * E + c];
        for (int off = 32; off >= 1; off >>= 1) t = t + __shfl_xor(t, off);
        if (lane == 0) s_s[j] = silu_f(t + b1[j]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256) {
        float t = b2[c];
        for (int j = 0; j < S; ++j) t = t + s_s[j] * w2t[(size_t)j * E + c];
        gate[(size_t)b * E + c] = sigmoid_f(t);
    }
}

// ------------------------------------------------------------------------------------------------
// head tail: global average pool over hw pixels of feat [B][hw][C] -> pooled [B][C]
__global__ void k_avgpool(const float *__restrict__ feat, int hw, int C, float inv_hw, float *__restrict__ pooled) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float t = 0.0f;
    for (int p = 0; p < hw; ++p) t = t + feat[((size_t)b * hw + p) * C + c];
    pooled[(size_t)b * C + c] = t * inv_hw;
}

// Linear(1280, D) + tanh + u8 quantiser.  wt: [Cin][D] (transposed).  block per image, thread per output.
__global__ __launch_bounds__(256) void k_fc_tanh_quant(const float *__restrict__ pooled, int Cin, int D,
                                                       const float *__restrict__ wt, const float *__restrict__ bias,
                                                       float *__restrict__ out_f32, uint8_t *__restrict__ out_u8) {
    __shared__ float s_x[1280];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < Cin; c += blockDim.x) s_x[c] = pooled[(size_t)b * Cin + c];
    __syncthreads();
    for (int dd = threadIdx.x; dd < D; dd += blockDim.x) {
        float t = bias[dd];
        for (int c = 0; c < Cin; ++c) {
            const float p = s_x[c] * wt[(size_t)c * D + dd];
            t = t + p;
        }
        const float y = tanhf(t);
        if (out_f32) out_f32[(size_t)b * D + dd] = y;
        out_u8[(size_t)b * D + dd] = quantize_u8(y);
    }
}

}  // namespace pbe
