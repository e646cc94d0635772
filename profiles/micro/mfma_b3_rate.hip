// The "f32 product from bf16 / f16 pieces" arithmetic on gfx950, measured:
//   hipcc --offload-arch=gfx950 -O3 mfma_b3_rate.hip -o mfma_b3_rate && ./mfma_b3_rate
//  (a) bare v_mfma_f32_16x16x32_bf16 loops, operands in registers, 1 / 2 / 4 waves per SIMD -> clocks per MFMA per SIMD;
//  (b) one wave per SIMD: k independent vector instructions (v_fma_f32, and the split's own mix: v_and / v_sub / v_perm) between
//      consecutive bf16 MFMAs -> how much of the piece split hides under the matrix pipe inside a wave;
//  (d) two KINDS of waves on the same SIMDs (waves 0-3 bf16 MFMA, waves 4-7 v_fma_f32) -> does the bf16 matrix pipe run
//      beside vector arithmetic of OTHER waves (the f32-input MFMA does not: profiles/r03_mfma_f32_rate.txt);
//  (e) accuracy: out[n][m] = sum_k w[n][k] a[m][k] for K = 192 / 672 / 1152 / 1280 on random operands, evaluated
//        - as the f32 fmaf chain of the f32 MFMA (what the embed half uses today),
//        - from three bf16 pieces per operand (truncation split and round-to-nearest split), six piece products per K = 32 step,
//        - from two f16 pieces per operand (low piece scaled by 2^11, separate accumulator), three products per step,
//      each against the exact (f64) sum: max and rms of |err| / (sum_k |w a|), i.e. relative to the data's own scale.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 4096;

enum Stream { S_MB, S_FMA, S_MB_FMA, S_MB_SPLIT };

template <int ST, int K>
__device__ __forceinline__ void body(f32x4 (&acc)[4], float (&v)[8], uint32_t (&u)[8], u32x4 a, u32x4 b, float fa, float fb) {
    if constexpr (ST == S_MB) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
    } else if constexpr (ST == S_FMA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(fa), "v"(fb));
    } else if constexpr (ST == S_MB_FMA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + k) & 7]) : "v"(fa), "v"(fb));
        }
    } else if constexpr (ST == S_MB_SPLIT) {
        // the split's instruction mix, K of them per MFMA slot in rotation: and, sub, and, sub, perm
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int sel = (i * K + k) % 5, r = (i + k) & 7;
                if (sel == 0 || sel == 2) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[r]));
                else if (sel == 1 || sel == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[r]) : "v"(fa));
                else asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(a.x));
            }
        }
    }
}

template <int A, int B, int K>
__global__ __launch_bounds__(512) void k(float *out, float fa, float fb, int iters_a, int iters_b) {
    f32x4 acc[4];
    float v[8];
    uint32_t u[8];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{fa * i, fb, fa, fb * i};
    for (int i = 0; i < 8; ++i) { v[i] = 0.5f + 0.01f * (float)((threadIdx.x + i) & 31); u[i] = 0x3f800000u + threadIdx.x * 977u + i; }
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = 0x3f803f80u + ((threadIdx.x * 131u + i * 7u) & 0x007f007fu); b[i] = 0x3c003c00u + ((threadIdx.x * 37u + i) & 0x007f007fu); }
    const bool second = B >= 0 && threadIdx.x >= 256;
    if (!second) {
        for (int it = 0; it < iters_a; ++it) body<A, K>(acc, v, u, a, b, fa, fb);
    } else {
        if constexpr (B >= 0)
            for (int it = 0; it < iters_b; ++it) body<B, K>(acc, v, u, a, b, fa, fb);
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) r += acc[i].x + acc[i].w;
    for (int i = 0; i < 8; ++i) r += v[i] + __uint_as_float(u[i] & 0x3fffffffu);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class F>
float run(F f, int blocks, float *d, int ia, int ib, int threads = 256) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    f<<<blocks, threads>>>(d, 0.999f, 0.001f, ia, ib);
    (void)hipEventRecord(e0);
    f<<<blocks, threads>>>(d, 0.999f, 0.001f, ia, ib);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// ---- (e) accuracy -------------------------------------------------------------------------------------------------------------
// one wave per 16 x 16 output tile; w: [N][K], a: [M][K] row-major f32; out[variant][M][N]
// lane (li = lane & 15, kk = lane >> 4): A operand row n0 + li, B operand column m0 + li, k = 32 s + 8 kk .. + 7
__device__ __forceinline__ uint32_t pk_hi(float x0, float x1) { return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u); }
__device__ __forceinline__ float top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
__device__ __forceinline__ float rne16(float x) {  // round to nearest even bf16, as a float
    const uint32_t u = __float_as_uint(x);
    return __uint_as_float((u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
}
struct Pieces { bf16x8 h, m, l; };
template <bool RNE>
__device__ __forceinline__ Pieces split3(const float (&x)[8]) {
    uint32_t hb[8], mb[8], lb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float h = RNE ? rne16(x[e]) : top16(x[e]);
        const float r1 = x[e] - h;
        const float m = RNE ? rne16(r1) : top16(r1);
        const float r2 = r1 - m;
        const float l = RNE ? rne16(r2) : top16(r2);
        hb[e] = __float_as_uint(h); mb[e] = __float_as_uint(m); lb[e] = __float_as_uint(l);
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ph[j] = __builtin_amdgcn_perm(hb[2 * j + 1], hb[2 * j], 0x07060302u);
        pm[j] = __builtin_amdgcn_perm(mb[2 * j + 1], mb[2 * j], 0x07060302u);
        pl[j] = __builtin_amdgcn_perm(lb[2 * j + 1], lb[2 * j], 0x07060302u);
    }
    Pieces p;
    p.h = __builtin_bit_cast(bf16x8, ph); p.m = __builtin_bit_cast(bf16x8, pm); p.l = __builtin_bit_cast(bf16x8, pl);
    return p;
}

__global__ __launch_bounds__(64) void k_acc(const float *__restrict__ w, const float *__restrict__ a, int M, int N, int K, float *__restrict__ out) {
    const int lane = threadIdx.x, li = lane & 15, kk = lane >> 4;
    const int n0 = blockIdx.y * 16, m0 = blockIdx.x * 16;
    const float *wr = w + (size_t)(n0 + li) * K, *ar = a + (size_t)(m0 + li) * K;
    f32x4 c32 = {0, 0, 0, 0}, ct = {0, 0, 0, 0}, cr = {0, 0, 0, 0}, ch = {0, 0, 0, 0}, cl = {0, 0, 0, 0}, cb = {0, 0, 0, 0};
    // f32 MFMA chain, the embed half's k order: k = 16 s + 4 kk + e
    for (int s = 0; s < K / 16; ++s)
        for (int e = 0; e < 4; ++e) c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[16 * s + 4 * kk + e], ar[16 * s + 4 * kk + e], c32, 0, 0, 0);
    for (int s = 0; s < K / 32; ++s) {
        float wx[8], ax[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { wx[e] = wr[32 * s + 8 * kk + e]; ax[e] = ar[32 * s + 8 * kk + e]; }
        {
            const Pieces pw = split3<false>(wx), pa = split3<false>(ax);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.l, pa.h, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.h, pa.l, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.m, pa.m, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.m, pa.h, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.h, pa.m, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.h, pa.h, ct, 0, 0, 0);
            // single bf16 product (what "bf16 MFMA" alone would give), for scale
            cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.h, pa.h, cb, 0, 0, 0);
        }
        {
            const Pieces pw = split3<true>(wx), pa = split3<true>(ax);
            cr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.l, pa.h, cr, 0, 0, 0);
            cr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.h, pa.l, cr, 0, 0, 0);
            cr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.m, pa.m, cr, 0, 0, 0);
            cr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.m, pa.h, cr, 0, 0, 0);
            cr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.h, pa.m, cr, 0, 0, 0);
            cr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw.h, pa.h, cr, 0, 0, 0);
        }
        {   // two f16 pieces, the low one scaled by 2^11: x = h + l * 2^-11
            f16x8 wh, wl, ah, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wh[e] = (_Float16)wx[e]; wl[e] = (_Float16)((wx[e] - (float)wh[e]) * 2048.0f);
                ah[e] = (_Float16)ax[e]; al[e] = (_Float16)((ax[e] - (float)ah[e]) * 2048.0f);
            }
            cl = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah, cl, 0, 0, 0);
            cl = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al, cl, 0, 0, 0);
            ch = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah, ch, 0, 0, 0);
        }
    }
    const size_t MN = (size_t)M * N;
    for (int j = 0; j < 4; ++j) {
        const size_t o = (size_t)(m0 + li) * N + n0 + 4 * kk + j;
        out[0 * MN + o] = c32[j];
        out[1 * MN + o] = ct[j];
        out[2 * MN + o] = cr[j];
        out[3 * MN + o] = ch[j] + cl[j] * (1.0f / 2048.0f);
        out[4 * MN + o] = cb[j];
    }
}

static uint64_t rng_state = 0x5EED0042ull;
static double urand() {
    rng_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0;
}
static double nrand() { return std::sqrt(-2.0 * std::log(urand() + 1e-300)) * std::cos(6.283185307179586 * urand()); }

int main() {
    float *d;
    (void)hipMalloc(&d, 8192 * 512 * sizeof(float));
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs; ITERS %d per wave; clocks quoted at a nominal 2.4 GHz\n", p.gcnArchName, cus, ITERS);
    printf("## (a) bare v_mfma_f32_16x16x32_bf16, operands in registers (16384 FLOP each)\n");
    for (int w : {1, 2, 4}) {
        const float ms = run(k<S_MB, -1, 0>, cus * w, d, ITERS, 0);
        printf("%d wave(s)/SIMD: %.1f TFLOP/s (%.2f clk per MFMA per SIMD); six of them per 32 k of a 16 x 16 tile = %.1f TFLOP/s of f32-equivalent products\n", w,
               (double)cus * w * 4 * ITERS * 8 * 16384.0 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)ITERS * 8 * w) * 2.4,
               (double)cus * w * 4 * ITERS * 8 * 16384.0 / 6.0 / (ms * 1e-3) / 1e12);
    }
    auto clk = [&](float ms) { return ms * 1e6 / ((double)ITERS * 8) * 2.4; };
    printf("## (b) one wave per SIMD: k vector instructions between consecutive bf16 MFMAs (clk per MFMA slot; bare = 16)\n");
    printf("v_fma_f32:        k=0 %.1f  k=1 %.1f  k=2 %.1f  k=3 %.1f  k=4 %.1f  k=6 %.1f  k=8 %.1f\n", clk(run(k<S_MB_FMA, -1, 0>, cus, d, ITERS, 0)),
           clk(run(k<S_MB_FMA, -1, 1>, cus, d, ITERS, 0)), clk(run(k<S_MB_FMA, -1, 2>, cus, d, ITERS, 0)), clk(run(k<S_MB_FMA, -1, 3>, cus, d, ITERS, 0)),
           clk(run(k<S_MB_FMA, -1, 4>, cus, d, ITERS, 0)), clk(run(k<S_MB_FMA, -1, 6>, cus, d, ITERS, 0)), clk(run(k<S_MB_FMA, -1, 8>, cus, d, ITERS, 0)));
    printf("split mix (and/sub/and/sub/perm): k=1 %.1f  k=2 %.1f  k=3 %.1f  k=4 %.1f  k=5 %.1f  k=8 %.1f\n", clk(run(k<S_MB_SPLIT, -1, 1>, cus, d, ITERS, 0)),
           clk(run(k<S_MB_SPLIT, -1, 2>, cus, d, ITERS, 0)), clk(run(k<S_MB_SPLIT, -1, 3>, cus, d, ITERS, 0)), clk(run(k<S_MB_SPLIT, -1, 4>, cus, d, ITERS, 0)),
           clk(run(k<S_MB_SPLIT, -1, 5>, cus, d, ITERS, 0)), clk(run(k<S_MB_SPLIT, -1, 8>, cus, d, ITERS, 0)));
    printf("two waves per SIMD, v_fma_f32 (clk per MFMA slot per SIMD): k=2 %.1f  k=4 %.1f  k=8 %.1f\n", clk(run(k<S_MB_FMA, -1, 2>, cus * 2, d, ITERS, 0)) / 2,
           clk(run(k<S_MB_FMA, -1, 4>, cus * 2, d, ITERS, 0)) / 2, clk(run(k<S_MB_FMA, -1, 8>, cus * 2, d, ITERS, 0)) / 2);
    printf("## (d) two kinds of waves on the same SIMDs: 8-wave workgroups, waves 0-3 bf16 MFMA, waves 4-7 v_fma_f32, 1 workgroup per CU\n");
    {
        const int ia = ITERS * 2, ib = ITERS * 8;
        const float ta = run(k<S_MB, -1, 0>, cus, d, ia, 0), tb = run(k<S_FMA, -1, 0>, cus, d, ib, 0), tab = run(k<S_MB, S_FMA, 0>, cus, d, ia, ib, 512);
        printf("A = bf16 MFMA, B = fma   A alone (1 wave/SIMD) %.3f ms  B alone %.3f ms  A beside B %.3f ms   (sum %.3f, max %.3f)\n", ta, tb, tab, ta + tb, ta > tb ? ta : tb);
    }
    printf("## (e) accuracy of sum_k w a against the exact sum, relative to sum_k |w a|: max / rms over 64 x 64 outputs\n");
    for (int dist = 0; dist < 2; ++dist)
        for (int K : {192, 672, 1152, 1280}) {
            const int M = 64, N = 64;
            std::vector<float> w((size_t)N * K), a((size_t)M * K);
            for (auto &x : w) x = (float)(nrand() * 0.05);
            // dist 0: activations like SiLU outputs times a gate (mostly small positive, a few large); dist 1: zero-mean normal
            for (auto &x : a) { const double z = nrand() * 2.0; x = dist == 0 ? (float)((z / (1.0 + std::exp(-z))) * urand()) : (float)z; }
            float *dw, *da, *dout;
            (void)hipMalloc(&dw, w.size() * 4); (void)hipMalloc(&da, a.size() * 4); (void)hipMalloc(&dout, 5 * (size_t)M * N * 4);
            (void)hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
            k_acc<<<dim3(M / 16, N / 16), 64>>>(dw, da, M, N, K, dout);
            std::vector<float> o(5 * (size_t)M * N);
            (void)hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
            const char *names[5] = {"f32 MFMA chain", "3 x bf16 trunc, 6 products", "3 x bf16 rne, 6 products", "2 x f16 scaled, 3 products", "1 x bf16 (hi hi only)"};
            printf("K = %4d, %s activations:\n", K, dist == 0 ? "SiLU-like" : "normal");
            for (int v = 0; v < 5; ++v) {
                double mx = 0, sq = 0, mxabs = 0;
                for (int m = 0; m < M; ++m)
                    for (int n = 0; n < N; ++n) {
                        double ex = 0, sc = 0;
                        for (int kq = 0; kq < K; ++kq) { const double pr = (double)w[(size_t)n * K + kq] * (double)a[(size_t)m * K + kq]; ex += pr; sc += std::fabs(pr); }
                        const double err = std::fabs((double)o[(size_t)v * M * N + (size_t)m * N + n] - ex);
                        mx = std::max(mx, err / sc); sq += (err / sc) * (err / sc); mxabs = std::max(mxabs, err);
                    }
                printf("  %-28s max %.3e  rms %.3e  (max abs %.3e)\n", names[v], mx, std::sqrt(sq / (M * N)), mxabs);
            }
            (void)hipFree(dw); (void)hipFree(da); (void)hipFree(dout);
        }
    return 0;
}
