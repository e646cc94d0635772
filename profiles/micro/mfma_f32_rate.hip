// What the embed half's two pipes sustain on gfx950, measured (the f32-input MFMA rate, and what shares it):
//   hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate && ./mfma_f32_rate
//  (a) bare v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 loops on random operands held in registers, 1 / 2 / 4 waves per SIMD
//      -> TFLOP/s of the whole chip (the 157.3 TFLOP/s denominator of the embed roofline);
//  (b) the same loop with k independent v_fma_f32 between consecutive MFMAs of ONE wave -> how much vector work hides under
//      the matrix pipe inside a wave;
//  (c) transcendental issue cost: v_exp_f32, v_rcp_f32, and the 5-instruction SiLU of pb_embed_kernels.h (mul, exp, add,
//      rcp, mul) -- ns per wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD;
//  (d) two KINDS of waves on the same SIMDs (waves 0-3 of an 8-wave workgroup one instruction stream, waves 4-7 another): MFMA beside fma,
//      MFMA beside exp, exp beside fma -> do the pipes run concurrently ACROSS waves (time = max) or share issue (time = sum)?
// All streams are inline asm (the compiler neither fuses nor reorders them).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITERS = 4096;

enum Stream { S_MFMA16, S_MFMA32, S_FMA, S_EXP, S_RCP, S_SILU, S_MFMA16_FMA };

// one loop iteration = UNIT instructions of the stream's kind
template <int ST, int K>
__device__ __forceinline__ void body(f32x4 (&acc)[4], f32x16 (&big)[2], float (&v)[8], float a, float b) {
    if constexpr (ST == S_MFMA16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
    } else if constexpr (ST == S_MFMA32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(big[i & 1]) : "v"(a), "v"(b));
    } else if constexpr (ST == S_FMA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
    } else if constexpr (ST == S_EXP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (ST == S_RCP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (ST == S_SILU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t;
            asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(t) : "v"(v[i]));
            asm volatile("v_exp_f32 %0, %0" : "+v"(t));
            asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(t));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(t));
        }
    } else if constexpr (ST == S_MFMA16_FMA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + k) & 7]) : "v"(a), "v"(b));
        }
    }
}

// B = -1: every wave runs A.  Otherwise the workgroup has 8 waves: waves 0-3 run stream A, waves 4-7 stream B (a 512-thread
// workgroup puts waves w and w + 4 on the same SIMD, so every SIMD holds one wave of each kind)
template <int A, int B, int K>
__global__ __launch_bounds__(512) void k(float *out, float a, float b, int iters_a, int iters_b) {
    f32x4 acc[4];
    f32x16 big[2];
    float v[8];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{a * i, b, a, b * i};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 16; ++j) big[i][j] = a * j + i;
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * (float)((threadIdx.x + i) & 31);
    const float ra = a * (1.0f + (threadIdx.x & 63) * 0.013f), rb = b * (1.0f - (threadIdx.x & 31) * 0.017f);  // "random" operands
    const bool second = B >= 0 && threadIdx.x >= 256;
    if (!second) {
        for (int it = 0; it < iters_a; ++it) body<A, K>(acc, big, v, ra, rb);
    } else {
        if constexpr (B >= 0)
            for (int it = 0; it < iters_b; ++it) body<B, K>(acc, big, v, ra, rb);
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) r += acc[i].x + acc[i].w;
    for (int i = 0; i < 2; ++i) r += big[i][0] + big[i][15];
    for (int i = 0; i < 8; ++i) r += v[i];
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

int main() {
    float *d;
    (void)hipMalloc(&d, 8192 * 256 * sizeof(float));
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs; ITERS %d per wave; clocks quoted at a nominal 2.4 GHz\n", p.gcnArchName, cus, ITERS);
    printf("## (a) bare f32-input MFMA, operands in registers\n");
    for (int w : {1, 2, 4}) {
        const float ms16 = run(k<S_MFMA16, -1, 0>, cus * w, d, ITERS, 0);
        const float ms32 = run(k<S_MFMA32, -1, 0>, cus * w, d, ITERS, 0);
        const double f16 = (double)cus * w * 4 * ITERS * 8 * 2048.0 / (ms16 * 1e-3) / 1e12;
        const double f32 = (double)cus * w * 4 * ITERS * 4 * 4096.0 / (ms32 * 1e-3) / 1e12;
        printf("%d wave(s)/SIMD: 16x16x4 %.1f TFLOP/s (%.2f clk per MFMA per SIMD)   32x32x2 %.1f TFLOP/s (%.2f clk)\n", w, f16,
               ms16 * 1e6 / ((double)ITERS * 8 * w) * 2.4, f32, ms32 * 1e6 / ((double)ITERS * 4 * w) * 2.4);
    }
    printf("## (b) one wave per SIMD: k v_fma_f32 between consecutive 16x16x4 MFMAs (clk per MFMA slot; bare = 32)\n");
    {
        const float m0 = run(k<S_MFMA16_FMA, -1, 0>, cus, d, ITERS, 0), m2 = run(k<S_MFMA16_FMA, -1, 2>, cus, d, ITERS, 0),
                    m4 = run(k<S_MFMA16_FMA, -1, 4>, cus, d, ITERS, 0), m6 = run(k<S_MFMA16_FMA, -1, 6>, cus, d, ITERS, 0),
                    m8 = run(k<S_MFMA16_FMA, -1, 8>, cus, d, ITERS, 0);
        auto clk = [&](float ms) { return ms * 1e6 / ((double)ITERS * 8) * 2.4; };
        printf("k=0 %.1f  k=2 %.1f  k=4 %.1f  k=6 %.1f  k=8 %.1f\n", clk(m0), clk(m2), clk(m4), clk(m6), clk(m8));
        const float n2 = run(k<S_MFMA16_FMA, -1, 2>, cus * 2, d, ITERS, 0), n4 = run(k<S_MFMA16_FMA, -1, 4>, cus * 2, d, ITERS, 0),
                    n8 = run(k<S_MFMA16_FMA, -1, 8>, cus * 2, d, ITERS, 0);
        printf("two waves per SIMD (clk per MFMA slot per SIMD; bare = 32): k=2 %.1f  k=4 %.1f  k=8 %.1f\n", clk(n2) / 2, clk(n4) / 2, clk(n8) / 2);
    }
    printf("## (c) vector / transcendental issue, ns per wave-instruction per SIMD (clk)\n");
    auto row = [&](const char *name, auto kern, int per_iter) {
        printf("%-34s", name);
        for (int w : {1, 2, 4}) {
            const float ms = run(kern, cus * w, d, ITERS, 0);
            const double ns = ms * 1e6 / ((double)ITERS * per_iter * w);
            printf("  %dw %.3f ns (%.2f clk)", w, ns, ns * 2.4);
        }
        printf("\n");
    };
    row("v_fma_f32", k<S_FMA, -1, 0>, 8);
    row("v_exp_f32", k<S_EXP, -1, 0>, 8);
    row("v_rcp_f32", k<S_RCP, -1, 0>, 8);
    row("SiLU (5 instr), per SiLU", k<S_SILU, -1, 0>, 8);
    printf("## (d) two kinds of waves on the same SIMDs: 8-wave workgroups, waves 0-3 stream A, waves 4-7 stream B, 1 workgroup per CU\n");
    auto pair = [&](const char *name, auto ka, auto kb, auto kab, int ia, int ib) {
        const float ta = run(ka, cus, d, ia, 0), tb = run(kb, cus, d, ib, 0), tab = run(kab, cus, d, ia, ib, 512);
        printf("%-22s A alone (1 wave/SIMD) %.3f ms  B alone %.3f ms  A beside B %.3f ms   (sum %.3f, max %.3f)\n", name, ta, tb, tab, ta + tb,
               ta > tb ? ta : tb);
    };
    pair("A = MFMA, B = fma", k<S_MFMA16, -1, 0>, k<S_FMA, -1, 0>, k<S_MFMA16, S_FMA, 0>, ITERS, ITERS * 8);
    pair("A = MFMA, B = exp", k<S_MFMA16, -1, 0>, k<S_EXP, -1, 0>, k<S_MFMA16, S_EXP, 0>, ITERS, ITERS * 2);
    pair("A = MFMA, B = SiLU", k<S_MFMA16, -1, 0>, k<S_SILU, -1, 0>, k<S_MFMA16, S_SILU, 0>, ITERS, ITERS);
    pair("A = exp, B = fma", k<S_EXP, -1, 0>, k<S_FMA, -1, 0>, k<S_EXP, S_FMA, 0>, ITERS * 2, ITERS * 8);
    return 0;
}
