// Sustained issue rate of v_mfma_i32_16x16x64_i8 on gfx950 (hipcc --offload-arch=gfx950 -O3 mfma_i8_rate.hip -o mfma_i8_rate):
// every wave runs 8 independent accumulator chains from registers (no memory), 1, 2 or 4 waves per SIMD on all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 4096, N_ACC = 8;
template <int NV>  // NV independent v_fma_f32 per MFMA, interleaved: do vector instructions issue under the matrix pipe's time?
__global__ __launch_bounds__(256) void k(int *out, int seed) {
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)(seed + i) + threadIdx.x;
    i32x4 a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, b = {seed ^ 0x55, seed + 1, seed + 2, seed + 3};
    i32x4 acc[N_ACC];
    for (int i = 0; i < N_ACC; ++i) acc[i] = (i32x4){i, i, i, i};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < N_ACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[(i * NV + v) & 7]) : "v"(f[(i + v + 3) & 7]));
        }
    }
    int r = 0;
    for (int i = 0; i < N_ACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) r += (int)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    int *d; (void)hipMalloc(&d, 4096 * 256 * sizeof(int));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](int nv, int w) {
        auto launch = [&]() {
            if (nv == 0) k<0><<<256 * w, 256>>>(d, 3);
            else if (nv == 1) k<1><<<256 * w, 256>>>(d, 3);
            else if (nv == 2) k<2><<<256 * w, 256>>>(d, 3);
            else k<4><<<256 * w, 256>>>(d, 3);
        };
        launch();
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    for (int nv : {0, 1, 2, 4})
    for (int w : {1, 2, 4}) {
        const float ms = run(nv, w);
        const double n = (double)ITERS * N_ACC * w;  // MFMAs per SIMD
        const double ops = n * 1024 * 2.0 * 16 * 16 * 64;
        printf("%d v_fma per MFMA, %d waves/SIMD: %.3f ms, %.2f ns per MFMA per SIMD (%.1f clk at 2.4 GHz), %.2f POP/s\n", nv, w, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4,
               ops / (ms * 1e-3) / 1e15);
    }
    return 0;
}
