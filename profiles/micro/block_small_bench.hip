// Timing of k_block_small (pixelbox_amd/csrc/pb_block_small.h) on random data, whole and with phases removed (ABL).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I pixelbox_amd/csrc -I include profiles/micro/block_small_bench.hip -o /tmp/block_small_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include "pb_common.h"
#include "pb_embed_kernels.h"
#include "pb_gemm_p3.h"
#include "pb_block_small.h"
using namespace pbe;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static float *dev_rand(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
    float *d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}
static void *dev_rand_bf16(size_t n, float scale) {  // n bf16 values (top halves of random floats)
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) {
        const float f = scale * ((float)rand() / RAND_MAX - 0.5f);
        uint32_t u; memcpy(&u, &f, 4);
        h[i] = (uint16_t)(u >> 16);
    }
    void *d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
template <int KS, int COUT, bool RESID, int ABL, bool P3 = false, bool P3E = false>
static void run(const char *name, int n) {
    constexpr int CIN = 192, E = 1152, SP = 48;
    using GEO = BlockGeom<KS, CIN, E, COUT, 4, 2, SP, P3>;
    BlockW w{};
    const int n_wg = (n + 1) / 2;
    CK(hipMalloc(&w.dbg, (size_t)n_wg * 8 * 16 * 8)); CK(hipMemset(w.dbg, 0, (size_t)n_wg * 8 * 16 * 8));
    w.we2 = dev_rand((size_t)(CIN / 16) * (E / 16) * 256, 0.1f); w.be = dev_rand(E, 0.1f);
    w.dwc = dev_rand((size_t)E * GEO::KKP, 0.2f); w.bd = dev_rand(E, 0.1f);
    w.w1 = dev_rand((size_t)SP * E, 0.05f); w.b1 = dev_rand(SP, 0.1f); w.w2t = dev_rand((size_t)SP * E, 0.05f); w.b2 = dev_rand(E, 0.1f);
    w.we3 = dev_rand_bf16((size_t)(CIN / 32) * (E / 16) * 3 * 64 * 8, 0.1f);
    w.wp3 = dev_rand_bf16((size_t)(E / 32) * (COUT / 16) * 3 * 64 * 8, 0.05f);
    w.wp2 = dev_rand((size_t)(E / 16) * (COUT / 16) * 256, 0.05f); w.bp = dev_rand(COUT, 0.1f); w.nt16 = COUT / 16;
    float *x = dev_rand((size_t)n * 16 * CIN, 2.0f), *out;
    CK(hipMalloc(&out, (size_t)n * 16 * COUT * 4));
    auto kern = k_block_small<KS, CIN, E, COUT, 4, 2, SP, RESID, ABL, P3, P3E>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3((n + 1) / 2), dim3(512), GEO::LDS_BYTES, 0, x, w, out, n);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3((n + 1) / 2), dim3(512), GEO::LDS_BYTES, 0, x, w, out, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s n=%4d  %8.1f us\n", name, n, ms * 1000.f / reps);
    if (ABL & 32) {
        std::vector<unsigned long long> h((size_t)n_wg * 8 * 16);
        CK(hipMemcpy(h.data(), w.dbg, h.size() * 8, hipMemcpyDeviceToHost));
        static const char *ph[13] = {"prologue", "group top", "expand mfma", "silu+window", "barrier A", "filter", "barrier B", "means", "fc1",
                                     "units", "fc2+scale", "project", "epilogue"};
        double tot = 0;
        for (int i = 0; i < 13; ++i) {
            double sum = 0, w0 = 0, w7 = 0;
            for (int g = 0; g < n_wg; ++g)
                for (int wv = 0; wv < 8; ++wv) sum += (double)h[((size_t)g * 8 + wv) * 16 + i];
            w0 = (double)h[i]; w7 = (double)h[7 * 16 + i];
            printf("    %-12s %9.0f cycles (wg0 wave0 %8.0f, wave7 %8.0f)\n", ph[i], sum / (n_wg * 8.0), w0, w7);
            tot += sum / (n_wg * 8.0);
        }
        printf("    total        %9.0f cycles\n", tot);
    }
}
int main(int argc, char **argv) {
    if (argc > 1) {  // round 6: the P3 forms (project on the bf16 cores; + expand on them: P3E)
        for (int n : {512}) {
            run<5, 192, true, 0, true, false>("5x5 P3 whole", n);
            run<5, 192, true, 32, true, false>("5x5 P3 stamped", n);
            run<5, 192, true, 0, true, true>("5x5 P3+P3E whole", n);
            run<5, 192, true, 32, true, true>("5x5 P3+P3E stamped", n);
            run<5, 192, true, 1, true, true>("5x5 P3+P3E -expand mfma", n);
            run<5, 192, true, 16, true, true>("5x5 P3+P3E weights from L1", n);
            run<5, 192, true, 2, true, true>("5x5 P3+P3E -taps", n);
            run<5, 192, true, 64, true, true>("5x5 P3+P3E -tap loads", n);
            run<5, 192, true, 64 + 32, true, true>("5x5 P3+P3E -tap loads stamped", n);
            run<5, 192, true, 4, true, true>("5x5 P3+P3E -se", n);
            run<5, 192, true, 8, true, true>("5x5 P3+P3E -project mfma", n);
            run<5, 192, true, 15, true, true>("5x5 P3+P3E -all", n);
            run<3, 320, false, 0, true, true>("3x3/320 P3+P3E whole", n);
            run<3, 320, false, 32, true, true>("3x3/320 P3+P3E stamped", n);
        }
        return 0;
    }
    for (int n : {2, 512}) {
        run<5, 192, true, 0>("5x5 whole", n);
        run<5, 192, true, 32>("5x5 stamped", n);
        run<3, 320, false, 32>("3x3/320 stamped", n);
        run<5, 192, true, 16>("5x5 weights from L1", n);
        run<5, 192, true, 20>("5x5 weights from L1 -se", n);
        run<3, 320, false, 16>("3x3/320 weights from L1", n);
        run<5, 192, true, 1>("5x5 -expand mfma", n);
        run<5, 192, true, 2>("5x5 -taps", n);
        run<5, 192, true, 4>("5x5 -se", n);
        run<5, 192, true, 8>("5x5 -project mfma", n);
        run<5, 192, true, 3>("5x5 -expand -taps", n);
        run<5, 192, true, 15>("5x5 -all", n);
        run<3, 320, false, 0>("3x3/320 whole", n);
        run<3, 320, false, 4>("3x3/320 -se", n);
        run<3, 320, false, 8>("3x3/320 -project mfma", n);
    }
    return 0;
}
