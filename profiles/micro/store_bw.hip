// What the chip takes in plain stores: a kernel that only writes (16 B per lane, whole lines), one that copies, and one that writes
// while re-reading a small L2-resident block.  hipcc --offload-arch=gfx950 -O3 -o store_bw store_bw.hip && ./store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// each workgroup owns `per_wg` consecutive float4s; a wave instruction writes 1 KB
__global__ __launch_bounds__(256) void k_store(f32x4 *out, size_t per_wg, float v) {
    f32x4 *o = out + (size_t)blockIdx.x * per_wg + threadIdx.x;
    for (size_t i = 0; i < per_wg; i += 256) o[i] = (f32x4){v, v, v, v};
}
__global__ __launch_bounds__(256) void k_copy(f32x4 *out, const f32x4 *in, size_t per_wg) {
    f32x4 *o = out + (size_t)blockIdx.x * per_wg + threadIdx.x;
    const f32x4 *s = in + (size_t)blockIdx.x * per_wg + threadIdx.x;
    for (size_t i = 0; i < per_wg; i += 256) o[i] = s[i];
}
// the copy with non-temporal loads, non-temporal stores, or both (read-once / write-once streams beside each other)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy_nt(f32x4 *out, const f32x4 *in, size_t per_wg) {
    f32x4 *o = out + (size_t)blockIdx.x * per_wg + threadIdx.x;
    const f32x4 *s = in + (size_t)blockIdx.x * per_wg + threadIdx.x;
    for (size_t i = 0; i < per_wg; i += 1024) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = NTL ? __builtin_nontemporal_load(s + i + 256 * j) : s[i + 256 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (NTS) __builtin_nontemporal_store(v[j], o + i + 256 * j);
            else o[i + 256 * j] = v[j];
        }
    }
}
// strided pieces: each lane group of 4 writes 64 B of a `pitch`-byte record (the band kernel's NHWC pattern: 16 channels of E)
__global__ __launch_bounds__(256) void k_store_pieces(float *out, size_t recs_per_wg, int pitch_f, int piece) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *o = out + ((size_t)blockIdx.x * recs_per_wg + 16 * wave + (lane >> 2)) * pitch_f + 16 * piece + 4 * (lane & 3);
    for (size_t i = 0; i < recs_per_wg; i += 64) *reinterpret_cast<f32x4 *>(o + i * pitch_f) = (f32x4){1.f, 2.f, 3.f, 4.f};
}
int main() {
    const size_t bytes = (size_t)768 << 20;
    f32x4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : {50, 100, 200, 300, 600}) {
        for (int wgs : {1024, 2048, 8192, 32768}) {
            const size_t n4 = (mb << 20) / 16, per = n4 / wgs / 256 * 256;
            float ms_s = 0, ms_c = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_store, dim3(wgs), dim3(256), 0, 0, a, per, 1.f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_s, e0, e1));
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, a, b, per); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_c, e0, e1));
            }
            const double by = (double)per * wgs * 16;
            printf("%4zu MB, %5d workgroups: store %7.1f us = %5.2f TB/s written;  copy %7.1f us = %5.2f TB/s written (+ as much read)\n", mb, wgs, ms_s * 1e3,
                   by / ms_s / 1e9, ms_c * 1e3, by / ms_c / 1e9);
        }
    }
    for (size_t mb : {200, 300, 600}) {
        const int wgs = 8192;
        const size_t n4 = (mb << 20) / 16, per = n4 / wgs / 1024 * 1024;
        float ms[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 3; ++rep) {
#define RUN(I, NTL, NTS) CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_copy_nt<NTL, NTS>), dim3(wgs), dim3(256), 0, 0, a, b, per); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[I], e0, e1));
            RUN(0, false, false) RUN(1, true, false) RUN(2, false, true) RUN(3, true, true)
#undef RUN
        }
        const double by = (double)per * wgs * 16;
        printf("%4zu MB copied, four loads in flight per lane: plain %6.1f us = %4.2f TB/s each way; nt loads %6.1f us = %4.2f; nt stores %6.1f us = %4.2f; both %6.1f us = %4.2f\n", mb,
               ms[0] * 1e3, by / ms[0] / 1e9, ms[1] * 1e3, by / ms[1] / 1e9, ms[2] * 1e3, by / ms[2] / 1e9, ms[3] * 1e3, by / ms[3] / 1e9);
    }
    // NHWC pieces: 200 MB of 384-byte records, six launches-in-one? no: one launch per 64-byte piece, and all six pieces concurrently
    for (int pitch : {96, 144}) {
        const size_t recs = ((size_t)200 << 20) / (pitch * 4), wgs = 4096, per = recs / wgs / 64 * 64;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int p = 0; p < pitch / 16; ++p) hipLaunchKernelGGL(k_store_pieces, dim3(wgs), dim3(256), 0, 0, (float *)a, per, pitch, p);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("records of %d floats written as %d launches of 64-byte pieces: %7.1f us = %5.2f TB/s\n", pitch, pitch / 16, ms * 1e3, (double)per * wgs * pitch * 4 / ms / 1e9);
    }
    return 0;
}
