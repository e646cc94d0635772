// What a dependent chain of SMALL kernels costs per launch on one stream, and how much of it is instruction fetch:
//   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor && ./launch_floor
// (a) an empty kernel, (b) one workgroup doing three DEPENDENT 16-byte loads per lane from a 4 MB table (k_se's shape of memory
// access), (c) one wave running N unrolled dependent FMAs of straight-line code (N x 8 bytes of instructions), either the SAME
// kernel 48 times or 48 DIFFERENT instantiations in turn (a forward at batch 1 is ~50 different kernels, each run once).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_empty(float *o) { if (o == nullptr) o[0] = 1.f; }
__global__ __launch_bounds__(1024) void k_chain(const f32x4 *t, float *o, unsigned mask) {
    unsigned i = threadIdx.x;
    f32x4 a = t[i & mask];
    f32x4 b = t[(__float_as_uint(a.x) + i * 7u) & mask];
    f32x4 c = t[(__float_as_uint(b.y) + i * 13u) & mask];
    o[threadIdx.x] = c.x + c.w;
}
template <int N, int ID>
__global__ __launch_bounds__(64) void k_code(float *o, float a, float b) {
    float x = threadIdx.x + ID;
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    o[threadIdx.x] = x;
}
template <int N, int... IDS> void launch_all(float *o, hipStream_t s, std::integer_sequence<int, IDS...>) {
    (k_code<N, IDS><<<dim3(1), dim3(64), 0, s>>>(o, 0.999f, 0.001f), ...);
}
template <class F> double per_launch_us(F f, int n, hipStream_t s) {
    f();
    (void)hipStreamSynchronize(s);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        f();
        (void)hipStreamSynchronize(s);
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    return best / n;
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *o; CK(hipMalloc(&o, 1 << 20));
    f32x4 *t; CK(hipMalloc(&t, 4 << 20)); CK(hipMemset(t, 0x11, 4 << 20));
    printf("empty kernel, 48 in a row:                       %6.2f us per launch\n", per_launch_us([&] { for (int i = 0; i < 48; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, o); }, 48, s));
    printf("three dependent loads (1024 threads), 48 in a row: %6.2f us per launch\n", per_launch_us([&] { for (int i = 0; i < 48; ++i) hipLaunchKernelGGL(k_chain, dim3(1), dim3(1024), 0, s, t, o, (4u << 20) / 16 - 1); }, 48, s));
    {   // the same 48 empty / chain kernels as a captured graph, replayed
        for (int kind = 0; kind < 2; ++kind) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 48; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, o);
                else hipLaunchKernelGGL(k_chain, dim3(1), dim3(1024), 0, s, t, o, (4u << 20) / 16 - 1);
            }
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            printf("%s, 48 as ONE replayed graph:            %6.2f us per node\n", kind ? "three dependent loads" : "empty kernel         ",
                   per_launch_us([&] { (void)hipGraphLaunch(ge, s); }, 48, s));
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    }
    printf("straight-line code, one wave:\n");
#define ROW(N)                                                                                                                      \
    printf("  %5d FMAs (%3d KB of code): the same kernel x 48 %6.2f us per launch;  48 different kernels in turn %6.2f us per launch\n", N, N * 8 / 1024, \
           per_launch_us([&] { for (int i = 0; i < 48; ++i) hipLaunchKernelGGL((k_code<N, 0>), dim3(1), dim3(64), 0, s, o, 0.999f, 0.001f); }, 48, s),      \
           per_launch_us([&] { launch_all<N>(o, s, std::make_integer_sequence<int, 48>{}); }, 48, s));
    ROW(64) ROW(512) ROW(2048) ROW(8192)
    return 0;
}
