// Issue cost of the vector instructions the exhaustive pass folds with, on gfx950:
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
// Every kernel repeats ONE instruction (inline asm, so the compiler neither packs nor fuses anything) over N_ACC
// independent registers per lane, or over one register (a dependent chain), with 1, 2 or 4 waves per SIMD; the table
// printed is nanoseconds per instruction per wave and the same in clocks at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048, REP = 4;
enum Op { MUL, ADD_CHAIN, PK_MUL, PK_MUL_BCAST, PK_FMA_SGPR, PK_ADD_CHAIN, CVT_UB, PK_MUL_ADD_PAIR, FMA, EXP, RCP, SILU, CVT_RPI, EXP_MUL };
template <int OP, int N_ACC>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, unsigned w) {
    f32x2 acc[N_ACC];
    for (int i = 0; i < N_ACC; ++i) acc[i] = f32x2{(float)threadIdx.x + i, 1.0f + i};
    f32x2 m = {a, b};
    unsigned ww = w + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r)
#pragma unroll
            for (int i = 0; i < N_ACC; ++i) {
                if constexpr (OP == MUL || OP == ADD_CHAIN) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i].x) : "v"(m.x));
                if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].x) : "v"(m.x), "v"(m.y));
                if constexpr (OP == PK_MUL || OP == PK_ADD_CHAIN) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(m));
                if constexpr (OP == PK_MUL_BCAST) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(acc[i]) : "v"(m));
                if constexpr (OP == PK_FMA_SGPR) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "s"(m));
                if constexpr (OP == CVT_UB) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(acc[i].x) : "v"(ww));
                if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(acc[i].x));
                if constexpr (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i].x));
                if constexpr (OP == EXP_MUL) {  // does a full-rate instruction of the same wave hide under a transcendental?
                    asm volatile("v_exp_f32 %0, %0" : "+v"(acc[i].x));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i].y) : "v"(m.x));
                }
                if constexpr (OP == CVT_RPI) asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(acc[i].y) : "v"(acc[i].x));
                if constexpr (OP == SILU) {  // silu_f of pb_embed_common.h: x * rcp(1 + exp2(x * -log2 e)), five instructions
                    float t, u;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(acc[i].x), "v"(m.y));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(t));
                    asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(u) : "v"(t));
                    asm volatile("v_rcp_f32 %0, %0" : "+v"(u));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i].x) : "v"(u));
                }
                if constexpr (OP == PK_MUL_ADD_PAIR) {
                    f32x2 p;
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(m), "v"(acc[(i + 1) % N_ACC]));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(p));
                }
            }
    }
    float r = 0;
    for (int i = 0; i < N_ACC; ++i) r += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F> float run(F f, int blocks, float *d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f<<<blocks, 256>>>(d, 0.999f, 0.001f, 12345u);
    (void)hipEventRecord(e0);
    f<<<blocks, 256>>>(d, 0.999f, 0.001f, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int OP, int N_ACC> void row(const char *name, float *d, int per_iter_instr) {
    printf("%-46s", name);
    for (int w : {1, 2, 4}) {
        const float ms = run(k<OP, N_ACC>, 256 * w, d);
        const double ns = ms * 1e6 / ((double)ITERS * REP * N_ACC * per_iter_instr * w);
        printf("  %dw/SIMD %.3f ns (%.2f clk)", w, ns, ns * 2.4);
    }
    printf("\n");
}
int main() {
    float *d; (void)hipMalloc(&d, 4096 * 256 * sizeof(float));
    row<MUL, 8>("v_mul_f32, 8 independent", d, 1);
    row<FMA, 8>("v_fma_f32, 8 independent", d, 1);
    row<ADD_CHAIN, 1>("v_mul_f32, dependent chain", d, 1);
    row<PK_MUL, 8>("v_pk_mul_f32, 8 independent", d, 1);
    row<PK_MUL_BCAST, 8>("v_pk_mul_f32 op_sel broadcast, 8 independent", d, 1);
    row<PK_FMA_SGPR, 8>("v_pk_fma_f32 with an SGPR pair, 8 independent", d, 1);
    row<PK_ADD_CHAIN, 1>("v_pk_mul_f32, dependent chain", d, 1);
    row<CVT_UB, 8>("v_cvt_f32_ubyte1, 8 independent", d, 1);
    row<EXP, 8>("v_exp_f32, 8 independent", d, 1);
    row<RCP, 8>("v_rcp_f32, 8 independent", d, 1);
    row<CVT_RPI, 8>("v_cvt_rpi_i32_f32, 8 independent", d, 1);
    row<EXP_MUL, 8>("v_exp_f32 + independent v_mul_f32, per PAIR", d, 1);
    row<SILU, 8>("SiLU (mul, exp, add, rcp, mul), 8 independent, per VALUE", d, 1);
    row<PK_MUL_ADD_PAIR, 4>("v_pk_mul + dependent v_pk_add, 4 chains", d, 2);
    row<PK_MUL_ADD_PAIR, 2>("v_pk_mul + dependent v_pk_add, 2 chains", d, 2);
    return 0;
}
