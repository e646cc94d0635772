// pb_gemm_p3_launch.h -- host interface of the tiled P3 GEMM (kernels: pb_gemm_p3.h; instantiations and dispatch:
// pb_gemm_p3.hip, a translation unit of its own so that it compiles beside pb_embed.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pbe {

struct P3Args {
    const float *act;
    long M;
    int K;
    const void *wt3;  // fragment order: [ceil(K / 32)][tiles16][3 planes][64 lanes] x 16 B
    int tiles16;
    const float *bias;
    int N;
    const float *gate;  // null: no squeeze-excite scale on the operand
    int hw;
    const float *resid;
    int do_silu;
    float *out;
    float scale;      // EPI 1: 1 / (pixels per image)
    uint8_t *out_u8;  // EPI 2
};

// nw = 1: the one-wave form without LDS (DIRECT); else 4 or 8 waves per workgroup sharing the weight fragments through LDS.
// epi: 0 store, 1 head conv + average pool of a 4 x 4 map, 2 Linear + tanh + quantiser.
bool p3_has(int nr, int mr, int nw, int epi, bool gate, bool ktail);
// launches (no error check, no sync); false when the shape is not instantiated
bool p3_launch(int nr, int mr, int nw, int epi, hipStream_t st, const P3Args &a);

}  // namespace pbe
