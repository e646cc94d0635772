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
#include "pb_embed_common.h"
#include "pb_p3_common.h"

namespace pbe {

typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// The squeeze-excite gate computed in the TAIL of the kernel that produced the pooled sums, by the workgroup that finishes an
// image last -- for the blocks whose excite weights are small (squeeze width <= 16: the stem-fused first block and the five
// LDS-ring fronts, 2 * SP * E * 4 B <= 31 KB per image; the late blocks' 123-442 KB per image would make the tail a long
// serial piece of one workgroup, and stay with k_se, which shares the loads between images) (-6 dependent launches per forward; the reference evaluates the same layers inside one `MODEL.run`,
// src/image_hashes/efficientnet.rs:34).  Same arithmetic, in the same order, as k_se above -- the order is defined on channel
// quads, not on threads: FC1 sums a unit's products per quad (x, y, z, w in sequence), then over the 64 quads of a block by
// the xor butterfly of a wave, then over the blocks in order; FC2 sums the units in groups of 16 in sequence and the groups
// in order behind the bias -- so a forward gives the same bits whether a layer's gate comes from k_se or from here.
//
// Hand-off (guide: inter-workgroup communication, the counter form): every workgroup stores its pooled sums WRITE-THROUGH
// (8-byte agent-scope atomic stores: se_part_store), each storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup
// meets at a barrier, ONE lane adds 1 to the image's counter (agent-scope atomic); the workgroup whose add returns
// total - 1 is the last: one lane runs an agent-scope acquire, the workgroup meets again, and every load of another
// workgroup's sums is an 8-byte agent-scope atomic load (L1 bypassed).  No placement or dispatch order is assumed.  The last
// workgroup zeroes the counter for the next launch (the host also clears the counters at the start of every forward).
struct SeTail {
    const float *w1, *b1, *w2t, *b2;  // as k_se: [SP][E], [SP], [SP][E], [E]
    float *gate;                      // [B][E]
    unsigned *cnt;                    // [B] arrival counters, zero between launches
    float inv_hw;
    int sp;                           // 0: no tail (k_se runs as a kernel of its own); else 8 / 16 / 32 / 48
};

__device__ __forceinline__ void se_part_store(long long *p, const ll4 &v) {
    __hip_atomic_store(p + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ ll4 se_part_load(const long long *p) {
    ll4 v;
    v.x = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

// Arrival of this workgroup for one counter; true in the workgroup that arrives last (uniform over the workgroup).  Every
// wave that stored sums must call it (all threads of the workgroup do).  `s_flag`: one LDS word nobody else uses meanwhile.
__device__ __forceinline__ bool se_arrive(unsigned *cnt, unsigned total, unsigned *s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == total - 1u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
        *s_flag = last ? 1u : 0u;
    }
    __syncthreads();
    return *s_flag != 0u;
}

// Gate of ONE image by the calling workgroup (NT threads, a multiple of 64): part = the image's [n_tiles][E] pooled sums,
// gate = its [E] output, scr = (E / 256 + 1) * SP + SP floats of LDS scratch (16-byte aligned).
template <int SP, int NT>
__device__ __forceinline__ void se_gate_image(const long long *__restrict__ part, int n_tiles, int E, const SeTail &se,
                                              float *__restrict__ gate, float *scr) {
    constexpr int JG = SP < 16 ? SP : 16, G = SP / JG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_quads = E >> 2, n_blocks = (n_quads + 63) >> 6;
    float *s_p = scr;                  // [n_blocks][SP]
    float *s_s = scr + n_blocks * SP;  // [SP]
    const double sc = (1.0 / 16777216.0) * (double)se.inv_hw;
    // ---- FC1: block of 64 quads per wave turn
    for (int blk = wave; blk < n_blocks; blk += NT / 64) {
        const int cq = blk * 64 + lane;
        const bool on = cq < n_quads;
        const int c = on ? 4 * cq : 0;
        ll4 t = {0, 0, 0, 0};
        for (int tl = 0; tl < n_tiles; ++tl) se_add(t, se_part_load(part + (size_t)tl * E + c));
        f32x4 m = {(float)((double)t.x * sc), (float)((double)t.y * sc), (float)((double)t.z * sc), (float)((double)t.w * sc)};
        if (!on) m = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int jj = 0; jj < SP; ++jj) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(se.w1 + (size_t)jj * E + c);
            float a = m.x * wv.x;
            a = a + m.y * wv.y; a = a + m.z * wv.z; a = a + m.w * wv.w;
            for (int off = 32; off >= 1; off >>= 1) a = a + __shfl_xor(a, off);
            if (lane == 0) s_p[blk * SP + jj] = a;
        }
    }
    __syncthreads();
    if (tid < SP) {
        float v = s_p[tid];
        for (int blk = 1; blk < n_blocks; ++blk) v = v + s_p[blk * SP + tid];
        s_s[tid] = silu_f(v + se.b1[tid]);
    }
    __syncthreads();
    // ---- FC2 + sigmoid, a channel quad per thread turn
    for (int cq = tid; cq < n_quads; cq += NT) {
        const int c = 4 * cq;
        f32x4 v = *reinterpret_cast<const f32x4 *>(se.b2 + c);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int j = 0; j < JG; ++j) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(se.w2t + (size_t)(g * JG + j) * E + c);
                const float sj = s_s[g * JG + j];
                acc.x = acc.x + sj * wv.x; acc.y = acc.y + sj * wv.y; acc.z = acc.z + sj * wv.z; acc.w = acc.w + sj * wv.w;
            }
            v.x = v.x + acc.x; v.y = v.y + acc.y; v.z = v.z + acc.z; v.w = v.w + acc.w;
        }
        const f32x4 r = {sigmoid_f(v.x), sigmoid_f(v.y), sigmoid_f(v.z), sigmoid_f(v.w)};
        *reinterpret_cast<f32x4 *>(gate + c) = r;
    }
    __syncthreads();  // scr may be reused by the caller (next image)
}

template <int NT>
__device__ __forceinline__ void se_gate_image_sp(const long long *part, int n_tiles, int E, const SeTail &se, float *gate, float *scr) {
    if (se.sp == 8) se_gate_image<8, NT>(part, n_tiles, E, se, gate, scr);
    else if (se.sp == 16) se_gate_image<16, NT>(part, n_tiles, E, se, gate, scr);
    else if (se.sp == 32) se_gate_image<32, NT>(part, n_tiles, E, se, gate, scr);
    else se_gate_image<48, NT>(part, n_tiles, E, se, gate, scr);
}

// ------------------------------------------------------------------------------------------------
// stem: u8 NHWC [B,H,W,3] -> f32 NHWC [B,H/2,W/2,32]; 3x3 stride 2 pad 1, + bias, SiLU (efficientnet.rs:19-29 feeds it
// px / 255).  Round 5: the stem on the bf16 matrix cores, from the BYTES.
//
// THE ARITHMETIC OF THE STEM (one definition: stem_tile below; k_stem and k_stem_dw both call it, so both give the same bits):
//   out[p][co] = silu( S + bias[co] ),   S = the value of an f32 accumulator that starts at 0 and receives THREE
//   v_mfma_f32_16x16x32_bf16 in this order:  (w'_lo, px)  (w'_mid, px)  (w'_hi, px)
//   where px is the pixel byte itself (0..255: eight significant bits, EXACT as a bf16), w' = fl(w / 255.0f) on the host,
//   split exactly into three bf16 pieces (hi + mid + lo, truncation split as in pb_gemm_p3.h), all 27 taps in the ONE k-step
//   of 32 (five zero slots), every piece product exact in f32 (8 x 8 significant bits), the matrix pipe accumulating in f32.
// Against the reference's fl(px / 255) * w summed in f32 (the oracle) this differs by rounding only, and less: the reference
// rounds px / 255 AND every product, here nothing is rounded before the accumulator (one rounding on w / 255).  Rounds 1-4
// gathered fl(px / 255) through a 256-entry LDS table into an f32 MFMA chain of 8 steps: 39 % of the kernel's LDS cycles were
// bank conflicts on that table (profiles/r04_embed_layers.txt), 16 f32 MFMAs (512 clocks) per 16-pixel tile where this is 6
// bf16 ones (~110).
//
// Operand maps (v_mfma_f32_16x16x32_bf16: lane (li, kq) holds k = 8 kq + j, j = 0..7, of A row li / B column li):
//   A = weights (row = output channel 16 c + li), B = pixels (column = output pixel x = 16 tx + li), so a lane ends with 4
//   consecutive channels 16 c + 4 kq .. + 3 of its pixel (float4 epilogue, as before).
//   k slot -> tap:  kq = 0, 1, 2: input row ky = kq, bytes j = 0..7 of the pixel's 9-byte window (kx = j / 3, ci = j % 3);
//                   kq = 3: j = 0, 1, 2 -> byte 8 (kx = 2, ci = 2) of rows ky = 0, 1, 2; j = 3..7 empty.
// Input rows are staged in LDS as bytes: per row 4 pad bytes (bytes 1..3 = pixel -1 = 0) then the W * 3 row bytes, RSB =
// W * 3 + 4 per row; the window of output pixel x starts at staged byte 6 x + 1 (never dword-aligned): a lane reads the three
// aligned dwords around its 8 bytes and shifts them out (v_alignbyte); the kq = 3 lanes read one dword of each of the three
// rows and pick one byte of each (two v_perm with a per-lane selector).  18 vector instructions + 3 LDS reads per fragment.
struct StemFrag {
    u32x4s w[2][3];    // [channel tile][plane hi / mid / lo] weight fragments (stem_w3, 16 B per lane each)
    f32x4 bv[2];       // bias of the lane's 4 channels per channel tile
    int a0, a1, a2;    // byte offsets of the lane's three dwords from (staged row 0 of the window rows) + 96 tx
    uint32_t sh;       // byte shift of the window inside its first dword: (6 li + 1) & 3
    uint32_t sel_a, sel_b, mask_hi;
};
__device__ __forceinline__ void stem_frag_init(StemFrag &f, const u32x4s *__restrict__ w3, const float *__restrict__ bias, int lane, int RSB) {
    const int li = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) f.w[c][p] = w3[(c * 3 + p) * 64 + lane];
#pragma unroll
    for (int c = 0; c < 2; ++c) f.bv[c] = *reinterpret_cast<const f32x4 *>(bias + 16 * c + 4 * kq);
    const int ws = 6 * li + 1, wb = ws & ~3;
    f.sh = (uint32_t)(ws & 3);
    if (kq < 3) {
        f.a0 = kq * RSB + wb; f.a1 = f.a0 + 4; f.a2 = f.a0 + 8;
        f.sel_a = 0x03020100u; f.sel_b = 0x03020100u; f.mask_hi = 0xFFFFFFFFu;
    } else {
        f.a0 = wb + 8; f.a1 = RSB + wb + 8; f.a2 = 2 * RSB + wb + 8;
        f.sel_a = 0x0C0C0400u;  // [lo.b0, hi.b0, 0, 0]
        f.sel_b = 0x0C040100u;  // [.b0, .b1, t2.b0, 0]
        f.mask_hi = 0u;
    }
}
// rows: staged byte 0 of the tile's first window row (input row 2 y - 1), advanced by 96 tx (16 pixels x 6 bytes); 4-byte aligned
__device__ __forceinline__ void stem_tile(const StemFrag &f, const uint8_t *rows, f32x4 out[2]) {
    const uint32_t d0 = *reinterpret_cast<const uint32_t *>(rows + f.a0);
    const uint32_t d1 = *reinterpret_cast<const uint32_t *>(rows + f.a1);
    const uint32_t d2 = *reinterpret_cast<const uint32_t *>(rows + f.a2);
    uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, f.sh);
    uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, f.sh);
    const uint32_t t2 = d2 >> (8u * f.sh);
    lo = __builtin_amdgcn_perm(t2, __builtin_amdgcn_perm(hi, lo, f.sel_a), f.sel_b);
    hi &= f.mask_hi;
    float b[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b[j] = (float)((lo >> (8 * j)) & 0xFFu);      // v_cvt_f32_ubyteN
        b[4 + j] = (float)((hi >> (8 * j)) & 0xFFu);
    }
    u32x4s px;
#pragma unroll
    for (int j = 0; j < 4; ++j) px[j] = __builtin_amdgcn_perm(__float_as_uint(b[2 * j + 1]), __float_as_uint(b[2 * j]), 0x07060302u);  // top halves: exact bf16
    typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, f.w[c][2]), __builtin_bit_cast(bf16x8s, px), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, f.w[c][1]), __builtin_bit_cast(bf16x8s, px), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, f.w[c][0]), __builtin_bit_cast(bf16x8s, px), acc, 0, 0, 0);
#if defined(PB_STEM_ABL) && (PB_STEM_ABL & 2)
        out[c] = (f32x4){acc.x + f.bv[c].x, acc.y + f.bv[c].y, acc.z + f.bv[c].z, acc.w + f.bv[c].w};  // ablation: no SiLU behind the stem
#else
        out[c] = (f32x4){silu_f(acc.x + f.bv[c].x), silu_f(acc.y + f.bv[c].y), silu_f(acc.z + f.bv[c].z), silu_f(acc.w + f.bv[c].w)};
#endif
    }
}

// k_stem: the stem alone (inputs wider than the fused kernel's LDS ring allows, or PB_NO_STEM_FUSION).  A block owns one
// output row (b, y): its three input rows are staged as bytes (coalesced dwords), a wave takes 16 consecutive output pixels
// per trip (stem_tile).  Input sizes are multiples of 32 (checked at load): rows are whole dwords and whole MFMA tiles.
// dynamic LDS = 3 * (W * 3 + 4) bytes.
__global__ __launch_bounds__(256) void k_stem(const uint8_t *__restrict__ img, int B, int H, int W,
                                              const u32x4s *__restrict__ w3, const float *__restrict__ bias,
                                              float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_rows[];  // [3][RSB]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int Ho = H / 2, Wo = W / 2;
    const int RSB = W * 3 + 4;
    StemFrag f;
    stem_frag_init(f, w3, bias, lane, RSB);
    const int row_dwords = W * 3 / 4;
    for (int ry = blockIdx.x; ry < B * Ho; ry += gridDim.x) {
        const int b = ry / Ho, y = ry - b * Ho;
        // stage input rows 2y - 1, 2y, 2y + 1 (rows above the image: zero bytes)
        for (int i = threadIdx.x; i < 3 * (row_dwords + 1); i += 256) {
            const int r = i / (row_dwords + 1), dq = i - r * (row_dwords + 1);  // dq = 0: the pad dword
            const int iy = 2 * y - 1 + r;
            uint32_t u = 0;
            if (dq > 0 && iy >= 0) u = *reinterpret_cast<const uint32_t *>(img + ((size_t)b * H + iy) * W * 3 + 4 * (dq - 1));
            *reinterpret_cast<uint32_t *>(s_rows + (size_t)r * RSB + 4 * dq) = u;
        }
        __syncthreads();
        for (int tx = wave; tx < Wo / 16; tx += 4) {
            f32x4 r[2];
            stem_tile(f, s_rows + 96 * tx, r);
            float *o = out + ((size_t)ry * Wo + tx * 16 + li) * 32;
#pragma unroll
            for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4 *>(o + 16 * c + 4 * kq) = r[c];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// stem + the first block's depthwise 3x3 (stride 1) in one kernel: the stem's output (64 x 64 x 32 f32 per image,
// 268 MB per 512-image batch, written once and read once by nothing but that depthwise conv -- the first block has
// no expansion and no residual) never leaves the CU.  A block owns a band of output rows.  It first loads ALL the
// input rows the band needs (2 rows per stem row + 1, as raw bytes, coalesced dwords, one memory round trip for the
// whole band) into LDS; then, per stem row, stem_tile on those bytes, the activated row written into a 3-row LDS ring
// ([pixel + 1][36 floats]: 4 floats of padding per pixel keep the MFMA-layout float4 writes conflict-free; one zero pixel
// on each side and zero rows outside the image are the depthwise zero padding), and once three rows are in the ring the
// depthwise filter of the middle row from LDS (a thread = one pixel x channel quad -- the quad is the thread's for the
// whole kernel, so its 9 taps and bias live in registers; bias-first (ky, kx) accumulation, one fused multiply-add per
// tap as in every depthwise form), + SiLU, store, SE partial sums.  Stem rows at band edges are recomputed by the
// neighbouring band.  Identical arithmetic to k_stem + k_dwconv: same bits.
// grid = (n_bands, B); dynamic LDS = (2 rows_per_band + 5) * RSB bytes + 3 * (W / 2 + 2) * 36 floats.
// RPP: stem rows per phase (1 or 2).  A phase = the tiles of RPP stem rows (Wo / 16 tiles each, dealt over the 4 waves), a barrier,
// the depthwise filter of the RPP output rows whose three stem rows are now in the ring (RPP + 2 slots), a barrier.  With one row
// per phase a wave had ONE tile (3 LDS reads -> 18 vector instructions -> 2 x 3 dependent MFMAs -> 8 SiLU) and two filter items
// between two barriers and three waves per SIMD to hide it under: RPP = 2 halves the barriers and gives every wave two independent
// tiles / four items per phase (measured: profiles/r05_stem.txt).
template <int RPP>
__global__ __launch_bounds__(256) void k_stem_dw(const uint8_t *__restrict__ img, int B, int H, int W,
                                                 const u32x4s *__restrict__ w3, const float *__restrict__ bias,
                                                 const float *__restrict__ dw_w, const float *__restrict__ dw_b,
                                                 float *__restrict__ out, long long *__restrict__ part, int n_bands,
                                                 int rows_per_band, SeTail se) {
    constexpr int RING = RPP + 2;
    extern __shared__ __attribute__((aligned(16))) float s_sd[];
    const int Ho = H / 2, Wo = W / 2;
    const int RSB = W * 3 + 4;     // bytes per staged input row: 4 pad bytes (bytes 1..3 = pixel -1 = 0), then the row
    const int RP = (Wo + 2) * 36;  // floats per ring row
    float *s_ring = s_sd;
    uint8_t *s_in = reinterpret_cast<uint8_t *>(s_sd + RING * RP);
    const int tid = threadIdx.x;
    for (int i = tid; i < RING * 2 * 36; i += 256) {  // the ring's border pixels (columns -1 and Wo) stay zero
        const int r = i / 72, side = (i % 72) / 36, fl = i % 36;
        s_ring[r * RP + (side ? (Wo + 1) * 36 : 0) + fl] = 0.0f;
    }
    const int b = blockIdx.y, band = blockIdx.x;
    const int y0 = band * rows_per_band;
    const int y1 = (y0 + rows_per_band) < Ho ? (y0 + rows_per_band) : Ho;
    // input rows 2 (y0 - 1) - 1 .. 2 y1 + 1 -> staged rows 0 .. ; rows outside the image are zero bytes
    const int iy0 = 2 * (y0 - 1) - 1;
    const int n_in = 2 * (y1 - y0 + 2) + 1;
    {
        // The band's input rows are ONE contiguous block of the image (rows iy0 .. iy0 + n_in - 1, W * 3 bytes each; W * 3 is a multiple
        // of 96, so rows start 16-byte aligned): 16-byte pieces, ALL of a thread's loads requested before its first LDS store.  (Rounds
        // 1-4 copied a dword per loop turn with a runtime division in the index, and hipcc waited for each turn's load before the next
        // turn's: 14 dependent round trips of ~1.5 us per workgroup -- with everything else of the kernel removed it still took 57 of
        // its 108 us, profiles/r05_stem.txt.)  A piece lands in its staged row behind the 4 pad bytes (the pad dword itself is zeroed
        // by the thread that stores the row's first piece); rows outside the image are zero.
        typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
        const int row_pieces = W * 3 / 16;            // 16-byte pieces per row
        const int n_pieces = n_in * row_pieces;
        constexpr int MAXP = 8;                       // pieces per thread and trip (8 x 256 x 16 B = 32 KB per trip: a band of 37 rows of 384 B in one)
        const uint8_t *src0 = img + ((size_t)b * H) * W * 3;
        for (int base = 0; base < n_pieces; base += MAXP * 256) {
            u32x4v v[MAXP];
            int rr[MAXP], cc[MAXP];
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                const int i = base + j * 256 + tid;
                const int r = i / row_pieces, c = i - r * row_pieces;
                rr[j] = r;
                cc[j] = c;
                const int iy = iy0 + r;
                v[j] = (u32x4v){0u, 0u, 0u, 0u};
                if (i < n_pieces && iy >= 0 && iy < H) v[j] = *reinterpret_cast<const u32x4v *>(src0 + ((size_t)iy * row_pieces + c) * 16);
            }
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                const int i = base + j * 256 + tid;
                if (i < n_pieces) {
                    uint32_t *d = reinterpret_cast<uint32_t *>(s_in + (size_t)rr[j] * RSB + 4 + 16 * cc[j]);  // 4-byte aligned: RSB = 4 (mod 16)
                    d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
                    if (cc[j] == 0) d[-1] = 0u;  // the row's pad dword (bytes 1..3 = pixel -1)
                }
            }
        }
    }
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what it indexes stays in SGPRs
    const int li = lane & 15, kq = lane >> 4;
    StemFrag f;
    stem_frag_init(f, w3, bias, lane, RSB);
    const int quad = tid & 7;
    f32x4 tw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tw[t] = *reinterpret_cast<const f32x4 *>(dw_w + t * 32 + quad * 4);
    const f32x4 tb = *reinterpret_cast<const f32x4 *>(dw_b + quad * 4);
    ll4 psum = {0, 0, 0, 0};
    int qmax = 0;  // largest converted output seen (se_range_check)
    const int tiles = Wo / 16;
    __syncthreads();
    for (int sy = y0 - 1; sy <= y1; sy += RPP) {
        // ---- stem rows sy .. sy + RPP - 1 (rows beyond y1: not needed by this band; rows outside the image: zeros)
#pragma unroll
        for (int rr = 0; rr < RPP; ++rr) {
            const int row = sy + rr;
            if (row > y1) continue;  // uniform
            const bool in_img = row >= 0 && row < Ho;
            const uint8_t *rows0 = s_in + (size_t)(2 * (row - y0 + 1)) * RSB;  // staged byte 0 of input row 2 row - 1
            float *ring = s_ring + ((row + RING) % RING) * RP;
            for (int tx = wave; tx < tiles; tx += 4) {
                f32x4 r[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#if defined(PB_STEM_ABL) && (PB_STEM_ABL & 32)
                if (in_img) r[0].x = (float)rows0[96 * tx + lane];  // ablation: no stem tile (one byte read)
#else
                if (in_img) stem_tile(f, rows0 + 96 * tx, r);
#endif
                float *rp = ring + (tx * 16 + li + 1) * 36 + 4 * kq;
#pragma unroll
                for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4 *>(rp + 16 * c) = r[c];
            }
        }
        __syncthreads();
        // ---- depthwise rows sy - 1 .. sy + RPP - 2
#pragma unroll
        for (int rr = 0; rr < RPP; ++rr) {
            const int oy = sy - 1 + rr;
            if (oy < y0 || oy >= y1) continue;  // uniform
            const float *r0 = s_ring + ((oy - 1 + RING) % RING) * RP + 4 * quad, *r1 = s_ring + ((oy + RING) % RING) * RP + 4 * quad,
                        *r2 = s_ring + ((oy + 1 + RING) % RING) * RP + 4 * quad;
            for (int px = tid >> 3; px < Wo; px += 32) {
                f32x4 acc = tb;
#if defined(PB_STEM_ABL) && (PB_STEM_ABL & 4)
                dw_tap(acc, *reinterpret_cast<const f32x4 *>(r1 + px * 36 + 36), tw[4]);  // ablation: one tap
#else
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float *rrp = (ky == 0 ? r0 : (ky == 1 ? r1 : r2)) + px * 36;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) dw_tap(acc, *reinterpret_cast<const f32x4 *>(rrp + kx * 36), tw[ky * 3 + kx]);
                }
#endif
#if defined(PB_STEM_ABL) && (PB_STEM_ABL & 8)
                const f32x4 r = acc;  // ablation: no SiLU behind the filter
#else
                const f32x4 r = {silu_f(acc.x), silu_f(acc.y), silu_f(acc.z), silu_f(acc.w)};
#endif
#if defined(PB_STEM_ABL) && (PB_STEM_ABL & 1)
                if (r.x == 12345.678f)  // ablation: no output stores
#endif
                *reinterpret_cast<f32x4 *>(out + (((size_t)b * Ho + oy) * Wo + px) * 32 + 4 * quad) = r;
#if defined(PB_STEM_ABL) && (PB_STEM_ABL & 16)
                psum.x += __float_as_int(r.x) ^ __float_as_int(r.y) ^ __float_as_int(r.z) ^ __float_as_int(r.w);  // ablation: no fixed-point sums
#else
                se_acc(psum, qmax, r);
#endif
            }
        }
        __syncthreads();  // the next phase overwrites ring slots the filter just read
    }
    se_range_check(qmax, part - 1);
    ll4 *s_red = reinterpret_cast<ll4 *>(s_sd);  // [256]: over the ring (free by now: the loop ends with a barrier)
    s_red[tid] = psum;
    __syncthreads();
    if (tid < 8) {
        ll4 t = s_red[tid];
        for (int j = 1; j < 32; ++j) se_add(t, s_red[j * 8 + tid]);
        se_part_store(part + ((size_t)b * n_bands + band) * 32 + 4 * tid, t);
    }
    // the band that completes the image computes the first block's squeeze-excite gate (se_gate_image; the ring is free by now)
    if (se.sp) {
        __syncthreads();
        if (se_arrive(se.cnt + b, (unsigned)n_bands, reinterpret_cast<unsigned *>(s_sd)))
            se_gate_image_sp<256>(part + (size_t)b * n_bands * 32, n_bands, 32, se, se.gate + (size_t)b * 32, s_sd + 4);
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
constexpr int G_KC = 64;  // K chunk staged in LDS (double-buffered: one barrier per chunk)

// Pipeline: (a) the weight chunk c+1 is fetched global -> registers while chunk c is computed from LDS and
// written to the other LDS buffer afterwards (one barrier per chunk); (b) the activation operand of k-step
// t+PD is requested before the MFMAs of k-step t (register ring), so the global-load latency of the streamed
// operand hides behind 4*MR*NR MFMAs per step times PD steps.
// PDX: activation prefetch distance in k-steps (0: the default, 2 for MR = 4 and 4 otherwise; 8 / 16: the eight-wave form of
// the late layers, whose operand arrives from beyond the XCD's L2 -- with 1 KiB per wave and k-step, 8 waves x 4 steps
// keep 32 KiB per CU in flight, which at ~1.5 us of loaded latency is ~20 GB/s per CU, the rate those layers ran at; the
// ring is registers, and an eight-wave workgroup per CU has 256 of them per lane).
template <int MR, int NR, bool GATE, int NW = 4, int PDX = 0>
__global__ __launch_bounds__(64 * NW) void k_gemm1x1(const float *__restrict__ act, int M, int K,
                                                 const float *__restrict__ wt, int Kpad, int Npad,
                                                 const float *__restrict__ bias, int N,
                                                 const float *__restrict__ gate, int hw,
                                                 const float *__restrict__ resid, int do_silu,
                                                 float *__restrict__ out) {
    constexpr int NT = 16 * NR;
    constexpr int LDW = NT + 4;  // +4: rows k and k+4 land 16 banks apart (conflict-free ds_read_b32)
    constexpr int PD = PDX ? PDX : (MR == 4 ? 2 : 4);  // activation prefetch distance in k-steps: divides, or is a multiple of, the
                                                        // G_KC/16 = 4 steps of a chunk (ring slot = step % PD, a compile-time index)
    constexpr int CPG = PD > 4 ? PD / 4 : 1;            // chunks per turn of the main loop (the slots repeat after PD steps)
    static_assert(PD == 2 || PD == 4 || PD == 8 || PD == 16, "prefetch distance");
    constexpr int NTHR = 64 * NW;  // NW = 8: the same 16*MR rows per wave, twice the waves per block (the late layers have
                                   // few row tiles: more waves per CU hide the operand latency without changing the k order)
    constexpr int WREGS = (G_KC * (NT / 4) + NTHR - 1) / NTHR;  // float4 per thread per weight chunk
    __shared__ __attribute__((aligned(16))) float s_w[2][G_KC * LDW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: what it indexes stays in SGPRs
    const int li = lane & 15;   // pixel within a tile (activation operand) / channel within a tile (weight operand)
    const int kk = lane >> 4;   // k slot
    const int n0 = blockIdx.y * NT;
    const long m_block = (long)blockIdx.x * (16 * NW * MR);
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
        grow[r] = GATE ? gate + (mc / hw) * K : nullptr;
    }
    f32x4 acc[MR][NR];
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int c = 0; c < NR; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int n_steps = Kpad / 16;
    const int n_chunks = (Kpad + G_KC - 1) / G_KC;
    // operand ring: slot (t % PD) holds k-step t; the SE gate travels beside it and is multiplied in only at
    // the point of use (a multiply right after the load would make the prefetch wait for its own data)
    f32x4 aring[PD][MR];
    f32x4 gring[GATE ? PD : 1][GATE ? MR : 1];
    auto load_act = [&](int t, int slot) {
        const int kbase = t * 16 + 4 * kk;
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            // rows >= M and k >= K read a valid dummy address and are zeroed at use (no branch around the load)
            const bool ok = mval[r] && kbase < K;
            const int kb = ok ? kbase : 0;
            aring[slot][r] = *reinterpret_cast<const f32x4 *>(arow[r] + kb);
            if constexpr (GATE) gring[slot][r] = *reinterpret_cast<const f32x4 *>(grow[r] + kb);
        }
    };
#pragma unroll
    for (int t = 0; t < PD; ++t)
        if (t < n_steps) load_act(t, t);

    f32x4 wreg[WREGS];
    auto load_w = [&](int chunk) {
        const int k0 = chunk * G_KC;
#pragma unroll
        for (int j = 0; j < WREGS; ++j) {
            const int i = threadIdx.x + j * NTHR;
            const int kr = i / (NT / 4), c4 = i % (NT / 4);
            const int krc = (k0 + kr) < Kpad ? (k0 + kr) : (Kpad - 1);  // clamp: rows beyond Kpad are never used
            wreg[j] = *reinterpret_cast<const f32x4 *>(wt + (size_t)krc * Npad + n0 + c4 * 4);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WREGS; ++j) {
            const int i = threadIdx.x + j * NTHR;
            const int kr = i / (NT / 4), c4 = i % (NT / 4);
            if (kr < G_KC) *reinterpret_cast<f32x4 *>(&s_w[buf][kr * LDW + c4 * 4]) = wreg[j];
        }
    };
    load_w(0);
    store_w(0);
    __syncthreads();

    // The full chunks run in a loop of their own and the tail chunk (Kpad % 64 != 0) after it: with both forms in ONE
    // loop body the ring slots are phis of two paths and hipcc copies all of them at the back edge behind vmcnt(0).
    auto do_chunk = [&](int chunk, auto full, auto cpos) __attribute__((always_inline)) {
        constexpr int SLOT0 = (4 * decltype(cpos)::value) % PD;  // ring slot of the chunk's first step
        const int k0 = chunk * G_KC;
        const int kc = (Kpad - k0) < G_KC ? (Kpad - k0) : G_KC;
        // full chunks fetch and stage the next weight chunk unconditionally (after the last one: the same chunk again, into
        // the buffer nobody reads): under a condition hipcc's wait-count pass must assume the path "fetched but not
        // stored" and drains every outstanding load (vmcnt(0)) at the top of each chunk, operand ring included
        if constexpr (decltype(full)::value) load_w(chunk + 1 < n_chunks ? chunk + 1 : n_chunks - 1);
        const float *sw = s_w[chunk & 1] + li;
        // one k-step: operands of step t from ring slot u, request of step t + PD into the same slot, 4 NR LDS reads,
        // 4 MR NR MFMAs.  The request is UNCONDITIONAL (past the last step it re-reads the last one, never used): a
        // load under `if (t + PD < n_steps)` makes the slot a phi of (loaded, kept), which hipcc resolves by loading
        // into a temporary and copying it into the slot after the step's MFMAs -- behind s_waitcnt vmcnt(0), i.e. every
        // step waited for the load it had issued 4 MR NR MFMAs earlier and the ring hid one step of latency, not PD.
        auto k_step = [&](int s, int u) __attribute__((always_inline)) {
            const int t = (k0 + s) / 16;  // global k-step; t % PD == u because PD divides the 4 steps of a chunk
            const int kbase = t * 16 + 4 * kk;
            f32x4 a[MR];
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                a[r] = aring[u][r];
                if constexpr (GATE) {
                    const f32x4 g = gring[u][r];
                    a[r].x = a[r].x * g.x; a[r].y = a[r].y * g.y; a[r].z = a[r].z * g.z; a[r].w = a[r].w * g.w;
                }
                if (!(mval[r] && kbase < K)) a[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            load_act((t + PD < n_steps) ? (t + PD) : (n_steps - 1), u);
            // weight fragments of the whole k-step: all 4*NR LDS reads are issued before the first MFMA
            // (sched_barrier keeps hipcc from sinking each read next to its use, which exposes the LDS
            // latency between every pair of MFMAs)
            float wv[4][NR];
            const float *wbase = sw + (s + 4 * kk) * LDW;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < NR; ++c) wv[e][c] = wbase[e * LDW + c * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int r = 0; r < MR; ++r) {
                    const float av = e == 0 ? a[r].x : (e == 1 ? a[r].y : (e == 2 ? a[r].z : a[r].w));
#pragma unroll
                    for (int c = 0; c < NR; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e][c], av, acc[r][c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (decltype(full)::value) {  // four unguarded steps
#pragma unroll
            for (int q = 0; q < G_KC / 16; ++q) k_step(16 * q, (SLOT0 + q) % PD);
        } else {  // tail chunk, once per kernel
#pragma unroll
            for (int q = 0; q < G_KC / 16; ++q)
                if (16 * q < kc) k_step(16 * q, (SLOT0 + q) % PD);
        }
        if constexpr (decltype(full)::value) store_w((chunk + 1) & 1);
        __syncthreads();
    };
    const int n_full = Kpad / G_KC;
    if constexpr (CPG == 1) {
        for (int chunk = 0; chunk < n_full; ++chunk) do_chunk(chunk, std::true_type{}, std::integral_constant<int, 0>{});
        if (n_full < n_chunks) do_chunk(n_full, std::false_type{}, std::integral_constant<int, 0>{});
    } else {
        int chunk = 0;
        for (; chunk + CPG <= n_full; chunk += CPG) {
            do_chunk(chunk, std::true_type{}, std::integral_constant<int, 0>{});
            do_chunk(chunk + 1, std::true_type{}, std::integral_constant<int, 1>{});
            if constexpr (CPG == 4) {
                do_chunk(chunk + 2, std::true_type{}, std::integral_constant<int, 2>{});
                do_chunk(chunk + 3, std::true_type{}, std::integral_constant<int, 3>{});
            }
        }
        // the chunks left over (fewer than CPG, once per kernel): the same bodies at their positions in the turn
        const int rem = n_full - chunk;
        if (rem > 0) do_chunk(chunk, std::true_type{}, std::integral_constant<int, 0>{});
        if constexpr (CPG == 4) {
            if (rem > 1) do_chunk(chunk + 1, std::true_type{}, std::integral_constant<int, 1>{});
            if (rem > 2) do_chunk(chunk + 2, std::true_type{}, std::integral_constant<int, 2>{});
        }
        if (n_full < n_chunks) {
            if (rem == 0) do_chunk(n_full, std::false_type{}, std::integral_constant<int, 0>{});
            else if (rem == 1) do_chunk(n_full, std::false_type{}, std::integral_constant<int, 1>{});
            else if constexpr (CPG == 4) {
                if (rem == 2) do_chunk(n_full, std::false_type{}, std::integral_constant<int, 2>{});
                else do_chunk(n_full, std::false_type{}, std::integral_constant<int, 3>{});
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
// k_gemm_t: the same product as k_gemm1x1<MR = 1> -- same operand maps, same k order, bit-identical results -- with the
// per-k-step overhead taken out of the loop.  In k_gemm1x1 a k-step of 4 NR MFMAs also issues ~20 vector and ~8 scalar
// instructions (row / k masks and clamped 64-bit addresses under saveexec, the gate multiplies, 2 NR two-dword LDS reads that
// the first MFMA waits for), and on this chip vector instructions do not hide under f32 MFMAs (profiles/micro/
// mfma_f32_rate.hip: the two share the SIMD's f32 datapath).  Here:
//  * the weights arrive pre-arranged for the fragment reads: wt2[chunk of 64 k][16-column tile][kk][li][s][e] holds
//    w[k = 64 chunk + 16 s + 4 kk + e][n = 16 tile + li] (zero beyond K / N), so a lane's four e-values of a k-step are ONE
//    ds_read_b128 (NR per k-step instead of 4 NR dwords) and the staging is a straight 16-byte copy; LDS rows of 5 slots
//    (80 B) per (kk, li) make both the staging stores and the fragment reads conflict-free;
//  * the fragments of k-step s + 1 are read while the MFMAs of step s run (two register sets);
//  * rows beyond M are clamped ONCE (they compute on row M - 1 and store nothing), k-steps beyond K exist only in the last
//    chunk (a uniform trip count), so the loop carries no masks; the activation / gate addresses of a chunk's four steps are
//    immediates off one pointer that advances 256 B per chunk.
// Needs K % 16 == 0 (every project / head / FC layer of EfficientNet-B0; the thin expand layers keep k_gemm1x1).
// grid = (ceil(M / (16 NW)), Npad / (16 NR)); block = 64 NW.
// EPI selects the epilogue: 0 bias (+ SiLU) (+ residual), store the [M][N] result;
//   1 (the head conv of a 4 x 4 map: hw = 16, a wave's 16 rows are ONE image): bias + SiLU, then the global average pool in
//     the accumulators -- the 16 pixels summed in pixel order across the lanes (v_add_f32 with a row_shr:1 DPP source, the
//     order of k_avgpool's loop), times `scale` = 1/16 -- and only the pooled [M / 16][N] row is stored: the head's
//     [M][1280] activation (42 MB per 512 images) is neither written nor read back, and k_avgpool's launch is gone;
//   2 (the final Linear): bias, tanh, the u8 quantiser of efficientnet.rs:39, stored to out (f32, optional) and out_u8:
//     k_tanh_quant's launch is gone.
template <int NR, bool GATE, int NW, int EPI = 0>
__global__ __launch_bounds__(64 * NW) void k_gemm_t(const float *__restrict__ act, int M, int K, const float *__restrict__ wt2,
                                                   int tiles16, const float *__restrict__ bias, int N,
                                                   const float *__restrict__ gate, int hw, const float *__restrict__ resid,
                                                   int do_silu, float *__restrict__ out, float scale = 0.f,
                                                   uint8_t *__restrict__ out_u8 = nullptr) {
    constexpr int NTHR = 64 * NW;
    constexpr int CH4 = NR * 256;                         // float4 per weight chunk of this block (NR tiles x 4 kk x 16 li x 4 s)
    constexpr int WREGS = NR;                             // staged by the first 256 threads, NR float4 each (whole waves: no lane predicate)
    constexpr int BUF = NR * 64 * 5;                      // float4 slots per LDS buffer
    __shared__ __attribute__((aligned(16))) f32x4 s_w[2 * BUF];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int c0 = blockIdx.y * NR;  // first 16-column tile
    const long mrow = (long)blockIdx.x * (16 * NW) + wave * 16 + li;
    const bool mval = mrow < M;
    const long mc = mval ? mrow : (long)M - 1;
    const float *ap = act + mc * K + 4 * kk;                             // + 16 floats per k-step
    const float *gp = GATE ? gate + (mc / hw) * K + 4 * kk : nullptr;
    const int n_steps = K >> 4, n_chunks = (n_steps + 3) >> 2;
    f32x4 acc[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // two operand sets of four k-steps each: the MFMAs of chunk c read one set while the steps of chunk c + 1 are requested
    // into the other (a single set, each slot re-requested once read, makes the new value, the old one and their gated
    // product overlap in time: hipcc rotates registers and copies the ring at the loop's back edge behind near-complete
    // waits for the loads just issued -- seen in the ISA)
    f32x4 ar[2][4], gr[2][GATE ? 4 : 1];
    // steps of the first chunk (a chunk shorter than 4 steps re-reads its last step: in bounds, never used)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tc = q < n_steps ? q : n_steps - 1;
        ar[0][q] = *reinterpret_cast<const f32x4 *>(ap + 16 * tc);
        if constexpr (GATE) gr[0][q] = *reinterpret_cast<const f32x4 *>(gp + 16 * tc);
    }

    f32x4 wreg[WREGS];
    const float *wsrc = wt2 + (size_t)c0 * 1024;  // + chunk * tiles16 * 1024 floats
    const int stg = threadIdx.x & 255;
    auto load_w = [&](int chunk) __attribute__((always_inline)) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(wsrc + (size_t)chunk * tiles16 * 1024);
        if (NW == 4 || wave < 4) {  // wave-uniform
#pragma unroll
            for (int j = 0; j < WREGS; ++j) wreg[j] = src[stg + j * 256];
        }
    };
    auto store_w = [&](int buf) __attribute__((always_inline)) {
        if (NW == 4 || wave < 4) {
#pragma unroll
            for (int j = 0; j < WREGS; ++j) {
                const int i = stg + j * 256;  // = (c * 64 + kk * 16 + li) * 4 + s
                s_w[buf * BUF + (i >> 2) * 5 + (i & 3)] = wreg[j];
            }
        }
    };
    load_w(0);
    store_w(0);
    __syncthreads();
    const unsigned rd0 = (unsigned)((kk * 16 + li) * 5);  // + c * 320 + s, + buf * BUF
    f32x4 wa[NR], wb[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) wa[c] = s_w[rd0 + c * 320];
    // One chunk of STEPS k-steps.  The full chunks run in a loop of their own and the short last chunk (K % 64 != 0) after
    // it: with both in ONE loop body the ring slots are phis of two paths, and hipcc resolves them by loading into
    // temporaries that it copies into the slots behind s_waitcnt vmcnt(0) -- every chunk would wait for the loads it has
    // just issued (seen in the ISA of the first version of this kernel, as in k_gemm1x1).
    auto do_chunk = [&](int chunk, auto steps_c, auto set_c) __attribute__((always_inline)) {
        constexpr int STEPS = decltype(steps_c)::value;
        constexpr int CUR = decltype(set_c)::value, NXT = CUR ^ 1;
        const unsigned rb = rd0 + (unsigned)(chunk & 1) * BUF, rn = rd0 + (unsigned)((chunk + 1) & 1) * BUF;
        load_w(chunk + 1 < n_chunks ? chunk + 1 : chunk);  // unconditional (a conditional fetch makes hipcc drain every outstanding load per chunk)
        // one k-step: operands of step q from ring slot q, the same step of the NEXT chunk requested into that slot (past
        // the end: the last step again, never used), the fragments of step q + 1 read into the other register set, then
        // 4 NR MFMAs from this one
        auto k_step = [&](auto qc, f32x4 (&wcur)[NR], f32x4 (&wnxt)[NR]) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            f32x4 a = ar[CUR][q];
            if constexpr (GATE) {
                const f32x4 g = gr[CUR][q];
                a.x = a.x * g.x; a.y = a.y * g.y; a.z = a.z * g.z; a.w = a.w * g.w;
            }
            {
                const int tn = 4 * chunk + 4 + q;  // global k-step requested
                const int tc = tn < n_steps ? tn : n_steps - 1;
                ar[NXT][q] = *reinterpret_cast<const f32x4 *>(ap + 16 * tc);
                if constexpr (GATE) gr[NXT][q] = *reinterpret_cast<const f32x4 *>(gp + 16 * tc);
            }
            if constexpr (q + 1 < STEPS) {
#pragma unroll
                for (int c = 0; c < NR; ++c) wnxt[c] = s_w[rb + c * 320 + q + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float av = e == 0 ? a.x : (e == 1 ? a.y : (e == 2 ? a.z : a.w));
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    const float wv = e == 0 ? wcur[c].x : (e == 1 ? wcur[c].y : (e == 2 ? wcur[c].z : wcur[c].w));
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, av, acc[c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        k_step(std::integral_constant<int, 0>{}, wa, wb);
        if constexpr (STEPS > 1) k_step(std::integral_constant<int, 1>{}, wb, wa);
        if constexpr (STEPS > 2) k_step(std::integral_constant<int, 2>{}, wa, wb);
        if constexpr (STEPS > 3) k_step(std::integral_constant<int, 3>{}, wb, wa);
        store_w((chunk + 1) & 1);
        __syncthreads();
        // first fragments of the next chunk (after the barrier: the buffer has just been written)
#pragma unroll
        for (int c = 0; c < NR; ++c) wa[c] = s_w[rn + c * 320];
    };
    const int n_full = n_steps >> 2;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>;
    int chunk = 0;
    for (; chunk + 2 <= n_full; chunk += 2) {
        do_chunk(chunk, I4{}, I0{});
        do_chunk(chunk + 1, I4{}, I1{});
    }
    {   // once per kernel: an odd full chunk, then the short last chunk (K % 64 != 0) -- on whichever set holds its steps
        const int rem = n_steps & 3;
        if (chunk < n_full) {
            do_chunk(chunk, I4{}, I0{});
            if (rem == 1) do_chunk(n_full, std::integral_constant<int, 1>{}, I1{});
            else if (rem == 2) do_chunk(n_full, std::integral_constant<int, 2>{}, I1{});
            else if (rem == 3) do_chunk(n_full, std::integral_constant<int, 3>{}, I1{});
        } else {
            if (rem == 1) do_chunk(n_full, std::integral_constant<int, 1>{}, I0{});
            else if (rem == 2) do_chunk(n_full, std::integral_constant<int, 2>{}, I0{});
            else if (rem == 3) do_chunk(n_full, std::integral_constant<int, 3>{}, I0{});
        }
    }
    if constexpr (EPI == 1) {
        // M is a multiple of 16 here (whole images), so a tile is valid or not as a whole: no lane leaves before the shifts
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const int n = (c0 + c) * 16 + kk * 4;
            const int nc = n < N ? n : 0;
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + nc);
            f32x4 v = acc[c];
            v.x = silu_f(v.x + b.x); v.y = silu_f(v.y + b.y); v.z = silu_f(v.z + b.z); v.w = silu_f(v.w + b.w);
            // t = 0; for p in 0..15: t = t + v[p]  (k_avgpool's order): after step j lane j of the 16-lane row holds the sum of
            // pixels 0..j; lane 0 keeps 0 + v[0]
            f32x4 t = {0.0f + v.x, 0.0f + v.y, 0.0f + v.z, 0.0f + v.w};
#pragma unroll
            for (int j = 1; j < 16; ++j) {  // unconditional: a lane that already holds its prefix sum recomputes the same value (see k_gemm_p3)
                t.x = dpp_shr1(t.x) + v.x; t.y = dpp_shr1(t.y) + v.y; t.z = dpp_shr1(t.z) + v.z; t.w = dpp_shr1(t.w) + v.w;
            }
            if (mval && li == 15 && n < N) {
                const f32x4 r = {t.x * scale, t.y * scale, t.z * scale, t.w * scale};
                *reinterpret_cast<f32x4 *>(out + (mrow >> 4) * N + n) = r;
            }
        }
        return;
    }
    if (!mval) return;
#pragma unroll
    for (int c = 0; c < NR; ++c) {
        const int n = (c0 + c) * 16 + kk * 4;
        if (n >= N) continue;  // N % 4 == 0
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + n);
        f32x4 v = acc[c];
        v.x = v.x + b.x; v.y = v.y + b.y; v.z = v.z + b.z; v.w = v.w + b.w;
        if constexpr (EPI == 2) {
            const f32x4 y = {tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w)};
            if (out) *reinterpret_cast<f32x4 *>(out + mrow * N + n) = y;
            const uint32_t pk = (uint32_t)quantize_u8(y.x) | ((uint32_t)quantize_u8(y.y) << 8) | ((uint32_t)quantize_u8(y.z) << 16) |
                                ((uint32_t)quantize_u8(y.w) << 24);
            *reinterpret_cast<uint32_t *>(out_u8 + mrow * N + n) = pk;
            continue;
        }
        if (do_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        if (resid) {
            const f32x4 rv = *reinterpret_cast<const f32x4 *>(resid + mrow * N + n);
            v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w;
        }
        *reinterpret_cast<f32x4 *>(out + mrow * N + n) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// k_gemm_stream: the gated project GEMMs of the EARLY blocks (K = 32 ... 240, N <= 48, hundreds of thousands of pixel rows)
// as a stream.  Those layers are memory-bound -- 0.25-0.4 GB read per launch for 2-4 GFLOP -- and the tiled kernels above pay
// a workgroup's set-up (weight staging, barrier, gate pointers) for every 64 rows.  Here a wave keeps the WHOLE weight matrix
// as MFMA fragments in registers (KS * NT float4 per lane, from Gemm::wt4), walks `tiles_per_wave` consecutive 16-row tiles
// of one image with the next tile's activation (and residual) fragments requested before the current tile's MFMAs, and holds
// the image's squeeze-excite gate in registers; no LDS, no barrier.  Same operand maps, same k order, gate multiplied into
// the activation fragment at use: bit-identical to k_gemm1x1 / k_gemm_t.
// Needs K = 16 KS, hw % (16 tiles_per_wave) == 0 (a wave never straddles two images), M % 16 == 0.
// grid = ceil(M / 16 / tiles_per_wave / 4); block = 256.
template <int KS, int NT, bool RESID>
__global__ __launch_bounds__(256) void k_gemm_stream(const float *__restrict__ act, long M, const float *__restrict__ wt4, int nt16,
                                                     const float *__restrict__ bias, int N, const float *__restrict__ gate, int hw,
                                                     const float *__restrict__ resid, float *__restrict__ out, int tiles_per_wave) {
    constexpr int K = 16 * KS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const long n_tiles = M >> 4;
    const long t0 = ((long)blockIdx.x * 4 + wave) * tiles_per_wave;
    if (t0 >= n_tiles) return;
    const long t1 = t0 + tiles_per_wave < n_tiles ? t0 + tiles_per_wave : n_tiles;
    f32x4 wf[KS][NT], bq[NT], g[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int c = 0; c < NT; ++c) wf[s][c] = *reinterpret_cast<const f32x4 *>(wt4 + (((size_t)s * nt16 + c) * 64 + lane) * 4);
#pragma unroll
    for (int c = 0; c < NT; ++c) bq[c] = *reinterpret_cast<const f32x4 *>(bias + 16 * c + 4 * kk);
    {
        const float *gp = gate + ((t0 * 16) / hw) * K + 4 * kk;
#pragma unroll
        for (int s = 0; s < KS; ++s) g[s] = *reinterpret_cast<const f32x4 *>(gp + 16 * s);
    }
    const float *ap = act + (t0 * 16 + li) * K + 4 * kk;  // + 16 K floats per tile
    const float *rp = RESID ? resid + (t0 * 16 + li) * N + 4 * kk : nullptr;
    float *op = out + (t0 * 16 + li) * N + 4 * kk;
    f32x4 a[2][KS], rv[2][RESID ? NT : 1];
    auto request = [&](long t, auto setc) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
        const long d = (t < t1 ? t : t1 - 1) - t0;  // past the end: the last tile again, never used
#pragma unroll
        for (int s = 0; s < KS; ++s) a[SET][s] = *reinterpret_cast<const f32x4 *>(ap + d * 16 * K + 16 * s);
        if constexpr (RESID) {
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const int n = 16 * c + 4 * kk;
                rv[SET][c] = *reinterpret_cast<const f32x4 *>(rp + d * 16 * N + (n < N ? 16 * c : 0));
            }
        }
    };
    auto tile = [&](long t, auto setc) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
        request(t + 1, std::integral_constant<int, SET ^ 1>{});
        f32x4 acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 v = a[SET][s];
            v.x = v.x * g[s].x; v.y = v.y * g[s].y; v.z = v.z * g[s].z; v.w = v.w * g[s].w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float av = e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    const f32x4 wq = wf[s][c];
                    const float wv = e == 0 ? wq.x : (e == 1 ? wq.y : (e == 2 ? wq.z : wq.w));
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, av, acc[c], 0, 0, 0);
                }
            }
        }
        const long d = t - t0;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const int n = 16 * c + 4 * kk;
            if (n >= N) continue;  // N % 4 == 0
            f32x4 v = acc[c];
            v.x = v.x + bq[c].x; v.y = v.y + bq[c].y; v.z = v.z + bq[c].z; v.w = v.w + bq[c].w;
            if constexpr (RESID) {
                const f32x4 r = rv[SET][c];
                v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w;
            }
            *reinterpret_cast<f32x4 *>(op + d * 16 * N + 16 * c) = v;
        }
    };
    request(t0, std::integral_constant<int, 0>{});
    for (long t = t0; t < t1; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < t1) tile(t + 1, std::integral_constant<int, 1>{});
        else break;
    }
}

// ------------------------------------------------------------------------------------------------
// The same 1x1 convolution for a handful of pixel rows (one image: M = 16 .. 256): one wave per workgroup owns one
// 16-row x 16-channel tile and reads its weight operand straight from global memory (4 dwords per k-step, 64-byte
// segments across the 16 channel lanes), PD k-steps ahead, with no LDS staging and no barriers.  k_gemm1x1 on such a
// problem is a latency chain -- per 64-deep chunk a weight fetch, an LDS store and a workgroup barrier for ONE busy
// wave (15 us for K = 1152 at M = 16); here the chain is the K / 4 dependent MFMAs themselves.  Operand maps, k order,
// gate-then-mask and epilogue are those of k_gemm1x1: the outputs are bit-identical.
// grid = (ceil(M / 16), Npad / 16); block = 64.
template <bool GATE>
__global__ __launch_bounds__(64) void k_gemm_thin(const float *__restrict__ act, int M, int K, const float *__restrict__ wt,
                                                  int Kpad, int Npad, const float *__restrict__ bias, int N,
                                                  const float *__restrict__ gate, int hw, const float *__restrict__ resid,
                                                  int do_silu, float *__restrict__ out) {
    constexpr int PD = 16;  // k-steps in flight: at batch 1 the weights come from HBM / Infinity Cache (each byte is used once per forward)
    const int lane = threadIdx.x & 63, li = lane & 15, kk = lane >> 4;
    const int n0 = blockIdx.y * 16;
    const long mrow = (long)blockIdx.x * 16 + li;
    const bool mval = mrow < M;
    const long mc = mval ? mrow : 0;
    const float *arow = act + mc * K;
    const float *grow = GATE ? gate + (mc / hw) * K : nullptr;
    const float *wcol = wt + n0 + li;
    const int n_steps = Kpad / 16;
    f32x4 ar[PD], gr[GATE ? PD : 1];
    float wr[PD][4];
    auto load_step = [&](int t, int slot) __attribute__((always_inline)) {
        const int kbase = t * 16 + 4 * kk;
        const int kb = (mval && kbase < K) ? kbase : 0;
        ar[slot] = *reinterpret_cast<const f32x4 *>(arow + kb);
        if constexpr (GATE) gr[slot] = *reinterpret_cast<const f32x4 *>(grow + kb);
#pragma unroll
        for (int e = 0; e < 4; ++e) wr[slot][e] = wcol[(size_t)(kbase + e) * Npad];
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto k_step = [&](int t, int slot) __attribute__((always_inline)) {
        const int kbase = t * 16 + 4 * kk;
        f32x4 a = ar[slot];
        if constexpr (GATE) {
            const f32x4 g = gr[slot];
            a.x = a.x * g.x; a.y = a.y * g.y; a.z = a.z * g.z; a.w = a.w * g.w;
        }
        if (!(mval && kbase < K)) a = (f32x4){0.f, 0.f, 0.f, 0.f};
        float w4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = wr[slot][e];
        load_step((t + PD < n_steps) ? (t + PD) : (n_steps - 1), slot);  // unconditional: see k_gemm1x1
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[0], a.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[1], a.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[2], a.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[3], a.w, acc, 0, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < PD; ++t) load_step(t < n_steps ? t : n_steps - 1, t);
    int t0 = 0;
    for (; t0 + PD <= n_steps; t0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) k_step(t0 + u, u);
    }
#pragma unroll
    for (int u = 0; u < PD; ++u)
        if (t0 + u < n_steps) k_step(t0 + u, u);
    const int n = n0 + kk * 4;
    if (!mval || n >= N) return;  // N % 4 == 0
    const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + n);
    f32x4 v = acc;
    v.x = v.x + b.x; v.y = v.y + b.y; v.z = v.z + b.z; v.w = v.w + b.w;
    if (do_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
    if (resid) {
        const f32x4 rv = *reinterpret_cast<const f32x4 *>(resid + mrow * N + n);
        v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w;
    }
    *reinterpret_cast<f32x4 *>(out + mrow * N + n) = v;
}

// ------------------------------------------------------------------------------------------------
// depthwise KSxKS conv, stride S, pad (KS-1)/2, + bias + SiLU, NHWC, with the squeeze-excite pooling
// partial sums fused: part[b][tile][c] = sum of the outputs of this block's pixels (fixed order ->
// deterministic).  Register tiling: a thread owns a strip of TX = 4 adjacent output pixels of one row for one
// channel quad; per filter row it loads the (TX-1)*S + KS input float4 once and reuses them across the strip
// (5x5 s1: 40 loads per 4 outputs instead of 100), filter taps come from LDS.  Accumulation order per output is
// (ky, kx), as in the oracle.
// w: [KS*KS][C] tap-major.  grid = (tiles_per_image, B, zsplit); blockDim = cq_per_block * slots with
// cq_per_block = (C/4) / zsplit channel quads per block (exact), slots = 256 / cq_per_block strips in flight.
template <int KS, int S>
__global__ __launch_bounds__(256) void k_dwconv(const float *__restrict__ in, int H, int W, int C,
                                                const float *__restrict__ w, const float *__restrict__ bias,
                                                float *__restrict__ out, int Ho, int Wo, int strips_per_tile,
                                                long long *__restrict__ part, int n_tiles, int cq_per_block) {
    constexpr int PAD = (KS - 1) / 2;
    constexpr int TX = 4;
    constexpr int NX = (TX - 1) * S + KS;
    __shared__ ll4 s_red[256];
    extern __shared__ f32x4 s_wt[];  // [KS*KS][cq_per_block]
    const int slots = blockDim.x / cq_per_block;
    const int cq_l = threadIdx.x % cq_per_block;
    const int slot = threadIdx.x / cq_per_block;
    const int cq = blockIdx.z * cq_per_block + cq_l;
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    const int c0 = cq * 4;
    for (int i = threadIdx.x; i < KS * KS * cq_per_block; i += blockDim.x) {
        const int t = i / cq_per_block, q = i % cq_per_block;
        s_wt[i] = *reinterpret_cast<const f32x4 *>(w + (size_t)t * C + (blockIdx.z * cq_per_block + q) * 4);
    }
    __syncthreads();
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + c0);
    const float *ib = in + (size_t)b * H * W * C + c0;
    float *ob = out + (size_t)b * Ho * Wo * C + c0;
    const int strips_x = (Wo + TX - 1) / TX;
    const int n_strips = Ho * strips_x;
    ll4 psum = {0, 0, 0, 0};
    int qmax = 0;  // largest converted output seen (se_range_check)
    const int st_begin = tile * strips_per_tile;
    const int st_end = (st_begin + strips_per_tile) < n_strips ? (st_begin + strips_per_tile) : n_strips;
    for (int st = st_begin + slot; st < st_end; st += slots) {
        const int y = st / strips_x, x0 = (st % strips_x) * TX;
        f32x4 acc[TX];
#pragma unroll
        for (int t = 0; t < TX; ++t) acc[t] = bv;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = y * S + ky - PAD;
            if (iy < 0 || iy >= H) continue;
            const float *rowp = ib + (size_t)iy * W * C;
            f32x4 seg[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const int ix = x0 * S - PAD + j;
                seg[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ix >= 0 && ix < W) seg[j] = *reinterpret_cast<const f32x4 *>(rowp + (size_t)ix * C);
            }
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const f32x4 wv = s_wt[(ky * KS + kx) * cq_per_block + cq_l];
#pragma unroll
                for (int t = 0; t < TX; ++t) {
                    // out-of-range taps hold zeros: acc + 0*w == acc, the same value the oracle's skip gives
                    const f32x4 v = seg[t * S + kx];
                    dw_tap(acc[t], v, wv);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TX; ++t) {
            if (x0 + t < Wo) {
                f32x4 o = {silu_f(acc[t].x), silu_f(acc[t].y), silu_f(acc[t].z), silu_f(acc[t].w)};
                *reinterpret_cast<f32x4 *>(ob + ((size_t)y * Wo + x0 + t) * C) = o;
                se_acc(psum, qmax, o);
            }
        }
    }
    se_range_check(qmax, part - 1);
    s_red[threadIdx.x] = psum;
    __syncthreads();
    if (slot == 0) {
        ll4 t = s_red[cq_l];
        for (int sl = 1; sl < slots; ++sl) se_add(t, s_red[sl * cq_per_block + cq_l]);
        *reinterpret_cast<ll4 *>(part + ((size_t)b * n_tiles + tile) * C + c0) = t;
    }
}

// ------------------------------------------------------------------------------------------------
// depthwise conv, rolling-window form for the large early maps: a thread owns a COLUMN strip of TX adjacent
// outputs for one channel quad and walks down the rows of its band keeping the KS x ((TX-1)*S+KS) input window
// in registers -- every new output row loads only S new input rows (3x3 s1: 1.5 float4 loads per output quad
// instead of 4.5; 5x5 s1: 3 instead of 10).  Same accumulation order (ky, kx), same fused bias + SiLU + SE
// partial sums as k_dwconv.  grid = (bands, B, zsplit); blockDim = cq_per_block * strips_x.
template <int KS, int S, int TX>
__global__ __launch_bounds__(256) void k_dwconv_roll(const float *__restrict__ in, int H, int W, int C,
                                                     const float *__restrict__ w, const float *__restrict__ bias,
                                                     float *__restrict__ out, int Ho, int Wo, int rows_per_band,
                                                     long long *__restrict__ part, int n_bands, int cq_per_block) {
    constexpr int PAD = (KS - 1) / 2;
    constexpr int NX = (TX - 1) * S + KS;
    __shared__ ll4 s_red[256];
    extern __shared__ f32x4 s_wt[];  // [KS*KS][cq_per_block]
    const int strips_x = blockDim.x / cq_per_block;
    const int cq_l = threadIdx.x % cq_per_block;
    const int sx = threadIdx.x / cq_per_block;
    const int cq = blockIdx.z * cq_per_block + cq_l;
    const int b = blockIdx.y;
    const int band = blockIdx.x;
    const int c0 = cq * 4;
    for (int i = threadIdx.x; i < KS * KS * cq_per_block; i += blockDim.x) {
        const int t = i / cq_per_block, q = i % cq_per_block;
        s_wt[i] = *reinterpret_cast<const f32x4 *>(w + (size_t)t * C + (blockIdx.z * cq_per_block + q) * 4);
    }
    __syncthreads();
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + c0);
    const float *ib = in + (size_t)b * H * W * C + c0;
    float *ob = out + (size_t)b * Ho * Wo * C + c0;
    const int x0 = sx * TX;
    const int ix0 = x0 * S - PAD;
    const int y_begin = band * rows_per_band;
    const int y_end = (y_begin + rows_per_band) < Ho ? (y_begin + rows_per_band) : Ho;
    ll4 psum = {0, 0, 0, 0};
    int qmax = 0;  // largest converted output seen (se_range_check)
    f32x4 win[KS][NX];
    auto load_row = [&](int iy, f32x4 (&dst)[NX]) {
        const bool rv = iy >= 0 && iy < H;
        const float *rowp = ib + (size_t)(rv ? iy : 0) * W * C;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int ix = ix0 + j;
            dst[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (rv && ix >= 0 && ix < W) dst[j] = *reinterpret_cast<const f32x4 *>(rowp + (size_t)ix * C);
        }
    };
    if (y_begin < y_end && x0 < Wo) {
#pragma unroll
        for (int r = 0; r < KS; ++r) load_row(y_begin * S - PAD + r, win[r]);
        for (int y = y_begin; y < y_end; ++y) {
            f32x4 acc[TX];
#pragma unroll
            for (int t = 0; t < TX; ++t) acc[t] = bv;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const f32x4 wv = s_wt[(ky * KS + kx) * cq_per_block + cq_l];
#pragma unroll
                    for (int t = 0; t < TX; ++t) {
                        const f32x4 v = win[ky][t * S + kx];
                        dw_tap(acc[t], v, wv);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                if (x0 + t < Wo) {
                    f32x4 o = {silu_f(acc[t].x), silu_f(acc[t].y), silu_f(acc[t].z), silu_f(acc[t].w)};
                    *reinterpret_cast<f32x4 *>(ob + ((size_t)y * Wo + x0 + t) * C) = o;
                    se_acc(psum, qmax, o);
                }
            }
            if (y + 1 < y_end) {  // slide the window down by S rows
#pragma unroll
                for (int r = 0; r + S < KS; ++r)
#pragma unroll
                    for (int j = 0; j < NX; ++j) win[r][j] = win[r + S][j];
#pragma unroll
                for (int r = (KS - S > 0 ? KS - S : 0); r < KS; ++r) load_row((y + 1) * S - PAD + r, win[r]);
            }
        }
    }
    se_range_check(qmax, part - 1);
    s_red[threadIdx.x] = psum;
    __syncthreads();
    if (sx == 0) {
        ll4 t = s_red[cq_l];
        for (int sl = 1; sl < strips_x; ++sl) se_add(t, s_red[sl * cq_per_block + cq_l]);
        *reinterpret_cast<ll4 *>(part + ((size_t)b * n_bands + band) * C + c0) = t;
    }
}

// ------------------------------------------------------------------------------------------------
// Depthwise conv for SMALL maps (<= 16 x 16 input): one block stages the whole zero-padded map of its channel
// quads in LDS -- every input element is read from global memory exactly once, coalesced -- and all KS*KS taps of
// every output come from LDS.  (The strip kernel re-reads each input ~10x through L1/L2, which is what bounds the
// 8x8 and 4x4 layers.)  Same bias-first (ky, kx) accumulation with separately rounded products as k_dwconv, zero
// taps outside the image (acc + 0*w == acc): bit-identical.  The block sees all pixels of its channels, so the SE
// partial sum is the whole sum (n_tiles = 1).
// grid = (1, B, C/4/cqb); blockDim = cqb * slots; dynamic LDS = ((H+2P)(W+2P) + KS*KS) * cqb * 16 bytes.
template <int KS, int S>
__global__ __launch_bounds__(256) void k_dwconv_lds(const float *__restrict__ in, int H, int W, int C,
                                                    const float *__restrict__ w, const float *__restrict__ bias,
                                                    float *__restrict__ out, int Ho, int Wo,
                                                    long long *__restrict__ part, int cqb) {
    constexpr int PAD = (KS - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) f32x4 s_lds_dw[];
    __shared__ ll4 s_red[256];
    const int Hp = H + 2 * PAD, Wp = W + 2 * PAD;
    f32x4 *s_in = s_lds_dw;                    // [Hp * Wp][cqb]
    f32x4 *s_w = s_lds_dw + Hp * Wp * cqb;     // [KS * KS][cqb]
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int cq_l = tid % cqb, slot = tid / cqb, slots = blockDim.x / cqb;
    const int c0 = 4 * (blockIdx.z * cqb + cq_l);
    const float *ib = in + (size_t)b * H * W * C + c0;
    for (int p = slot; p < Hp * Wp; p += slots) {
        const int y = p / Wp - PAD, x = p % Wp - PAD;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < H && x >= 0 && x < W) v = *reinterpret_cast<const f32x4 *>(ib + ((size_t)y * W + x) * C);
        s_in[p * cqb + cq_l] = v;
    }
    for (int t = slot; t < KS * KS; t += slots) s_w[t * cqb + cq_l] = *reinterpret_cast<const f32x4 *>(w + (size_t)t * C + c0);
    __syncthreads();
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + c0);
    float *ob = out + (size_t)b * Ho * Wo * C + c0;
    ll4 psum = {0, 0, 0, 0};
    int qmax = 0;  // largest converted output seen (se_range_check)
    for (int o = slot; o < Ho * Wo; o += slots) {
        const int oy = o / Wo, ox = o % Wo;
        f32x4 acc = bv;
        const f32x4 *ip = s_in + ((oy * S) * Wp + ox * S) * cqb + cq_l;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const f32x4 v = ip[(ky * Wp + kx) * cqb];
                const f32x4 wv = s_w[(ky * KS + kx) * cqb + cq_l];
                dw_tap(acc, v, wv);
            }
        const f32x4 r = {silu_f(acc.x), silu_f(acc.y), silu_f(acc.z), silu_f(acc.w)};
        *reinterpret_cast<f32x4 *>(ob + (size_t)o * C) = r;
        se_acc(psum, qmax, r);
    }
    se_range_check(qmax, part - 1);
    s_red[tid] = psum;
    __syncthreads();
    if (slot == 0) {
        ll4 t = s_red[cq_l];
        for (int sl = 1; sl < slots; ++sl) se_add(t, s_red[sl * cqb + cq_l]);
        *reinterpret_cast<ll4 *>(part + (size_t)b * C + c0) = t;
    }
}

// ------------------------------------------------------------------------------------------------
// Rolling fused MBConv front: expand 1x1 (+bias+SiLU) -> depthwise KSxKS stride S (+bias+SiLU) -> SE partial sums
// with the expanded activation living only in REGISTERS.
// A wave owns a strip of 16 adjacent input columns and 16 NC expanded channels and walks down the rows of its band.
// Per input row it runs the expand on the matrix cores with the 16 columns as the MFMA's column index, so lane
// (li, kq) ends up holding channels 16c + 4kq .. +3 of column li -- exactly the operand map of k_gemm1x1, with the
// weight fragments held in registers for the whole walk.  The last KS expanded rows stay in a register ring; the
// depthwise filter takes its x-neighbours from the adjacent lanes of the 16-lane row with DPP row shifts (folded
// into the fused multiply-adds: v_fmac_f32_dpp, see fmac_shift) and its y-neighbours from the ring.  A strip yields
// OW = (16 - KS) / S + 1 output columns (14 / 7 / 12 / 6 for 3x3 s1 / 3x3 s2 / 5x5 s1 / 5x5 s2); the x halo is
// recomputed (MFMA work is not the limit here), the y halo only at band boundaries.  No LDS traffic apart from
// the broadcast reads of the filter taps.  Same k order, tap order and roundings as the
// two-kernel path: the outputs are bit-identical to k_gemm1x1 + k_dwconv.
// grid = (ceil(n_strips * n_bands / 4), B, E / (16 NC)); block = 4 independent waves of one channel group.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// the value held by lane li + d of the 16-lane row, 0 outside the row (d is a constant after unrolling)
__device__ __forceinline__ float row_shift(float v, int d) {
    switch (d) {
        case -2: return dpp_f32<0x112>(v);  // row_shr:2
        case -1: return dpp_f32<0x111>(v);  // row_shr:1
        case 1: return dpp_f32<0x101>(v);   // row_shl:1
        case 2: return dpp_f32<0x102>(v);   // row_shl:2
        default: return v;
    }
}

// acc += (value of x in lane li + d of the 16-lane row, 0 outside the row) * w as ONE instruction: v_fmac_f32 with the
// DPP row shift on its first source.  hipcc folds update_dpp into v_mul_f32 but not into the tied-operand v_fmac, and
// a v_mov_b32_dpp per shifted tap doubles the filter's issue slots.  The DPP source must not have been written by
// the two VALU instructions before (hardware does not interlock that; hipcc inserts the wait states for its own DPP
// moves but cannot see inside asm): every use below reads a ring row that an earlier phase of the row iteration
// produced.  Same value as dw_tap on row_shift(x, d).
__device__ __forceinline__ float fmac_shift(float acc, float x, float w, int d) {
#ifdef PB_DW_UNFUSED
    return acc + row_shift(x, d) * w;
#else
    switch (d) {
        case -2: asm("v_fmac_f32_dpp %0, %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(w)); break;
        case -1: asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(w)); break;
        case 1: asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(w)); break;
        case 2: asm("v_fmac_f32_dpp %0, %1, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(w)); break;
        default: acc = __builtin_fmaf(x, w, acc); break;
    }
    return acc;
#endif
}

template <int KS, int S, int KC, int NC>
__global__ __launch_bounds__(256, NC == 1 ? 4 : 2) void k_front_roll(
    const float *__restrict__ x, int H, int W, int Cin, const float *__restrict__ wt, int Epad,
    const float *__restrict__ bias_e, const float *__restrict__ dw_w, const float *__restrict__ dw_b, int E,
    float *__restrict__ out, int Ho, int Wo, long long *__restrict__ part, int n_strips, int n_bands, int rows_per_band) {
    constexpr int EC = 16 * NC;
    constexpr int PAD = (KS - 1) / 2;
    constexpr int OW = (16 - KS) / S + 1;
    __shared__ __attribute__((aligned(16))) float s_dw[KS * KS * EC + 2 * EC];  // taps, then expand bias, depthwise bias
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what it indexes stays in SGPRs
    const int li = lane & 15, kq = lane >> 4;
    const int b = blockIdx.y;
    const int e0 = blockIdx.z * EC;
    for (int i = tid; i < KS * KS * (EC / 4); i += 256) {
        const int t = i / (EC / 4), c4 = i % (EC / 4);
        *reinterpret_cast<f32x4 *>(s_dw + t * EC + c4 * 4) = *reinterpret_cast<const f32x4 *>(dw_w + (size_t)t * E + e0 + c4 * 4);
    }
    if (tid < EC) {
        s_dw[KS * KS * EC + tid] = bias_e[e0 + tid];
        s_dw[KS * KS * EC + EC + tid] = dw_b[e0 + tid];
    }
    __syncthreads();
    const int item = blockIdx.x * 4 + wave;
    if (item >= n_strips * n_bands) return;
    const int strip = item % n_strips, band = item / n_strips;
    const int oy_b = band * rows_per_band;
    const int oy_e = (oy_b + rows_per_band) < Ho ? (oy_b + rows_per_band) : Ho;
    const int xpos = strip * OW * S - PAD + li;  // this lane's input column
    const bool xok = xpos >= 0 && xpos < W;
    const int jo = li - PAD;                     // output lanes sit on the centre tap of their window
    const int ox = strip * OW + jo / S;
    const bool is_out = jo >= 0 && (jo % S) == 0 && (jo / S) < OW && ox < Wo;
    const float *xb = x + (size_t)b * H * W * Cin;

    // weight fragments for the whole walk: A operand, lane (li, kq) holds wt[k = 16s + 4kq + e][e0 + 16c + li]
    float wreg[KC][4][NC];
#pragma unroll
    for (int s2 = 0; s2 < KC; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < NC; ++c) wreg[s2][e][c] = wt[(size_t)(16 * s2 + 4 * kq + e) * Epad + e0 + 16 * c + li];

    auto load_row = [&](int iy, f32x4 (&xa)[KC]) {
        const bool ok = xok && iy >= 0 && iy < H;
        const float *p = xb + ((size_t)(ok ? iy : 0) * W + (ok ? xpos : 0)) * Cin;
#pragma unroll
        for (int s2 = 0; s2 < KC; ++s2) {
            const int k = 16 * s2 + 4 * kq;
            // out-of-range lanes read a valid dummy address and are zeroed (no branch around the load)
            f32x4 v = *reinterpret_cast<const f32x4 *>(p + (k < Cin ? k : 0));
            if (!(ok && k < Cin)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            xa[s2] = v;
        }
    };
    auto expand_row = [&](int iy, const f32x4 (&xa)[KC], f32x4 (&ev)[NC]) {
        f32x4 acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < KC; ++s2)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float av = e == 0 ? xa[s2].x : (e == 1 ? xa[s2].y : (e == 2 ? xa[s2].z : xa[s2].w));
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s2][e][c], av, acc[c], 0, 0, 0);
            }
        const bool ok = xok && iy >= 0 && iy < H;  // the depthwise zero padding applies to the EXPANDED activation
        // biases are re-read from LDS (kept out of long-lived registers, see tap_off below); the opaque value is the quad
        // index, not the offset, so that hipcc still sees a 16-byte-aligned address and reads with ds_read_b128
        int kq_b = kq;
        asm volatile("" : "+v"(kq_b));
        const int b_off = 4 * kq_b;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 v = acc[c];
            const f32x4 bev = *reinterpret_cast<const f32x4 *>(s_dw + KS * KS * EC + 16 * c + b_off);
            v.x = silu_f(v.x + bev.x); v.y = silu_f(v.y + bev.y); v.z = silu_f(v.z + bev.z); v.w = silu_f(v.w + bev.w);
            if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            ev[c] = v;
        }
    };

    // a 3x3 filter of one channel group is 36 registers: kept for the whole walk (the kernel has them to spare);
    // the larger ones are re-read from LDS every row, see tap_off below
    constexpr bool TAPS_IN_REGS = KS * KS * NC <= 9;
    f32x4 wtap[TAPS_IN_REGS ? KS * KS : 1][NC];
    if constexpr (TAPS_IN_REGS) {
#pragma unroll
        for (int t = 0; t < KS * KS; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) wtap[t][c] = *reinterpret_cast<const f32x4 *>(s_dw + t * EC + 16 * c + 4 * kq);
    }
    f32x4 ring[KS][NC];
    f32x4 xa[S][KC];
    ll4 psum[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) psum[c] = (ll4){0, 0, 0, 0};
    int qmax = 0;  // largest converted output seen (se_range_check)
    // prime: rows oy_b*S - PAD .. + (KS - S - 1) go to ring[S ..]; the loop shifts them down before use
    const int iy_first = oy_b * S - PAD;
#pragma unroll
    for (int r = 0; r < KS - S; ++r) {
        f32x4 xp[KC];
        load_row(iy_first + r, xp);
        expand_row(iy_first + r, xp, ring[S + r]);
    }
#pragma unroll
    for (int j = 0; j < S; ++j) load_row(iy_first + KS - S + j, xa[j]);
    for (int oy = oy_b; oy < oy_e; ++oy) {
        const int iy_new = oy * S - PAD + KS - S;  // first of the S new input rows of this output row
#pragma unroll
        for (int r = 0; r < KS - S; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) ring[r][c] = ring[r + S][c];
#pragma unroll
        for (int j = 0; j < S; ++j) expand_row(iy_new + j, xa[j], ring[KS - S + j]);
        if (oy + 1 < oy_e) {
#pragma unroll
            for (int j = 0; j < S; ++j) load_row(iy_new + S + j, xa[j]);  // next output row's inputs, in flight during the filter
        }
        // the taps are re-read from LDS every row (broadcast reads, 4 addresses per wave); the opaque offset keeps
        // hipcc from hoisting all KS*KS*NC float4 out of the row loop into registers (and spilling)
        // (opaque quad index, aligned offset: with the offset itself opaque every tap read became two ds_read2_b32 plus an
        // address add -- 72 + 42 of the ~1 050 instructions of a row of the 48-channel form)
        int kq_t = kq;
        asm volatile("" : "+v"(kq_t));
        const int tap_off = 4 * kq_t;
        f32x4 o[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c] = *reinterpret_cast<const f32x4 *>(s_dw + KS * KS * EC + EC + 16 * c + tap_off);
        // fresh SSA names for the ring: otherwise hipcc recognises that a row's shifted copies were already
        // computed for the previous output row and keeps all KS*(KS-1) of them alive across iterations
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                asm volatile("" : "+v"(ring[ky][c].x), "+v"(ring[ky][c].y), "+v"(ring[ky][c].z), "+v"(ring[ky][c].w));
        // tap t+1 is requested while tap t is applied; the scheduling barriers keep hipcc from issuing all
        // KS*KS*NC reads up front (100+ live registers)
        f32x4 wn[NC];
        if constexpr (!TAPS_IN_REGS) {
#pragma unroll
            for (int c = 0; c < NC; ++c) wn[c] = *reinterpret_cast<const f32x4 *>(s_dw + 16 * c + tap_off);
        }
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) {
            const int ky = t / KS, kx = t % KS;
            f32x4 wc[NC];
            if constexpr (TAPS_IN_REGS) {
#pragma unroll
                for (int c = 0; c < NC; ++c) wc[c] = wtap[t][c];
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) wc[c] = wn[c];
                if (t + 1 < KS * KS) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) wn[c] = *reinterpret_cast<const f32x4 *>(s_dw + (t + 1) * EC + 16 * c + tap_off);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 v = ring[ky][c];
                o[c].x = fmac_shift(o[c].x, v.x, wc[c].x, kx - PAD); o[c].y = fmac_shift(o[c].y, v.y, wc[c].y, kx - PAD);
                o[c].z = fmac_shift(o[c].z, v.z, wc[c].z, kx - PAD); o[c].w = fmac_shift(o[c].w, v.w, wc[c].w, kx - PAD);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // pin the filter arithmetic here: hipcc otherwise sinks it into the is_out branch below while the DPP moves
        // and tap reads (which cannot move into divergent control flow) stay outside with all their results live
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(o[c].x), "+v"(o[c].y), "+v"(o[c].z), "+v"(o[c].w));
        if (is_out) {
            float *op = out + (((size_t)b * Ho + oy) * Wo + ox) * E + e0 + 4 * kq;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 r = {silu_f(o[c].x), silu_f(o[c].y), silu_f(o[c].z), silu_f(o[c].w)};
                *reinterpret_cast<f32x4 *>(op + 16 * c) = r;
                se_acc(psum[c], qmax, r);
            }
        }
    }
    se_range_check(qmax, part - 1);
    // SE partial of this (strip, band): sum over the 16 columns of the row (exact integer adds)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            psum[c].x += __shfl_xor(psum[c].x, m, 64);
            psum[c].y += __shfl_xor(psum[c].y, m, 64);
            psum[c].z += __shfl_xor(psum[c].z, m, 64);
            psum[c].w += __shfl_xor(psum[c].w, m, 64);
        }
        if (li == 0) *reinterpret_cast<ll4 *>(part + ((size_t)b * (n_strips * n_bands) + item) * E + e0 + 16 * c + 4 * kq) = psum[c];
    }
}

// ------------------------------------------------------------------------------------------------
// Fused MBConv front for SMALL maps (16 x 16, 8 x 8 and 4 x 4 inputs: the twelve late blocks).  There the 6x-expanded
// activation of one image is small (64 px x 48 channels = 12 KB per channel group), so a workgroup computes it
// whole -- expand 1x1 on the f32 MFMA with the SAME operand maps and k order as k_gemm1x1 (weights of the channel
// group resident in LDS for the workgroup's whole life, activations streamed from global as float4) -- keeps it in
// LDS inside a zero ring (the depthwise padding), and runs the depthwise filter + SiLU + SE sums from there (tap
// order and roundings of k_dwconv).  The expanded tensor's HBM round trip and the depthwise launch disappear.
// A group = 64 MR pixel rows = one 16 x 16 image (MR = 4), one or two 8 x 8 images, four or eight 4 x 4 images; wave w
// owns MR row tiles of 16.  A workgroup
// walks `groups_per_wg` consecutive groups with its weights staged once.  SE sums: int64 LDS atomics (integer
// adds: order-free), complete per image, so n_tiles = 1.  Outputs are bit-identical to k_gemm1x1 + k_dwconv.
// grid = (ceil(n_groups / groups_per_wg), 1, E / (16 NR)); block = 256.
// dynamic LDS (floats): Kpad * (NT + 4) weights | G * (H+2P) * ((W+2P) | 1) * NT expanded window | KS*KS*NT taps | 2 NT biases
#ifdef PB_SM_STAMP_E
__device__ unsigned long long g_sm_stamp[65536 * 12];  // [workgroup * 4 + wave][slot]
#define PB_ST(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define PB_ST(i) do {} while (0)
#endif
// P3E (round 6): the expand layer is a P3 layer (pb_gemm_p3.h: K >= 80 -- blocks 6-15).  The group's activation operands are held as
// raw floats while the previous group filters (as before) and split into the three bf16 planes at the start of the expand -- the
// CONSUMER-side split (a lane splits its own 8 k per step: 44 vector instructions, once per group and step, used for NR tiles); the
// weights sit in LDS as fragment planes ([k-step of 32][tile][plane][lane] x 16 B: staging is a straight copy of the layer's wt3
// block, a fragment read one conflict-free ds_read_b128), a k-step is p3_step's six bf16 MFMAs per tile.  Same bits as k_gemm_p3.
template <int KS, int S, int NR, int MR, int NS, bool P3E = false>
__global__ __launch_bounds__(256) void k_mbconv_small(
    const float *__restrict__ x, int H, int W, int Cin, const float *__restrict__ wt, int Kpad, int Epad,
    const float *__restrict__ bias_e, const float *__restrict__ dw_w, const float *__restrict__ dw_b, int E,
    float *__restrict__ out, int Ho, int Wo, long long *__restrict__ part, int n_img, int groups_per_wg,
    const u32x4 *__restrict__ wt3 = nullptr, int tiles16 = 0) {
    static_assert(!P3E || NS > 0, "P3E: the operands-in-registers form");
    constexpr int KS32 = (NS + 1) / 2;  // P3E: k-steps of 32 (NS = Kpad / 16 steps of 16)
    constexpr int NT = 16 * NR, LDW = NT + 4, CQ = NT / 4;
    constexpr int PAD = (KS - 1) / 2;
    constexpr int ROWS = 64 * MR;  // pixel rows per group: MR row tiles per wave (the weight fragments of a k-step serve all MR)
    extern __shared__ __attribute__((aligned(16))) float s_ms[];
    __shared__ unsigned long long s_se[2][8][NT];  // [group parity][image of the group][channel]: 2^-24 fixed-point sums (se_acc)
    const int P = H * W;       // 256, 64 or 16
    const int G = ROWS / P;    // images per group (1 .. 8)
    const int Hp = H + 2 * PAD, Wp = (W + 2 * PAD) | 1;  // odd row pitch: see the depthwise phase
    float *s_w = s_ms;                          // [Kpad][LDW]; P3E: [KS32][NR][3][64 lanes] x 16 B
    float *s_e = s_w + (P3E ? KS32 * NR * 768 : Kpad * LDW);  // [G][Hp * Wp][NT]
    float *s_dw = s_e + G * Hp * Wp * NT;       // [KS * KS][NT]
    float *s_b = s_dw + KS * KS * NT;           // [NT] expand bias, [NT] depthwise bias
    const int tid = threadIdx.x;
    int qmax = 0;  // largest converted output seen (se_range_check)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what it indexes stays in SGPRs
    const int li = lane & 15, kq = lane >> 4;
    const int e0 = blockIdx.z * NT;
#ifdef PB_SM_STAMP_E
    unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_t;
#endif
    const int n_groups = (n_img + G - 1) / G;
    const int g_first = blockIdx.x * groups_per_wg;
    const int n_steps = Kpad / 16;
    // this lane's MR pixel rows of the group: row = 16 (MR wave + r) + li -> image g_img of the group, pixel pix
    int g_img[MR], pix[MR];
    float *e_dst[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const int row = 16 * (MR * wave + r) + li;
        g_img[r] = row / P;
        pix[r] = row % P;
        e_dst[r] = s_e + (g_img[r] * Hp * Wp + (pix[r] / W + PAD) * Wp + (pix[r] % W + PAD)) * NT;
    }
    // NS > 0 (= Kpad / 16): the lane's activation operands of a whole group (MR rows x NS float4) are loaded in ONE
    // burst and held in registers -- requested for the first group before the weight staging, for every next group
    // right after the current group's last MFMA, so that the round trip runs under the depthwise phase.  The
    // streaming form (NS = 0: one k-step ahead, 8 MFMAs = 256 cycles of cover per load) left the MFMA pipe waiting
    // for L2 at every step: the expand phase alone ran at 45 % of the MFMA rate.  Loaded values are masked (image
    // past the batch, k past Cin) at use, not at load: a select after the load would wait for it on the spot.
    f32x4 a_all[MR][P3E ? 2 * KS32 : (NS > 0 ? NS : 1)];
    auto load_group = [&](int gi2) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const int img = gi2 * G + g_img[r];
            const float *arow = x + ((size_t)(img < n_img ? img : 0) * P + pix[r]) * Cin;
            if constexpr (P3E) {  // lane (li, kq) of a k-step holds k = 32 s + 8 kq .. + 7 (Cin % 8 == 0: an 8-group is inside the row or past it)
#pragma unroll
                for (int s2 = 0; s2 < KS32; ++s2) {
                    const int kb = 32 * s2 + 8 * kq;
                    const int kc = kb < Cin ? kb : 0;
                    a_all[r][2 * s2] = *reinterpret_cast<const f32x4 *>(arow + kc);
                    a_all[r][2 * s2 + 1] = *reinterpret_cast<const f32x4 *>(arow + kc + 4);
                }
            } else {
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const int kb = 16 * s2 + 4 * kq;
                a_all[r][s2] = *reinterpret_cast<const f32x4 *>(arow + (kb < Cin ? kb : 0));
            }
            }
        }
    };
    if constexpr (NS > 0) {
        if (g_first < n_groups) load_group(g_first);
    }
    // ---- once per workgroup: weights of this channel group, taps, biases, zero ring
    // (every global load of the staging is requested before the first LDS store -- the first four weight quads per
    // thread, the taps, the biases: one round trip to L2 instead of one per loop turn and per array)
    {
        constexpr int NTAP4 = KS * KS * (NT / 4);  // <= 300 float4
        constexpr int NTJ = (NTAP4 + 255) / 256;
        if constexpr (P3E) {  // the layer's fragment planes of this channel group: straight 16-byte copies, four in flight per thread
            constexpr int NF = KS32 * NR * 192;  // 16-byte pieces
            const int t0 = e0 / 16;
            for (int i0 = tid; i0 < NF; i0 += 4 * 256) {
                u32x4 f4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + 256 * j, ic = i < NF ? i : 0;
                    const int sstep = ic / (NR * 192), rem = ic % (NR * 192);
                    f4[j] = wt3[((size_t)sstep * tiles16 + t0) * 192 + rem];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + 256 * j;
                    if (i < NF) reinterpret_cast<u32x4 *>(s_w)[i] = f4[j];
                }
            }
        }
        const int n_w4 = P3E ? 0 : Kpad * (NT / 4);
        f32x4 w4[4], t4[NTJ];
        float be = 0.f, bd = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            const int ic = i < n_w4 ? i : 0;
            w4[j] = *reinterpret_cast<const f32x4 *>(wt + (size_t)(ic / (NT / 4)) * Epad + e0 + (ic % (NT / 4)) * 4);
        }
#pragma unroll
        for (int j = 0; j < NTJ; ++j) {
            const int i = tid + 256 * j;
            const int ic = i < NTAP4 ? i : 0;
            t4[j] = *reinterpret_cast<const f32x4 *>(dw_w + (size_t)(ic / (NT / 4)) * E + e0 + (ic % (NT / 4)) * 4);
        }
        if (tid < NT) {
            be = bias_e[e0 + tid];
            bd = dw_b[e0 + tid];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            if (i < n_w4) *reinterpret_cast<f32x4 *>(s_w + (i / (NT / 4)) * LDW + (i % (NT / 4)) * 4) = w4[j];
        }
#pragma unroll
        for (int j = 0; j < NTJ; ++j) {
            const int i = tid + 256 * j;
            if (i < NTAP4) *reinterpret_cast<f32x4 *>(s_dw + (i / (NT / 4)) * NT + (i % (NT / 4)) * 4) = t4[j];
        }
        if (tid < NT) {
            s_b[tid] = be;
            s_b[NT + tid] = bd;
        }
        for (int i0 = tid + 4 * 256; i0 < n_w4; i0 += 4 * 256) {  // weight slices beyond 16 KB (K > 128 or 48 channels)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + 256 * j;
                const int ic = i < n_w4 ? i : 0;
                w4[j] = *reinterpret_cast<const f32x4 *>(wt + (size_t)(ic / (NT / 4)) * Epad + e0 + (ic % (NT / 4)) * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + 256 * j;
                if (i < n_w4) *reinterpret_cast<f32x4 *>(s_w + (i / (NT / 4)) * LDW + (i % (NT / 4)) * 4) = w4[j];
            }
        }
    }
    PB_ST(10);
    for (int i = tid; i < G * Hp * Wp * (NT / 4); i += 256) *reinterpret_cast<f32x4 *>(s_e + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 2 * 8 * NT; i += 256) (&s_se[0][0][0])[i] = 0ull;
    PB_ST(11);
    __syncthreads();
    PB_ST(0);
    for (int gi = g_first; gi < g_first + groups_per_wg && gi < n_groups; ++gi) {
        const int sp = (gi - g_first) & 1;
        // ---- expand: MR x 16 pixel rows x NT channels per wave, K = Cin
        if constexpr (NS > 0) {
            f32x4 acc[MR][NR];
            bool iv[MR];
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                iv[r] = gi * G + g_img[r] < n_img;
#pragma unroll
                for (int c = 0; c < NR; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (P3E) {
                const u32x4 *sw3 = reinterpret_cast<const u32x4 *>(s_w) + lane;
#pragma unroll
                for (int s = 0; s < KS32; ++s) {
                    // this step's fragments (all planes of all NR tiles, one burst), then the split of this step's operands
                    u32x4 wq[NR][3];
#pragma unroll
                    for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                        for (int c = 0; c < NR; ++c) wq[c][pl] = sw3[((s * NR + c) * 3 + pl) * 64];
                    P3Act pa[MR];
#pragma unroll
                    for (int r = 0; r < MR; ++r) {
                        f32x4 a0 = a_all[r][2 * s], a1 = a_all[r][2 * s + 1];
                        if (!(iv[r] && 32 * s + 8 * kq < Cin)) a0 = a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                        pa[r] = p3_split8(a0, a1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#define PB_SM_PASS(WP_, AP_) \
    _Pragma("unroll") for (int r = 0; r < MR; ++r) _Pragma("unroll") for (int c = 0; c < NR; ++c) acc[r][c] = p3_mfma(wq[c][WP_], pa[r].AP_, acc[r][c]);
                    PB_SM_PASS(2, h)
                    PB_SM_PASS(1, m)
                    PB_SM_PASS(1, h)
                    PB_SM_PASS(0, l)
                    PB_SM_PASS(0, m)
                    PB_SM_PASS(0, h)
#undef PB_SM_PASS
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int kbase = 16 * s + 4 * kq;
                f32x4 a[MR];
#pragma unroll
                for (int r = 0; r < MR; ++r) {
                    a[r] = a_all[r][s];
                    if (!(iv[r] && kbase < Cin)) a[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                const float *wbase = s_w + kbase * LDW + li;
                float wv[4][NR];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < NR; ++c) wv[e][c] = wbase[e * LDW + c * 16];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < MR; ++r) {
                        const float av = e == 0 ? a[r].x : (e == 1 ? a[r].y : (e == 2 ? a[r].z : a[r].w));
#pragma unroll
                        for (int c = 0; c < NR; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e][c], av, acc[r][c], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            PB_ST(1);
            if (gi + 1 < g_first + groups_per_wg && gi + 1 < n_groups) load_group(gi + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < MR; ++r)
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    const f32x4 bev = *reinterpret_cast<const f32x4 *>(s_b + 16 * c + 4 * kq);
                    f32x4 v = acc[r][c];
                    v.x = silu_f(v.x + bev.x); v.y = silu_f(v.y + bev.y); v.z = silu_f(v.z + bev.z); v.w = silu_f(v.w + bev.w);
                    *reinterpret_cast<f32x4 *>(e_dst[r] + 16 * c + 4 * kq) = v;
                }
        } else {
            bool iv[MR];
            const float *arow[MR];
            f32x4 a_nxt[MR];
            f32x4 acc[MR][NR];
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                const int img = gi * G + g_img[r];
                iv[r] = img < n_img;
                arow[r] = x + ((size_t)(iv[r] ? img : 0) * P + pix[r]) * Cin;
                a_nxt[r] = *reinterpret_cast<const f32x4 *>(arow[r] + (4 * kq < Cin ? 4 * kq : 0));
#pragma unroll
                for (int c = 0; c < NR; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int s = 0; s < n_steps; ++s) {
                const int kbase = 16 * s + 4 * kq;
                f32x4 a[MR];
#pragma unroll
                for (int r = 0; r < MR; ++r) {
                    a[r] = a_nxt[r];
                    if (!(iv[r] && kbase < Cin)) a[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (s + 1 < n_steps) {
                        const int kn = kbase + 16;
                        a_nxt[r] = *reinterpret_cast<const f32x4 *>(arow[r] + (kn < Cin ? kn : 0));
                    }
                }
                const float *wbase = s_w + kbase * LDW + li;
                float wv[4][NR];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < NR; ++c) wv[e][c] = wbase[e * LDW + c * 16];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < MR; ++r) {
                        const float av = e == 0 ? a[r].x : (e == 1 ? a[r].y : (e == 2 ? a[r].z : a[r].w));
#pragma unroll
                        for (int c = 0; c < NR; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e][c], av, acc[r][c], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            PB_ST(1);
#pragma unroll
            for (int r = 0; r < MR; ++r)
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    const f32x4 bev = *reinterpret_cast<const f32x4 *>(s_b + 16 * c + 4 * kq);
                    f32x4 v = acc[r][c];
                    v.x = silu_f(v.x + bev.x); v.y = silu_f(v.y + bev.y); v.z = silu_f(v.z + bev.z); v.w = silu_f(v.w + bev.w);
                    *reinterpret_cast<f32x4 *>(e_dst[r] + 16 * c + 4 * kq) = v;
                }
        }
        PB_ST(2);
        __syncthreads();
        PB_ST(3);
        // ---- depthwise from LDS: items = (image of the group, pair of adjacent outputs of a row, channel quad) for
        // stride 1, single outputs for stride 2.  Per filter row a thread reads the pair's KS + 1 window values and the
        // KS taps once (16-byte LDS reads) and applies them to both outputs from registers: 11 reads per 10 tap
        // applications (5x5) where one output per thread took 20 -- with fused taps the LDS reads were half of the
        // phase's issue slots.  Wider strips leave waves without work (an 8 x 8 map x 8 channel quads is 256 pairs:
        // one per thread) and ran slower.  Per output the taps are still applied ky-major, kx-minor.  Lanes run
        // over the channel quads first, then over output rows: with the odd row pitch the rows a 16-lane read
        // group touches alternate between the two halves of the 256-byte bank row.
        const int HoWo = Ho * Wo;
        constexpr int TX = S == 1 ? 2 : 1;
        constexpr int NIN = (TX - 1) * S + KS;
        const int lg_ho = 31 - __builtin_clz(Ho);
        const int n_sx = Wo / TX, lg_sx = 31 - __builtin_clz(n_sx);  // Ho, Wo: powers of two (small_eligible)
        for (int it = tid; it < G * n_sx * Ho * CQ; it += 256) {
            const int cq = it % CQ, po = it / CQ, oy = po & (Ho - 1), sx = (po >> lg_ho) & (n_sx - 1), g2 = po >> (lg_ho + lg_sx);
            const int img2 = gi * G + g2;
            if (img2 >= n_img) continue;
            const int ox0 = sx * TX;
            const float *ip = s_e + (g2 * Hp * Wp + (oy * S) * Wp + ox0 * S) * NT + 4 * cq;
            f32x4 acc[TX];
            {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(s_b + NT + 4 * cq);
#pragma unroll
                for (int j = 0; j < TX; ++j) acc[j] = bv;
            }
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                f32x4 v[NIN], wv[KS];
#pragma unroll
                for (int i = 0; i < NIN; ++i) v[i] = *reinterpret_cast<const f32x4 *>(ip + (ky * Wp + i) * NT);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) wv[kx] = *reinterpret_cast<const f32x4 *>(s_dw + (ky * KS + kx) * NT + 4 * cq);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int j = 0; j < TX; ++j) dw_tap(acc[j], v[j * S + kx], wv[kx]);
                // one filter row at a time (hipcc otherwise interleaves the rows and keeps every read live): the pins
                // close the row's arithmetic, the barrier keeps the next row's reads behind it
                if constexpr (TX > 1) {
#pragma unroll
                    for (int j = 0; j < TX; ++j) asm volatile("" : "+v"(acc[j].x), "+v"(acc[j].y), "+v"(acc[j].z), "+v"(acc[j].w));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ll4 q4 = {0, 0, 0, 0};
            float *op = out + ((size_t)img2 * HoWo + oy * Wo + ox0) * E + e0 + 4 * cq;
#pragma unroll
            for (int j = 0; j < TX; ++j) {
                const f32x4 r4 = {silu_f(acc[j].x), silu_f(acc[j].y), silu_f(acc[j].z), silu_f(acc[j].w)};
#if defined(PB_SM_ABL) && (PB_SM_ABL & 1)
                if (r4.x == 12345.678f)  // ablation: no output stores
#endif
                *reinterpret_cast<f32x4 *>(op + (size_t)j * E) = r4;
                se_acc(q4, qmax, r4);
            }
            atomicAdd(&s_se[sp][g2][4 * cq + 0], (unsigned long long)q4.x);
            atomicAdd(&s_se[sp][g2][4 * cq + 1], (unsigned long long)q4.y);
            atomicAdd(&s_se[sp][g2][4 * cq + 2], (unsigned long long)q4.z);
            atomicAdd(&s_se[sp][g2][4 * cq + 3], (unsigned long long)q4.w);
        }
        PB_ST(4);
        __syncthreads();
        PB_ST(5);
        for (int i = tid; i < G * NT; i += 256) {
            const int g2 = i / NT, ch = i % NT;
            const int img2 = gi * G + g2;
            if (img2 < n_img) part[(size_t)img2 * E + e0 + ch] = (long long)s_se[sp][g2][ch];
            s_se[sp][g2][ch] = 0ull;
        }
        // no barrier here: the next group's sums go to the other s_se buffer (this one is added to again two barriers
        // from now), and its expand phase overwrites a window that every wave has finished reading (barrier above)
        PB_ST(6);
#ifdef PB_SM_STAMP_E
        st_[7] += 1;
#endif
    }
    se_range_check(qmax, part - 1);
#ifdef PB_SM_STAMP_E
    if (lane == 0 && E == PB_SM_STAMP_E && KS == PB_SM_STAMP_KS && S == 1) {
        const size_t w_ = ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 4 + wave) & 65535;
        for (int i = 0; i < 8; ++i) g_sm_stamp[w_ * 12 + i] = st_[i];
        g_sm_stamp[w_ * 12 + 10] = st_[10];
        g_sm_stamp[w_ * 12 + 11] = st_[11];
        g_sm_stamp[w_ * 12 + 8] = __builtin_amdgcn_s_memtime() - st_t0;
        g_sm_stamp[w_ * 12 + 9] = 1ull;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_se for a FEW images (round 6): the same gates, bit for bit, from NWG workgroups per image instead of one.  At batch 1 k_se is
// ONE workgroup pulling up to 442 KB of weights through one CU (7-10 us, 16 launches per forward: 109 of the one-image call's
// 380 us); here workgroup w of an image computes the squeeze units [w JW, (w + 1) JW) -- the whole sum of a unit, with k_se's thread
// layout over the channel quads, so the order of every addition is k_se's -- publishes them as 8-byte {value, 1} granules (one
// sc1 store each), collects all SP of its image (one lane per granule polls with sc1 loads: the data-tagged hand-off of the
// MI355X guide) and computes the gates of ITS slice of the channel quads (FC2 over all units in k_se's group order).  A workgroup
// reads 1 / NWG of either weight matrix.  The granules and the arrival counter are left zero by the image's last-arriving
// workgroup (every workgroup takes its number after it has read all granules), so a replayed graph -- whose kernel arguments are
// frozen -- starts clean.  All NWG x n workgroups must be resident together: the host uses this form for n <= 8 images only.
constexpr int SEM_WG = 8;
template <int SP>
__global__ __launch_bounds__(320) void k_se_multi(const long long *__restrict__ part, int n_tiles, int E, float inv_hw,
                                                  const float *__restrict__ w1, const float *__restrict__ b1,
                                                  const float *__restrict__ w2t, const float *__restrict__ b2,
                                                  float *__restrict__ gate, unsigned long long *__restrict__ xchg,
                                                  unsigned *__restrict__ arrive) {
    constexpr int JG = SP < 16 ? SP : 16, G = SP / JG;  // k_se's unit groups (FC2 adds their partial sums in group order)
    constexpr int JW = SP / SEM_WG;                     // units per workgroup
    static_assert(SP % SEM_WG == 0, "whole units per workgroup");
    __shared__ float s_part[5][JW];
    __shared__ float s_s[SP];
    const int w = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_quads = E >> 2, n_waves = (n_quads + 63) >> 6;
    const bool on = tid < n_quads;
    const int c = on ? 4 * tid : 0;
    unsigned long long *xg = xchg + (size_t)b * 64;
    // the weights of both layers are requested FIRST (they depend on nothing): FC1's land under the pooled sums' round trip, FC2's
    // under FC1 and the hand-off
    const int per = (n_quads + SEM_WG - 1) / SEM_WG;
    const int cq2 = w * per + tid;
    const bool on2 = tid < per && cq2 < n_quads;
    const int c2 = on2 ? 4 * cq2 : 0;
    f32x4 wv1[JW], wv2[G][JG];
#pragma unroll
    for (int j = 0; j < JW; ++j) wv1[j] = *reinterpret_cast<const f32x4 *>(w1 + (size_t)(w * JW + j) * E + c);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < JG; ++j) wv2[g][j] = *reinterpret_cast<const f32x4 *>(w2t + (size_t)(g * JG + j) * E + c2);
    f32x4 bv2 = *reinterpret_cast<const f32x4 *>(b2 + c2);
    __builtin_amdgcn_sched_barrier(0);
    // ---- squeeze: mean of this thread's quad (k_se's arithmetic)
    f32x4 m;
    {
        ll4 t = {0, 0, 0, 0};
        const long long *pp = part + (size_t)b * n_tiles * E + c;
        int tl = 0;
        for (; tl + 8 <= n_tiles; tl += 8) {
            ll4 u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = *reinterpret_cast<const ll4 *>(pp + (size_t)(tl + j) * E);
#pragma unroll
            for (int j = 0; j < 8; ++j) se_add(t, u[j]);
        }
        for (; tl < n_tiles; ++tl) se_add(t, *reinterpret_cast<const ll4 *>(pp + (size_t)tl * E));
        const double sc = (1.0 / 16777216.0) * (double)inv_hw;
        m = (f32x4){(float)((double)t.x * sc), (float)((double)t.y * sc), (float)((double)t.z * sc), (float)((double)t.w * sc)};
        if (!on) m = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // ---- FC1 of this workgroup's units: quad product, the wave's xor butterfly, the waves in index order (k_se's order)
    {
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            float a = m.x * wv1[j].x;
            a = a + m.y * wv1[j].y; a = a + m.z * wv1[j].z; a = a + m.w * wv1[j].w;
            for (int off = 32; off >= 1; off >>= 1) a = a + __shfl_xor(a, off);
            if (lane == 0) s_part[wave][j] = a;
        }
    }
    __syncthreads();
    if (tid < JW) {
        const int jj = w * JW + tid;
        float v = s_part[0][tid];
        for (int wv2 = 1; wv2 < n_waves; ++wv2) v = v + s_part[wv2][tid];
        const float sj = silu_f(v + b1[jj]);
        const unsigned long long gnl = (unsigned long long)__float_as_uint(sj) | (1ull << 32);
        __hip_atomic_store(xg + jj, gnl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // one 8-byte write-through store: arrives whole
    }
    // ---- all SP units of the image: one lane per granule polls
    if (tid < SP) {
        unsigned long long gnl;
        while (((gnl = __hip_atomic_load(xg + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) == 0ull) __builtin_amdgcn_s_sleep(1);
        s_s[tid] = __uint_as_float((uint32_t)gnl);
    }
    __syncthreads();
    // every granule of the image has been read by this workgroup: take a number; the last one leaves the exchange zero
    if (tid == 0) {
        const unsigned pos = __hip_atomic_fetch_add(arrive + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pos + 1u == (unsigned)SEM_WG) {
            for (int j = 0; j < SP; ++j) __hip_atomic_store(xg + j, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(arrive + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- FC2 for this workgroup's slice of the channel quads: groups of JG units, partial sums added in group order (k_se's order)
    if (on2) {
        f32x4 v = bv2;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < JG; ++j) {
                const float sj = s_s[g * JG + j];
                acc.x = acc.x + sj * wv2[g][j].x; acc.y = acc.y + sj * wv2[g][j].y;
                acc.z = acc.z + sj * wv2[g][j].z; acc.w = acc.w + sj * wv2[g][j].w;
            }
            v.x = v.x + acc.x; v.y = v.y + acc.y; v.z = v.z + acc.z; v.w = v.w + acc.w;
        }
        const f32x4 r = {sigmoid_f(v.x), sigmoid_f(v.y), sigmoid_f(v.z), sigmoid_f(v.w)};
        *reinterpret_cast<f32x4 *>(gate + (size_t)b * E + c2) = r;
    }
}

// ------------------------------------------------------------------------------------------------
// squeeze-excite gates for one image per block: mean over pixels (exact integer sum of the tile partials),
// FC(E->S)+SiLU, FC(S->E)+sigmoid.  w1: [S][E]; w2t: [S][E] (transposed se_expand); gate: [B][E].
// The kernel is a chain of three dependent global-memory round trips (partials, w1, w2t) and nothing else, so it is
// laid out to make each of them ONE trip: a thread owns a channel QUAD and a group of JG squeeze units
// (blockDim = QP * G with QP = quads rounded up to a wave multiple and G = SP / JG groups), issues its JG 16-byte
// weight loads back to back and only then uses them.  Reductions run in a fixed order (shuffles inside a wave,
// then the waves of a group in index order; FC2's G partial sums in group order): deterministic and independent
// of the batch.  SP = S rounded up to 8/16/32/48; w1, b1 and w2t are zero-padded to SP rows on the host, so no
// load is conditional on a runtime value.
// IMG images per block share every weight load (the kernel moves 2 * SP * E * 4 bytes of weights per block from
// L2: at 512 blocks of one image that is 226 MB for the widest layers and sets the time); each image's arithmetic
// and its order are the same for every IMG, so the gates do not depend on it.
template <int SP, int IMG>
__global__ __launch_bounds__(1024) void k_se(const long long *__restrict__ part, int n_tiles, int E, float inv_hw,
                                             const float *__restrict__ w1, const float *__restrict__ b1,
                                             const float *__restrict__ w2t, const float *__restrict__ b2,
                                             float *__restrict__ gate, int QP, int n_img) {
    constexpr int JG = SP < 16 ? SP : 16;  // squeeze units per thread
    constexpr int G = SP / JG;             // thread groups
    __shared__ float s_part[IMG][16][JG];  // [image][wave][j of the wave's group]
    __shared__ float s_s[IMG][SP];
    __shared__ f32x4 s_t[IMG][G > 1 ? G - 1 : 1][320];  // FC2 partial sums of groups 1.. (QP <= 320)
    const int b0 = blockIdx.x * IMG;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what it indexes stays in SGPRs
    const int g = tid / QP, cq = tid - g * QP;  // QP is a multiple of 64: a wave never straddles two groups
    const int n_quads = E >> 2;
    const bool on = cq < n_quads;
    const int c = on ? 4 * cq : 0;
    const int j0 = g * JG;
    // ---- squeeze: mean of the quad (every group recomputes it: n_tiles 32-byte loads per image)
    f32x4 m[IMG];
#pragma unroll
    for (int i = 0; i < IMG; ++i) {
        const int b = (b0 + i) < n_img ? (b0 + i) : (n_img - 1);  // a padded slot repeats the last image
        ll4 t = {0, 0, 0, 0};  // exact: fixed-point partial sums (see se_acc); integer adds, any order
        {
            // eight tiles per trip, all their loads requested before the first add: one tile per trip made a small batch's
            // SE kernel a chain of up to 32 dependent round trips (14 us at batch 1)
            const long long *pp = part + (size_t)b * n_tiles * E + c;
            int tl = 0;
            for (; tl + 8 <= n_tiles; tl += 8) {
                ll4 u[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) u[j] = *reinterpret_cast<const ll4 *>(pp + (size_t)(tl + j) * E);
#pragma unroll
                for (int j = 0; j < 8; ++j) se_add(t, u[j]);
            }
            for (; tl < n_tiles; ++tl) se_add(t, *reinterpret_cast<const ll4 *>(pp + (size_t)tl * E));
        }
        const double sc = (1.0 / 16777216.0) * (double)inv_hw;
        m[i] = (f32x4){(float)((double)t.x * sc), (float)((double)t.y * sc), (float)((double)t.z * sc), (float)((double)t.w * sc)};
        if (!on) m[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // ---- FC1 partial sums over this thread's quad for its JG units
    float p[IMG][JG];
    {
        f32x4 wv[JG];
#pragma unroll
        for (int j = 0; j < JG; ++j) wv[j] = *reinterpret_cast<const f32x4 *>(w1 + (size_t)(j0 + j) * E + c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < IMG; ++i)
#pragma unroll
            for (int j = 0; j < JG; ++j) {
                float a = m[i].x * wv[j].x;
                a = a + m[i].y * wv[j].y; a = a + m[i].z * wv[j].z; a = a + m[i].w * wv[j].w;
                p[i][j] = a;
            }
    }
#pragma unroll
    for (int i = 0; i < IMG; ++i)
#pragma unroll
        for (int j = 0; j < JG; ++j) {
            float v = p[i][j];
            for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
            if (lane == 0) s_part[i][wave][j] = v;
        }
    __syncthreads();
    if (tid < SP * IMG) {
        const int i = tid / SP, jj = tid - i * SP;
        const int gg = jj / JG, j = jj - gg * JG;
        const int w0 = gg * (QP >> 6), w1n = w0 + (QP >> 6);  // the waves of group gg, in order
        float v = s_part[i][w0][j];
        for (int w = w0 + 1; w < w1n; ++w) v = v + s_part[i][w][j];
        s_s[i][jj] = silu_f(v + b1[jj]);  // padded rows: silu(0 + 0) = 0
    }
    __syncthreads();
    // ---- FC2: partial sum of this thread's JG units for its quad; group 0 adds the others in order
    f32x4 acc[IMG];
    {
        f32x4 wv[JG];
#pragma unroll
        for (int j = 0; j < JG; ++j) wv[j] = *reinterpret_cast<const f32x4 *>(w2t + (size_t)(j0 + j) * E + c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < IMG; ++i) {
            acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < JG; ++j) {
                const float sj = s_s[i][j0 + j];
                acc[i].x = acc[i].x + sj * wv[j].x; acc[i].y = acc[i].y + sj * wv[j].y;
                acc[i].z = acc[i].z + sj * wv[j].z; acc[i].w = acc[i].w + sj * wv[j].w;
            }
        }
    }
    if constexpr (G > 1) {
        if (g > 0) {
#pragma unroll
            for (int i = 0; i < IMG; ++i) s_t[i][g - 1][cq] = acc[i];
        }
        __syncthreads();
    }
    if (g == 0 && on) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(b2 + c);
#pragma unroll
        for (int i = 0; i < IMG; ++i) {
            if (b0 + i >= n_img) break;
            f32x4 v = bv;
            v.x = v.x + acc[i].x; v.y = v.y + acc[i].y; v.z = v.z + acc[i].z; v.w = v.w + acc[i].w;
            if constexpr (G > 1) {
#pragma unroll
                for (int gg = 1; gg < G; ++gg) {
                    const f32x4 o = s_t[i][gg - 1][cq];
                    v.x = v.x + o.x; v.y = v.y + o.y; v.z = v.z + o.z; v.w = v.w + o.w;
                }
            }
            const f32x4 r = {sigmoid_f(v.x), sigmoid_f(v.y), sigmoid_f(v.z), sigmoid_f(v.w)};
            *reinterpret_cast<f32x4 *>(gate + (size_t)(b0 + i) * E + c) = r;
        }
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

// ------------------------------------------------------------------------------------------------
// Pre-processing of an arbitrary-size RGB8 image (efficientnet.rs:20): `resize_to_fill(W, H, Triangle)` of the
// image crate 0.25.x -- scale to cover, separable triangle filter (vertical pass into f32, horizontal pass back to
// u8), centre crop.  Restated from the crate's published algorithm (src/imageops/sample.rs vertical_sample /
// horizontal_sample, src/math/utils.rs resize_dimensions); the tests hold a CPU restatement of the same text and the
// two agree bit for bit (same f32 operations in the same order).  Only the rows / columns that survive the crop
// are computed.
__device__ __forceinline__ float triangle_kernel(float x) {
    const float a = fabsf(x);
    return a < 1.0f ? 1.0f - a : 0.0f;
}
// tap window of output sample o when in_size samples become out_size: [left, right), centre `input`, scale sratio
__device__ __forceinline__ void resize_window(uint32_t o, uint32_t in_size, uint32_t out_size, int &left, int &right,
                                              float &input, float &sratio) {
    const float ratio = (float)in_size / (float)out_size;
    sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 1.0f * sratio;
    float in0 = ((float)o + 0.5f) * ratio;
    long long l = (long long)floorf(in0 - src_support);
    l = l < 0 ? 0 : (l > (long long)in_size - 1 ? (long long)in_size - 1 : l);
    long long r = (long long)ceilf(in0 + src_support);
    r = r < l + 1 ? l + 1 : (r > (long long)in_size ? (long long)in_size : r);
    left = (int)l;
    right = (int)r;
    input = in0 - 0.5f;
}
// One image of a batch: where its bytes sit in the staged source block, its geometry (src/math/utils.rs resize_dimensions with
// fill = true, src/dynimage.rs resize_to_fill's centre crop: computed on the host), where its vertical-pass rows go, and which
// slot of the network's input batch receives the result.
struct ResizeDesc {
    unsigned long long src_off;  // bytes into the source block: u8 [h][w][3]
    unsigned long long tmp_off;  // floats into the scratch block: f32 [H][w][3] (unused when resample = 0)
    uint32_t w, h, w2, h2;       // source size, size after scaling to cover W x H
    uint32_t cx, cy;             // crop origin in the scaled image
    uint32_t resample;           // 0: the source already has the scaled size (imageops::resize copies): crop only
    uint32_t slot;               // image index in the destination batch
};
// vertical pass of a whole batch: src u8 [h][w][3] -> tmp f32 [H][w][3], output rows cy .. cy + H - 1 of the h2-row image.
// grid = (ceil(max w / 256), H, images); same arithmetic, in the same order, for an image whatever batch it is part of.
__global__ __launch_bounds__(256) void k_resize_v(const uint8_t *__restrict__ src_base, const ResizeDesc *__restrict__ desc,
                                                  float *__restrict__ tmp_base) {
    const ResizeDesc d = desc[blockIdx.z];
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t row = blockIdx.y;
    if (!d.resample || x >= d.w) return;
    const uint8_t *src = src_base + d.src_off;
    int left, right;
    float input, sratio;
    resize_window(d.cy + row, d.h, d.h2, left, right, input, sratio);
    float sum = 0.0f;
    for (int i = left; i < right; ++i) sum = sum + triangle_kernel(((float)i - input) / sratio);
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    for (int i = left; i < right; ++i) {
        const float wgt = triangle_kernel(((float)i - input) / sratio) / sum;
        const uint8_t *p = src + ((size_t)i * d.w + x) * 3;
        const float m0 = (float)p[0] * wgt, m1 = (float)p[1] * wgt, m2 = (float)p[2] * wgt;
        t0 = t0 + m0; t1 = t1 + m1; t2 = t2 + m2;
    }
    float *o = tmp_base + d.tmp_off + ((size_t)row * d.w + x) * 3;
    o[0] = t0; o[1] = t1; o[2] = t2;
}
// horizontal pass + crop of a whole batch: tmp f32 [H][w][3] -> dst u8 [slot][H][W][3], output columns cx .. cx + W - 1 of the
// w2-column image; an image that needs no resampling is cropped straight from its source.  grid = (ceil(W H / 256), images).
__global__ __launch_bounds__(256) void k_resize_h(const uint8_t *__restrict__ src_base, const float *__restrict__ tmp_base,
                                                  const ResizeDesc *__restrict__ desc, uint32_t W, uint32_t H,
                                                  uint8_t *__restrict__ dst_base) {
    const ResizeDesc d = desc[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const uint32_t y = i / W, xo = i % W;
    uint8_t *o = dst_base + ((size_t)d.slot * W * H + i) * 3;
    if (!d.resample) {  // same-size source: crop only
        const uint8_t *p = src_base + d.src_off + ((size_t)(d.cy + y) * d.w + d.cx + xo) * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        return;
    }
    const float *tmp = tmp_base + d.tmp_off;
    int left, right;
    float input, sratio;
    resize_window(d.cx + xo, d.w, d.w2, left, right, input, sratio);
    float sum = 0.0f;
    for (int k = left; k < right; ++k) sum = sum + triangle_kernel(((float)k - input) / sratio);
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    for (int k = left; k < right; ++k) {
        const float wgt = triangle_kernel(((float)k - input) / sratio) / sum;
        const float *p = tmp + ((size_t)y * d.w + k) * 3;
        const float m0 = p[0] * wgt, m1 = p[1] * wgt, m2 = p[2] * wgt;
        t0 = t0 + m0; t1 = t1 + m1; t2 = t2 + m2;
    }
    auto to_u8 = [](float t) -> uint8_t {
        t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
        return (uint8_t)roundf(t);  // f32::round: half away from zero
    };
    o[0] = to_u8(t0); o[1] = to_u8(t1); o[2] = to_u8(t2);
}

// Both passes of one output row in ONE workgroup (round 5): the vertical pass of the source columns the row's horizontal windows
// touch goes into LDS instead of the f32 scratch image (H x w x 3 floats per image written and read back: 400 of the 530 MB the two
// kernels move for 512 images of 256 x 256), the horizontal pass reads it from there.  Same operations in the same order on the same
// values as k_resize_v + k_resize_h (a scratch value is formed by the same loop and read back unchanged), so the same bytes; the
// filter weights of the row are formed once per workgroup instead of once per thread.  The host uses it when the widest span of a
// sub-batch fits the LDS it asks for (s_span floats x 3), else the two-kernel path.  grid = (H, images); block = 256.
__global__ __launch_bounds__(256) void k_resize_fused(const uint8_t *__restrict__ src_base, const ResizeDesc *__restrict__ desc, uint32_t W,
                                                      uint32_t H, uint8_t *__restrict__ dst_base) {
    extern __shared__ __attribute__((aligned(16))) float s_rs[];  // [span][3] vertical sums, then nothing else
    __shared__ float s_wv[64];
    const ResizeDesc d = desc[blockIdx.y];
    const uint32_t y = blockIdx.x;
    uint8_t *orow = dst_base + ((size_t)d.slot * W * H + (size_t)y * W) * 3;
    if (!d.resample) {  // same-size source: crop only
        const uint8_t *p = src_base + d.src_off + ((size_t)(d.cy + y) * d.w + d.cx) * 3;
        for (uint32_t i = threadIdx.x; i < W * 3; i += 256) orow[i] = p[i];
        return;
    }
    const uint8_t *src = src_base + d.src_off;
    // the source columns this row's horizontal windows cover: [xlo, xhi)
    int l0, r0, l1, r1;
    float in0, sr0;
    resize_window(d.cx, d.w, d.w2, l0, r0, in0, sr0);
    resize_window(d.cx + W - 1, d.w, d.w2, l1, r1, in0, sr0);
    const int xlo = l0, xhi = r1;
    // vertical weights of this output row (the same values every thread of k_resize_v forms for itself)
    int left, right;
    float input, sratio;
    resize_window(d.cy + y, d.h, d.h2, left, right, input, sratio);
    const int nv = right - left;
    if (nv <= 64) {
        float sum = 0.0f;
        for (int i = left; i < right; ++i) sum = sum + triangle_kernel(((float)i - input) / sratio);
        if ((int)threadIdx.x < nv) s_wv[threadIdx.x] = triangle_kernel(((float)(left + (int)threadIdx.x) - input) / sratio) / sum;
        __syncthreads();
        for (int x = xlo + (int)threadIdx.x; x < xhi; x += 256) {
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
            for (int i = 0; i < nv; ++i) {
                const float wgt = s_wv[i];
                const uint8_t *p = src + ((size_t)(left + i) * d.w + x) * 3;
                const float m0 = (float)p[0] * wgt, m1 = (float)p[1] * wgt, m2 = (float)p[2] * wgt;
                t0 = t0 + m0; t1 = t1 + m1; t2 = t2 + m2;
            }
            float *o = s_rs + (size_t)(x - xlo) * 3;
            o[0] = t0; o[1] = t1; o[2] = t2;
        }
    } else {  // a very tall source: more taps than the weight table holds -- every thread forms them, as k_resize_v does
        float sum = 0.0f;
        for (int i = left; i < right; ++i) sum = sum + triangle_kernel(((float)i - input) / sratio);
        for (int x = xlo + (int)threadIdx.x; x < xhi; x += 256) {
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
            for (int i = left; i < right; ++i) {
                const float wgt = triangle_kernel(((float)i - input) / sratio) / sum;
                const uint8_t *p = src + ((size_t)i * d.w + x) * 3;
                const float m0 = (float)p[0] * wgt, m1 = (float)p[1] * wgt, m2 = (float)p[2] * wgt;
                t0 = t0 + m0; t1 = t1 + m1; t2 = t2 + m2;
            }
            float *o = s_rs + (size_t)(x - xlo) * 3;
            o[0] = t0; o[1] = t1; o[2] = t2;
        }
    }
    __syncthreads();
    for (uint32_t xo = threadIdx.x; xo < W; xo += 256) {
        int hl, hr;
        float hin, hsr;
        resize_window(d.cx + xo, d.w, d.w2, hl, hr, hin, hsr);
        float sum = 0.0f;
        for (int k = hl; k < hr; ++k) sum = sum + triangle_kernel(((float)k - hin) / hsr);
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
        for (int k = hl; k < hr; ++k) {
            const float wgt = triangle_kernel(((float)k - hin) / hsr) / sum;
            const float *p = s_rs + (size_t)(k - xlo) * 3;
            const float m0 = p[0] * wgt, m1 = p[1] * wgt, m2 = p[2] * wgt;
            t0 = t0 + m0; t1 = t1 + m1; t2 = t2 + m2;
        }
        auto to_u8 = [](float t) -> uint8_t {
            t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
            return (uint8_t)roundf(t);  // f32::round: half away from zero
        };
        uint8_t *o = orow + (size_t)xo * 3;
        o[0] = to_u8(t0); o[1] = to_u8(t1); o[2] = to_u8(t2);
    }
}

// tanh + u8 quantiser (efficientnet.rs:39) on the Linear(1280, D) outputs (computed by k_gemm1x1, bias included)
__global__ void k_tanh_quant(const float *__restrict__ pre, long n, float *__restrict__ out_f32,
                             uint8_t *__restrict__ out_u8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float y = tanhf(pre[i]);
    if (out_f32) out_f32[i] = y;
    out_u8[i] = quantize_u8(y);
}

}  // namespace pbe
