/*
 * pb_oracle_effnet_body.h -- the network of pb_oracle_effnet.c, written once over the scalar type REAL
 * (TEST INFRASTRUCTURE; see pb_oracle.h).  Included twice:
 *   pb_oracle_effnet.c      REAL = float   the oracle proper (naive f32, separate multiply and add, the order below)
 *   pb_oracle_effnet_f64.c  REAL = double  the same network evaluated in f64 from the same f32 weights and u8 pixels:
 *                                          a THIRD POINT under the 1e-5 embed bar -- how far each f32 evaluation
 *                                          (this oracle's, the HIP path's) sits from the value of the function itself
 * The includer defines REAL, R(x) (a literal of that type), REXP / RTANH (exp / tanh of that type), FN(name) (symbol
 * suffixing) and PBO_FORWARD (the exported entry point's name).  With REAL = float every expression below is, token for
 * token, what pb_oracle_effnet.c held before the f64 mode was added; tests/test_oracle.py pins its outputs to the goldens.
 *
 * Restates what `MODEL.run` computes in the reference (src/image_hashes/efficientnet.rs:10-14,34): the network that
 * resources/train.py:30-46 builds and :167-174 exports -- torchvision efficientnet_b0().features -> AdaptiveAvgPool2d(1)
 * -> Flatten -> Linear(1280, D) -> Tanh, eval mode, BatchNorm folded into the conv -- plus the pre-processing of
 * efficientnet.rs:19-29 (px as f32 / 255.0, RGB).
 *
 * Accumulation order (documented, arbitrary): acc = bias; then taps in (ky, kx, ci) order for dense convs, (ky, kx) for
 * depthwise, ci ascending for 1x1 / FC; separate multiply and add (no FMA).
 */

typedef struct {
    int expand, k, stride, cin, cout, repeats;
} FN(stage_t);

/* torchvision efficientnet_b0 inverted-residual setting */
static const FN(stage_t) FN(STAGES)[7] = {
    {1, 3, 1, 32, 16, 1},  {6, 3, 2, 16, 24, 2},   {6, 5, 2, 24, 40, 2},  {6, 3, 2, 40, 80, 3},
    {6, 5, 1, 80, 112, 3}, {6, 5, 2, 112, 192, 4}, {6, 3, 1, 192, 320, 1},
};

static inline REAL FN(silu)(REAL x) { return x / (R(1.0) + REXP(-x)); }
static inline REAL FN(sigmoid)(REAL x) { return R(1.0) / (R(1.0) + REXP(-x)); }

/* out[p][co] = act(b[co] + sum_ci in[p][ci] * w[co][ci]);  w is OI (torch layout) */
static void FN(conv1x1)(const REAL *in, int npix, int cin, const float *w, const float *b, int cout,
                        REAL *out, int act /*0 none, 1 silu*/) {
    REAL *wt = (REAL *)malloc(sizeof(REAL) * (size_t)cin * cout);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) wt[(size_t)ci * cout + co] = w[(size_t)co * cin + ci];
    for (int p = 0; p < npix; ++p) {
        REAL *o = out + (size_t)p * cout;
        const REAL *x = in + (size_t)p * cin;
        for (int co = 0; co < cout; ++co) o[co] = b[co];
        for (int ci = 0; ci < cin; ++ci) {
            const REAL a = x[ci];
            const REAL *wr = wt + (size_t)ci * cout;
            for (int co = 0; co < cout; ++co) {
                REAL prod = a * wr[co];
                o[co] = o[co] + prod;
            }
        }
        if (act)
            for (int co = 0; co < cout; ++co) o[co] = FN(silu)(o[co]);
    }
    free(wt);
}

/* depthwise kxk, stride s, pad (k-1)/2, + bias + SiLU. w is [C][k][k]. */
static void FN(dwconv)(const REAL *in, int h, int wd, int c, const float *w, const float *b, int k,
                       int s, REAL *out, int ho, int wo) {
    const int pad = (k - 1) / 2;
    for (int y = 0; y < ho; ++y)
        for (int x = 0; x < wo; ++x) {
            REAL *o = out + ((size_t)y * wo + x) * c;
            for (int ch = 0; ch < c; ++ch) o[ch] = b[ch];
            for (int ky = 0; ky < k; ++ky) {
                int iy = y * s + ky - pad;
                if (iy < 0 || iy >= h) continue;
                for (int kx = 0; kx < k; ++kx) {
                    int ix = x * s + kx - pad;
                    if (ix < 0 || ix >= wd) continue;
                    const REAL *ip = in + ((size_t)iy * wd + ix) * c;
                    const float *wp = w + ky * k + kx;
                    for (int ch = 0; ch < c; ++ch) {
                        REAL prod = ip[ch] * wp[(size_t)ch * k * k];
                        o[ch] = o[ch] + prod;
                    }
                }
            }
            for (int ch = 0; ch < c; ++ch) o[ch] = FN(silu)(o[ch]);
        }
}

/* stem: 3x3 s2 p1, 3 -> 32, + bias + SiLU; input u8 HWC; w is [32][3][3][3] (OIHW) */
static void FN(stem)(const uint8_t *img, int h, int wd, const float *w, const float *b, REAL *out,
                     int ho, int wo) {
    for (int y = 0; y < ho; ++y)
        for (int x = 0; x < wo; ++x) {
            REAL *o = out + ((size_t)y * wo + x) * 32;
            for (int co = 0; co < 32; ++co) o[co] = b[co];
            for (int ky = 0; ky < 3; ++ky) {
                int iy = y * 2 + ky - 1;
                if (iy < 0 || iy >= h) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    int ix = x * 2 + kx - 1;
                    if (ix < 0 || ix >= wd) continue;
                    for (int ci = 0; ci < 3; ++ci) {
                        /* efficientnet.rs:27  img[(x,y)][c] as f32 / 255.0 */
                        REAL a = (REAL)img[((size_t)iy * wd + ix) * 3 + ci] / R(255.0);
                        for (int co = 0; co < 32; ++co) {
                            REAL prod = a * w[((co * 3 + ci) * 3 + ky) * 3 + kx];
                            o[co] = o[co] + prod;
                        }
                    }
                }
            }
            for (int co = 0; co < 32; ++co) o[co] = FN(silu)(o[co]);
        }
}

static size_t FN(blob_floats)(int D) {
    size_t n = 32 * 27 + 32;
    for (int s = 0; s < 7; ++s) {
        const FN(stage_t) *st = &FN(STAGES)[s];
        for (int r = 0; r < st->repeats; ++r) {
            int cin = r == 0 ? st->cin : st->cout;
            int e = cin * st->expand;
            int sq = cin / 4 > 1 ? cin / 4 : 1;
            if (st->expand != 1) n += (size_t)e * cin + e;
            n += (size_t)e * st->k * st->k + e;
            n += (size_t)sq * e + sq;
            n += (size_t)e * sq + e;
            n += (size_t)st->cout * e + st->cout;
        }
    }
    n += 1280 * 320 + 1280;
    n += (size_t)D * 1280 + D;
    return n;
}

#ifndef PBXW_HEADER_BYTES
#define PBXW_HEADER_BYTES 32
#endif

/* Returns 0 on success. out[D] = tanh output; img is H*W*3 u8 (RGB, HWC). */
int PBO_FORWARD(const uint8_t *blob, size_t blob_len, const uint8_t *img, REAL *out) {
    if (blob_len < PBXW_HEADER_BYTES || memcmp(blob, "PBXW0001", 8) != 0) return -1;
    uint32_t hdr[4];
    uint64_t nfl;
    memcpy(hdr, blob + 8, 16);
    memcpy(&nfl, blob + 24, 8);
    const int H = (int)hdr[0], W = (int)hdr[1], D = (int)hdr[2];
    if (nfl != FN(blob_floats)(D) || blob_len != PBXW_HEADER_BYTES + nfl * 4) return -2;
    if (H % 32 || W % 32) return -3;
    const float *p = (const float *)(blob + PBXW_HEADER_BYTES);

    int h = H / 2, w = W / 2, c = 32;
    size_t maxact = (size_t)h * w * 96 * 2; /* >= largest expanded activation (stage 2: 96 ch @ H/2) */
    REAL *x = (REAL *)malloc(sizeof(REAL) * maxact);
    REAL *t1 = (REAL *)malloc(sizeof(REAL) * maxact);
    REAL *t2 = (REAL *)malloc(sizeof(REAL) * maxact);
    REAL sebuf[1152 * 2 + 64];

    FN(stem)(img, H, W, p, p + 32 * 27, x, h, w);
    p += 32 * 27 + 32;

    for (int s = 0; s < 7; ++s) {
        const FN(stage_t) *st = &FN(STAGES)[s];
        for (int r = 0; r < st->repeats; ++r) {
            const int cin = r == 0 ? st->cin : st->cout;
            const int stride = r == 0 ? st->stride : 1;
            const int e = cin * st->expand;
            const int sq = cin / 4 > 1 ? cin / 4 : 1;
            const int k = st->k;
            const int ho = (h + stride - 1) / stride, wo = (w + stride - 1) / stride;
            const REAL *ein = x;
            if (st->expand != 1) {
                FN(conv1x1)(x, h * w, cin, p, p + (size_t)e * cin, e, t1, 1);
                p += (size_t)e * cin + e;
                ein = t1;
            }
            FN(dwconv)(ein, h, w, e, p, p + (size_t)e * k * k, k, stride, t2, ho, wo);
            p += (size_t)e * k * k + e;
            /* squeeze-excite: mean over pixels (row-major order), FC+SiLU, FC+sigmoid, scale */
            REAL *mean = sebuf, *sv = sebuf + 1152, *gate = sebuf + 1152 + 64;
            for (int ch = 0; ch < e; ++ch) mean[ch] = R(0.0);
            for (int px = 0; px < ho * wo; ++px)
                for (int ch = 0; ch < e; ++ch) mean[ch] = mean[ch] + t2[(size_t)px * e + ch];
            const REAL inv = R(1.0) / (REAL)(ho * wo);
            for (int ch = 0; ch < e; ++ch) mean[ch] = mean[ch] * inv;
            FN(conv1x1)(mean, 1, e, p, p + (size_t)sq * e, sq, sv, 1);
            p += (size_t)sq * e + sq;
            FN(conv1x1)(sv, 1, sq, p, p + (size_t)e * sq, e, gate, 0);
            p += (size_t)e * sq + e;
            for (int ch = 0; ch < e; ++ch) gate[ch] = FN(sigmoid)(gate[ch]);
            for (int px = 0; px < ho * wo; ++px)
                for (int ch = 0; ch < e; ++ch) t2[(size_t)px * e + ch] = t2[(size_t)px * e + ch] * gate[ch];
            /* project (no activation) + residual */
            FN(conv1x1)(t2, ho * wo, e, p, p + (size_t)st->cout * e, st->cout, t1, 0);
            p += (size_t)st->cout * e + st->cout;
            if (stride == 1 && cin == st->cout) {
                for (size_t i = 0; i < (size_t)ho * wo * st->cout; ++i) t1[i] = x[i] + t1[i];
            }
            REAL *tmp = x;
            x = t1;
            t1 = tmp;
            h = ho;
            w = wo;
            c = st->cout;
        }
    }
    /* head conv 1x1 320 -> 1280 + SiLU, global average pool, Linear + tanh */
    FN(conv1x1)(x, h * w, c, p, p + 1280 * 320, 1280, t1, 1);
    p += 1280 * 320 + 1280;
    REAL *pool = t2;
    for (int ch = 0; ch < 1280; ++ch) pool[ch] = R(0.0);
    for (int px = 0; px < h * w; ++px)
        for (int ch = 0; ch < 1280; ++ch) pool[ch] = pool[ch] + t1[(size_t)px * 1280 + ch];
    const REAL invp = R(1.0) / (REAL)(h * w);
    for (int ch = 0; ch < 1280; ++ch) pool[ch] = pool[ch] * invp;
    FN(conv1x1)(pool, 1, 1280, p, p + (size_t)D * 1280, D, out, 0);
    for (int i = 0; i < D; ++i) out[i] = RTANH(out[i]);
    free(x);
    free(t1);
    free(t2);
    return 0;
}
