/*
 * pb_oracle.c -- CPU ORACLE (test infrastructure; see pb_oracle.h).
 * Plain-C restatement of the reference's scan arithmetic, quantiser and query
 * semantics.  Strict IEEE f32, left-to-right folds, no FMA contraction.
 */
#include "pb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- engine.rs:576 ------------------------------------------------------ */
static float g_lut[256];
static int g_lut_ready = 0;

static void lut_init(void) {
    for (int v = 0; v < 256; ++v) {
        volatile float t = (float)v / 255.0f; /* `*v as f32 / 255.0` */
        volatile float t2 = t * 2.0f;        /* `* 2.0`  (exact)    */
        g_lut[v] = t2 - 1.0f;                /* `- 1.0`             */
    }
    g_lut_ready = 1;
}

void pbo_dequant_lut(float lut[256]) {
    if (!g_lut_ready) lut_init();
    memcpy(lut, g_lut, sizeof(g_lut));
}

/* engine.rs:580-581  iter().fold(0f32, |initial, x| initial + x*x) */
static float fold_sq(const uint8_t *a, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float x = g_lut[a[i]];
        float p = x * x;
        acc = acc + p;
    }
    return acc;
}

/* engine.rs:585  zip().fold(0f32, |initial, (a, b)| initial + (a*b)) -- zip stops at min len */
static float fold_dot(const uint8_t *a, size_t na, const uint8_t *b, size_t nb) {
    size_t n = na < nb ? na : nb;
    float acc = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = g_lut[a[i]] * g_lut[b[i]];
        acc = acc + p;
    }
    return acc;
}

/* engine.rs:581-586: returns 0 via *degenerate when magnitude < 1e-6 */
static float cos_sim(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, int *degenerate) {
    if (!g_lut_ready) lut_init();
    float sa = sqrtf(fold_sq(a, na));
    float sb = sqrtf(fold_sq(b, nb));
    float magnitude = sa * sb;
    *degenerate = 0;
    if (magnitude < 1e-6f) { /* engine.rs:582 */
        *degenerate = 1;
        return 0.0f;
    }
    float dot = fold_dot(a, na, b, nb);
    return dot / magnitude; /* engine.rs:586 */
}

float pbo_cosine_similarity(const uint8_t *a, size_t na, const uint8_t *b, size_t nb) {
    int deg;
    return cos_sim(a, na, b, nb, &deg);
}

/* engine.rs:572-588 */
float pbo_cosine_distance(const uint8_t *a, size_t na, const uint8_t *b, size_t nb) {
    int deg;
    float cs = cos_sim(a, na, b, nb, &deg);
    if (deg) return 0.0f; /* engine.rs:583 */
    /* f32::max ignores NaN: cs.max(1e-6) */
    float m = (cs > 1e-6f) ? cs : 1e-6f;
    if (cs != cs) m = 1e-6f;
    float r = 1.0f / m;
    return r - 1.0f; /* engine.rs:587 */
}

/* engine.rs:590-592 */
float pbo_byte_distance(const uint8_t *a, size_t na, const uint8_t *b, size_t nb) {
    size_t n = na < nb ? na : nb;
    float acc = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float d = fabsf((float)a[i] - (float)b[i]);
        acc = acc + d;
    }
    return acc / (255.0f * (float)na);
}

/* engine.rs:594-604 */
float pbo_hamming_distance(const uint8_t *a, size_t na, const uint8_t *b, size_t nb) {
    size_t n = na < nb ? na : nb;
    uint8_t sum = 0; /* `.sum::<u8>()` -- wraps in a release build */
    for (size_t i = 0; i < n; ++i) {
        uint8_t diff = a[i] ^ b[i];
        uint8_t bits = 0;
        while (diff != 0) {
            bits += diff & 1;
            diff >>= 1;
        }
        sum = (uint8_t)(sum + bits);
    }
    return (float)sum / (8.0f * (float)na);
}

/* ---- efficientnet.rs:39 -------------------------------------------------- */
uint8_t pbo_quantize1(float f) {
    float t = f * 128.0f;
    /* f32::max / f32::min return the non-NaN operand */
    t = (t != t) ? -128.0f : (t > -128.0f ? t : -128.0f);
    t = (t < 128.0f) ? t : 128.0f;
    /* `as i8`: truncate toward zero, saturating */
    int i;
    if (t >= 127.0f) i = 127;
    else if (t <= -128.0f) i = -128;
    else i = (int)t;
    /* 128u8.saturating_add_signed(i) */
    int u = 128 + i;
    if (u < 0) u = 0;
    if (u > 255) u = 255;
    return (uint8_t)u;
}

void pbo_quantize(const float *f, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) out[i] = pbo_quantize1(f[i]);
}

/* ---- engine.rs:375-390 ---------------------------------------------------- */
typedef struct {
    float dist;
    int64_t id;
} pbo_hit;

static int hit_less(const pbo_hit *x, const pbo_hit *y) {
    if (x->dist < y->dist) return 1;
    if (x->dist > y->dist) return 0;
    return x->id < y->id;
}

void pbo_scan_all(const uint8_t *query, const uint8_t *rows, size_t n, size_t d, float *out_dist) {
    for (size_t r = 0; r < n; ++r) out_dist[r] = pbo_cosine_distance(query, d, rows + r * d, d);
}

size_t pbo_scan_topk(const uint8_t *query, const uint8_t *rows, const int64_t *ids,
                     size_t n, size_t d, size_t k, double max_dist,
                     int64_t *out_ids, float *out_dist) {
    if (k == 0) return 0;
    pbo_hit *best = (pbo_hit *)malloc(sizeof(pbo_hit) * k);
    size_t cnt = 0;
    for (size_t r = 0; r < n; ++r) {
        float dist = pbo_cosine_distance(query, d, rows + r * d, d);
        if (!((double)dist < max_dist)) continue; /* WHERE dist < ?  (f64) */
        pbo_hit h;
        h.dist = dist;
        h.id = ids ? ids[r] : (int64_t)r;
        if (cnt == k && !hit_less(&h, &best[k - 1])) continue;
        /* insertion into the sorted prefix: ORDER BY dist ASC (ties: id asc) LIMIT k */
        size_t pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && hit_less(&h, &best[pos - 1])) {
            if (pos < k) best[pos] = best[pos - 1];
            --pos;
        }
        best[pos] = h;
        if (cnt < k) ++cnt;
    }
    for (size_t i = 0; i < cnt; ++i) {
        out_ids[i] = best[i].id;
        out_dist[i] = best[i].dist;
    }
    free(best);
    return cnt;
}

/* the same query shape with byte_distance / hamming_distance as the UDF (engine.rs:590-604, 624-663) */
size_t pbo_scan_topk_metric(int metric, const uint8_t *query, const uint8_t *rows, const int64_t *ids, size_t n, size_t d,
                            size_t k, double max_dist, int64_t *out_ids, float *out_dist) {
    if (metric == 0) return pbo_scan_topk(query, rows, ids, n, d, k, max_dist, out_ids, out_dist);
    if (k == 0) return 0;
    pbo_hit *best = (pbo_hit *)malloc(sizeof(pbo_hit) * k);
    size_t cnt = 0;
    for (size_t r = 0; r < n; ++r) {
        float dist = metric == 1 ? pbo_byte_distance(query, d, rows + r * d, d) : pbo_hamming_distance(query, d, rows + r * d, d);
        if (!((double)dist < max_dist)) continue;
        pbo_hit h;
        h.dist = dist;
        h.id = ids ? ids[r] : (int64_t)r;
        if (cnt == k && !hit_less(&h, &best[k - 1])) continue;
        size_t pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && hit_less(&h, &best[pos - 1])) {
            if (pos < k) best[pos] = best[pos - 1];
            --pos;
        }
        best[pos] = h;
        if (cnt < k) ++cnt;
    }
    for (size_t i = 0; i < cnt; ++i) {
        out_ids[i] = best[i].id;
        out_dist[i] = best[i].dist;
    }
    free(best);
    return cnt;
}

/* ---- synthetic data -------------------------------------------------------- */
uint64_t pbo_splitmix64_at(uint64_t seed, uint64_t word_index) {
    uint64_t z = seed + (word_index + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void pbo_fill_synthetic(uint64_t seed, uint64_t byte_offset, size_t nbytes, uint8_t *out) {
    size_t j = 0;
    while (j < nbytes) {
        uint64_t g = byte_offset + j;
        uint64_t z = pbo_splitmix64_at(seed, g >> 3);
        unsigned b = (unsigned)(g & 7);
        for (; b < 8 && j < nbytes; ++b, ++j) out[j] = (uint8_t)(z >> (8 * b));
    }
}
