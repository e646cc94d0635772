/*
 * pb_oracle_effnet_f64.c -- the embed network of pb_oracle_effnet.c evaluated in f64
 * (TEST INFRASTRUCTURE; see pb_oracle.h).
 *
 * Same network, same weights (the blob's f32 values, widened exactly), same u8 pixels (px / 255.0 in f64),
 * every product, sum, exp and tanh in double.  It is the value of the function `MODEL.run` evaluates
 * (src/image_hashes/efficientnet.rs:31-42) to ~1e-15, i.e. the point both f32 evaluations -- the oracle's naive
 * loops and the HIP path's matrix-core sums -- are approximations OF.  tests/test_embed_gpu.py holds the HIP path
 * to `max|hip - f64| <= 1.5 max|oracle_f32 - f64|` per image, saturated images included;
 * bench.py reports the three distances outside the timed region.
 */
#include "pb_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define R(x) x
#define REXP exp
#define RTANH tanh
#define FN(name) name##_f64
#define PBO_FORWARD pbo_effnet_forward_f64
#include "pb_oracle_effnet_body.h"

typedef struct {
    const uint8_t *blob;
    size_t blob_len;
    const uint8_t *imgs;
    size_t img_bytes;
    int D;
    size_t n, tid, nthreads;
    double *out;
    int rc;
} job64_t;

static void *worker64(void *arg) {
    job64_t *j = (job64_t *)arg;
    for (size_t i = j->tid; i < j->n; i += j->nthreads) {
        int rc = pbo_effnet_forward_f64(j->blob, j->blob_len, j->imgs + i * j->img_bytes, j->out + i * j->D);
        if (rc) {
            j->rc = rc;
            break;
        }
    }
    return NULL;
}

/* n images -> out_f64[n][D] (tanh outputs, before the quantiser), batch-1 per call on nthreads threads */
int pbo_effnet_batch_f64(const uint8_t *blob, size_t blob_len, const uint8_t *imgs, size_t n, int nthreads, double *out_f64) {
    if (blob_len < PBXW_HEADER_BYTES) return -1;
    uint32_t hdr[4];
    memcpy(hdr, blob + 8, 16);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t th[64];
    job64_t jobs[64];
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (job64_t){blob, blob_len, imgs, (size_t)hdr[0] * hdr[1] * 3, (int)hdr[2], n, (size_t)t, (size_t)nthreads, out_f64, 0};
        pthread_create(&th[t], NULL, worker64, &jobs[t]);
    }
    int rc = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    return rc;
}
