/*
 * pb_oracle_effnet.c -- CPU ORACLE for the embed half of the hot path
 * (TEST INFRASTRUCTURE; see pb_oracle.h).
 *
 * Restates, in naive f32 C, what `MODEL.run` computes in the reference
 * (src/image_hashes/efficientnet.rs:10-14,34): the network that
 * resources/train.py:30-46 builds and :167-174 exports --
 *   torchvision efficientnet_b0().features -> AdaptiveAvgPool2d(1) -> Flatten
 *   -> Linear(1280, D) -> Tanh,   eval mode, BatchNorm folded into the conv.
 * plus the pre-processing of efficientnet.rs:19-29 (px as f32 / 255.0, RGB) and the
 * quantiser of efficientnet.rs:39 (pbo_quantize in pb_oracle.c).
 *
 * PARITY UNPINNED: the arithmetic in the reference is done by tract-onnx 0.22.0
 * (Cargo.toml:25), which is not under /root/reference, whose f32 accumulation order is
 * unspecified, and whose ONNX weights are git-ignored (.gitignore:6).  The reference
 * pins no embedding value (efficientnet.rs:54-67 checks determinism only).  This file is
 * validated against an independent f32 implementation (torch CPU conv2d, hand-assembled
 * B0; tests/golden/gen_golden.py) to 1e-5, which is the bar the HIP path is held to.
 *
 * Accumulation order chosen here (documented, arbitrary): acc = bias; then taps in
 * (ky, kx, ci) order for dense convs, (ky, kx) for depthwise, ci ascending for 1x1 / FC;
 * separate f32 multiply and add (no FMA).  The network itself is written once, over the scalar
 * type, in pb_oracle_effnet_body.h; this file instantiates it in f32 (the oracle) and
 * pb_oracle_effnet_f64.c in f64 (the third point under the embed bar).
 *
 * Activations are NHWC f32.  Weight blob layout: see pixelbox_amd/weights.py (PBXW0001).
 */
#include "pb_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define R(x) x##f
#define REXP expf
#define RTANH tanhf
#define FN(name) name##_f32
#define PBO_FORWARD pbo_effnet_forward
#include "pb_oracle_effnet_body.h"
#undef REAL
#undef R
#undef REXP
#undef RTANH
#undef FN
#undef PBO_FORWARD

/* mlhash over a batch (efficientnet.rs:31-42), batch-1 per call on `nthreads` threads
 * (PARALLEL_FILE_PROCESSORS = 4, engine.rs:22). out_u8[n][D]; out_f32 may be NULL. */
typedef struct {
    const uint8_t *blob;
    size_t blob_len;
    const uint8_t *imgs;
    size_t img_bytes;
    int D;
    size_t n, tid, nthreads;
    uint8_t *out_u8;
    float *out_f32;
    int rc;
} job_t;

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    float *f = (float *)malloc(sizeof(float) * j->D);
    for (size_t i = j->tid; i < j->n; i += j->nthreads) {
        int rc = pbo_effnet_forward(j->blob, j->blob_len, j->imgs + i * j->img_bytes, f);
        if (rc) {
            j->rc = rc;
            break;
        }
        if (j->out_f32) memcpy(j->out_f32 + i * j->D, f, sizeof(float) * j->D);
        pbo_quantize(f, j->D, j->out_u8 + i * j->D);
    }
    free(f);
    return NULL;
}

int pbo_mlhash_batch(const uint8_t *blob, size_t blob_len, const uint8_t *imgs, size_t n,
                     int nthreads, uint8_t *out_u8, float *out_f32) {
    if (blob_len < PBXW_HEADER_BYTES) return -1;
    uint32_t hdr[4];
    memcpy(hdr, blob + 8, 16);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t th[64];
    job_t jobs[64];
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (job_t){blob, blob_len, imgs, (size_t)hdr[0] * hdr[1] * 3, (int)hdr[2], n,
                          (size_t)t, (size_t)nthreads, out_u8, out_f32, 0};
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    int rc = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    return rc;
}
