#!/usr/bin/env python3
"""bench.py's host-image ingest leg (256 x 256 images, batches of 512 through pb_embed_batch_images_device, an embedder per host thread)
for several thread counts; PB_TRACE_TUNE=3 prints the packing times.  Under rocprofv3 --kernel-trace: profiles/trace_durations.py sums
the kernels."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pixelbox_amd import capi, synth, weights as W

blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
nb = 512
h = w = 256
per = h * w * 3
n = int(os.environ.get("PB_PROBE_IMAGES", "16384"))
pool = synth.fill_synthetic(synth.SEED_IMAGES + 7, 0, 64 * per).reshape(64, h, w, 3)
images = [np.ascontiguousarray(np.roll(pool[i % 64], i // 64, axis=1)) for i in range(n)]
batches = [capi.Embedder.image_batch_args(images[i : i + nb]) for i in range(0, n, nb)]
for NT in [int(x) for x in os.environ.get("PB_PROBE_THREADS", "8,12,16,4").split(",")]:
    embs = [capi.Embedder(blob, max_batch=nb) for _ in range(NT)]
    for t in range(NT):
        embs[t].embed_images_device(batches[t % len(batches)])
    torch.cuda.synchronize()

    do_append = bool(os.environ.get("PB_PROBE_APPEND"))
    index = capi.Index(256, n + 16) if do_append else None
    if do_append and os.environ.get("PB_PROBE_APPEND") == "async":
        index.set_option(capi.PB_OPT_APPEND_ASYNC, 1)
    lock = threading.Lock()
    next_id = [1]

    def worker(t):
        for bi in range(t, len(batches), NT):
            d_ptr = embs[t].embed_images_device(batches[bi])
            if do_append:
                cnt = batches[bi][3]
                with lock:
                    ids = np.arange(next_id[0], next_id[0] + cnt, dtype=np.int64)
                    next_id[0] += cnt
                    index.append_device(ids, d_ptr)

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(NT)]
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{NT:2d} threads: {n / dt:9.0f} images/s ({dt / len(batches) * 1e3:.3f} ms per batch of {nb} in aggregate)", flush=True)
    del embs
