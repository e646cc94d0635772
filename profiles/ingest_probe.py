#!/usr/bin/env python3
"""One thread, one embedder: ms per pb_embed_batch_images_device call of 256 host images, per image size, against the
same batch through pb_embed_batch (already 128 x 128: no resize) and the plain memcpy rate of this host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pixelbox_amd import capi, synth, weights as W

blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
nb = int(os.environ.get("PB_PROBE_BATCH", "256"))
emb = capi.Embedder(blob, max_batch=nb)
for (h, w) in ((128, 128), (256, 256), (480, 640)):
    per = h * w * 3
    pool = synth.fill_synthetic(synth.SEED_IMAGES + 7, 0, 32 * per).reshape(32, h, w, 3)
    images = [np.ascontiguousarray(np.roll(pool[i % 32], i // 32, axis=1)) for i in range(nb)]
    args = capi.Embedder.image_batch_args(images)
    for _ in range(2):
        emb.embed_images_device(args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 6
    for _ in range(reps):
        emb.embed_images_device(args)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    big = np.concatenate([im.reshape(-1) for im in images])
    dst = np.empty_like(big)
    t0 = time.perf_counter()
    np.copyto(dst, big)
    cp = (time.perf_counter() - t0) * 1e3
    print(f"{w}x{h}: {ms:.3f} ms per call of {nb} images = {nb / ms * 1e3:.0f} images/s ({nb * per / 1e6:.0f} MB per call; one-thread memcpy of that: {cp:.2f} ms)")
imgs = synth.fill_synthetic(synth.SEED_IMAGES, 0, nb * 128 * 128 * 3).reshape(nb, 128, 128, 3)
emb.embed(imgs, want_f32=False)
t0 = time.perf_counter()
for _ in range(6):
    emb.embed(imgs, want_f32=False)
print(f"pb_embed_batch of {nb} 128x128 images (host buffers, no resize): {(time.perf_counter() - t0) * 1e3 / 6:.3f} ms per call")

# ---- several host threads, an embedder each (the bench leg's shape): aggregate rate per thread count
import threading

d_imgs = torch.randint(0, 256, (nb, 128, 128, 3), dtype=torch.uint8, device="cuda")
d_out = torch.empty((nb, 256), dtype=torch.uint8, device="cuda")
emb.embed_device(d_imgs.data_ptr(), nb, d_out.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    emb.embed_device(d_imgs.data_ptr(), nb, d_out.data_ptr())
torch.cuda.synchronize()
print(f"forward alone, {nb} device-resident images: {(time.perf_counter() - t0) * 100:.3f} ms")
h, w = 256, 256
per = h * w * 3
pool = synth.fill_synthetic(synth.SEED_IMAGES + 7, 0, 32 * per).reshape(32, h, w, 3)
for nt in (1, 2, 4, 8):
    embs = [capi.Embedder(blob, max_batch=nb) for _ in range(nt)]
    argsl = []
    for t in range(nt):
        images = [np.ascontiguousarray(np.roll(pool[i % 32], i // 32 + 7 * t, axis=1)) for i in range(nb)]
        argsl.append(capi.Embedder.image_batch_args(images))
    for t in range(nt):
        embs[t].embed_images_device(argsl[t])
        embs[t].embed_images_device(argsl[t])
    reps = 12

    def work(t):
        for _ in range(reps):
            embs[t].embed_images_device(argsl[t])

    th = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nt} thread(s), {w}x{h}: {nt * reps * nb / dt:.0f} images/s ({dt / (nt * reps) * 1e3:.3f} ms per batch of {nb} in aggregate)")
    del embs, argsl
