#!/usr/bin/env python3
"""Does overlapping two embed forwards on two streams beat one forward of twice the batch?  Two embedders (own stream and
buffers each) driven from two host threads against one embedder with the whole batch."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pixelbox_amd import capi, synth, weights as W

blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
N = int(os.environ.get("N", "512"))
reps = 20
imgs = torch.randint(0, 256, (N, 128, 128, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((N, 256), dtype=torch.uint8, device="cuda")


def run(emb, ptr_in, n, ptr_out, reps):
    for _ in range(reps):
        emb.embed_device(ptr_in, n, ptr_out)


one = capi.Embedder(blob, max_batch=N)
run(one, imgs.data_ptr(), N, out.data_ptr(), 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(one, imgs.data_ptr(), N, out.data_ptr(), reps)
torch.cuda.synchronize()
t_one = (time.perf_counter() - t0) / reps
print(f"one embedder, batch {N}: {t_one * 1e3:.3f} ms per {N} images = {N / t_one:,.0f} img/s")
for parts in (2, 4):
    h = N // parts
    embs = [capi.Embedder(blob, max_batch=h) for _ in range(parts)]
    for i, e in enumerate(embs):
        run(e, imgs[i * h:].data_ptr(), h, out[i * h:].data_ptr(), 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(e, imgs[i * h:].data_ptr(), h, out[i * h:].data_ptr(), reps)) for i, e in enumerate(embs)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    t_par = (time.perf_counter() - t0) / reps
    print(f"{parts} embedders, batch {h} each, concurrently: {t_par * 1e3:.3f} ms per {N} images = {N / t_par:,.0f} img/s")
