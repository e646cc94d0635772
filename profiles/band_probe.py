#!/usr/bin/env python3
"""A/B of the embed forward with and without the LDS-ring front kernel (k_front_band / PB_NO_BAND), same process:
bit comparison of the float outputs at several batch sizes, then interleaved timing of batch-512 forwards
(`PB_TRACE_TUNE=1` in the environment additionally prints every candidate the per-layer timing loops measured)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pixelbox_amd import capi, synth, weights as W

batch = int(os.environ.get("PB_PROBE_BATCH", "512"))
blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)


def make(env):
    for k in ("PB_NO_BAND", "PB_FORCE_BAND", "PB_FOLD_SE", "PB_NO_GEMM_T", "PB_NO_TAIL_FUSION", "PB_NO_BLOCK_FUSION", "PB_NO_GEMM_STREAM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = capi.Embedder(blob, max_batch=batch)
    for k in env:
        os.environ.pop(k, None)
    return e


variants = {"old": {"PB_NO_BAND": "1", "PB_NO_GEMM_T": "1", "PB_NO_TAIL_FUSION": "1", "PB_NO_BLOCK_FUSION": "1", "PB_NO_GEMM_STREAM": "1"}, "no_block_fusion": {"PB_NO_BLOCK_FUSION": "1"}, "no_gemm_stream": {"PB_NO_GEMM_STREAM": "1"}, "tuned": {}}
only = os.environ.get("PB_PROBE_VARIANTS")
if only:
    variants = {k: v for k, v in variants.items() if k in only.split(",")}
embs = {k: make(v) for k, v in variants.items()}
imgs_h = synth.synthetic_images(synth.SEED_IMAGES, 0, batch, 128, 128)
imgs = torch.from_numpy(imgs_h).cuda()
out = torch.empty((batch, 256), dtype=torch.uint8, device="cuda")
outf = torch.empty((batch, 256), dtype=torch.float32, device="cuda")

ref = None
for name, e in embs.items():
    res = {}
    for n in (1, 3, 64, batch):
        e.embed_device(imgs.data_ptr(), n, out.data_ptr(), outf.data_ptr())
        torch.cuda.synchronize()
        res[n] = (outf[:n].cpu().numpy().copy(), out[:n].cpu().numpy().copy())
    # batch invariance inside the variant
    for n in (1, 3, 64):
        assert np.array_equal(res[n][0].view(np.uint32), res[batch][0][:n].view(np.uint32)), (name, n)
    if ref is None:
        ref = res
    else:
        for n in res:
            same = np.array_equal(res[n][0].view(np.uint32), ref[n][0].view(np.uint32))
            print(f"{name} vs {list(embs)[0]} at batch {n}: float bits identical = {same}, max |diff| = "
                  f"{float(np.abs(res[n][0] - ref[n][0]).max()):.3g}, bytes identical = {np.array_equal(res[n][1], ref[n][1])}")
print("batch invariance inside every variant: ok")

times = {k: [] for k in embs}
for rnd in range(6):
    for name, e in embs.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            e.embed_device(imgs.data_ptr(), batch, out.data_ptr())
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / 4 * 1e3)
for name, t in times.items():
    print(f"{name}: ms/batch median {sorted(t)[len(t) // 2]:.4f} min {min(t):.4f}  ({batch / sorted(t)[len(t) // 2] * 1e3:.0f} img/s)")

# batch-1 latency (pb_mlhash's device part): median of 200 synchronous forwards of one image
for name, e in embs.items():
    lat = []
    for i in range(220):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.embed_device(imgs.data_ptr(), 1, out.data_ptr())
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = sorted(lat[20:])
    print(f"{name}: batch-1 forward median {lat[len(lat) // 2]:.4f} ms, min {lat[0]:.4f} ms")
