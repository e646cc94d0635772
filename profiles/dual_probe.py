#!/usr/bin/env python3
"""Forward pass as ONE chain of launches against TWO half-batches side by side (PB_OPT_EMBED_DUAL): ms per batch at several
batch sizes, and the bytes of the two forms against each other."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pixelbox_amd import capi, synth, weights as W

blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
for batch in (512, 256, 128, 64, 1024):
    emb = capi.Embedder(blob, max_batch=batch)
    imgs = torch.randint(0, 256, (batch, 128, 128, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty((batch, 256), dtype=torch.uint8, device="cuda")
    outf = torch.empty((batch, 256), dtype=torch.float32, device="cuda")
    ref = None
    for dual in (0, 2, 0, 2):
        emb.set_option(capi.PB_OPT_EMBED_DUAL, dual)
        for _ in range(5):
            emb.embed_device(imgs.data_ptr(), batch, out.data_ptr(), outf.data_ptr())
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                emb.embed_device(imgs.data_ptr(), batch, out.data_ptr(), outf.data_ptr())
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
        o = outf.cpu().numpy().copy()
        if ref is None:
            ref = o
        print(f"batch {batch:5d} dual {dual}: {best:.4f} ms / batch = {batch / best:.1f} k img/s   bits equal to the first: {bool(np.array_equal(ref.view(np.uint32), o.view(np.uint32)))}", flush=True)
