#!/usr/bin/env python3
"""Batch-512 forward with the network's front (stem .. block PB_FRONT_BLOCKS - 1) run over sub-batches of S images
(PB_OPT_EMBED_FRONT_SUB): time per S, and the bytes against the one-pass form.  One process per PB_FRONT_BLOCKS value
(the split point is fixed at create)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pixelbox_amd import capi, synth, weights as W

batch = 512
blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
emb = capi.Embedder(blob, max_batch=batch)
imgs = torch.randint(0, 256, (batch, 128, 128, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((batch, 256), dtype=torch.uint8, device="cuda")
outf = torch.empty((batch, 256), dtype=torch.float32, device="cuda")
ref = None
subs = [int(x) for x in os.environ.get("PB_PROBE_SUBS", "0,64,128,256,0,128").split(",")]
for sub in subs:
    emb.set_option(capi.PB_OPT_EMBED_FRONT_SUB, sub)
    for _ in range(4):
        emb.embed_device(imgs.data_ptr(), batch, out.data_ptr(), outf.data_ptr())
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(10):
            emb.embed_device(imgs.data_ptr(), batch, out.data_ptr(), outf.data_ptr())
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    o = outf.cpu().numpy().copy()
    if ref is None:
        ref = o
    same = bool(np.array_equal(ref.view(np.uint32), o.view(np.uint32)))
    print(f"front_blocks {os.environ.get('PB_FRONT_BLOCKS', 'default')} sub {sub:4d}: {best:.4f} ms / 512   bits equal to first: {same}", flush=True)
