#!/usr/bin/env python3
"""Runs the batch-512 embed forward a few times (after the per-layer tuning) so a rocprofv3 --kernel-trace of this
script ends with steady-state forwards; profiles/embed_layers.py prints the last one layer by layer."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pixelbox_amd import capi, synth, weights as W

batch = int(os.environ.get("PB_PROBE_BATCH", "512"))
reps = int(os.environ.get("PB_PROBE_REPS", "6"))
blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
emb = capi.Embedder(blob, max_batch=batch)
imgs = torch.randint(0, 256, (batch, 128, 128, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((batch, 256), dtype=torch.uint8, device="cuda")
for _ in range(3):
    emb.embed_device(imgs.data_ptr(), batch, out.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    emb.embed_device(imgs.data_ptr(), batch, out.data_ptr())
torch.cuda.synchronize()
print("ms/batch", (time.perf_counter() - t0) / reps * 1e3)
