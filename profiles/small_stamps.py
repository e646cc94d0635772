#!/usr/bin/env python3
"""Phase stamps of k_mbconv_small (library built with -DPB_SM_STAMP_E=<E> -DPB_SM_STAMP_KS=<k>): cycles per wave and
group spent in each phase of the layer with that expanded width, from one steady-state batch-512 forward."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pixelbox_amd import capi, synth, weights as W

batch = 512
blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
emb = capi.Embedder(blob, max_batch=batch)
imgs = torch.randint(0, 256, (batch, 128, 128, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((batch, 256), dtype=torch.uint8, device="cuda")
for _ in range(3):
    emb.embed_device(imgs.data_ptr(), batch, out.data_ptr())
torch.cuda.synchronize()
L = capi.lib()
buf = (C.c_ulonglong * (65536 * 12))()
rc = L.pb_debug_small_stamps(buf, 1)
print('rc', rc, L.pb_last_error())
emb.embed_device(imgs.data_ptr(), batch, out.data_ptr())
torch.cuda.synchronize()
rc = L.pb_debug_small_stamps(buf, 0)
print('rc', rc, L.pb_last_error())
import numpy as np
v = np.ctypeslib.as_array(buf).reshape(65536, 12).astype(np.float64).sum(axis=0).tolist()
names = ["staging", "expand mfma", "expand epilogue", "barrier 1", "depthwise", "barrier 2", "part + barrier 3"]
waves, groups = v[9], v[7]
print(f"waves {waves}, wave-groups {groups}, lifetime per wave {v[8] / max(waves, 1):.0f} cycles")
print(f"  {'weights -> LDS':18s} {v[10] / max(waves, 1):8.0f} per wave")
print(f"  {'taps, zero ring':18s} {v[11] / max(waves, 1):8.0f} per wave")
print(f"  {'first barrier':18s} {v[0] / max(waves, 1):8.0f} per wave")
for i in range(1, 7):
    print(f"  {names[i]:18s} {v[i] / max(groups, 1):8.0f} per wave and group")
