#!/usr/bin/env python3
"""Three-way distances of the embedding over the bench's N synthetic images (test infrastructure: uses oracle/):
HIP path, CPU oracle (f32), and the same network evaluated in f64 (oracle/pb_oracle_effnet_f64.c) -- per image
max |hip - f64|, max |oracle - f64|, max |hip - oracle| over the D outputs, saturated images flagged."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import capi as oracle
from pixelbox_amd import capi, synth, weights as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "bench"  # bench: bench.py's images (the raw byte stream); parity: synth.synthetic_images (brightness windows, some saturate)
nthr = min(64, os.cpu_count() or 8)
blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
if kind == "bench":
    imgs = synth.fill_synthetic(synth.SEED_IMAGES, 0, n * 128 * 128 * 3).reshape(n, 128, 128, 3)
else:
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, n, 128, 128)
emb = capi.Embedder(blob, max_batch=min(n, 512))
u8, f = emb.embed(imgs)
ref_u8, ref_f = oracle.mlhash_batch(blob, imgs, 256, nthreads=nthr)
f64 = oracle.effnet_batch_f64(blob, imgs, 256, nthreads=nthr)
e_hip = np.abs(f.astype(np.float64) - f64).max(axis=1)
e_orc = np.abs(ref_f.astype(np.float64) - f64).max(axis=1)
e_ho = np.abs(f - ref_f).max(axis=1)
sat = np.abs(f64).max(axis=1) >= 0.999
ratio = e_hip / np.maximum(e_orc, 1e-30)
print(f"{kind} images {n}, saturated {int(sat.sum())}")
print(f"max |hip - f64|     all {e_hip.max():.3e}  unsaturated {e_hip[~sat].max():.3e}  median {np.median(e_hip):.3e}")
print(f"max |oracle - f64|  all {e_orc.max():.3e}  unsaturated {e_orc[~sat].max():.3e}  median {np.median(e_orc):.3e}")
print(f"max |hip - oracle|  all {e_ho.max():.3e}  unsaturated {e_ho[~sat].max():.3e}")
print(f"per image hip/oracle error ratio: max {ratio.max():.2f}  p99 {np.percentile(ratio, 99):.2f}  median {np.median(ratio):.2f}; images with ratio > 1.5: {int((ratio > 1.5).sum())}")
worst = np.argsort(-ratio)[:8]
for i in worst:
    print(f"  image {i}: hip {e_hip[i]:.3e} oracle {e_orc[i]:.3e} ratio {ratio[i]:.2f} saturated {bool(sat[i])} max|f64| {np.abs(f64[i]).max():.6f}")
print(f"bytes differing hip vs oracle {int((u8 != ref_u8).sum())} of {u8.size}")
q64 = np.clip(np.trunc(np.clip(f64 * 128.0, -128, 128)), -128, 127).astype(np.int64) + 128
print(f"bytes differing hip vs quantised f64 {int((u8 != q64).sum())}, oracle vs quantised f64 {int((ref_u8 != q64).sum())}")
