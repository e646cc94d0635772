#!/usr/bin/env python3
"""max |GPU embedding - CPU oracle embedding| over N synthetic images (test infrastructure: uses oracle/)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import capi as oracle
from pixelbox_amd import capi, synth, weights as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, n, 128, 128)
emb = capi.Embedder(blob, max_batch=64)
u8, f = emb.embed(imgs)
ref_u8, ref_f = oracle.mlhash_batch(blob, imgs, 256, nthreads=8)
err = np.abs(f - ref_f).max(axis=1)
unsat = np.abs(ref_f).max(axis=1) < 0.999
print(f"images {n}, unsaturated {int(unsat.sum())}: max |d| {err[unsat].max():.3e} (all images {err.max():.3e}); bytes differing {int((u8 != ref_u8).sum())} of {u8.size}")
