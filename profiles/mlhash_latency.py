import os,sys,time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pixelbox_amd import capi, synth, weights as W
blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
emb = capi.Embedder(blob, max_batch=8)
img = synth.synthetic_images(synth.SEED_IMAGES, 0, 1, 128, 128)[0]
for _ in range(20): emb.mlhash(img)
t0=time.perf_counter()
for _ in range(200): emb.mlhash(img)
print("mlhash latency ms", (time.perf_counter()-t0)/200*1e3)
