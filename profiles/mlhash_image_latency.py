"""pb_mlhash_image (resize_to_fill on the GPU + the network) of one 256 x 256 image and of an exact-size one, ms per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixelbox_amd import capi, synth, weights as W
blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
emb = capi.Embedder(blob, max_batch=8)
big = synth.fill_synthetic(synth.SEED_IMAGES, 0, 256 * 256 * 3).reshape(256, 256, 3)
img = synth.synthetic_images(synth.SEED_IMAGES, 0, 1, 128, 128)[0]
for name, x in (("256x256", big), ("128x128", img)):
    for _ in range(20): emb.mlhash_image(x)
    t0 = time.perf_counter()
    for _ in range(200): emb.mlhash_image(x)
    print(f"mlhash_image {name} latency ms {(time.perf_counter() - t0) / 200 * 1e3:.4f}")
