import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pixelbox_amd import capi, synth, weights as W

blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
emb = capi.Embedder(blob, max_batch=512)
rng = np.random.default_rng(1)
N = 2048

def stats(name, imgs):
    u8, f = emb.embed(imgs)
    uniq = len({h.tobytes() for h in u8})
    x = f - f.mean(0)
    s = np.linalg.svd(x, compute_uv=False)
    ev = s**2 / (s**2).sum()
    print(f"{name:28s} unique hashes {uniq}/{len(u8)}, per-dim std median {f.std(0).mean():.4f}, top-1 PCA share {ev[0]:.3f}, dims for 90% {int(np.searchsorted(np.cumsum(ev), 0.9)) + 1}, |f| max {np.abs(f).max():.3f}")

stats("current generator", synth.synthetic_images(synth.SEED_IMAGES, 0, N, 128, 128))
stats("uniform noise", rng.integers(0, 256, (N, 128, 128, 3), dtype=np.uint8))
# blobs + gradients
yy, xx = np.mgrid[0:128, 0:128].astype(np.float32)
imgs = np.zeros((N, 128, 128, 3), np.float32)
for i in range(N):
    im = np.zeros((128, 128, 3), np.float32)
    g = rng.uniform(-1, 1, (2, 3)); im += (xx[..., None] * g[0] + yy[..., None] * g[1]) * 0.8 + rng.uniform(40, 200, 3)
    for _ in range(rng.integers(2, 7)):
        cx, cy, r = rng.uniform(0, 128), rng.uniform(0, 128), rng.uniform(6, 40)
        col = rng.uniform(-150, 150, 3)
        im += np.exp(-(((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)))[..., None] * col
    fx, fy, ph = rng.uniform(0.02, 0.6), rng.uniform(0.02, 0.6), rng.uniform(0, 6.28)
    im += (np.sin(xx * fx + yy * fy + ph) * rng.uniform(0, 60))[..., None] * rng.uniform(-1, 1, 3)
    imgs[i] = im
stats("blobs+gradient+wave", np.clip(imgs, 0, 255).astype(np.uint8))
