"""Which queries of the end-to-end leg miss the certificate, and why: per uncertified query the number of table rows that are byte-identical
to it, and the number whose integer cosine lies within 2e-6 of the k-th best (the set any filter would have to re-score exactly)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pixelbox_amd import capi, synth, weights

n, nb, d = int(os.environ.get("PB_PROBE_IMAGES", "1000000")), 512, 256
emb = capi.Embedder(weights.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, d, fc_gain=3.0), max_batch=nb, device=0)
ix = capi.Index(d, n)
imgs = torch.empty((nb, 128, 128, 3), dtype=torch.uint8, device="cuda:0")
out = torch.empty((nb, d), dtype=torch.uint8, device="cuda:0")
for first in range(0, n, nb):
    count = min(nb, n - first)
    capi.fill_synthetic_scenes_device(0, synth.SEED_IMAGES, first, count, 128, 128, imgs.data_ptr(), 4)
    emb.embed_device(imgs.data_ptr(), count, out.data_ptr())
    ix.append_device(np.arange(first + 1, first + count + 1, dtype=np.int64), out.data_ptr())
nq = 1000
pick = (np.arange(nq, dtype=np.int64) * n) // nq
t_ids, t_rows = ix.read(0, len(ix))
qh = t_rows[pick].copy()
ix.search(qh[:128], 100, 1e3)
ix.stats(reset=True)
ix.search(qh, 100, 1e3)
st = ix.stats()
print(f"burst of {nq}: certified {st.fast_path} second_chance {st.second_chance} exhaustive {st.fallback}")
# which ones: bursts of 8 queries (the concurrent path's minimum), the same gate per query
unc = []
for i in range(0, nq, 8):
    ix.stats(reset=True)
    ix.search(qh[i:i + 8], 100, 1e3)
    s2 = ix.stats()
    if s2.fallback:
        for j in range(i, min(i + 8, nq)):
            ix.stats(reset=True)
            ix.search(np.repeat(qh[j:j + 1], 8, axis=0), 100, 1e3)
            if ix.stats().fallback:
                unc.append(j)
print(f"uncertified when asked 8 at a time (a query repeated 8 times): {len(unc)}")
R = t_rows.astype(np.int32) * 2 - 255
nr = np.sqrt((R.astype(np.int64) ** 2).sum(axis=1).astype(np.float64))
hist_dup, hist_tie = [], []
for j in unc[:40]:
    q = qh[j].astype(np.int32) * 2 - 255
    cs = (R @ q).astype(np.float64) / (nr * np.sqrt(float((q.astype(np.int64) ** 2).sum())))
    kth = np.partition(cs, -100)[-100]
    dup = int((t_rows == qh[j]).all(axis=1).sum())
    tie = int((cs >= kth - 2e-6).sum())
    near = int((cs >= 0.999).sum())
    hist_dup.append(dup); hist_tie.append(tie)
    print(f"query {j}: byte-identical rows {dup}, rows with cos >= k-th - 2e-6: {tie}, rows with cos >= 0.999: {near}, k-th cos {kth:.7f}")
print("median identical", np.median(hist_dup), "median tie set", np.median(hist_tie))
