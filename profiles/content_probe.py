"""Does the filter pass's rate depend on the table's CONTENT?  ('a table of constant bytes is scanned at half the
rate of a random one, reason not found'.)  4M-row tables (1 GB: four Infinity Caches) with different fills, the 64-pass looped launch
timed on each, with queries that keep top-100 lists (max_dist 1e3) and with queries nothing can pass (max_dist 1e-9)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixelbox_amd import capi, synth

rows = int(os.environ.get("PB_PROBE_ROWS", "4000000"))
rng = np.random.default_rng(3)
q = synth.fill_synthetic(synth.SEED_QUERY, 0, 2 * 64 * 256).reshape(2, 64, 256)
ids = np.arange(1, rows + 1, dtype=np.int64)
pat = rng.integers(0, 256, 256, dtype=np.uint8)
fills = [
    ("random bytes", lambda: synth.fill_synthetic(synth.SEED_INDEX, 0, rows * 256).reshape(rows, 256)),
    ("all 0x00", lambda: np.zeros((rows, 256), np.uint8)),
    ("all 0x80", lambda: np.full((rows, 256), 0x80, np.uint8)),
    ("all 0xFF", lambda: np.full((rows, 256), 0xFF, np.uint8)),
    ("every row the same random 256 bytes", lambda: np.broadcast_to(pat, (rows, 256)).copy()),
    ("random, but only 2 distinct byte values (0x40 / 0xC0)", lambda: np.where(synth.fill_synthetic(synth.SEED_INDEX, 0, rows * 256).reshape(rows, 256) & 1, 0xC0, 0x40).astype(np.uint8)),
    ("random low nibble, high nibble 0x8", lambda: (synth.fill_synthetic(synth.SEED_INDEX, 0, rows * 256).reshape(rows, 256) & 0x0F | 0x80).astype(np.uint8)),
]
for name, make in fills:
    tab = make()
    ix = capi.Index(256, rows)
    ix.load(ids, tab)
    ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
    res = []
    for md in (1e3, 1e-9):
        ix.search(q[0], 100, md)
        ix.stats(reset=True); ix.set_option(capi.PB_OPT_PROFILE, 1)
        for r in range(3): ix.search(q[r & 1], 100, md)
        st = ix.stats(); ix.set_option(capi.PB_OPT_PROFILE, 0)
        res.append((st.profiled_ms / st.profiled_launches, st.fast_path, st.fallback, st.queries))
    gb = rows * 256 * 64 / 1e9
    print(f"{name:55s}: max_dist 1e3 {res[0][0]:7.3f} ms = {gb / res[0][0]:5.2f} TB/s (certified {res[0][1]}/{res[0][3]}, exhaustive {res[0][2]});  "
          f"nothing passes {res[1][0]:7.3f} ms = {gb / res[1][0]:5.2f} TB/s", flush=True)
    del ix, tab
