#!/usr/bin/env python3
"""byte_distance / hamming_distance scans over a 10M x 256 phashes-like table: coalesced exact-key pass vs exhaustive."""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pixelbox_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for metric, name in ((capi.PB_METRIC_BYTE, "byte_distance"), (capi.PB_METRIC_HAMMING, "hamming_distance")):
    ix = capi.Index(256, n, metric=metric)
    ix.fill_synthetic(synth.SEED_INDEX, 0, n, 1)
    q = synth.fill_synthetic(synth.SEED_QUERY, 0, 32 * 256).reshape(32, 256)
    for path, label in ((0, "coalesced pass"), (1, "exhaustive pass")):
        ix.set_option(capi.PB_OPT_SEARCH_PATH, path)
        ix.search(q, 100, 0.45)
        ix.stats(reset=True)
        t0 = time.perf_counter()
        ix.search(q, 100, 0.45)
        dt = time.perf_counter() - t0
        st = ix.stats()
        print(f"{name:17s} {label:16s}: {dt / 32 * 1e3:7.3f} ms/query = {n * 256 / (dt / 32) / 1e12:5.2f} TB/s of table bytes; "
              f"fast {st.fast_path} exhaustive {st.fallback}")
    del ix
