import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pixelbox_amd import capi, synth
for rows in (10_000_000, 1_250_000):
    ix = capi.Index(256, rows); ix.fill_synthetic(synth.SEED_INDEX, 0, rows, 1); ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
    q = synth.fill_synthetic(synth.SEED_QUERY, 0, 8 * 64 * 256).reshape(8, 64, 256)
    for r in range(3): ix.search(q[r], 100, 1e3)
    ix.stats(reset=True); ix.set_option(capi.PB_OPT_PROFILE, 1)
    t0 = time.perf_counter()
    for r in range(16): ix.search(q[r % 8], 100, 1e3)
    dt = (time.perf_counter() - t0) / 16
    st = ix.stats()
    print(f"{os.environ.get('PB_LOOP_STATIC','tickets'):>8s} rows {rows}: step {dt*1e3:.3f} ms, kernel {st.profiled_ms/st.profiled_launches:.4f} ms per 64 passes = {st.profiled_bytes/(st.profiled_ms*1e-3)/1e12:.3f} TB/s, certified {st.fast_path}/{st.queries}")
    del ix
