import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pixelbox_amd import capi, synth
for rows in (1_000_000, 1_250_000, 500_000):
    ix = capi.Index(256, rows); ix.fill_synthetic(synth.SEED_INDEX, 0, rows, 1); ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
    q = synth.fill_synthetic(synth.SEED_QUERY + 5, 0, 64 * 256).reshape(64, 256)
    for i in range(5): ix.search(q[i:i+1], 100, 1e3)
    ix.stats(reset=True); ix.set_option(capi.PB_OPT_PROFILE, 1)
    w = []
    for i in range(64):
        call, *_ = ix.prepared_search(q[i:i+1], 100, 1e3)
        t0 = time.perf_counter(); call(); w.append((time.perf_counter() - t0) * 1e3)
    st = ix.stats(); w.sort()
    print(f"{os.environ.get('PB_LOOP_STATIC','auto'):>7s} rows {rows}: kernel {st.profiled_ms/st.profiled_launches*1e3:.1f} us = {rows*256/(st.profiled_ms/st.profiled_launches*1e-3)/1e12:.2f} TB/s, call median {w[32]*1e3:.1f} us, certified {st.fast_path}/{st.queries}")
