"""Exhaustive pass (k_scan_exact_co) over a 10M x 256 table: kernel time per query for QN = 1 and 2.
PIXELBOX_LIB=<ablation build> selects another build of the library (see pb_scan_kernels.h PB_XC_ABL)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelbox_amd import capi, synth

rows = int(os.environ.get("PB_PROBE_ROWS", "10000000"))
ix = capi.Index(256, rows)
ix.fill_synthetic(synth.SEED_INDEX, 0, rows, 1)
ix.set_option(capi.PB_OPT_SEARCH_PATH, 1)
q = synth.fill_synthetic(synth.SEED_QUERY, 0, 64 * 256).reshape(64, 256)
for qn in (1, 2, 4):
    ix.set_option(capi.PB_OPT_EXACT_QN, qn)
    ix.search(q[:8])
    ix.stats(reset=True)
    ix.set_option(capi.PB_OPT_PROFILE, 1)
    t0 = time.perf_counter()
    ix.search(q)
    dt = time.perf_counter() - t0
    ix.set_option(capi.PB_OPT_PROFILE, 0)
    st = ix.stats()
    per = st.profiled_ms / 64
    print(f"{os.environ.get('PIXELBOX_LIB', 'default'):>40s} QN {qn}: 64 queries wall {dt * 1e3:7.2f} ms, kernel {per:.4f} ms per query = "
          f"{rows * 256 / (per * 1e-3) / 1e12:.2f} TB/s algorithmic per query, {rows * 256 / (per * qn * 1e-3) / 1e12:.2f} TB/s streamed")
