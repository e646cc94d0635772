"""Slow and fast tables: one process creates PB_PROBE_TABLES 10M-row tables, keeps them all, and times the 64-pass
looped filter launch over each (kernel time from the library's own events, PB_OPT_PROFILE), in the default tile mapping and with
wave-fastest tiles (PB_OPT_SCAN_VARIANT bit 3: a workgroup reads 8 adjacent 8-KiB pieces instead of pieces 2 MB apart).  Run under
rocprofv3 --pmc <translation counters> the dispatches line up with the tables in launch order (3 launches per table and mapping,
the last one of each is the one compared)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixelbox_amd import capi, synth

import ctypes as C
if os.environ.get("PB_PROBE_FRAGMENT"):
    # what a box that has run other jobs looks like to the allocator: many blocks of 1-8 MB over most of the memory, every other one
    # freed (the survivors stay allocated for the life of this process), so that a 2.56 GB table is pieced together from the holes
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    rng = np.random.default_rng(7)
    total_gb = float(os.environ.get("PB_PROBE_FRAGMENT_GB", "120"))
    blocks, tot = [], 0
    while tot < total_gb * (1 << 30):
        sz = int(rng.integers(1, 9)) << 20
        p_ = C.c_void_p()
        if hip.hipMalloc(C.byref(p_), sz) != 0:
            break
        blocks.append(p_)
        tot += sz
    keep = []
    for i, b in enumerate(blocks):
        if i & 1:
            hip.hipFree(b)
        else:
            keep.append(b)
    print(f"fragmented: {len(blocks)} blocks of 1-8 MB ({tot / 2**30:.1f} GB), every other one freed", flush=True)
rows = int(os.environ.get("PB_PROBE_ROWS", "10000000"))
n_tab = int(os.environ.get("PB_PROBE_TABLES", "6"))
q = synth.fill_synthetic(synth.SEED_QUERY, 0, 4 * 64 * 256).reshape(4, 64, 256)
tabs = []
for t in range(n_tab):
    ix = capi.Index(256, rows)
    ix.fill_synthetic(synth.SEED_INDEX, 0, rows, 1)
    ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
    tabs.append(ix)
for rnd in range(2):
    for t, ix in enumerate(tabs):
        res = []
        for variant in (0, 8):
            ix.set_option(4, variant)  # PB_OPT_SCAN_VARIANT
            ix.search(q[0], 100, 1e3)
            ix.stats(reset=True); ix.set_option(capi.PB_OPT_PROFILE, 1)
            for r in range(2): ix.search(q[1 + r], 100, 1e3)
            st = ix.stats(); ix.set_option(capi.PB_OPT_PROFILE, 0)
            res.append(st.profiled_ms / st.profiled_launches)
        ix.set_option(4, 0)
        print(f"round {rnd} table {t}: default mapping {res[0]:.3f} ms per 64 passes, wave-fastest tiles {res[1]:.3f} ms", flush=True)
