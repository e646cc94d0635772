#!/usr/bin/env python3
"""Interleaved A/B sweep of k_scan_filter variants in ONE process (guide rule 24): prints median GB/s per variant."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixelbox_amd import capi, synth

n, d, B = int(os.environ.get("SWEEP_ROWS", "10000000")), 256, 16
ix = capi.Index(d, n)
ix.fill_synthetic(synth.SEED_INDEX, 0, n, 1)
q = synth.fill_synthetic(synth.SEED_QUERY, 0, 64 * 16 * d).reshape(64, 16, d)
# (variant bits, workgroups per CU, waves per workgroup)
# + explicit grid (0 = wg/cu * CUs); variant bit 3 = wave-fastest tile mapping
# variant bits: 1 = plain loads, 2/4 = U16/U4, 8 = wave-fastest mapping
variants = [(0, 1, 8, 0), (1, 1, 8, 0), (0, 1, 16, 0), (1, 1, 16, 0), (0, 2, 8, 0), (1, 2, 8, 0), (0, 1, 4, 0), (0, 2, 4, 0), (1, 2, 4, 0), (0, 1, 8, 224), (0, 1, 8, 240), (2, 1, 16, 0), (8, 1, 8, 0), (4, 2, 8, 0), (5, 2, 8, 0)]
B = int(os.environ.get("SWEEP_B", "1"))
res = {k: [] for k in variants}
ix.set_option(capi.PB_OPT_PROFILE, 1)
for rnd in range(5):
    for (v, w, nw, g) in variants:
        ix.set_option(4, v); ix.set_option(5, w); ix.set_option(6, nw); ix.set_option(7, g)
        ix.stats(reset=True)
        for rep in range(4 if B == 1 else 1):
            ix.search(q[(rnd * 7 + v + rep) % 64][:B], 100, 1e3)
        st = ix.stats()
        if rnd:  # first round = warm-up
            res[(v, w, nw, g)].append(st.profiled_bytes / (st.profiled_ms * 1e-3) / 1e9)
names = {0: "U8+nt", 1: "U8", 2: "U16+nt", 3: "U16", 4: "U4+nt", 5: "U4"}
for (v, w, nw, g), xs in res.items():
    xs = sorted(xs)
    print(f"grid={g:3d} map={'wave' if v & 8 else 'wg  '} wg/cu={w} waves/wg={nw:2d} {names[v & 7]:7s} median {xs[len(xs)//2]:7.1f} GB/s  min {xs[0]:7.1f} max {xs[-1]:7.1f}")
