"""The looped filter launch (64 passes per launch) with workgroup tickets (default) or fixed strides (PB_LOOP_STATIC=1),
over a 10M-row table and a 1.25M-row shard; PB_PROBE_VARIANTS=1 also tries other launch shapes (loads in flight per lane,
waves per workgroup, workgroups per CU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixelbox_amd import capi, synth

shapes = [("default", {})]
if os.environ.get("PB_PROBE_VARIANTS"):
    shapes += [("U=16", {4: 2}), ("U=4", {4: 4}), ("4 waves x 2 wg/cu", {6: 4, 5: 2}), ("8 waves x 2 wg/cu", {5: 2}), ("U=16 x 2 wg/cu", {4: 2, 5: 2})]
for rows in [int(x) for x in os.environ.get("PB_PROBE_ROWS", "10000000,1250000").split(",")]:
    ix = capi.Index(256, rows); ix.fill_synthetic(synth.SEED_INDEX, 0, rows, 1); ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
    q = synth.fill_synthetic(synth.SEED_QUERY, 0, 8 * 64 * 256).reshape(8, 64, 256)
    for name, opts in shapes:
        for o in (4, 5, 6):
            ix.set_option(o, {4: 0, 5: 1, 6: 8}[o])
        for o, v in opts.items():
            ix.set_option(o, v)
        for r in range(3): ix.search(q[r], 100, 1e3)
        ix.stats(reset=True); ix.set_option(capi.PB_OPT_PROFILE, 1)
        t0 = time.perf_counter()
        for r in range(16): ix.search(q[r % 8], 100, 1e3)
        dt = (time.perf_counter() - t0) / 16
        st = ix.stats(); ix.set_option(capi.PB_OPT_PROFILE, 0)
        print(f"{os.environ.get('PB_LOOP_STATIC','tickets'):>8s} {name:>18s} rows {rows}: step {dt*1e3:.3f} ms, kernel {st.profiled_ms/st.profiled_launches:.4f} ms per 64 passes = {st.profiled_bytes/(st.profiled_ms*1e-3)/1e12:.3f} TB/s, certified {st.fast_path}/{st.queries}")
    del ix
