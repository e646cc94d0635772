"""Where does a one-query call spend its time?  (roofline_single_call of bench.py)

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_single -- python3 profiles/single_call_probe.py
    python3 profiles/single_call_probe.py --summarize gpurun_out/prof_single

Runs N one-query pb_index_search calls over a 10M x 256 table and prints the median wall time per call; with
--summarize it reads the kernel trace of such a run and prints, per call, the kernels' durations and the gaps between
them (filter -> select, and from one call's select to the next call's filter), i.e. device time vs. everything else.
"""
import glob
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    from pixelbox_amd import capi, synth

    rows = int(os.environ.get("PB_PROBE_ROWS", "10000000"))
    n = int(os.environ.get("PB_PROBE_CALLS", "40"))
    ix = capi.Index(256, rows)
    ix.fill_synthetic(synth.SEED_INDEX, 0, rows, 1)
    ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
    q = synth.fill_synthetic(synth.SEED_QUERY + 3, 0, n * 256).reshape(n, 256)
    for i in range(3):
        ix.search(q[i:i + 1], 100, 1e3)
    wall = []
    for i in range(n):
        t0 = time.perf_counter()
        ix.search(q[i:i + 1], 100, 1e3)
        wall.append((time.perf_counter() - t0) * 1e3)
    wall.sort()
    print(f"rows {rows}: one-query call wall ms: median {wall[len(wall) // 2]:.4f} min {wall[0]:.4f} p90 {wall[int(len(wall) * 0.9)]:.4f}")
    # the C call alone (arguments converted beforehand: what a compiled host pays)
    bare = []
    for i in range(n):
        call, ids, dist, cnt = ix.prepared_search(q[i:i + 1], 100, 1e3)
        t0 = time.perf_counter()
        call()
        bare.append((time.perf_counter() - t0) * 1e3)
    bare.sort()
    print(f"rows {rows}: bare pb_index_search call ms: median {bare[len(bare) // 2]:.4f} min {bare[0]:.4f} p90 {bare[int(len(bare) * 0.9)]:.4f}")
    st = ix.stats()
    print(f"certified {st.fast_path} of {st.queries}")


def summarize(d):
    import csv

    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    calls = []
    for i, (s, e, name) in enumerate(rows[:-1]):  # a call = the filter launch and the select launch behind it
        if "k_scan_filter" in name and "k_select_rescore" in rows[i + 1][2]:
            calls.append([(s, e, "k_scan_filter"), (rows[i + 1][0], rows[i + 1][1], "k_select_rescore")])
    out = []
    for j in range(max(1, len(calls) - 30), len(calls)):
        c = calls[j]
        durs = [(x[1] - x[0]) / 1e3 for x in c]
        gaps = [(c[1][0] - c[0][1]) / 1e3, (c[0][0] - calls[j - 1][1][1]) / 1e3]  # filter -> select, previous call's end -> this filter
        out.append(([x[2] for x in c], durs, gaps, (c[1][1] - c[0][0]) / 1e3))
    if not out:
        print("no calls found")
        return
    med = lambda v: sorted(v)[len(v) // 2]
    print("kernels:", out[0][0])
    print("durations us (median):", [round(med([o[1][i] for o in out]), 2) for i in range(2)])
    print("gap filter -> select, gap previous call's last kernel -> this call's filter (host turn-around), us (median):",
          [round(med([o[2][i] for o in out]), 2) for i in range(2)])
    print("first kernel start -> last kernel end us (median):", round(med([o[3] for o in out]), 2))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        run()
