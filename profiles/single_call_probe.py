"""Where does a one-query call spend its time?  (roofline_single_call of bench.py)

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_single -- python3 profiles/single_call_probe.py
    python3 profiles/single_call_probe.py --summarize gpurun_out/prof_single

Runs N one-query pb_index_search calls over a 10M x 256 table and prints the median wall time per call; with
--summarize it reads the kernel trace of such a run and prints, per call, the kernels' durations and the gaps between
them (stage -> filter -> select), i.e. device time vs. everything else.
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
    cur = []
    for s, e, name in rows:
        short = name.split("(")[0].split("::")[-1]
        if "k_stage_query" in name:
            cur = [(s, e, short)]
            calls.append(cur)
        elif cur is not None and cur:
            cur.append((s, e, short))
    out = []
    for c in calls[-30:]:
        if len(c) < 3:
            continue
        names = [x[2][:22] for x in c[:3]]
        durs = [(x[1] - x[0]) / 1e3 for x in c[:3]]
        gaps = [(c[i + 1][0] - c[i][1]) / 1e3 for i in range(2)]
        out.append((names, durs, gaps, (c[2][1] - c[0][0]) / 1e3))
    if not out:
        print("no calls found")
        return
    med = lambda v: sorted(v)[len(v) // 2]
    print("kernels:", out[0][0])
    print("durations us (median):", [round(med([o[1][i] for o in out]), 2) for i in range(3)])
    print("gaps us (median):", [round(med([o[2][i] for o in out]), 2) for i in range(2)])
    print("first kernel start -> last kernel end us (median):", round(med([o[3] for o in out]), 2))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        run()
