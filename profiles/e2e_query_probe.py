"""Where does the query phase of the end-to-end leg (bench.py end_to_end: 1M embedded images, 1000 queries) go?

    python3 profiles/e2e_query_probe.py                     # wall times of the 1000-query call, path counters
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e2e -o e2e -- python3 profiles/e2e_query_probe.py
    python3 profiles/e2e_query_probe.py --summarize gpurun_out/prof_e2e

Builds the same table as the bench leg (hashes of the synthetic images, ids = image number + 1), saves it to
gpurun_out/e2e_table.npz for other probes, then answers the 1000 queries PB_PROBE_REPS times.  With --summarize it prints
the scan kernels' total time per repetition from a kernel trace of such a run.
"""
import glob
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = int(os.environ.get("PB_PROBE_REPS", "4"))


def run():
    import torch

    from pixelbox_amd import capi, synth, weights

    n, nb, d = int(os.environ.get("PB_PROBE_IMAGES", "1000000")), 512, 256
    gain = float(os.environ.get("PIXELBOX_E2E_FC_GAIN", "3.0"))  # as bench.py's end-to-end leg: structured scenes, scaled final Linear
    emb = capi.Embedder(weights.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, d, fc_gain=gain), max_batch=nb, device=0)
    ix = capi.Index(d, n)
    imgs = torch.empty((nb, 128, 128, 3), dtype=torch.uint8, device="cuda:0")
    out = torch.empty((nb, d), dtype=torch.uint8, device="cuda:0")
    for first in range(0, n, nb):
        count = min(nb, n - first)
        capi.fill_synthetic_scenes_device(0, synth.SEED_IMAGES, first, count, 128, 128, imgs.data_ptr(), 4)
        emb.embed_device(imgs.data_ptr(), count, out.data_ptr())
        ix.append_device(np.arange(first + 1, first + count + 1, dtype=np.int64), out.data_ptr())
    nq = 1000
    pick = (np.arange(nq, dtype=np.int64) * n) // nq
    t_ids, t_rows = ix.read(0, len(ix))
    qh = t_rows[pick].copy()  # ids are image number + 1 in insertion order; duplicates of a hash are kept (distinct ids)
    assert np.array_equal(t_ids[pick], pick + 1)
    os.makedirs("gpurun_out", exist_ok=True)
    if os.environ.get("PB_PROBE_SAVE"):
        np.savez_compressed("gpurun_out/e2e_table.npz", ids=t_ids, rows=t_rows, queries=qh)
    del t_rows
    ix.search(qh[:128], 100, 1e3)
    for r in range(REPS):
        ix.stats(reset=True)
        t0 = time.perf_counter()
        ix.search(qh, 100, 1e3)
        dt = (time.perf_counter() - t0) * 1e3
        st = ix.stats()
        print(f"rep {r}: {dt:.2f} ms  certified {st.fast_path} second_chance {st.second_chance} exhaustive {st.fallback}")


def summarize(d):
    import csv

    tot = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0]
                if "pb_embed" in name or "pbe::" in name:
                    continue
                t = tot.setdefault(name, [0, 0])
                t[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                t[1] += 1
    for name, (ns, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{ns / 1e6:9.3f} ms {c:7d} calls  {name[:110]}")


def timeline(d, last_ms=32.0):
    """Kernels and copies of the last `last_ms` of the trace: start offset, duration, gap to the previous one."""
    import csv

    ev = []
    for pat in ("*kernel_trace.csv", "*memory_copy_trace.csv"):
        for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    name = r.get("Kernel_Name") or ("copy " + r.get("Direction", ""))
                    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0][-60:]))
    ev.sort()
    t_end = ev[-1][1]
    ev = [e for e in ev if e[0] >= t_end - last_ms * 1e6]
    prev = ev[0][0]
    for s, e, name in ev:
        print(f"{(s - ev[0][0]) / 1e3:10.1f} us  +{(s - prev) / 1e3:8.1f} gap  {(e - s) / 1e3:9.1f} us  {name}")
        prev = e


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--timeline":
        timeline(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        run()
