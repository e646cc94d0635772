#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files (one counter per pass) into a small JSON.

    python profiles/summarize_pmc.py OUT.json FETCH_SIZE=path/to/counter_collection.csv WRITE_SIZE=path/...

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced (16 B/lane) streaming read, so it is
doubled for kernels whose reads are of that shape (flagged per kernel below); WRITE_SIZE is exact.
"""
import collections
import csv
import json
import sys

WIDE_STREAM_KERNELS = ("k_scan_filter", "k_scan_multi")  # 16 B/lane coalesced streaming reads


def main():
    out_path = sys.argv[1]
    res = collections.defaultdict(dict)
    for arg in sys.argv[2:]:
        name, path = arg.split("=", 1)
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name:
                # launches of different shapes (e.g. 64-query steps vs single-query latency probes) are kept apart
                key = r["Kernel_Name"].split("(")[0].replace("void ", "") + " grid=" + r["Grid_Size"]
                agg[key].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            res[k][name + "_KiB_avg_per_launch"] = sum(v) / len(v)
            res[k][name + "_launches"] = len(v)
    for k, d in res.items():
        f = d.get("FETCH_SIZE_KiB_avg_per_launch")
        w = d.get("WRITE_SIZE_KiB_avg_per_launch", 0.0)
        if f is not None:
            corr = 2.0 if any(s in k for s in WIDE_STREAM_KERNELS) else 1.0
            d["fetch_correction"] = corr
            d["hbm_bytes_per_launch"] = int(f * 1024 * corr + w * 1024)
            if corr == 1.0:
                # every other kernel also reads with 16-byte-per-lane loads somewhere (k_row_norms streams the whole table that way), and
                # the guide's gfx950 note (FETCH_SIZE under-reports wide loads by up to 2x) applies to those requests too: the raw figure
                # is a LOWER bound, twice it an upper bound (k_row_norms "2.04 GB" for a 2.56 GB table is the lower one)
                d["hbm_bytes_per_launch_upper"] = int(f * 1024 * 2.0 + w * 1024)
                d["fetch_correction_note"] = "raw FETCH_SIZE: lower bound; hbm_bytes_per_launch_upper doubles it (16-byte loads are under-counted on gfx950)"
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
