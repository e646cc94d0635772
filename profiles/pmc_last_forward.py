#!/usr/bin/env python3
"""Per-dispatch PMC counters of the LAST embed forward of a profiles/embed_probe.py run.

    rocprofv3 --pmc C1 C2 ... --output-format csv -d D -o p -- python3 profiles/embed_probe.py
    python3 profiles/pmc_last_forward.py D/.../p_counter_collection.csv [kernel substring]

Prints one line per dispatch (kernel, grid) with every collected counter; used to look at what bounds a layer
(TA busy / stalled, TCP -> TCC requests and latency, L2 hits and misses, ...)."""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        key = int(r["Dispatch_Id"])
        d = per.setdefault(key, {"name": r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pbe::", ""), "grid": r["Grid_Size"]})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)
    starts = [i for i in ids if "k_stem" in per[i]["name"]]
    s = starts[-1]
    for i in ids:
        if i < s:
            continue
        d = per[i]
        if want and want not in d["name"]:
            continue
        extra = " ".join(f"{k}={v:.4g}" for k, v in d.items() if k not in ("name", "grid"))
        print(f"{d['name'][:34]:34s} grid={d['grid']:>9s} {extra}")
        if "k_tanh_quant" in d["name"] or ("k_gemm_t" in d["name"] and d["name"].rstrip(">").endswith(", 2")):
            break


if __name__ == "__main__":
    main()
