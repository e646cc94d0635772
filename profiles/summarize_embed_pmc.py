#!/usr/bin/env python3
"""HBM traffic of ONE steady-state batch-512 embed forward from rocprofv3 --pmc passes over profiles/embed_probe.py.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d D1 -o f -- python3 profiles/embed_probe.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d D2 -o w -- python3 profiles/embed_probe.py
    python3 profiles/summarize_embed_pmc.py OUT.json FETCH_SIZE=D1/.../f_counter_collection.csv WRITE_SIZE=D2/.../w_counter_collection.csv

The last forward of the run (last k_stem_dw / k_stem dispatch .. k_tanh_quant) is summed per kernel family.  Units per
/opt/skills/guides/MI355X_MICROARCH.md section HBM: KiB; FETCH_SIZE is reported RAW and doubled -- the guide's gfx950
correction is established for wide (16 B per lane) coalesced streaming reads, which is what these kernels issue for
activations (float4 per lane) but not for every weight / tap read, so the truth lies between the two; WRITE_SIZE is exact
for 16-byte-per-lane stores.  Infinity-Cache hits are counted (guide), so a tensor written by one kernel and read by the
next while it is still cached shows up in both columns although it may never have reached HBM.
"""
import collections
import csv
import json
import sys


def last_forward(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    starts = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"]]
    s = starts[-1]
    out = []
    for r in rows[s:]:
        if "pbe::" not in r["Kernel_Name"]:
            break
        out.append((r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pbe::", ""), float(r["Counter_Value"])))
        if "k_tanh_quant" in r["Kernel_Name"]:
            break
    return out


def main():
    out_path = sys.argv[1]
    per = {}
    for arg in sys.argv[2:]:
        name, path = arg.split("=", 1)
        per[name] = last_forward(path, name)
    fam = collections.defaultdict(lambda: {"launches": 0, "FETCH_KiB_raw": 0.0, "WRITE_KiB": 0.0})
    for name, rows in per.items():
        for k, v in rows:
            f = k.split("<")[0]
            if name == "FETCH_SIZE":
                fam[f]["launches"] += 1
                fam[f]["FETCH_KiB_raw"] += v
            else:
                fam[f]["WRITE_KiB"] += v
    tot_f = sum(d["FETCH_KiB_raw"] for d in fam.values()) * 1024
    tot_w = sum(d["WRITE_KiB"] for d in fam.values()) * 1024
    res = {"per_kernel_family": fam, "launches_per_forward": len(per.get("FETCH_SIZE", [])),
           "fetch_bytes_raw": int(tot_f), "fetch_bytes_doubled": int(2 * tot_f), "write_bytes": int(tot_w),
           "hbm_bytes_per_forward_low": int(tot_f + tot_w), "hbm_bytes_per_forward_high": int(2 * tot_f + tot_w)}
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
