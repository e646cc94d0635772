#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.

    python profiles/summarize_mfma.py OUT.json path/to/counter_collection.csv

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): GRBM_GUI_ACTIVE is reported as the sum
over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), SQ_VALU_MFMA_BUSY_CYCLES as the sum over all SIMDs.
"""
import collections
import csv
import json
import sys


def main():
    out, path = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    rows = list(csv.DictReader(open(path)))
    # round 6: only the LAST forward of the run -- the dispatches from its stem kernel on -- so that the per-layer
    # timing loops' candidate launches (every form of every layer, several batch buckets) are not pooled with the forward that runs
    stems = [int(r["Dispatch_Id"]) for r in rows if "k_stem" in r["Kernel_Name"]]
    first = max(stems) if stems else 0
    rows = [r for r in rows if int(r["Dispatch_Id"]) >= first]
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            launches[k] += 1
    res = {}
    tot = collections.defaultdict(lambda: [0.0, 0.0])
    for k, d in sorted(agg.items()):
        gui, mf = d.get("GRBM_GUI_ACTIVE", 0.0), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        res[k] = {"GRBM_GUI_ACTIVE": gui, "SQ_VALU_MFMA_BUSY_CYCLES": mf, "launches": launches[k],
                  "mfma_busy_frac": round(mf / (gui / 8 * 1024), 4) if gui else 0.0}
        for group, pred in (("_all_embed_kernels", k.startswith("pbe::")), ("_gemm_kernels_only", "k_gemm1x1" in k),
                            ("_fused_front_kernels", "k_front_roll" in k), ("_burst_collect_kernel", "k_scan_multi_wg" in k)):
            if pred:
                tot[group][0] += mf
                tot[group][1] += gui
    for g, (mf, gui) in tot.items():
        res[g] = {"mfma_busy_frac": round(mf / (gui / 8 * 1024), 4) if gui else 0.0}
    res["_note"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 "
                    "of profiles/embed_probe.py: the dispatches of its LAST batch-512 forward only (from the stem kernel on; the timing loops' candidate launches are excluded). mfma_busy_frac = "
                    "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs).")
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res):
        if k.startswith("_") and k != "_note":
            print(k, res[k])


if __name__ == "__main__":
    main()
