#!/usr/bin/env python3
"""Per-layer timing of one embed forward from a rocprofv3 --kernel-trace CSV (last k_stem .. k_fc_tanh_quant)."""
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    idx = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"]]
    big = max(int(rows[i]["Grid_Size_X"]) for i in idx)
    s = [i for i in idx if int(rows[i]["Grid_Size_X"]) == big][-1]  # last forward of the largest batch
    tot, by = 0.0, {}
    for r in rows[s : s + 80]:
        n = r["Kernel_Name"]
        if "pbe::" not in n:
            break
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
        tot += dur
        short = n.split("(")[0].replace("void pbe::", "").replace("pbe::", "")
        fam = short.split("<")[0]
        by[fam] = by.get(fam, 0.0) + dur
        print(f"{short:26s} grid={r['Grid_Size_X']:>9s}x{r['Grid_Size_Y']:>4s}x{r['Grid_Size_Z']:>2s} wg={r['Workgroup_Size_X']:>4s} vgpr={r['VGPR_Count']:>4s} {dur:8.1f} us")
        if "k_tanh_quant" in n:
            break
    print("total us", round(tot, 1), {k: round(v, 1) for k, v in by.items()})


if __name__ == "__main__":
    main(sys.argv[1])
