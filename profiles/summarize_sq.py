#!/usr/bin/env python3
"""Where do the waves of each kernel spend their cycles?  From a rocprofv3 pass
    --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
(MI355X_MICROARCH.md, rocprofv3 PMC slots: WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall,
ACTIVE_INST_ANY = issuing; the three add up to WAVE_CYCLES).

    python profiles/summarize_sq.py OUT.json path/to/counter_collection.csv
"""
import collections
import csv
import json
import sys


def main():
    out, path = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        fam = k.split("<")[0]
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    res = {}
    for fam, d in sorted(agg.items()):
        wc = d.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        res[fam] = {"parked_waitcnt_or_barrier": round(d.get("SQ_WAIT_ANY", 0) / wc, 3),
                    "issue_stall": round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                    "issuing": round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                    "issuing_valu": round(d.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                    "lds_issue_stall": round(d.get("SQ_WAIT_INST_LDS", 0) / wc, 3)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print(f"{k:28s} {v}")


if __name__ == "__main__":
    main()
