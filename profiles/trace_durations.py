#!/usr/bin/env python3
"""Print per-call durations (us) of kernels whose name contains argv[2] from a rocprofv3 kernel_trace.csv (argv[1])."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print([round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if sys.argv[2] in r["Kernel_Name"]])
