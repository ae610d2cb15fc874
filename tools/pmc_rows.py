#!/usr/bin/env python
"""Reduce a rocprofv3 --pmc counter_collection CSV to per-row averages of one kernel.
Usage: python tools/pmc_rows.py <counter_collection.csv> <kernel substring> <rows x problems swept by ONE launch>"""
import csv
import sys
from collections import defaultdict

path, kern, work = sys.argv[1], sys.argv[2], float(sys.argv[3])
tot, calls = defaultdict(float), defaultdict(int)
with open(path) as fh:
    for row in csv.DictReader(fh):
        if kern in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            calls[row["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} total {tot[k]:.4g}  launches {calls[k]}  per row and problem {tot[k] / (work * calls[k]):10.3f}")
