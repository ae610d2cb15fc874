#!/usr/bin/env python
"""HBM bytes per launch of the kernels whose names contain the given substrings, from the two rocprofv3 PMC
passes FETCH_SIZE and WRITE_SIZE (separate runs: the TCC block cannot hold both).  FETCH_SIZE is doubled
(gfx950 tallies wide streaming reads at half their bytes: MI355X_MICROARCH.md, HBM section); both counters
are reported in KiB.  Usage: python tools/pmc_bytes.py <fetch.csv> <write.csv> <out.json> <substring> ..."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, name):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == name:
                tot[row["Kernel_Name"]] += float(row["Counter_Value"])
                n[row["Kernel_Name"]] += 1
    return tot, n


fetch_csv, write_csv, out = sys.argv[1:4]
ft, fn = per_kernel(fetch_csv, "FETCH_SIZE")
wt, wn = per_kernel(write_csv, "WRITE_SIZE")
res = {"fetch_correction": 2.0, "kernels": {}}
for sub in sys.argv[4:]:
    names = [k for k in ft if sub in k]
    f = sum(ft[k] for k in names) / max(1, sum(fn[k] for k in names))
    w = sum(wt[k] for k in names) / max(1, sum(wn[k] for k in names))
    res["kernels"][sub] = {"launches": sum(fn[k] for k in names), "fetch_size_kb_per_launch": f,
                           "write_size_kb_per_launch": w, "hbm_GB_per_launch": (2.0 * f + w) * 1024.0 / 1e9}
res["hbm_GB_sum_one_launch_each"] = sum(v["hbm_GB_per_launch"] for v in res["kernels"].values())
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
