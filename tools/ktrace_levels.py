#!/usr/bin/env python
"""Durations of the last evaluation's launches in a rocprofv3 kernel trace, in launch order:
python tools/ktrace_levels.py trace.csv [kernel-substring ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:] or ["k_"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in keys)]
# the last evaluation: from the last k_factor7/k_factorw launch that follows a k_corr/k_reduce... simply the tail
n = int(sys.argv[0] and 60)
t_prev = None
for r in sel[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if t_prev is None else (s - t_prev) / 1e3
    t_prev = e
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
    print(f"{name:40s} grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8s} wg {r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')):>4s}  {(e - s) / 1e3:8.1f} us   gap {gap:6.1f} us")
