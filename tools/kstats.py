#!/usr/bin/env python
"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py file.csv [rows] [calls-divisor]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.3f} ms" + (f"  ({tot / 1e6 / div:.3f} ms per unit of {div:g})" if div != 1 else ""))
for r in rows[:top]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'{name[:64]:64s} calls {int(r["Calls"]):5d}  total {float(r["TotalDurationNs"]) / 1e6:9.3f} ms  '
          f'avg {float(r["AverageNs"]) / 1e3:9.1f} us  {float(r["Percentage"]):5.1f}%')
