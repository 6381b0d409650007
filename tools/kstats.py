"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "")
    name = name.split("(")[0][:60]
    print(f'{name:60s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"]) / 1e3:9.2f} us  min {float(r["MinNs"]) / 1e3:9.2f}  max {float(r["MaxNs"]) / 1e3:9.2f}')
