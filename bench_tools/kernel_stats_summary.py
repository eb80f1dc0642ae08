"""Readable digest of a rocprofv3 --kernel-trace --stats kernel_stats.csv.  Usage: kernel_stats_summary.py <csv> [top]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
total = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    name = r["Name"]
    m = re.search(r"(\w+<[^(]*>|\w+)\(", name.replace("(anonymous namespace)::", ""))
    short = (m.group(1) if m else name)[:70]
    print("%-70s calls %5s  avg %9.1f us  %5.1f %%" % (short, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                      float(r["TotalDurationNs"]) / total * 100))
