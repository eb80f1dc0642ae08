"""The kernels of the LAST `count` launches of a rocprofv3 --kernel-trace run in launch order: duration, idle gap before each,
grid in workgroups -- for chains of small launches whose cost is latency, not throughput.

  python bench_tools/timeline_digest.py <kernel_trace.csv> <count>
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|heamd::|^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    count = int(sys.argv[2])
    rep = rows[-count:]
    busy = gaps = 0
    prev = None
    for r in rep:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0 if prev is None else max(0, s - prev)
        busy += e - s
        gaps += gap
        lanes = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        print("%-80s %8.1f us  gap %6.1f us  %6d workgroups of %d" % (short(r["Kernel_Name"]), (e - s) / 1e3, gap / 1e3,
                                                                       grid // max(lanes, 1), lanes))
        prev = e if prev is None else max(prev, e)
    print("busy %.1f us  gaps %.1f us  span %.1f us" % (busy / 1e3, gaps / 1e3, (prev - int(rep[0]["Start_Timestamp"])) / 1e3))


if __name__ == "__main__":
    main()
