"""Start / end of every kernel of the LAST ct x ct of a rocprofv3 --kernel-trace kernel_trace.csv relative to the first one's
start, with the stream (queue) it ran on -- what shows two kernels of one call running side by side.

  python bench_tools/kernel_overlap.py <kernel_trace.csv> [name-of-the-first-kernel-of-a-call]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|heamd::|^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:60]


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "behz_rows_fused<13, 10, 4"
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    if not starts:
        raise SystemExit("no kernel named like %r" % first)
    begin = starts[-1]
    # a call = the kernels from a little before its first row-fused launch (the lifts may start first) to its floor kernel
    while begin > 0 and "lift_kernel" in rows[begin - 1]["Kernel_Name"]:
        begin -= 1
    origin = int(rows[begin]["Start_Timestamp"])
    for r in rows[begin:]:
        start, end = int(r["Start_Timestamp"]) - origin, int(r["End_Timestamp"]) - origin
        queue = r.get("Queue_Id", r.get("Stream_Id", "?"))
        print("%-60s queue %-4s start %9.1f us  end %9.1f us  (%8.1f us)" % (short(r["Kernel_Name"]), queue, start / 1e3, end / 1e3,
                                                                        (end - start) / 1e3))
        if "floor_kernel" in r["Kernel_Name"]:
            break


if __name__ == "__main__":
    main()
