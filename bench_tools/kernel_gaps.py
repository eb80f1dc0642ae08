"""Timeline digest of a rocprofv3 --kernel-trace kernel_trace.csv: for the LAST repetition of a repeated call sequence, the
kernels in launch order with their durations and the idle gap before each, and the totals -- how much of a latency-bound
tail is kernels and how much is the space between them.

  python bench_tools/kernel_gaps.py <kernel_trace.csv> <name-of-the-first-kernel-of-a-repetition> [max-rows]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|heamd::|^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:72]


def main():
    path, first = sys.argv[1], sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    if not starts:
        raise SystemExit("no kernel named like %r" % first)
    # the last complete repetition: from the last `first` kernel to the end, or between the last two
    begin = starts[-1]
    rep = rows[begin:]
    busy = gaps = 0
    previous_end = None
    for k, r in enumerate(rep):
        start, end = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0 if previous_end is None else max(0, start - previous_end)
        busy += end - start
        gaps += gap
        if k < limit:
            print("%-72s %9.1f us   gap before %7.1f us" % (short(r["Kernel_Name"]), (end - start) / 1e3, gap / 1e3))
        previous_end = end if previous_end is None else max(previous_end, end)
    span = previous_end - int(rep[0]["Start_Timestamp"])
    print("kernels %d  busy %.1f us  gaps %.1f us  span %.1f us  (gaps = %.1f %% of the span)" % (
        len(rep), busy / 1e3, gaps / 1e3, span / 1e3, 100.0 * gaps / max(span, 1)))


if __name__ == "__main__":
    main()
