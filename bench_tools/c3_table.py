"""Merges two (kernel_stats.csv, traffic.json) pairs of the ct x ct + relinearize pipeline into one before / after table:
average microseconds per launch over 1024 products, launches per product batch, and HBM-side bytes per product
(2 x FETCH_SIZE + WRITE_SIZE, bench_tools/pmc_traffic.py).

  python bench_tools/c3_table.py before_stats.csv before_traffic.json after_stats.csv after_traffic.json LABEL
"""
import csv
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from pmc_traffic import short_name  # noqa: E402

SKIP = ("at::", "rocclr", "distribution_elementwise", "remainder_kernel")


def load(stats_path, traffic_path):
    rows = {}
    stats = [r for r in csv.DictReader(open(stats_path)) if not any(s in r["Name"] for s in SKIP)]
    # the profile target times ct x ct and relinearize a different number of times: launches per product batch are counted
    # against the one-per-call kernel of each half (the floor; the decomposition's forward transform)
    calls = {short_name(r["Name"]): int(r["Calls"]) for r in stats}
    mul_calls = next(c for n, c in calls.items() if n.startswith("floor_kernel"))
    relin_calls = next(c for n, c in calls.items() if n.startswith("ntt_forward_tiled") and n.endswith(", 1, 2>"))
    for r in stats:
        name = short_name(r["Name"])
        relin = (name.startswith("key_switch_finish") or (name.startswith("ntt_forward_tiled") and ", 1, " in name[-8:]) or
                 (name.startswith("ntt_inverse_tiled") and (", 2, " in name[-8:] or ", 4, " in name[-8:])))
        rows[name] = {"us": float(r["AverageNs"]) / 1e3, "per_batch": int(r["Calls"]) / (relin_calls if relin else mul_calls)}
    traffic = json.load(open(traffic_path))
    for name, t in traffic.items():
        if name.startswith("_") or name not in rows:
            continue
        per_dispatch = (t["fetch_KiB_per_dispatch"] + t["write_KiB_per_dispatch"]) * 1024.0
        rows[name]["bytes_per_product"] = per_dispatch * rows[name]["per_batch"] / 1024.0
    return rows


def main():
    before, after = load(sys.argv[1], sys.argv[2]), load(sys.argv[3], sys.argv[4])
    label = sys.argv[5] if len(sys.argv) > 5 else "before"
    print("# ct x ct + relinearize, N = 8192, L = 4, 1024 products per launch; before = variant `%s`, after = production" % label)
    print("%-46s %10s %10s   %12s %12s" % ("kernel (launches per batch)", "before us", "after us", "before MB/pr", "after MB/pr"))
    totals = [0.0, 0.0, 0.0, 0.0]
    for name in sorted(set(before) | set(after)):
        b, a = before.get(name), after.get(name)
        cells = []
        for k, row in enumerate((b, a)):
            cells.append("%10.1f" % (row["us"] * row["per_batch"]) if row else "%10s" % "-")
            totals[k] += row["us"] * row["per_batch"] if row else 0.0
        for k, row in enumerate((b, a)):
            mb = row.get("bytes_per_product") if row else None
            cells.append("%12.3f" % (mb / 1e6) if mb else "%12s" % "-")
            totals[2 + k] += mb / 1e6 if mb else 0.0
        launches = (a or b)["per_batch"]
        print("%-46s %s %s   %s %s" % ("%s (x%g)" % (name[:40], launches), cells[0], cells[1], cells[2], cells[3]))
    print("%-46s %10.1f %10.1f   %12.3f %12.3f" % ("TOTAL", *totals))
    print("# %.1f -> %.1f k products/s by kernel time alone" % (1024 / totals[0] * 1e3, 1024 / totals[1] * 1e3))


if __name__ == "__main__":
    main()
