"""HBM-side traffic per kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (/ TCC_HIT_sum TCC_MISS_sum) passes.

  python bench_tools/pmc_traffic.py <dir-with-one-subdirectory-per-pass> [units-per-run]

Prints, per kernel (template arguments kept, namespaces and parameter lists dropped): dispatches, KiB fetched and
written per dispatch and in total.  FETCH_SIZE is doubled (gfx950: the counter tallies 128-byte requests at 64 bytes,
/opt/skills/guides/MI355X_MICROARCH.md "HBM"); WRITE_SIZE is taken as reported (1.000 x the algorithmic bytes on the NTT).
With units-per-run the grand total is also printed per unit (e.g. per ciphertext product).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short_name(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"heamd::", "", name)
    depth, out = 0, []
    for ch in name:  # cut the parameter list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def main():
    root = sys.argv[1]
    units = float(sys.argv[2]) if len(sys.argv) > 2 else None
    table = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                table[short_name(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    total_fetch = total_write = 0.0
    report = {}
    for kernel in sorted(table):
        c = table[kernel]
        n = max(len(v) for v in c.values())
        fetch = 2.0 * sum(c.get("FETCH_SIZE", []))
        write = sum(c.get("WRITE_SIZE", []))
        hit, miss = sum(c.get("TCC_HIT_sum", [])), sum(c.get("TCC_MISS_sum", []))
        total_fetch += fetch
        total_write += write
        rate = hit / (hit + miss) if hit + miss > 0 else float("nan")
        report[kernel] = {"dispatches": n, "fetch_KiB_per_dispatch": fetch / n, "write_KiB_per_dispatch": write / n,
                          "tcc_hit_rate": rate}
        print(f"{kernel[:110]:110s} n={n:4d}  fetch {fetch / n / 1024:10.1f} MiB  write {write / n / 1024:10.1f} MiB"
              f"  per dispatch   L2 hit {rate:5.3f}")
    print(f"TOTAL fetch {total_fetch / 1048576:.3f} GiB  write {total_write / 1048576:.3f} GiB")
    if units:
        per_unit = (total_fetch + total_write) * 1024.0 / units
        print(f"per unit ({units:.0f} units): {per_unit:.0f} bytes")
        report["_per_unit_bytes"] = per_unit
    report["_total_fetch_KiB"] = total_fetch
    report["_total_write_KiB"] = total_write
    json.dump(report, open(os.path.join(root, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
