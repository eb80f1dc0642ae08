"""Aggregates rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages per dispatch."""
import collections
import csv
import glob
import os
import re
import sys

out = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "ntt_"
table = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for path in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if needle not in name:
                continue
            m = re.search(r"(\w*%s\w*(<[^>]*>)?)" % re.escape(needle), name)
            short = m.group(1) if m else name
            table[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[short] = (row["Grid_Size"], row["Workgroup_Size"], row["LDS_Block_Size"], row["VGPR_Count"], row["SGPR_Count"])
for kernel in sorted(table):
    print(f"== {kernel}  grid={meta[kernel][0]} wg={meta[kernel][1]} lds={meta[kernel][2]} vgpr={meta[kernel][3]} sgpr={meta[kernel][4]}")
    for counter in sorted(table[kernel]):
        v = table[kernel][counter]
        print(f"   {counter:32s} {sum(v) / len(v):18.1f}   (n={len(v)})")
