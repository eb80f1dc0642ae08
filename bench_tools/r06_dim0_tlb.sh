#!/bin/bash
# Round 6: is the dim-0 pass's dependence on where the database lies (bench_tools/dim0_placement_probe.py) a matter of address
# translation?  The probe under rocprofv3 --pmc (counters only, one pass each), per dispatch of the kernel: duration class and
# UTCL1 misses / UTCL2 busy cycles.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
T=${1:-r06y}; O=gpurun_out/$T; mkdir -p $O
i=0
for g in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_sum TCC_BUBBLE_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/tlb/pass$i -- python bench_tools/dim0_placement_probe.py > $O/tlb_pass$i.log 2>&1 || echo "pass $i failed"
  grep "^allocation" $O/tlb_pass$i.log | cut -c1-60,100-
  python - $O/tlb/pass$i <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
trace = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
counters = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(trace[0])):
    if "inner_product_plain_rows" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(counters[0])):
    if r["Dispatch_Id"] in dur:
        vals[r["Dispatch_Id"]][r["Counter_Name"]] = vals[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# one line per group of consecutive dispatches (an allocation's passes): mean duration and mean counters
ids = sorted(dur, key=int)
group, last = [], None
def flush(g):
    if not g: return
    names = sorted(vals[g[0]])
    print("dispatches %4s-%-4s  n=%2d  mean %.0f us  " % (g[0], g[-1], len(g), sum(dur[i] for i in g) / len(g)) +
          "  ".join("%s %.4g" % (n, sum(vals[i].get(n, 0) for i in g) / len(g)) for n in names))
for i in ids:
    if last is not None and abs(dur[i] - dur[last]) > 0.03 * dur[last] and len(group) >= 3:
        flush(group); group = []
    group.append(i); last = i
flush(group)
PY
done
rm -rf $O/tlb
