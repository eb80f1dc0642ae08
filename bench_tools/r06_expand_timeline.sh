#!/bin/bash
# Round 6: where a single query's expansion (320 outputs: the [256, 64] database of the whole-query leg) spends its time --
# kernels in launch order with the idle gaps between them.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
T=${1:-r06w}; O=gpurun_out/$T; mkdir -p $O
EXPAND_OUTPUTS=${2:-320} rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench_tools/expand_profile_target.py > $O/expand.txt 2>&1
grep "expand to" $O/expand.txt
csv=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$csv" > $O/expand_timeline.txt <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(name):
    name = re.sub(r"\(anonymous namespace\)::|heamd::|^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:80]
# a repetition = the kernels between two gaps > 200 us (the host's synchronize between timed calls is shorter than that
# only inside a call): take the last complete run of kernels
runs, current, prev_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and s - prev_end > 200000 and current:
        runs.append(current); current = []
    current.append((short(r["Kernel_Name"]), s, e, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
    prev_end = e if prev_end is None else max(prev_end, e)
if current: runs.append(current)
rep = runs[-1]
# the timed loop runs calls back to back: split the last run at its first kernel's name
first = rep[0][0]
idx = [i for i, k in enumerate(rep) if k[0] == first]
print("kernels in the last run %d, first kernel %s occurs %d times" % (len(rep), first, len(idx)))
busy = gaps = 0; prev = None
for name, s, e, grid, wg in rep:
    gap = 0 if prev is None else max(0, s - prev)
    busy += e - s; gaps += gap
    print("%-80s %8.1f us  gap %6.1f us  grid %s wg %s" % (name, (e - s) / 1e3, gap / 1e3, grid, wg))
    prev = e if prev is None else max(prev, e)
print("busy %.1f us  gaps %.1f us  span %.1f us" % (busy / 1e3, gaps / 1e3, (prev - rep[0][1]) / 1e3))
PY
tail -3 $O/expand_timeline.txt
rm -rf $O/trace
