#!/bin/bash
# Round 6, weak #1: the whole-PIR-query leg's spread, in the driver's own command (twice, fresh processes), after the other
# legs in one process, alone, and under a HIP-API trace.   bash bench_tools/r06_whole_query.sh TAG
cd "$GRAFT_REPO_ROOT"
T=${1:-r06a}
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$i.json 2> $O/bench_$i.err || tail -5 $O/bench_$i.err
  python - $O/bench_$i.json <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
w = d["extras"]["other_configs"]["config5_pir_whole_query_1gpu"]
print("bench.py whole query: mean %.2f min %.2f median %.2f max %.2f" % (w["ms_per_query"], w["ms_per_query_min"], w["ms_per_query_median"], w["ms_per_query_max"]))
print(" per call:", w["per_call_ms"])
print(" host    :", w["host_enqueue_ms"])
PY
done
timeout 600 python bench_tools/whole_query_spread.py after_legs 30 > $O/after_legs.txt 2>&1; tail -25 $O/after_legs.txt
timeout 600 python bench_tools/whole_query_spread.py alone 30 > $O/alone.txt 2>&1; tail -12 $O/alone.txt
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/trace -- python bench_tools/whole_query_spread.py alone 10 > $O/trace.txt 2>&1
f=$(find $O/trace -name "*hip_api_stats.csv" | head -1); cp "$f" $O/hip_api_stats.csv; head -30 $O/hip_api_stats.csv
rm -rf $O/trace
