#!/bin/bash
# The PIR chunk loop / four queries / whole query with the chunks answered in pieces of K (HEAMD_PIR_PIECE_CHUNKS; 0 = one
# piece, no lane): what the remaining dimensions beside the next piece's database pass are worth.
#   bash bench_tools/pir_overlap_sweep.sh TAG "0 1 2 4"
cd "$GRAFT_REPO_ROOT"
T=${1:-r06c}
O=gpurun_out/$T
mkdir -p $O
for k in ${2:-0 1 2 4}; do
  HEAMD_PIR_PIECE_CHUNKS=$k timeout 600 python bench_tools/path_bench.py \
    --only=config5_pir_chunk_loop_1gpu,config5_pir_4_queries_1gpu,config5_pir_whole_query_1gpu > $O/pieces_$k.json 2> $O/pieces_$k.err
  python - $O/pieces_$k.json $k <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
loop, four, whole = d["config5_pir_chunk_loop_1gpu"], d["config5_pir_4_queries_1gpu"], d["config5_pir_whole_query_1gpu"]
print("piece %s: chunk loop %.3f ms/chunk (%.3f of 8 TB/s, spread %s)  4 queries %.3f ms/chunk/query  whole query median %.2f max %.2f host max %.2f"
      % (sys.argv[2], loop["ms_per_chunk"], loop["frac_of_8TBps"], loop["spread_ms"], four["ms_per_chunk_per_query"],
         whole["ms_per_query_median"], whole["ms_per_query_max"], max(whole["host_enqueue_ms"])))
PY
done
