#!/bin/bash
# round 2, twelfth GPU visit: several queries over one database (parity, then rates)
export PYTHONPATH=swift-homomorphic-encryption_amd:$PYTHONPATH
mkdir -p gpurun_out/r02t
timeout 1200 python -m pytest tests/test_gpu_pir.py -m gpu -x -q > gpurun_out/r02t/tests.log 2>&1
tail -4 gpurun_out/r02t/tests.log
timeout 900 python - > gpurun_out/r02t/queries.json 2> gpurun_out/r02t/queries.err <<'PY'
import json, sys
sys.path.insert(0, "bench_tools")
import torch, heamd, path_bench as pb
out = {"loop_8_single_query": pb.config5_pir_chunk_loop(torch, heamd, d0=256, d1=64, chunks=8)}
for queries in (1, 2, 3, 4):
    out[f"queries_{queries}"] = pb.config5_pir_queries(torch, heamd, d0=256, d1=64, chunks=8, queries=queries)
print(json.dumps(out, indent=1))
PY
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02t/queries.json"))
for k, v in d.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
tail -3 gpurun_out/r02t/queries.err
