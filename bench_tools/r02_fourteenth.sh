#!/bin/bash
# round 2, fourteenth GPU visit: the whole server side of a Query in one call
export PYTHONPATH=swift-homomorphic-encryption_amd:$PYTHONPATH
mkdir -p gpurun_out/r02v
timeout 900 python - > gpurun_out/r02v/whole_query.json 2> gpurun_out/r02v/whole_query.err <<'PY'
import json, sys
sys.path.insert(0, "bench_tools")
import torch, heamd, path_bench as pb
out = {}
for indices in (1, 2, 4, 8):
    out[f"indices_{indices}"] = pb.config5_pir_whole_query(torch, heamd, d0=256, d1=64, chunks=8, indices=indices)
print(json.dumps(out, indent=1))
PY
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02v/whole_query.json"))
for k, v in d.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
tail -3 gpurun_out/r02v/whole_query.err
