#!/bin/bash
# round 2, thirteenth GPU visit: packed resident database (parity, then rates)
export PYTHONPATH=swift-homomorphic-encryption_amd:$PYTHONPATH
mkdir -p gpurun_out/r02u
timeout 1200 python -m pytest tests/test_gpu_pir.py tests/test_gpu_bfv.py -m gpu -x -q > gpurun_out/r02u/tests.log 2>&1
tail -6 gpurun_out/r02u/tests.log
timeout 900 python - > gpurun_out/r02u/packed.json 2> gpurun_out/r02u/packed.err <<'PY'
import json, sys
sys.path.insert(0, "bench_tools")
import torch, heamd, path_bench as pb
out = {}
for count, columns in ((1024, 128), (256, 64)):
    for packed in (False, True, False, True):
        key = f"d0_{count}_d1_{columns}_{'packed' if packed else 'plain'}"
        r = pb.config5_inner_product(torch, heamd, count=count, columns=columns, packed=packed)
        out.setdefault(key, []).append({k: r[k] for k in ("ct_pt_mac_per_s", "database_GBps")})
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r02u/packed.json; tail -3 gpurun_out/r02u/packed.err
