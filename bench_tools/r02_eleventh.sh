#!/bin/bash
# round 2, eleventh GPU visit: several queries side by side in the dim-0 inner product
export PYTHONPATH=swift-homomorphic-encryption_amd:$PYTHONPATH
mkdir -p gpurun_out/r02s
timeout 1200 python -m pytest tests/test_gpu_bfv.py tests/test_gpu_pir.py tests/test_gpu_streams.py tests/test_gpu_word32.py -m gpu -x -q -k "inner_product_plain or pir or mul_plain" > gpurun_out/r02s/tests.log 2>&1
tail -4 gpurun_out/r02s/tests.log
timeout 900 python - > gpurun_out/r02s/queries.json 2> gpurun_out/r02s/queries.err <<'PY'
import json, sys
sys.path.insert(0, "bench_tools")
import torch, heamd, path_bench as pb
out = {}
for queries in (1, 2, 3, 4):
    out[f"d0_1024_d1_32_q{queries}"] = pb.config5_inner_product(torch, heamd, count=1024, columns=32, queries=queries)
    out[f"d0_256_d1_64_q{queries}"] = pb.config5_inner_product(torch, heamd, count=256, columns=64, queries=queries)
    out[f"d0_256_d1_64_q{queries}_masked"] = pb.config5_inner_product(torch, heamd, count=256, columns=64, queries=queries, masked=True)
print(json.dumps(out, indent=1))
PY
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02s/queries.json"))
for k, v in d.items():
    print(k, f"{v['ct_pt_mac_per_s'] / 1e6:8.2f} M MAC/s  database {v['database_GBps']:7.0f} GB/s")
PY
tail -3 gpurun_out/r02s/queries.err
