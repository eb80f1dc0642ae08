#!/bin/bash
# round 2, seventh GPU visit: batched chunk loop (shared-left inner product) parity + timing
export PYTHONPATH=swift-homomorphic-encryption_amd:$PYTHONPATH
mkdir -p gpurun_out/r02m
timeout 900 python -m pytest tests/test_gpu_pir.py tests/test_gpu_bfv.py -m gpu -x -q > gpurun_out/r02m/tests.log 2>&1
tail -5 gpurun_out/r02m/tests.log
timeout 600 python - > gpurun_out/r02m/chunk_loop.json 2> gpurun_out/r02m/chunk_loop.err <<'PY'
import json, sys
sys.path.insert(0, "bench_tools")
import torch, heamd, path_bench as pb
out = {"single_chunk": pb.config5_pir_chunk(torch, heamd, d0=256, d1=64)}
for chunks in (2, 8, 16):
    out[f"loop_{chunks}"] = pb.config5_pir_chunk_loop(torch, heamd, d0=256, d1=64, chunks=chunks)
out["loop_d0_1024_d1_128_2"] = pb.config5_pir_chunk_loop(torch, heamd, d0=1024, d1=128, chunks=2)
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r02m/chunk_loop.json; tail -5 gpurun_out/r02m/chunk_loop.err
