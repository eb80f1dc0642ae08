#!/bin/bash
# round 2, eighth GPU visit: tile serialize kernels + chain/stream seeded generator: parity, then timings
export PYTHONPATH=swift-homomorphic-encryption_amd:$PYTHONPATH
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_galois.py -m gpu -x -q > gpurun_out/r02n/tests.log 2>&1
tail -5 gpurun_out/r02n/tests.log
timeout 900 python bench_tools/wire_format_bench.py ab "$@" > gpurun_out/r02n/wire_format_ab2.txt 2>&1
cat gpurun_out/r02n/wire_format_ab2.txt
timeout 900 python bench_tools/next_rows_bench.py > gpurun_out/r02n/next_rows.txt 2>&1
grep -c row gpurun_out/r02n/next_rows.txt
