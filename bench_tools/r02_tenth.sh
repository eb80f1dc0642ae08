#!/bin/bash
# round 2, tenth GPU visit: expansion leaves written by the last level's key switch
export PYTHONPATH=swift-homomorphic-encryption_amd:$PYTHONPATH
mkdir -p gpurun_out/r02r
timeout 1200 python -m pytest tests/test_gpu_galois.py tests/test_gpu_pir.py -m gpu -x -q > gpurun_out/r02r/tests.log 2>&1
tail -4 gpurun_out/r02r/tests.log
timeout 600 python bench_tools/expand_batch_profile_target.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02r/expand_batch.txt
cat gpurun_out/r02r/expand_batch.txt
