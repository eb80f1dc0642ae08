#!/bin/bash
# round 2, ninth GPU visit: Bfv.applyGalois / PirUtil.expand without the rotated copy (parity, then timings)
export PYTHONPATH=swift-homomorphic-encryption_amd:$PYTHONPATH
mkdir -p gpurun_out/r02p
timeout 1200 python -m pytest tests/test_gpu_galois.py tests/test_gpu_pir.py tests/test_gpu_bfv.py tests/test_gpu_word32.py -m gpu -x -q > gpurun_out/r02p/tests.log 2>&1
tail -8 gpurun_out/r02p/tests.log
timeout 900 python bench_tools/next_rows_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02p/next_rows.txt
grep "N2" gpurun_out/r02p/next_rows.txt | cut -c1-200
timeout 600 python bench_tools/expand_batch_profile_target.py > gpurun_out/r02p/expand_batch.txt 2>&1
tail -8 gpurun_out/r02p/expand_batch.txt
