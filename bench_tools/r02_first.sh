#!/bin/bash
# first GPU pass of round 2: parity, then timings
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
tail -5 gpurun_out/pytest.log
timeout 300 python bench_tools/ntt_variants.py > gpurun_out/ntt_variants.txt 2>&1
cat gpurun_out/ntt_variants.txt
timeout 600 python bench.py --skip-other-configs --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
cat gpurun_out/bench_quick.json
