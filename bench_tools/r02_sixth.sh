#!/bin/bash
# GPU pass 6: parity with the packed UInt32 scheme entry points, UInt32 scheme bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
tail -30 gpurun_out/pytest.log | cut -c1-250
timeout 600 python bench_tools/word32_scheme_bench.py > gpurun_out/word32_scheme.json 2> gpurun_out/word32_scheme.err || tail -5 gpurun_out/word32_scheme.err
cat gpurun_out/word32_scheme.json
